#!/usr/bin/env python3
"""Time one f32-storage convolution of the headline step in isolation (GPU box), all three passes:
    python3 tools/conv_bench_f32.py k333|k233|k133|c1 [Cin] [Cout] [--B B] [--wsets S] [--T T] [--hw HW] [--math f22|f32|split]
                                    [--opt NAME=VALUE ...] [--reps R]
Defaults are config #2's shapes: k333 16->16 B=24 (6 weight sets) T=8; k233 96->96 B=4 (1 set) T=8; k133 16->96 B=4 T=4;
c1 = the Cout = 1 (2,3,3) logit convs (Cin 16: B=24, 6 sets; Cin 96: B=4), T=4 in, 2 frames used."""
import argparse, sys, torch
sys.path.insert(0, ".")
from idee_amd import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("name", choices=["k333", "k233", "k133", "c1"])
ap.add_argument("Cin", nargs="?", type=int, default=None)
ap.add_argument("Cout", nargs="?", type=int, default=None)
ap.add_argument("--B", type=int, default=None)
ap.add_argument("--wsets", type=int, default=None)
ap.add_argument("--T", type=int, default=None)
ap.add_argument("--hw", type=int, default=200)
ap.add_argument("--math", default="f22")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dflt = dict(k333=(16, 16, 24, 6, 8), k233=(96, 96, 4, 1, 8), k133=(16, 96, 4, 1, 4), c1=(16, 1, 24, 6, 2))[a.name]
Cin, Cout = a.Cin or dflt[0], a.Cout or dflt[1]
if a.name == "c1" and Cin != 16 and a.B is None:
    dflt = (Cin, 1, 4, 1, 2)
B, wsets, T = a.B or dflt[2], a.wsets or dflt[3], a.T or dflt[4]
H = W = a.hw
L.set_option("IDEE_CONV_MATH", a.math)
for o in a.opt:
    k, v = o.split("=", 1)
    L.set_option(k, v)
dev = "cuda"
mode = dict(k333=L.CONV_K333_REPLICATE, k233=L.CONV_K233_S2_ZERO, k133=L.CONV_K133_ZERO, c1=L.CONV_K233_S2_ZERO)[a.name]
kt = dict(k333=3, k233=2, k133=1, c1=2)[a.name]
To = (T - 2) // 2 + 1 if kt == 2 else T
ys = ops._cl_strides(To, H, W, Cout) if Cout != 1 else (To * H * W, H * W, W, 1, 1)
conv = ops.ConvLayer(mode, B, wsets, T, H, W, Cin, Cout, ops._cl_strides(T, H, W, Cin), ys, Cout != 1)
w = torch.randn((wsets, Cout, Cin, kt, 3, 3), device=dev) * 0.05
bias = torch.randn((wsets, Cout), device=dev) * 0.1
pf, pd = conv.pack(w)
x = torch.randn((B, T, H, W, Cin), device=dev)
y = torch.empty((B, To, H, W, Cout) if Cout != 1 else (B, To, H, W), device=dev)
dy = torch.randn_like(y)
dx = torch.empty_like(x)


def timeit(f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


flops = 2.0 * B * To * H * W * Cin * Cout * kt * 9
tag = f"{a.name} {Cin}->{Cout} B{B} T{T} {H}x{W} [{a.math}{' ' + ' '.join(a.opt) if a.opt else ''}]"
t = timeit(lambda: conv.fwd(x, pf, bias, y))
print(f"{tag} fwd   {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {(x.numel() + y.numel()) * 4 / t / 1e3:7.1f} GB/s")
t = timeit(lambda: conv.dgrad(dy, pd, x if Cout != 1 else None, dx))
print(f"{tag} dgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {(2 * x.numel() + y.numel()) * 4 / t / 1e3:7.1f} GB/s")
t = timeit(lambda: conv.wgrad(x, dy, w))
print(f"{tag} wgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  (incl. zero fill + unpack)")
