#!/usr/bin/env python3
"""Time the bf16 convolution passes of config #4 in isolation (GPU box):
    python3 tools/conv_bench.py [k333|k233|k133] [Cin] [Cout] [--reps R]
Shapes: B = 6 weight sets x 1 item, T = 16, 412 x 424 (k133: the heads' 16 -> 192 / 192 -> 16 convs run on T = 8)."""
import argparse, sys, torch
sys.path.insert(0, ".")
from idee_amd import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("name", nargs="?", default="k333", choices=["k333", "k233", "k133"])
ap.add_argument("Cin", nargs="?", type=int, default=32)
ap.add_argument("Cout", nargs="?", type=int, default=32)
ap.add_argument("--reps", type=int, default=10)      # (an option, so that a wrapper can append it after any number of positionals)
_a = ap.parse_args()
name, Cin, Cout, reps = _a.name, _a.Cin, _a.Cout, _a.reps
dev, BF = "cuda", torch.bfloat16
mode = dict(k333=L.CONV_K333_REPLICATE, k233=L.CONV_K233_S2_ZERO, k133=L.CONV_K133_ZERO)[name]
kt = dict(k333=3, k233=2, k133=1)[name]
B, T, H, W = 6, 16, 412, 424
To = (T - 2) // 2 + 1 if name == "k233" else T
conv = ops.ConvLayer(mode, B, B, T, H, W, Cin, Cout, ops._cl_strides(T, H, W, Cin), ops._cl_strides(To, H, W, Cout), True, act=BF)
w = torch.randn((B, Cout, Cin, kt, 3, 3), device=dev) * 0.05
bias = torch.randn((B, Cout), device=dev) * 0.1
pf, pd = conv.pack(w)
x = torch.randn((B, T, H, W, Cin), device=dev).to(BF)
y = torch.empty((B, To, H, W, Cout), device=dev, dtype=BF)
dy = torch.randn((B, To, H, W, Cout), device=dev).to(BF)
dx = torch.empty_like(x)


def timeit(f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


px = B * To * H * W
flops = 2.0 * px * Cin * Cout * kt * 9
t = timeit(lambda: conv.fwd(x, pf, bias, y))
print(f"{name} {Cin}->{Cout} fwd   {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {(x.numel() + y.numel()) * 2 / t / 1e3:7.1f} GB/s")
t = timeit(lambda: conv.dgrad(dy, pd, x, dx))
print(f"{name} {Cin}->{Cout} dgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {(2 * x.numel() + y.numel()) * 2 / t / 1e3:7.1f} GB/s")
t = timeit(lambda: conv.wgrad(x, dy, w))
print(f"{name} {Cin}->{Cout} wgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s")
