#!/usr/bin/env python3
"""Time the 192-wide bf16 convolution passes at config #4's shapes (GPU box):  python3 tools/conv_bench_c4.py"""
import sys, torch
sys.path.insert(0, ".")
from idee_amd import ops, _lib as L
dev, BF = "cuda", torch.bfloat16


def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, Cin, Cout, T in (("k233", 192, 192, 8), ("k133", 16, 192, 8)):
    mode = dict(k233=L.CONV_K233_S2_ZERO, k133=L.CONV_K133_ZERO)[name]
    kt = 2 if name == "k233" else 1
    B, H, W = 1, 412, 424
    To = (T - 2) // 2 + 1 if name == "k233" else T
    conv = ops.ConvLayer(mode, B, 1, T, H, W, Cin, Cout, ops._cl_strides(T, H, W, Cin), ops._cl_strides(To, H, W, Cout), True, act=BF)
    w = torch.randn((1, Cout, Cin, kt, 3, 3), device=dev) * 0.05
    bias = torch.randn((1, Cout), device=dev) * 0.1
    pf, pd = conv.pack(w)
    x = torch.relu(torch.randn((B, T, H, W, Cin), device=dev)).to(BF)
    y = torch.empty((B, To, H, W, Cout), device=dev, dtype=BF)
    dy = torch.randn((B, To, H, W, Cout), device=dev).to(BF)
    dx = torch.empty_like(x)
    flops = 2.0 * B * To * H * W * Cin * Cout * kt * 9
    t = timeit(lambda: conv.fwd(x, pf, bias, y)); print(f"{name} {Cin}->{Cout} fwd   {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s")
    t = timeit(lambda: conv.dgrad(dy, pd, x, dx)); print(f"{name} {Cin}->{Cout} dgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s")
    t = timeit(lambda: conv.wgrad(x, dy, w)); print(f"{name} {Cin}->{Cout} wgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s")
