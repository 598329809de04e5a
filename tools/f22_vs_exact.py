#!/usr/bin/env python3
"""Headline-shape training, the f16-piece f32-level kernels (IDEE_SWIN_MATH = IDEE_CONV_MATH = f22) against the exact-f32 kernels:
two models from one initialisation take the same STEPS synthetic batches, one per arithmetic (the options are switched between the
two steps); prints both losses per step and the relative difference of losses and of the parameters at the end.  A kernel that
returned a wrong tile once would show as a jump; f32-level agreement is ~1e-6 on the loss of the first steps, growing with the
trajectories' divergence.   python3 tools/f22_vs_exact.py [STEPS=20]"""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idee_amd import _lib as L
import bench
from idee_amd.config import default_config
from idee_amd.train import SyntheticTrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
enc, H, W, T, Cin, batch, over, kind, label = bench.CONFIGS["2"]
cfg = default_config(encoder=enc, in_channels=Cin, **over)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
a = SyntheticTrainStep(cfg, device=dev)
torch.manual_seed(0)
b = SyntheticTrainStep(cfg, device=dev)
sa, sb = a.model.state_dict(), b.model.state_dict()
assert all(torch.equal(sa[k], sb[k]) for k in sa), "the two models do not start equal"


def arith(m):
    L.set_option("IDEE_SWIN_MATH", m); L.set_option("IDEE_CONV_MATH", m)


worst = 0.0
for s in range(steps):
    x, masks = bench.synthetic_batch(cfg, kind, batch, T, H, W, s, dev)
    arith("f22"); la = float(a.step(x, *masks)[0])
    arith("f32"); lb = float(b.step(x, *masks)[0])
    rel = abs(la - lb) / max(abs(lb), 1e-30)
    worst = max(worst, rel)
    print(f"step {s:3d}  f22 {la:.7f}  exact {lb:.7f}  rel {rel:.2e}", flush=True)
    assert la == la and lb == lb, "NaN"
pa, pb = a.opt.flat_p, b.opt.flat_p
print(f"worst relative loss difference {worst:.2e}; parameters after {steps} steps: relative L2 difference "
      f"{float((pa - pb).norm() / pb.norm()):.2e}")
