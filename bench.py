#!/usr/bin/env python3
"""bench.py — train samples/s of the IDEE hot path on N MI355X (BASELINE.json metric).

One "step" = zero_grad -> VQ_model forward -> BCE(joint) + lambda*anomaly-L1 + sum_v BCE(head_v)
+ loss_z_q -> backward -> (gradient all-reduce) -> Adam, on one batch of synthetic clips already
resident in HBM.  Workload at every N: BASELINE config #2/#3 — Swin_3D "tiny" (the reference's
default sizes), V=6, C=1, T=8, 200x200, 4 clips per GPU, fp32 (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--hw H] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (duration from
HIP events recorded on the launch stream inside the timed region) and `cpu_baseline` (the CPU
oracle, oracle/idee_oracle.py, timed on this box's host cores on one clip — rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: exact-f32 MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU")
    ap.add_argument("--hw", type=int, default=200, help="grid height = width")
    ap.add_argument("--encoder", default="Swin_3D")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-entry-point time table (stderr)")
    return ap.parse_args()


def synthetic_batch(cfg, n, T, H, W, rank, device):
    """seeded randn clamped to the dataset's [-10, 10], 10 % / 20 % masks."""
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn((n, cfg.in_channels_dynamic, cfg.in_channels, T, H, W), generator=g).clamp_(-10, 10)
    me = (torch.rand((n, H, W), generator=g) < 0.10).float()
    ml = (torch.rand((n, H, W), generator=g) < 0.20).float()
    return x.to(device), me.to(device), ml.to(device)


# ---- algorithmic FLOPs per launch of each convolution entry point (2 * MACs of the reference conv) ----
def conv_flops(desc):
    taps = {0: 27, 1: 18, 2: 9}[desc.mode]                 # K333_REPLICATE, K233_S2_ZERO, K133_ZERO
    To = (desc.T - 2) // 2 + 1 if desc.mode == 1 else desc.T
    return 2.0 * desc.B * To * desc.H * desc.W * taps * desc.Cin * desc.Cout


def swin_flops(geom, backward):
    """Per token: qkv 6C^2 + attention 4*Nw*C + proj 2C^2 + MLP 16C^2; backward = 2x."""
    nw = geom.wd * geom.wh * geom.ww
    per_tok = 6 * geom.C ** 2 + 4 * nw * geom.C + 2 * geom.C ** 2 + 16 * geom.C ** 2
    f = float(geom.N * geom.V * geom.D * geom.H * geom.W) * per_tok
    return 2 * f if backward else f


_HEADLINE = False   # set in main(): the PMC summary was taken at the headline workload only


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes of this same command, corrected as MI355X_MICROARCH.md prescribes; see
    profiles/collect_pmc.py) — or None when the summary has no unambiguous row for this entry point (e.g. the joint
    head's two convolutions share one kernel name but have different shapes)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not _HEADLINE or not os.path.exists(path):
        return None
    try:
        rows = {r["kernel"]: r["hbm_bytes_per_launch"] for r in json.load(open(path))["kernels"]}
    except Exception:  # noqa: BLE001
        return None
    name = key.split("[")[0]
    nw = "32" if "w244" in key else "8"
    want = None
    if name == "idee_swin_block_bwd":       # two kernels per launch
        want = ["swin_mlp_bwd_kernel<16, 2>", f"swin_attn_bwd_kernel<16, 2, {nw}, 8, 2>"]
    elif name == "idee_swin_block_fwd":
        want = [f"swin_fwd_mfma_kernel<16, 2, {nw}, true>"]
    elif "k333 16->16" in key:
        want = [{"idee_conv_fwd": "conv16_lds_kernel<0>", "idee_conv_dgrad": "conv16_lds_kernel<1>",
                 "idee_conv_wgrad": "wgrad16_lds_kernel<0, 8>"}[name]]
    if not want or any(k not in rows for k in want):
        return None
    return round(sum(rows[k] for k in want))


def host_cores():
    """CPU cores this process may actually use: min(affinity, cgroup cpu.max quota) — the GPU box exposes 256 logical
    CPUs but grants a 16-CPU quota, and 128 torch threads on 16 CPUs run ~5x slower than 16 threads."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return n


def cpu_baseline(cfg, model, T, H, W):
    """The oracle (CPU restatement, pinned to the reference by tests/golden) on ONE clip of the
    same workload: forward + losses + backward + Adam, 1 warm-up + 2 timed steps."""
    from oracle import idee_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    ocfg = O.default_cfg(encoder=cfg.encoder, in_channels_dynamic=cfg.in_channels_dynamic, in_channels=cfg.in_channels)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, v in sd.items() if v.is_floating_point()]
    x, me, ml = synthetic_batch(cfg, 1, T, H, W, 0, "cpu")
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    times = []
    for step in range(1, 4):
        t0 = time.perf_counter()
        for k in names:
            sd[k] = sd[k].detach().requires_grad_()
        outs = O.vq_model_forward(x, sd, ocfg, training=True)
        total, _ = O.train_losses_synthetic(outs, me, ml, sd, ocfg)
        total.backward()
        with torch.no_grad():
            for k in names:
                p, m, v = O.adam_step(sd[k].detach(), sd[k].grad, state[k][0], state[k][1], step, cfg.lr, cfg.beta1,
                                      cfg.beta2, 1e-8, cfg.weight_decay)
                state[k] = (m, v)
                sd[k] = p
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    return dict(value=round(1.0 / t, 4), unit="samples/s", cores=cores, kind="port",
                sample=f"1 clip V={cfg.in_channels_dynamic} T={T} {H}x{W}: oracle fwd+losses+bwd+Adam, 1 warm-up + 2 timed steps")


def main():
    a = parse()
    from idee_amd import _lib, parallel
    from idee_amd.config import default_config
    from idee_amd.train import SyntheticTrainStep
    rank, world, local = parallel.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.check(_lib.load().idee_check_device(local), "idee_check_device")
    cfg = default_config(encoder=a.encoder, in_channels=1)
    T, H, W = 8, a.hw, a.hw
    global _HEADLINE
    _HEADLINE = (a.hw == 200 and a.batch == 4 and a.encoder == "Swin_3D")
    torch.manual_seed(0)
    ts = SyntheticTrainStep(cfg, device=dev)
    x, me, ml = synthetic_batch(cfg, a.batch, T, H, W, rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        ts.step(x, me, ml)
    timed = ["idee_conv_fwd", "idee_conv_dgrad", "idee_conv_wgrad", "idee_swin_block_fwd", "idee_swin_block_bwd",
             "idee_lfq_fwd", "idee_lfq_bwd", "idee_patch_embed_ln_fwd", "idee_patch_embed_ln_bwd", "idee_anomaly_l1_fwd",
             "idee_anomaly_l1_bwd", "idee_adam_step"]
    log = _lib.set_event_log(timed)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ts.step(x, me, ml)
    barrier()
    dt = time.perf_counter() - t0
    _lib.set_event_log(None)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)

    # ---- per-entry-point durations (HIP events on the launch stream), dominant kernel, roofline ----
    import ctypes
    per = {}
    for name, evs in log.items():
        for e0, e1, args in evs:
            ms = e0.elapsed_time(e1)
            key, flops = name, 0.0
            if name.startswith("idee_conv_"):
                d = ctypes.cast(args[0], ctypes.POINTER(_lib.ConvDesc)).contents
                key = f"{name}[{('k333', 'k233', 'k133')[d.mode]} {d.Cin}->{d.Cout} B{d.B} T{d.T}]"
                flops = conv_flops(d)
            elif name.startswith("idee_swin_block_"):
                g = ctypes.cast(args[-2], ctypes.POINTER(_lib.SwinGeom)).contents
                key = f"{name}[w{g.wd}{g.wh}{g.ww} s{g.sd}{g.sh}{g.sw}]"
                flops = swin_flops(g, name.endswith("bwd"))
            r = per.setdefault(key, [0.0, 0, 0.0])
            r[0] += ms; r[1] += 1; r[2] = flops
    rows = sorted(per.items(), key=lambda kv: -kv[1][0])
    if a.breakdown and rank == 0:
        tot = sum(v[0] for v in per.values())
        sys.stderr.write(f"{'entry point':62s} {'ms/step':>9s} {'calls':>6s} {'avg ms':>8s} {'TFLOP/s':>8s}\n")
        for k, (ms, n, fl) in rows:
            sys.stderr.write(f"{k:62s} {ms / a.steps:9.3f} {n // a.steps:6d} {ms / n:8.3f} {fl / (ms / n) / 1e9 if fl else 0:8.2f}\n")
        sys.stderr.write(f"sum of bracketed entry points {tot / a.steps:.3f} ms/step; wall {dt / a.steps * 1e3:.3f} ms/step\n")
    roof = None
    mfma_rows = [r for r in rows if r[1][2] > 0]
    if mfma_rows:
        k, (ms, n, fl) = mfma_rows[0]          # the entry point with the largest total time in the timed region
        avg_ms = ms / n
        roof = dict(kernel=k, bound="mfma", achieved=round(fl / avg_ms / 1e9, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(fl / avg_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4), traffic=pmc_traffic(k), avg_launch_ms=round(avg_ms, 4),
                    flops_per_launch=fl, share_of_step=round(ms / a.steps / (dt / a.steps * 1e3), 4))

    # ---- second half of BASELINE.json's metric: forward ms per clip (eval mode, no autograd), outside the timed region ----
    ts.model.eval()
    with torch.no_grad():
        for _ in range(2):
            ts.model(x)
        barrier()
        f0 = time.perf_counter()
        for _ in range(5):
            ts.model(x)
        barrier()
        fwd_ms_clip = (time.perf_counter() - f0) / 5 / a.batch * 1e3
    ts.model.train()

    if rank != 0:
        return
    out = dict(metric="train samples/sec (TxHxW clips); one step = fwd+losses+bwd+Adam", value=round(world * a.batch * a.steps / dt, 3),
               unit="samples/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=f"Synthetic CERRA {H}x{W}, T={T}, V={cfg.in_channels_dynamic}, {a.encoder} tiny "
                                    f"(reference defaults), batch {a.batch}/GPU", global_batch=world * a.batch,
                           parallelism=f"dp{world}"),
               roofline=roof, fwd_ms_per_clip=round(fwd_ms_clip, 3))
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, ts.model, T, H, W)
        except Exception as e:  # noqa: BLE001  (the GPU measurement above must still be reported)
            out["cpu_baseline"] = dict(value=None, unit="samples/s", cores=host_cores(), kind="port", sample=f"failed: {e!r}")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
