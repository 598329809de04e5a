#!/usr/bin/env python3
"""bench.py — train samples/s of the IDEE hot path on N MI355X (BASELINE.json metric).

One "step" = zero_grad -> VQ_model forward -> BCE(joint) + lambda*anomaly-L1 + sum_v BCE(head_v)
+ loss_z_q -> backward -> (gradient all-reduce) -> Adam, on one batch of synthetic clips already
resident in HBM.  Default workload (--config 2): BASELINE config #2/#3 — Swin_3D "tiny" (the reference's
default sizes), V=6, C=1, T=8, 200x200, 4 clips per GPU, fp32 (weak scaling).  --config 5: the same with the
CNN_3D encoder; --config 4: ERA5-Land-shaped 412x424, T=16, C=2, Swin "small" (embed [32,32], depths [2,2]) through
the train_ERA5_Land.py loss sequence; --config 1: the CNN_3D 64x64 plumbing case (at T=8: the reference's classifier
cannot run T=4).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--batch B] [--no-cpu-baseline] [--breakdown]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no torchrun environment, this script starts the N ranks itself (child processes, created
BEFORE this process touches the GPU; rendezvous on 127.0.0.1) and exits with their status; it never prints a line
for fewer ranks than were asked for.

Prints ONE JSON line (rank 0).  `value` / `ms_per_step` / `roofline` are the EXACT-F32 arithmetic (--math f32, the default:
the reference's fp32).  Extra objects: `roofline` for the dominant kernel FAMILY (all launches of the entry point with the
largest share of the step; duration from HIP events recorded on the launch stream inside the timed region; priced against the
ceiling of the pipe that ran, see roofline_of), `f22` (the f16-piece arithmetic of the same workload timed in the same run,
with its own roofline — a side line, narrower than f32) and `cpu_baseline` (the CPU oracle, oracle/idee_oracle.py, timed on
this box's host cores — rank 0, N=1 only).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: exact-f32 MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")             # --math f32 (the default line)
PMC_SUMMARY_F22 = os.path.join(ROOT, "profiles", "r05_pmc_traffic_f22.json")     # --math f22 (the `f22` side line)
PMC_SUMMARY_BF16 = os.path.join(ROOT, "profiles", "r05_pmc_traffic_config4_bf16.json")

CONFIGS = {
    # id: (encoder, H, W, T, C_in, per-GPU batch, model overrides, step kind, label)
    "1": ("CNN_3D", 64, 64, 8, 1, 2, {}, "synthetic", "Synthetic_dataset 64x64, T=8 (reference-valid stand-in for T=4), CNN_3D"),
    "2": ("Swin_3D", 200, 200, 8, 1, 4, {}, "synthetic", "Synthetic CERRA 200x200, T=8, Swin_3D tiny (reference defaults)"),
    "4": ("Swin_3D", 412, 424, 16, 2, 1, dict(en_embed_dim=[32, 32], en_depths=[2, 2], codebook_dim=32, cls_dim=32, delta_t=16),
          "era5", "ERA5-Land-shaped synthetic 412x424 (EUR-11 grid), T=16, C=2, Swin_3D small (embed [32,32], depths [2,2]), "
                  "train_ERA5_Land.py losses"),
    "5": ("CNN_3D", 200, 200, 8, 1, 4, {}, "synthetic", "Synthetic CERRA 200x200, T=8, CNN_3D encoder (reference defaults)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS), help="BASELINE.json config number")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the config's)")
    ap.add_argument("--hw", type=int, default=None, help="override grid height = width")
    ap.add_argument("--encoder", default=None, help="override the config's encoder")
    ap.add_argument("--math", default="f32", choices=["f32", "split", "fwd22", "f22"],
                    help="f32 (default = the headline): every GEMM-shaped kernel on the exact-f32 MFMA — the reference's arithmetic and the "
                         "library's own default; with one GPU the f16-piece line (`f22`) is measured in the same run and printed beside it as a "
                         "labelled side object with its own roofline.  f22: f32 storage and accumulation, GEMM operands as two f16 pieces "
                         "(operand rounding 2^-22: narrower than f32) with power-of-two prescaled weights, per-tile gradient scales and an "
                         "activation range guard, Swin block forward + backward and the 16->16 / 96-wide / (1,3,3)-forward convolutions "
                         "(IDEE_SWIN_MATH = IDEE_CONV_MATH = f22); `exact_f32` is then printed beside it.  fwd22: only the FORWARD passes on f16 "
                         "pieces.  split: f16 pieces forward, bf16 pieces (2^-18) backward.  Neither is ever the default line")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="activation storage between the fused units.  f32 (default): the reference's arithmetic.  bf16: BASELINE "
                         "config #4's precision — bf16 tokens, bf16 MFMA GEMMs / convolutions with f32 accumulate, f32 LFQ, losses, "
                         "master weights and optimizer (Swin_3D encoder)")
    ap.add_argument("--no-side-line", "--no-exact-line", dest="no_side_line", action="store_true",
                    help="skip the other arithmetic's line (--math f32: `f22`; --math f22: `exact_f32`) measured beside the headline")
    ap.add_argument("--codebook", default="LFQ", choices=["LFQ", "VQ"],
                    help="LFQ: what the reference's build.py wires.  VQ: the nearest-neighbour codebook of models/codebook/VQ.py in its "
                         "place (BASELINE config #4's '+ VQ codebook' variant line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fwd-eval", action="store_true", help="skip the eval-mode forward timing (profiling runs: every launch then belongs to a train step)")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-entry-point / per-shape time table (stderr)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# multi-rank launch (no GPU call may precede this in the parent)
# ---------------------------------------------------------------------------------------------
def visible_gpus():
    """GPUs this process may use, WITHOUT opening the HIP runtime in the launcher: the KFD topology in sysfs (a node with
    simd_count > 0 is a GPU), narrowed by HIP_/ROCR_/CUDA_VISIBLE_DEVICES.  None when sysfs is unreadable (then every
    child validates its own device index and the launcher only polices their exit codes)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:  # noqa: BLE001
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_ranks(a, argv=None, poll_s=0.2):
    """Start a.gpus child ranks of this script (one fresh process per GPU; this launcher never touches HIP), poll them, and
    on the FIRST non-zero exit terminate the siblings (a rank that dies before the first collective would otherwise leave
    them blocked in the rendezvous until the driver's limit) and return that status."""
    import socket
    forced = "IDEE_FORCE_DEVICE" in os.environ
    have = visible_gpus()
    if have is not None and have < a.gpus and not forced:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) are visible; refusing to report a smaller run\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = argv if argv is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   IDEE_BENCH_CHILD="1")
        procs.append(subprocess.Popen(cmd, env=env))
    rc, live = 0, list(procs)
    while live and rc == 0:
        time.sleep(poll_s)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = r
                break
    if rc:
        sys.stderr.write(f"bench.py: rank process {procs.index(p)} exited with status {rc}; stopping the other ranks\n")
        for q in live:
            q.terminate()                      # exact PIDs this launcher started
        deadline = time.time() + 10
        for q in live:
            try:
                q.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()
    return rc


def synthetic_batch(cfg, kind, n, T, H, W, rank, device):
    """seeded randn clamped to the dataset's [-10, 10], 10 % / 20 % masks (+5 % cold-surface masks)."""
    import torch
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn((n, cfg.in_channels_dynamic, cfg.in_channels, T, H, W), generator=g).clamp_(-10, 10)
    me = (torch.rand((n, H, W), generator=g) < 0.10).float()
    ml = (torch.rand((n, H, W), generator=g) < 0.20).float()
    if kind == "era5":
        cold = (torch.rand((n, H, W), generator=g) < 0.05).float()
        cold_t = (torch.rand((n, T, H, W), generator=g) < 0.05).float()
        return x.to(device), (me.to(device), ml.to(device), cold.to(device), cold_t.to(device))
    return x.to(device), (me.to(device), ml.to(device))


# ---- algorithmic FLOPs per launch of each convolution entry point (2 * MACs of the reference conv) ----
def conv_flops(desc):
    taps = {0: 27, 1: 18, 2: 9}[desc.mode]                 # K333_REPLICATE, K233_S2_ZERO, K133_ZERO
    To = (desc.T - 2) // 2 + 1 if desc.mode == 1 else desc.T
    return 2.0 * desc.B * To * desc.H * desc.W * taps * desc.Cin * desc.Cout


def conv_bytes(desc):
    """Algorithmic HBM bytes of one launch of any pass of the convolution (a fused unit reads its input once
    and writes its output once): the input tensor + the output tensor (forward: x in, y out; data gradient: dy in, dx out;
    weight gradient: x and dy in).  Weights, bias and the ReLU reference are not counted."""
    es = 2 if desc.act_dtype == 1 else 4
    To = (desc.T - 2) // 2 + 1 if desc.mode == 1 else desc.T
    return float(desc.B * desc.T * desc.H * desc.W * desc.Cin * es + desc.B * To * desc.H * desc.W * desc.Cout * (4 if desc.Cout == 1 else es))


def swin_bytes(geom, backward):
    """Algorithmic HBM bytes of a Swin block launch: forward x in, y and the saved pre-MLP state out (3 token tensors in a train
    step); backward x, saved state, dy in and dx out (4 token tensors; the dh hand-off between its two kernels is traffic, not
    algorithm)."""
    tok = float(geom.N * geom.V * geom.D * geom.H * geom.W * geom.C) * (2 if geom.act_dtype == 1 else 4)
    return (4 if backward else 3) * tok


def swin_flops(geom, backward):
    """Per token: qkv 6C^2 + attention 4*Nw*C + proj 2C^2 + MLP 16C^2; backward = 2x."""
    nw = geom.wd * geom.wh * geom.ww
    per_tok = 6 * geom.C ** 2 + 4 * nw * geom.C + 2 * geom.C ** 2 + 16 * geom.C ** 2
    f = float(geom.N * geom.V * geom.D * geom.H * geom.W) * per_tok
    return 2 * f if backward else f


# which device kernels run inside one launch of a C-ABI entry point (for the PMC traffic lookup)
FAMILY_KERNELS = {
    "idee_swin_block_fwd": ("swin_fwd_",),
    "idee_swin_block_bwd": ("swin_mlp_bwd", "swin_attn_bwd"),
    "idee_conv_fwd": ("conv_bf16_kernel<0", "conv_bf16_kernel<2", "conv16_lds_kernel<0", "conv16_lds_kernel<2", "conv16_glds_kernel<0", "conv16_split_kernel<0", "conv16_split_kernel<2",
                      "conv_wide_kernel<0", "conv_wide_kernel<2", "conv_wide_split_kernel<2", "c1_fwd", "conv_kernel<0", "conv_kernel<2"),
    "idee_conv_dgrad": ("conv_bf16_kernel<1", "conv_bf16_kernel<3", "conv16_lds_kernel<1", "conv16_lds_kernel<3", "conv16_glds_kernel<1", "conv16_split_kernel<1", "conv16_split_kernel<3",
                        "conv_wide_kernel<1", "conv_wide_kernel<3", "conv_wide_split_kernel<3", "c1_dgrad", "conv_kernel<1", "conv_kernel<3",
                        "init_w_edges"),
    "idee_conv_wgrad": ("wgrad_bf16_kernel", "wgrad16_strip_kernel", "wgrad16_split_kernel", "wgrad_wide_lds_kernel", "c1_wgrad", "wgrad_kernel"),
}


def pmc_traffic(family, launches_per_step, headline, summary=None):
    """Average HBM bytes per launch of the family from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes of this same command, corrected as MI355X_MICROARCH.md prescribes; profiles/collect_pmc.py) — or None
    when the summary is absent, was taken at another workload, or lists no kernel of the family.  -> (bytes, source)."""
    summary = summary or PMC_SUMMARY
    if not headline or not os.path.exists(summary) or family not in FAMILY_KERNELS:
        return None, None
    try:
        doc = json.load(open(summary))
        tot = 0.0
        for r in doc["kernels"]:
            if any(r["kernel"].startswith(p) for p in FAMILY_KERNELS[family]):
                tot += r["hbm_bytes_per_launch"] * r["launches_per_step"]
    except Exception:  # noqa: BLE001
        return None, None
    if tot <= 0 or not launches_per_step:
        return None, None
    return round(tot / launches_per_step), f"profiles/{os.path.basename(summary)} @ {doc.get('source_commit', '?')}"


def math_label(math, cfg, dtype="f32"):
    """What arithmetic the GEMM-shaped kernels of THIS run used (the library switches are pinned in main())."""
    swin = cfg.encoder == "Swin_3D"
    if dtype == "bf16":
        return dict(storage="bf16 tokens between the fused units (patch embedding -> Swin blocks -> 3^3 convs -> classifier) and their "
                            "gradients; f32 parameters, parameter gradients, Adam moments, LFQ projection / code bits, logits, losses",
                    conv="v_mfma_f32_16x16x32_bf16, f32 accumulate (weights rounded to bf16 once per step); the Cout = 1 logit convs "
                         "are f32 streaming dot products on bf16 inputs",
                    swin_gemm="linear layers, P.V and the token-reduction weight-gradient GEMMs on v_mfma_f32_16x16x32_bf16 with "
                              "one-piece bf16 operands, f32 accumulate; LayerNorm, softmax, GELU and the attention scores in f32")
    if math == "f32":
        d = dict(conv="exact f32 MFMA (v_mfma_f32_16x16x4_f32)", everything_else="f32")
        if swin:
            d["swin_gemm"] = "exact f32 MFMA (v_mfma_f32_16x16x4_f32) for every linear layer, the attention scores and P.V, at every width"
        return d
    if math in ("fwd22", "f22"):
        d = dict(conv="FORWARD of the 16->16 and 96-wide convs: two f16 pieces per operand, 3 piece products (operand rounding 2^-22, "
                      "weights pre-scaled by a power of two); every data and weight gradient exact f32 MFMA",
                 everything_else="f32")
        if swin:
            d["swin_gemm"] = ("forward: two f16 pieces per operand, f32 accumulate (operand rounding 2^-22), the attention scores (head_dim "
                              "8: a lane's two dims x three piece products in the K = 32 slots of one MFMA) and P.V included; backward: "
                              "exact f32 MFMA")
            if math == "f22":
                d["conv"] = ("16->16 and 96-wide convs, forward, data gradient and weight gradient, and the (1,3,3) forward and weight "
                             "gradient: two f16 pieces per operand, 3 piece products; weights prescaled by a power of two per set, dy scaled "
                             "by a power of two per work unit / chunk (data gradients, the scale only moving down) or per row (weight "
                             "gradients, accumulators rescaled along), activations under a range guard (split as they are until a staged "
                             "tile's largest |x| reaches 2^15, then scaled by an exact 2^k: any finite input keeps two normal pieces); data "
                             "gradient of the (1,3,3) sign-field convs and the Cout = 1 convs exact f32")
                d["swin_gemm"] = ("forward and backward: two f16 pieces per operand, f32 accumulate — all 4 piece products in the linear layers, "
                                  "3 (lo x lo, 2^-22 of a product, dropped) where two 16-token / 16-key blocks share the K = 32 slots of an "
                                  "MFMA: P.V, the attention backward's row pass, the token-reduction (weight-gradient) GEMMs; weight tensors "
                                  "prescaled by a power of two per tensor (largest |w| in [2^6, 2^7)), the incoming gradient scaled by a "
                                  "power of two per 32-token work item (largest |dy| in [2^1, 2^8)): operand rounding max(2^-22 |x|, "
                                  "2^-26 of the tile's / tensor's largest element); forward attention scores (head_dim 8) and P.V on f16 "
                                  "pieces too; the backward's S = K Q^T and dP = V dO^T (K = head_dim) on the exact f32 MFMA")
        return d
    d = dict(conv="16->16 convs (all three passes) and the forward / data gradient of the 96-wide convs: two 16-bit pieces per operand, "
                  "3 piece products (f16 fwd: 2^-22 operand rounding, bf16 bwd: 2^-18); every other conv kernel exact f32 MFMA",
             everything_else="f32")
    if swin:
        d["swin_gemm"] = ("C = 16 and C = 32: two 16-bit pieces per operand, all 4 piece products, f32 accumulate (f16 pieces fwd: 2^-22 "
                          "operand rounding, attention scores at head_dim 8 and P.V included; bf16 pieces bwd: 2^-18, its attention "
                          "scores on the f32 MFMA)")
    return d


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    return "unknown"


def cpu_baseline(cfg, kind, model, T, H, W):
    """The oracle (CPU restatement, pinned to the reference by tests/golden) on ONE clip of the same workload: forward +
    losses + backward + Adam, 3 warm-up + 5 timed steps, median; plus the eval-mode forward, median of
    3.  Workloads too large for one clip on the host (config 4) are timed on a centred crop and scaled by area."""
    import torch
    from oracle import idee_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    ocfg = O.default_cfg(encoder=cfg.encoder, in_channels_dynamic=cfg.in_channels_dynamic, in_channels=cfg.in_channels,
                         en_embed_dim=list(cfg.en_embed_dim), en_depths=list(cfg.en_depths), codebook_dim=cfg.codebook_dim,
                         cls_dim=cfg.cls_dim)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, v in sd.items() if v.is_floating_point()]
    scale, Hc, Wc = 1.0, H, W
    if H * W * T * cfg.en_embed_dim[-1] > 200 * 200 * 8 * 16 * 1.5:
        Hc, Wc = 104, 104
        scale = (Hc * Wc) / float(H * W)
    x, masks = synthetic_batch(cfg, kind, 1, T, Hc, Wc, 0, "cpu")
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    times = []
    for step in range(1, 9):
        t0 = time.perf_counter()
        for k in names:
            sd[k] = sd[k].detach().requires_grad_()
        outs = O.vq_model_forward(x, sd, ocfg, training=True)
        if kind == "era5":
            if outs[0].dim() == 5:      # T=16: two temporal output maps; score both, averaged
                T3 = outs[0].shape[2]
                total = sum(O.train_losses_era5((outs[0][:, :, i], [p[:, :, i] for p in outs[1]], outs[2], outs[3], outs[4]),
                                                *masks, sd, ocfg)[0] for i in range(T3)) / T3
            else:
                total, _ = O.train_losses_era5(outs, *masks, sd, ocfg)
        else:
            total, _ = O.train_losses_synthetic(outs, *masks, sd, ocfg)
        total.sum().backward()
        with torch.no_grad():
            for k in names:
                p, m, v = O.adam_step(sd[k].detach(), sd[k].grad, state[k][0], state[k][1], step, cfg.lr, cfg.beta1,
                                      cfg.beta2, 1e-8, cfg.weight_decay)
                state[k] = (m, v)
                sd[k] = p
        times.append(time.perf_counter() - t0)
    t = statistics.median(times[3:])
    ft = []
    with torch.no_grad():
        for _ in range(3):
            t0 = time.perf_counter()
            O.vq_model_forward(x, sd, ocfg, training=False)
            ft.append(time.perf_counter() - t0)
    crop = "" if scale == 1.0 else f" on a {Hc}x{Wc} crop, rate scaled by area ({scale:.4f})"
    return dict(value=round(scale / t, 4), unit="samples/s", cores=cores, kind="port", cpu=cpu_model(),
                fwd_ms_per_clip=round(statistics.median(ft) / scale * 1e3, 1),
                sample=f"1 clip V={cfg.in_channels_dynamic} T={T} {H}x{W}{crop}: oracle fwd+losses+bwd+Adam, 3 warm-up + 5 timed steps, "
                       f"median; eval forward median of 3")


DTYPE_TEXT = {
    "f32": "f32",
    "fwd22": "f32 storage/accumulate; forward convolutions and Swin linear layers on two-piece f16 split MFMAs (operand rounding "
             "2^-22), every backward pass exact f32",
    "f22": "f32 storage and accumulation; GEMM operands as two f16 pieces (operand rounding 2^-22: weights prescaled by exact powers of "
           "two per tensor, gradients per tile, convolution inputs under a range guard) in the Swin block and the 16->16 / 96-wide / "
           "(1,3,3)-forward convolutions; the backward's attention scores, the (1,3,3) data gradient and the Cout = 1 convolutions "
           "exact f32 (see `math`)",
    "split": "f32 storage/accumulate; convolutions and Swin linear layers on two-piece f16 (forward) / bf16 (backward) split MFMAs",
}

TIMED = ["idee_conv_fwd", "idee_conv_dgrad", "idee_conv_dgrad_add", "idee_conv_wgrad", "idee_swin_block_fwd", "idee_swin_block_bwd",
         "idee_lfq_fwd", "idee_lfq_bwd", "idee_patch_embed_ln_fwd", "idee_patch_embed_ln_bwd", "idee_anomaly_l1_fwd", "idee_anomaly_l1_signs_fwd",
         "idee_anomaly_l1_signs_bwd", "idee_anomaly_l1_bwd", "idee_adam_step", "idee_ln_relu_res_fwd", "idee_ln_relu_res_bwd"]

PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (v_mfma_f32_16x16x32_{f16,bf16})


def piece_products(name, shape, math):
    """How many 16-bit piece products per multiply-add the kernels of this launch execute on the f16 / bf16 MFMA under `math`
    (DESIGN.md §3d), or 0 when the launch runs on the exact-f32 MFMA / is a streaming kernel.  `shape`: ConvDesc or SwinGeom."""
    if math == "f32":
        return 0.0
    if name.startswith("idee_swin_block_"):
        if getattr(shape, "drop", 0.0) > 0 or getattr(shape, "attn_drop", 0.0) > 0:
            return 0.0                                   # element-wise dropout: the exact kernels under any option
        bwd = name.endswith("bwd")
        if math == "fwd22" and bwd:
            return 0.0
        nw, C = shape.wd * shape.wh * shape.ww, shape.C
        lin, att = 24.0 * C * C, 4.0 * nw * C
        # linear layers: all 4 piece products; attention scores / P.V and (backward) the token-reduction GEMMs: 3
        return (4 * lin + 3 * att) / (lin + att) if not bwd else 3.5
    if name.startswith("idee_conv_"):
        if shape.Cout == 1 or shape.Cin == 1:
            return 0.0                                   # streaming dot products / generic kernel
        if shape.mode == 2 and name.startswith("idee_conv_dgrad"):
            return 0.0                                   # (1,3,3) data gradient: the exact K-split kernel
        if math == "fwd22" and not name.endswith("fwd"):
            return 0.0
        return 3.0
    return 0.0


def family_table(log):
    """event log -> (per (entry point, shape): [ms, launches, flops/launch, bytes/launch, piece-product flops/launch],
    per entry point: [ms, launches, flops, bytes, piece-product flops] summed over the timed region)."""
    import ctypes
    from idee_amd import _lib
    per, fam = {}, {}
    math = _lib.get_option("IDEE_SWIN_MATH") or "f32"
    cmath = _lib.get_option("IDEE_CONV_MATH") or "f32"
    for name, evs in log.items():
        name = "idee_conv_dgrad" if name == "idee_conv_dgrad_add" else name      # the same kernels, out = dgrad + addend
        for e0, e1, args in evs:
            ms = e0.elapsed_time(e1)
            key, flops, nbytes, pflops = name, 0.0, 0.0, 0.0
            if name.startswith("idee_conv_"):
                d = ctypes.cast(args[0], ctypes.POINTER(_lib.ConvDesc)).contents
                key = f"{name}[{('k333', 'k233', 'k133')[d.mode]} {d.Cin}->{d.Cout} B{d.B} T{d.T}]"
                flops, nbytes = conv_flops(d), conv_bytes(d)
                pflops = flops * (piece_products(name, d, cmath) if d.act_dtype == 0 else 1.0)
            elif name.startswith("idee_swin_block_"):
                g = ctypes.cast(args[-2], ctypes.POINTER(_lib.SwinGeom)).contents
                key = f"{name}[w{g.wd}{g.wh}{g.ww} s{g.sd}{g.sh}{g.sw} C{g.C}]"
                bwd = name.endswith("bwd")
                flops, nbytes = swin_flops(g, bwd), swin_bytes(g, bwd)
                # executed GEMM work of the backward = forward recompute + data gradients + weight gradients = 3 x forward (credited: 2 x)
                pflops = (1.5 if bwd else 1.0) * flops * (piece_products(name, g, math) if g.act_dtype == 0 else 1.0)
            r = per.setdefault(key, [0.0, 0, 0.0, 0.0, 0.0])
            r[0] += ms; r[1] += 1; r[2] = flops; r[3] = nbytes; r[4] = pflops
            f = fam.setdefault(name, [0.0, 0, 0.0, 0.0, 0.0])
            f[0] += ms; f[1] += 1; f[2] += flops; f[3] += nbytes; f[4] += pflops
    return per, fam


def print_breakdown(per, fam, steps, dt):
    rows = sorted(per.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for v in per.values())
    sys.stderr.write(f"{'entry point':66s} {'ms/step':>9s} {'calls':>6s} {'avg ms':>8s} {'TFLOP/s':>8s} {'alg GB/s':>9s}\n")
    for k, (ms, n, fl, by, _pf) in rows:
        sys.stderr.write(f"{k:66s} {ms / steps:9.3f} {n // steps:6d} {ms / n:8.3f} {fl / (ms / n) / 1e9 if fl else 0:8.2f} "
                         f"{by / (ms / n) / 1e6 if by else 0:9.1f}\n")
    sys.stderr.write("-- families --\n")
    for k, (ms, n, fl, by, _pf) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        sys.stderr.write(f"{k:66s} {ms / steps:9.3f} {n // steps:6d} {ms / n:8.3f} {fl / ms / 1e9 if fl else 0:8.2f} "
                         f"{by / ms / 1e6 if by else 0:9.1f}\n")
    sys.stderr.write(f"sum of bracketed entry points {tot / steps:.3f} ms/step; wall {dt / steps * 1e3:.3f} ms/step\n")


def roofline_of(fam, math, dtype, steps, dt, headline, a):
    """`roofline` of the dominant kernel FAMILY (all launches of the entry point with the largest total time in the timed
    region; duration from the HIP events recorded on the launch stream), priced against the ceiling of the pipe that ran:

      exact f32 (--math f32):  reference-formulation FLOPs against the f32 MFMA peak, 157.3 TFLOP/s
                               (v_mfma_f32_16x16x4_f32 runs at the f32 vector rate) — bound "mfma";
      f16 pieces (--math f22 / fwd22 / split) and bf16 activations:  BOTH ceilings of those kernels — algorithmic HBM bytes
                               (unit inputs + outputs) per second against 8 TB/s, and the EXECUTED piece-product
                               FLOPs (reference FLOPs x piece products per multiply-add x 1.5 for the Swin backward's recompute)
                               against the dense f16 / bf16 MFMA peak, 2.5 PFLOP/s — `bound` / `frac` are those of the ceiling the
                               family sits closer to, the other one is printed beside it.  No line can print a fraction >= 1.

    A launch of the dominant family that stays on the exact-f32 MFMA under an f16-piece option (dropout blocks, the (1,3,3) data
    gradient) counts with its f32 FLOPs against the f32 peak only in an f32 run; under f22 its time is in the family's time and
    it adds no piece-product FLOPs."""
    cand = sorted(((k, v) for k, v in fam.items() if v[2] > 0), key=lambda kv: -kv[1][0])
    if not cand:
        return None
    k, (ms, n, fl, by, pf) = cand[0]
    lps = n // steps
    common = dict(kernel=k + " (all launches of this entry point in the step)", avg_launch_ms=round(ms / n, 4), launches_per_step=lps,
                  flops_per_launch=fl / n, bytes_per_launch=by / n, share_of_step=round(ms / steps / (dt / steps * 1e3), 4))
    if dtype == "bf16":
        traffic, src = pmc_traffic(k, lps, a.config == "4" and a.hw is None and a.batch is None, PMC_SUMMARY_BF16)
    else:
        traffic, src = pmc_traffic(k, lps, headline and math in ("f32", "f22"), PMC_SUMMARY_F22 if math == "f22" else PMC_SUMMARY)
    gbs, tf_eq = by / ms / 1e6, fl / ms / 1e9
    if dtype == "f32" and (math == "f32" or pf == 0):      # (pf == 0: every launch of the family ran on the exact-f32 MFMA)
        return dict(common, bound="mfma", achieved=round(tf_eq, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(tf_eq / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic, traffic_source=src,
                    pipe="v_mfma_f32_16x16x4_f32 (f32 MFMA = f32 vector rate; it does not co-execute with the VALU stream, so "
                         "LayerNorm / softmax / GELU time adds to the MFMA time in the fused Swin kernels)",
                    algorithmic_gbs=round(gbs, 1))
    ptf = pf / ms / 1e9
    hbm = dict(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
    mf = dict(achieved=round(ptf, 1), peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ptf / PEAK_F16_MFMA_TFLOPS, 4),
              note="executed 16-bit piece-product FLOPs on v_mfma_f32_16x16x32_{f16,bf16}")
    lead, other = (("hbm", hbm), ("mfma", mf)) if hbm["frac"] >= mf["frac"] else (("mfma", mf), ("hbm", hbm))
    return dict(common, bound=lead[0], achieved=lead[1]["achieved"], peak=lead[1]["peak"], unit=lead[1]["unit"], frac=lead[1]["frac"],
                traffic=traffic, traffic_source=src, other_ceiling=dict(bound=other[0], **other[1]),
                f32_equivalent_tflops=round(tf_eq, 2))


def main():
    a = parse()
    if a.dtype == "bf16":
        a.math = "f32"          # (the bf16 activation path has its own kernels; the f32-path switches are pinned to their defaults)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import torch
    from idee_amd import _lib, parallel
    # the library looks these options up at every launch.  BOTH are pinned (idee_set_option, in every rank process) to the
    # chosen arithmetic whatever the caller's environment holds, so the printed `dtype` / `math` always describe what ran.
    math_opt = a.math if a.math in ("split", "fwd22", "f22") else "f32"
    _lib.set_option("IDEE_CONV_MATH", math_opt)
    _lib.set_option("IDEE_SWIN_MATH", math_opt)
    from idee_amd.config import default_config
    from idee_amd.train import SyntheticTrainStep, ERA5TrainStep
    rank, world, local = parallel.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launch environment has WORLD_SIZE={world}; refusing to "
                         f"report a line for a different rank count")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.check(_lib.load().idee_check_device(local), "idee_check_device")
    enc, H, W, T, Cin, batch, over, kind, label = CONFIGS[a.config]
    enc = a.encoder or enc
    batch = a.batch or batch
    if a.hw:
        H = W = a.hw
    cfg = default_config(encoder=enc, in_channels=Cin, act_dtype=a.dtype, codebook_impl=a.codebook, **over)
    headline = (a.config == "2" and a.hw is None and batch == 4 and enc == "Swin_3D")
    torch.manual_seed(0)
    ts = (ERA5TrainStep if kind == "era5" else SyntheticTrainStep)(cfg, device=dev)
    x, masks = synthetic_batch(cfg, kind, batch, T, H, W, rank, dev)

    ranks_report = None
    if world > 1:      # what the communicator itself reports: (rank, device) of every member
        mine = torch.tensor([torch.distributed.get_rank(), local], device=dev, dtype=torch.int64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        ranks_report = [t.tolist() for t in allr]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_region(steps, warmup):
        """W untimed warm-up steps, then EXACTLY `steps` steps between two barriers (+ device synchronisation on both sides); every
        launch of the entry points in TIMED is bracketed with HIP events on the launch stream.  -> (seconds [max over ranks],
        per-step event intervals in ms, the event log)."""
        for _ in range(warmup):
            ts.step(x, *masks)
        log = _lib.set_event_log(TIMED)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            ts.step(x, *masks)
        marks[steps].record()
        barrier()
        dt = time.perf_counter() - t0
        _lib.set_event_log(None)
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt)
        return dt, step_ms, log

    dt, step_ms, log = timed_region(a.steps, a.warmup)
    per, fam = family_table(log)
    if a.breakdown and rank == 0:
        print_breakdown(per, fam, a.steps, dt)
    roof = roofline_of(fam, a.math, a.dtype, a.steps, dt, headline, a)

    # ---- second half of BASELINE.json's metric: forward ms per clip (eval mode, no autograd), outside the timed region ----
    fwd_ms_clip = float("nan")
    if not a.no_fwd_eval:
        ts.model.eval()
        with torch.no_grad():
            for _ in range(2):
                ts.model(x)
            barrier()
            f0 = time.perf_counter()
            for _ in range(5):
                ts.model(x)
            barrier()
            fwd_ms_clip = (time.perf_counter() - f0) / 5 / batch * 1e3
        ts.model.train()

    # ---- the OTHER arithmetic of the same workload, measured in this run and printed beside the headline: the headline
    # (`value`, `ms_per_step`, `roofline`) is the reference's arithmetic (exact f32, --math f32 = the default); the f16-piece
    # line (`f22`: operand rounding 2^-22, narrower than f32) is a labelled side object with its own roofline.
    # With --math f22 it is the other way round (`exact_f32` beside an f22 `value`).
    side_key, side = None, None
    if a.dtype == "f32" and a.math in ("f32", "f22") and world == 1 and not a.no_side_line and enc in ("Swin_3D", "CNN_3D"):
        other = "f22" if a.math == "f32" else "f32"
        _lib.set_option("IDEE_CONV_MATH", other)      # looked up by the library at every launch
        _lib.set_option("IDEE_SWIN_MATH", other)
        sdt, sstep_ms, slog = timed_region(a.steps, min(a.warmup, 5))
        sper, sfam = family_table(slog)
        if a.breakdown and rank == 0:
            sys.stderr.write(f"==== side line: --math {other} ====\n")
            print_breakdown(sper, sfam, a.steps, sdt)
        sroof = roofline_of(sfam, other, a.dtype, a.steps, sdt, headline, a)
        _lib.set_option("IDEE_CONV_MATH", math_opt)
        _lib.set_option("IDEE_SWIN_MATH", math_opt)
        side_key = "f22" if other == "f22" else "exact_f32"
        side = dict(value=round(batch * a.steps / sdt, 3), unit="samples/s", ms_per_step=round(sdt / a.steps * 1e3, 3),
                    ms_per_step_median=round(statistics.median(sstep_ms), 3), steps=a.steps,
                    dtype="f32" if other == "f32" else DTYPE_TEXT["f22"], math=math_label(other, cfg, a.dtype), roofline=sroof,
                    note=f"same process, model, batch and step count as `value`, IDEE_SWIN_MATH = IDEE_CONV_MATH = {other}, timed after the "
                         "headline region with the same event bracketing" +
                         ("; NOT the reference's arithmetic (two f16 pieces per GEMM operand: 22-23 of f32's 24 operand bits) — a side "
                          "line, never `value`" if other == "f22" else "; the library's default arithmetic"))

    if rank != 0:
        return
    out = dict(metric="train samples/sec (TxHxW clips); one step = fwd+losses+bwd+Adam", value=round(world * batch * a.steps / dt, 3),
               unit="samples/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3),
               ms_per_step_median=round(statistics.median(step_ms), 3), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype=("bf16" if a.dtype == "bf16" else DTYPE_TEXT[a.math]),
               math=math_label(a.math, cfg, a.dtype),
               codebook=a.codebook,
               data="synthetic",
               config=dict(workload=f"#{a.config}: {label}, V={cfg.in_channels_dynamic}, batch {batch}/GPU" +
                                    (f", grid overridden to {H}x{W}" if a.hw else ""), global_batch=world * batch,
                           parallelism=f"dp{world}"),
               roofline=roof, fwd_ms_per_clip=None if a.no_fwd_eval else round(fwd_ms_clip, 3), peak_hbm_alloc_gb=round(torch.cuda.max_memory_allocated() / 1e9, 2))
    if side is not None:
        out[side_key] = side
    if world > 1:
        out["backend"] = torch.distributed.get_backend()
        out["ranks"] = ranks_report
    if world == 1 and not a.no_cpu_baseline and a.codebook == "VQ":
        out["cpu_baseline"] = dict(value=None, unit="samples/s", cores=host_cores(), kind="port",
                                   sample="not timed: the oracle's model forward wires LFQ as the reference's build.py does; see the LFQ line")
    elif world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, kind, ts.model, T, H, W)
        except Exception as e:  # noqa: BLE001  (the GPU measurement above must still be reported)
            out["cpu_baseline"] = dict(value=None, unit="samples/s", cores=host_cores(), kind="port", sample=f"failed: {e!r}")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        if "torch" in sys.modules:
            import torch
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
