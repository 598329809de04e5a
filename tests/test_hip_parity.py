"""GPU parity: the HIP path (through the C-ABI, via idee_amd.ops / idee_amd.models) against
(a) the golden vectors generated from the real reference and (b) the CPU oracle on seeded
random inputs at sizes that exercise multi-tile paths.  Run with `-m gpu` on an MI355X.

Tolerances (fp32 path): activations 2e-5 abs + 2e-4 rel; gradients 1e-3 rel with an absolute
floor of 1e-4 x the largest reference magnitude of that tensor (atomically accumulated sums);
`anomaly` indices bit-exact except tokens whose projection |s| < 1e-5 (listed, not averaged).
Gradients behind a ReLU: the oracle's pre-activations are recorded and a case is used only when none
of them is a tie (conftest.tie_free_case walks seeds and prints what it inspected), so every gradient
is compared at the strict tolerance — there is no outlier budget anywhere in this file."""
import math
import os

import pytest
import torch

from conftest import load_golden, relu_record, tie_free_case, tie_relu, interval_close, variants

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _L():
    from idee_amd import _lib
    return _lib


def _ops():
    from idee_amd import ops
    return ops


def close(name, got, ref, atol=2e-5, rtol=2e-4):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    fin = torch.isfinite(ref)
    same_nonfinite = torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.equal(got[~fin & ~torch.isnan(ref)], ref[~fin & ~torch.isnan(ref)])
    err = torch.where(fin, (got - ref).abs(), torch.zeros_like(ref))
    tol = atol + rtol * torch.where(fin, ref.abs(), torch.zeros_like(ref))
    bad = (err > tol) | (fin & ~torch.isfinite(got))
    if bad.any() or not same_nonfinite:
        i = int(err.argmax())
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} out of tol (non-finite pattern equal: {same_nonfinite}); "
                             f"max abs err {float(err.max()):.3e} (got {float(got.flatten()[i]):.6g} ref "
                             f"{float(ref.flatten()[i]):.6g}), ref max {float(ref[fin].abs().max()) if fin.any() else float('nan'):.3e}")


def gclose(name, got, ref, rtol=1e-3, floor=1e-4):
    fin = torch.isfinite(ref)
    close(name, got, ref, atol=max(1e-6, floor * float(ref[fin].abs().max()) if fin.any() else 1e-6), rtol=rtol)


def dev(t):
    return t.to(DEV)


# ---------------------------------------------------------------------------------------------
def test_library_loaded_and_device():
    from idee_amd import _lib
    lib = _lib.load()
    assert lib.idee_check_device(torch.cuda.current_device()) == 0, _lib.last_error()


@pytest.mark.parametrize("tag", ["swin_c1", "swin_c2", "cnn_c1"])
def test_patch_embed(tag):
    ops = _ops()
    a, sd, grads = load_golden("patch_embed_" + tag)
    x = dev(a["x"]).unsqueeze(1)                                   # (N,1,Cin,T,H,W)
    w = dev(sd["proj.weight"]).flatten(1).unsqueeze(0).requires_grad_()   # (1,E,Cin)
    b = dev(sd["proj.bias"]).unsqueeze(0).requires_grad_() if "proj.bias" in sd else None
    y = ops.PatchEmbedLN.apply(x, w, b)                             # (N,1,T,H,W,E)
    close("y", y[:, 0].permute(0, 4, 1, 2, 3), a["y"])
    (y[:, 0].permute(0, 4, 1, 2, 3) * dev(a["r"])).sum().backward()
    gclose("dw", w.grad[0], grads["proj.weight"].flatten(1))
    if b is not None:
        gclose("db", b.grad[0], grads["proj.bias"])


def _swin_params(sd, V=1):
    names = ["attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "mlp.fc1.weight", "mlp.fc1.bias",
             "mlp.fc2.weight", "mlp.fc2.bias", "attn.relative_position_bias_table"]
    return [dev(sd[n]).unsqueeze(0).contiguous().requires_grad_() for n in names], names


@pytest.mark.parametrize("tag", ["c16_even", "c16_odd", "c16_d1", "c32_even"])
def test_patch_merging_golden(tag):
    from idee_amd.models.encoder.Swin_3D import PatchMerging
    a, sd, grads = load_golden("patch_merging_" + tag)
    C = a["x"].shape[-1]
    pm = PatchMerging(C)
    pm.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    pm = pm.to(DEV)
    x = dev(a["x"]).requires_grad_()
    y = pm(x)
    close("y", y, a["y"])
    (y * dev(a["r"])).sum().backward()
    gclose("dx", x.grad, a["dx"])
    for k, p in pm.named_parameters():
        gclose("d" + k, p.grad, grads[k])


@pytest.mark.parametrize("tag", ["w244_noshift", "w244_shift", "w244_shift_pad", "w244_noshift_pad", "w811",
                                 "w811_T16_shift", "w244_c32"])
@variants('default', 'swin_split', 'swin_f22')
def test_swin_block_golden(tag, kernel_variant):
    ops = _ops()
    a, sd, grads = load_golden("swin_block_" + tag)
    params, names = _swin_params(sd)
    x = dev(a["x"]).unsqueeze(1).contiguous().requires_grad_()      # (B,1,D,H,W,C)
    y = ops.SwinBlock.apply(x, *params, dev(sd["attn.relative_position_index"]), int(a["heads"]),
                            tuple(a["window"].tolist()), tuple(a["shift"].tolist()), None)
    close("y", y[:, 0], a["y"])
    (y[:, 0] * dev(a["r"])).sum().backward()
    gclose("dx", x.grad[:, 0], a["dx"])
    for p, n in zip(params, names):
        gclose("d" + n, p.grad[0], grads[n])


@pytest.mark.parametrize("profile", ["loss_scale", "ragged_rows", "ragged_windows", "huge_x"])
def test_swin_f22_error_against_float64(profile, set_opt):
    """IDEE_SWIN_MATH=f22 (every GEMM of the block on two f16 pieces per operand, weights prescaled by powers of two, the
    incoming gradient scaled per 32-token work item) measured against a FLOAT64 evaluation of the oracle, beside the exact-f32
    kernels and the two-piece bf16 backward (IDEE_SWIN_MATH=split) on the same inputs.  Weights at the reference's
    initialisation scale (trunc_normal std 0.02, Swin_3D.py:117, :533 — f16 lo pieces of unscaled weights would be subnormal);
    upstream gradient `loss_scale`: 1e-7 (a mean-reduced loss); `ragged_rows`: per-(d,h)-row magnitudes over 15 decades with
    all-zero frames in between (scale changes in both directions inside a wave, zero tiles, the +40 clamp); `ragged_windows`:
    one magnitude per (batch item, pair of frames of a shifted window) over 12 decades, one pair all zero — every work item
    of both backward kernels is then homogeneous, so each frame's gradient must come out at f32 level RELATIVE TO ITS OWN
    MAGNITUDE (what a per-launch scale could not give).  `huge_x`: block inputs of magnitude 1e5 with rows of 1e-3 between them
    (every GEMM operand of the block is an image of a LayerNorm output under the block's own weights, so no activation operand
    can leave f16's range whatever the input's magnitude: the f16 pieces need no activation guard here, unlike the convolutions).
    Asserted: relative L2 error of y, dx and every parameter gradient within 4x the exact-f32 kernels' own error (+ 2e-7);
    for `loss_scale` / `ragged_windows` also the branch gradient dx - dy of every frame (n, d) against that frame's largest
    |dy|: within 4x the exact kernels' worst frame (+ 1e-7)."""
    from oracle import idee_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed({"loss_scale": 5, "ragged_rows": 11, "ragged_windows": 13, "huge_x": 17}[profile])
    N, D, H, W, C, NH = 2, 8, 24, 40, 16, 2
    window, shift = (2, 4, 4), (1, 2, 2)
    x = torch.randn((N, 1, D, H, W, C), generator=g) * 1.5
    r = torch.randn((N, 1, D, H, W, C), generator=g)
    if profile == "huge_x":
        x = x * torch.where(torch.rand((N, 1, D, H, 1, 1), generator=g) < 0.7, 1e5, 1e-3)
    if profile in ("loss_scale", "huge_x"):
        r = r * 1e-7
    elif profile == "ragged_rows":
        mag = 10.0 ** (torch.rand((N, 1, D, H, 1, 1), generator=g) * 15.0 - 14.0)
        r = r * mag
        r[:, :, 3:5] = 0.0                                           # two all-zero frames
    else:
        # shifted (2,4,4) windows at shift 1 in d pair the frames {1,2},{3,4},{5,6},{7,0}
        grp = ((torch.arange(D) - 1) % D) // 2
        mag = 10.0 ** (torch.rand((N, D // 2), generator=g) * 12.0 - 11.0)
        mag[0, 1] = 0.0                                               # frames 3, 4 of item 0: all zero
        r = r * mag[:, grp].reshape(N, 1, D, 1, 1, 1)
    shapes = {"attn.qkv.weight": (48, 16), "attn.qkv.bias": (48,), "attn.proj.weight": (16, 16), "attn.proj.bias": (16,),
              "mlp.fc1.weight": (64, 16), "mlp.fc1.bias": (64,), "mlp.fc2.weight": (16, 64), "mlp.fc2.bias": (16,),
              "attn.relative_position_bias_table": (147, 2)}
    sd = {k: torch.randn(sh, generator=g) * 0.02 for k, sh in shapes.items()}
    index = O.relative_position_index(window)
    # float64 reference
    sd64 = {k: v.double().requires_grad_() for k, v in sd.items()}
    sd64["attn.relative_position_index"] = index
    x64 = x[:, 0].double().requires_grad_()
    y64 = O.swin_block(x64, sd64, "", NH, window, shift)
    assert y64.dtype == torch.float64
    (y64 * r[:, 0].double()).sum().backward()
    ref = {"y": y64.detach(), "dx": x64.grad}
    ref.update({"d" + k: sd64[k].grad for k in shapes})
    names = list(shapes)

    def run(mode):
        set_opt("IDEE_SWIN_MATH", mode)
        params = [sd[n].unsqueeze(0).to(DEV).requires_grad_() for n in names]
        xd = x.to(DEV).requires_grad_()
        y = ops.SwinBlock.apply(xd, *params, index.to(DEV), NH, window, shift, None)
        (y * r.to(DEV)).sum().backward()
        out = {"y": y[:, 0].detach().cpu().double(), "dx": xd.grad[:, 0].cpu().double()}
        out.update({"d" + n: p.grad[0].cpu().double() for p, n in zip(params, names)})
        assert all(bool(torch.isfinite(v).all()) for v in out.values()), mode
        return out

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-300))

    def frame_err(out):    # branch gradient (dx - dy) of each frame (n, d): largest error, and the frame's largest |dy|
        rr = r[:, 0].double()
        e = (out["dx"] - ref["dx"]).abs().amax(dim=(-1, -2, -3))
        return e, rr.abs().amax(dim=(-1, -2, -3))

    res = {m: run(m) for m in ("f32", "f22", "split")}
    print(f"\n[f22 vs float64, profile {profile}] relative L2 error: tensor  exact-f32  f22  bf16x2-split")
    for k in ref:
        e = {m: rel(res[m][k], ref[k]) for m in res}
        print(f"  {k:38s} {e['f32']:.2e}  {e['f22']:.2e}  {e['split']:.2e}")
        assert e["f22"] <= 4.0 * e["f32"] + 2e-7, (k, e)
    if profile in ("ragged_rows", "huge_x"):
        return
    e32, scale = frame_err(res["f32"])
    e22, _ = frame_err(res["f22"])
    live = scale > 0
    if (~live).any():                                               # an all-zero window pair has an exactly zero gradient
        assert float(e22[~live].max()) == 0.0
    q32, q22 = (e32[live] / scale[live]), (e22[live] / scale[live])
    print(f"  dx error per frame (n, d) / frame max|dy|: exact-f32 max {float(q32.max()):.2e}, f22 max {float(q22.max()):.2e} "
          f"({int(live.sum())} frames, dy magnitudes {float(scale[live].min()):.1e} .. {float(scale[live].max()):.1e})")
    assert float(q22.max()) <= 4.0 * float(q32.max()) + 1e-7, (float(q22.max()), float(q32.max()))


@variants('default', 'swin_split', 'swin_f22')
def test_swin_block_two_variables_vs_oracle(kernel_variant):
    """V = 2 weight sets, N = 2, a grid that is not a multiple of the window and spans many
    window groups (exercises b % V weight selection and the grid-stride loops)."""
    from oracle import idee_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    N, V, D, H, W, C, NH = 2, 2, 8, 22, 50, 16, 2
    x = torch.randn((N, V, D, H, W, C), generator=g)
    r = torch.randn((N, V, D, H, W, C), generator=g)
    shapes = {"attn.qkv.weight": (48, 16), "attn.qkv.bias": (48,), "attn.proj.weight": (16, 16), "attn.proj.bias": (16,),
              "mlp.fc1.weight": (64, 16), "mlp.fc1.bias": (64,), "mlp.fc2.weight": (16, 64), "mlp.fc2.bias": (16,),
              "attn.relative_position_bias_table": (147, 2)}
    sds = [{k: (torch.randn(s, generator=g) * (0.3 if len(s) > 1 else 0.1)).requires_grad_() for k, s in shapes.items()} for _ in range(V)]
    index = O.relative_position_index((2, 4, 4))
    xr = x.clone().requires_grad_()
    outs = []
    for v in range(V):
        sdv = dict(sds[v]); sdv["attn.relative_position_index"] = index
        outs.append(O.swin_block(xr[:, v], sdv, "", NH, (2, 4, 4), (1, 2, 2)))
    yo = torch.stack(outs, 1)
    (yo * r).sum().backward()
    names = list(shapes)
    params = [torch.stack([sds[v][n].detach() for v in range(V)]).to(DEV).requires_grad_() for n in names]
    xd = x.to(DEV).requires_grad_()
    y = ops.SwinBlock.apply(xd, *params, index.to(DEV), NH, (2, 4, 4), (1, 2, 2), None)
    close("y", y, yo)
    (y * r.to(DEV)).sum().backward()
    gclose("dx", xd.grad, xr.grad)
    for p, n in zip(params, names):
        gclose("d" + n, p.grad, torch.stack([sds[v][n].grad for v in range(V)]))


@variants('default', 'conv_split', 'conv_f22', 'conv16_regs', 'wgrad16_regs')
def test_proj_var_golden(kernel_variant):
    ops = _ops()
    a, sd, grads = load_golden("proj_var")
    x = dev(a["x"]).permute(0, 2, 3, 4, 1).unsqueeze(1).contiguous().requires_grad_()   # (N,1,T,H,W,C)
    ps = [dev(sd[k]).unsqueeze(0).contiguous().requires_grad_() for k in ("0.weight", "0.bias", "2.weight", "2.bias")]
    y = ops.ProjVar.apply(x, *ps)
    close("y", y[:, 0].permute(0, 4, 1, 2, 3), a["y"])
    (y[:, 0].permute(0, 4, 1, 2, 3) * dev(a["r"])).sum().backward()
    gclose("dx", x.grad[:, 0].permute(0, 4, 1, 2, 3), a["dx"])
    for p, k in zip(ps, ("0.weight", "0.bias", "2.weight", "2.bias")):
        gclose("d" + k, p.grad[0], grads[k])


@pytest.mark.parametrize("tag,cin", [("1_16", 1), ("16_16", 16)])
@variants('default', 'conv_split', 'conv_f22', 'wgrad16_regs')
def test_conv_block_golden(tag, cin, kernel_variant):
    """CNN_3D encoder block: [PatchEmbed+LN] -> 2 x (x + ReLU(LN_affine(conv3^3 replicate(x))))."""
    ops = _ops()
    a, sd, grads = load_golden("conv_block_" + tag)
    leaf = lambda t: dev(t).unsqueeze(0).contiguous().requires_grad_()
    ps = {k: leaf(sd[k]) for k in ("conv1.weight", "norm1.weight", "norm1.bias", "conv2.weight", "norm2.weight", "norm2.bias")}
    if cin == 16:
        x = dev(a["x"]).permute(0, 2, 3, 4, 1).unsqueeze(1).contiguous().requires_grad_()
        t = x
    else:
        x = dev(a["x"]).unsqueeze(1).contiguous()
        ps["downsample.proj.weight"] = dev(sd["downsample.proj.weight"]).flatten(1).unsqueeze(0).contiguous().requires_grad_()
        t = ops.PatchEmbedLN.apply(x, ps["downsample.proj.weight"], None)
    t = ops.ConvLNUnit.apply(t, ps["conv1.weight"], ps["norm1.weight"], ps["norm1.bias"])
    t = ops.ConvLNUnit.apply(t, ps["conv2.weight"], ps["norm2.weight"], ps["norm2.bias"])
    y = t[:, 0].permute(0, 4, 1, 2, 3)
    close("y", y, a["y"])
    (y * dev(a["r"])).sum().backward()
    from oracle import idee_oracle as O
    with relu_record() as rec:                      # the fixture generator asserts this too (make_fixtures.fx_conv_block)
        O.conv_block(a["x"], sd, "", cin, 16)
    assert rec.ties() == 0, "fixture has a ReLU tie"
    if cin == 16:
        gclose("dx", x.grad[:, 0].permute(0, 4, 1, 2, 3), a["dx"])
    for k, p in ps.items():
        gref = grads[k].flatten(1) if k == "downsample.proj.weight" else grads[k]
        gclose("d" + k, p.grad[0], gref)


@pytest.mark.parametrize("T,H,W", [(8, 9, 72), (3, 5, 200), (8, 6, 200), (1, 1, 1), (2, 2, 17), (3, 21, 40), (2, 3, 31), (2, 2, 2)])
@variants('default', 'conv_split', 'conv_f22', 'conv16_regs', 'wgrad16_regs')
def test_proj_var_vs_oracle(T, H, W, set_opt, kernel_variant):
    """Multi-tile widths (72 = 3+2 column blocks, 200 = 4+3+3+3 incl. both frame edges), thin extents
    (clamp-everything cases).  W = 31 (W % 16 == 15): the data gradient's right-hand virtual column falls into the NEXT
    16-column tile, so the edge columns take the atomic path instead of the in-wave fold; W = 2: the narrowest folded frame;
    W = 1: both virtual columns fold onto the one real column (atomics).  Strict gradient tolerance on a tie-free case."""
    from oracle import idee_oracle as O
    ops = _ops()
    N, V, C = 2, 2, 16
    if H == 21:      # rows walked in runs of 8, 8, 5: the strip kernels' run starts away from row 0 (the headline shape's 67-row
        set_opt("IDEE_CONV_HRUN", "8")    # runs are otherwise only reached by the full-size tests)

    def build(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((N, V, T, H, W, C), generator=g)
        r = torch.randn((N, V, T, H, W, C), generator=g)
        w1, w2 = (torch.randn((V, C, C, 3, 3, 3), generator=g) * 0.05 for _ in range(2))
        b1, b2 = (torch.randn((V, C), generator=g) * 0.1 for _ in range(2))
        leaves = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
        outs = []
        with relu_record() as rec:
            for v in range(V):
                sd = {"0.weight": leaves[1][v], "0.bias": leaves[2][v], "2.weight": leaves[3][v], "2.bias": leaves[4][v]}
                outs.append(O.proj_var(leaves[0][:, v].permute(0, 4, 1, 2, 3), sd, "").permute(0, 2, 3, 4, 1))
        return ((x, w1, b1, w2, b2), r, leaves, torch.stack(outs, 1)), rec

    (x, w1, b1, w2, b2), r, leaves, yo = tie_free_case(build, range(T * 1000 + W, T * 1000 + W + 8), f"proj_var {T}x{H}x{W}")
    (yo * r).sum().backward()
    dl = [t.to(DEV).requires_grad_() for t in (x, w1, b1, w2, b2)]
    y = ops.ProjVar.apply(*dl)
    close("y", y, yo)
    (y * r.to(DEV)).sum().backward()
    for n, a_, b_ in zip(("dx", "dw1", "db1", "dw2", "db2"), dl, leaves):
        gclose(n, a_.grad, b_.grad)


def _row_ties(rec, dims):
    """relu_record.ties() with the tie threshold relative to the mean |a| of each ROW (reduced over `dims`) instead of the whole
    tensor: for inputs whose rows differ by many decades a quiet row is not a tie just because another row is large."""
    from conftest import TIE_REL
    rec.ties = lambda: sum(int((t.abs() < TIE_REL * t.abs().mean(dim=dims, keepdim=True)).sum()) for t in rec.inputs)
    return rec


@pytest.mark.parametrize("profile", ["loss_scale", "ragged_rows", "huge_x", "ragged_x"])
def test_conv_f22_error_against_float64(profile, set_opt):
    """IDEE_CONV_MATH=f22 (3^3 replicate convs: forward, data gradient and weight gradient on two f16 pieces per operand; weights
    prescaled per set, dy scaled by a power of two per work unit / per row) against a FLOAT64 evaluation of the oracle's
    proj_var, beside the exact-f32 kernels and the bf16-piece backward (IDEE_CONV_MATH=split) on the same inputs.
    `loss_scale`: upstream gradient 1e-7; `ragged_rows`: one magnitude per (n, v, t, h) row over 12 decades, two all-zero
    frames (the data gradient's scale moves down inside a unit and rows are begun again; the weight gradient's accumulators
    are rescaled in both directions).  `huge_x`: activations of magnitude 3e5 (an unguarded f16 hi piece is inf from 65520 on:
    the forward and weight-gradient kernels' activation range guard, common.h ActGuard, must fire in every work unit);
    `ragged_x`: one activation magnitude per (n, v, t, h) row over 9 decades, 1e-2 .. 1e7, and one batch item at magnitude 1
    (the guard fires in the middle of units, moves down more than once, rows are staged again; units of the quiet item stay
    unguarded) — there the quiet item's outputs must also come out at f32 level relative to THEIR OWN magnitude.
    Asserted: relative L2 error of y, dx and the four parameter gradients within 4x the
    exact kernels' own error (+ 2e-7).  The case is ReLU-tie-free (float64 and f32 masks agree)."""
    from oracle import idee_oracle as O
    ops = _ops()
    N, V, C, T, H, W = 2, 2, 16, 4, 21, 72

    def build(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((N, V, T, H, W, C), generator=g)
        r = torch.randn((N, V, T, H, W, C), generator=g)
        if profile == "loss_scale":
            r = r * 1e-7
        elif profile == "ragged_rows":
            r = r * 10.0 ** (torch.rand((N, V, T, H, 1, 1), generator=g) * 12.0 - 11.0)
            r[0, 0, 1] = 0.0
            r[1, 1, 2] = 0.0
        elif profile == "huge_x":
            x = x * 3e5
        else:
            mag = 10.0 ** (torch.rand((N, V, T, H, 1, 1), generator=g) * 9.0 - 2.0)
            mag[1] = 1.0                          # item 1: every frame quiet (its units never see a large row)
            x = x * mag
        w1, w2 = (torch.randn((V, C, C, 3, 3, 3), generator=g) * 0.05 for _ in range(2))
        b1, b2 = (torch.randn((V, C), generator=g) * 0.1 for _ in range(2))
        leaves = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2)]
        outs = []
        with relu_record() as rec:
            for v in range(V):
                sd = {"0.weight": leaves[1][v], "0.bias": leaves[2][v], "2.weight": leaves[3][v], "2.bias": leaves[4][v]}
                outs.append(O.proj_var(leaves[0][:, v].permute(0, 4, 1, 2, 3), sd, "").permute(0, 2, 3, 4, 1))
        if profile == "ragged_x":
            _row_ties(rec, (1, 4))            # (N, C, T, H, W) pre-activations: one scale per (n, t, h) row
        return ((x, w1, b1, w2, b2), r, leaves, torch.stack(outs, 1)), rec

    (x, w1, b1, w2, b2), r, leaves, y64 = tie_free_case(build, range(500, 516), f"proj_var f64 {profile}")
    assert y64.dtype == torch.float64
    (y64 * r.double()).sum().backward()
    names = ("dx", "dw1", "db1", "dw2", "db2")
    ref = {"y": y64.detach()}
    ref.update({n: l.grad for n, l in zip(names, leaves)})

    def run(mode):
        set_opt("IDEE_CONV_MATH", mode)
        dl = [t.to(DEV).requires_grad_() for t in (x, w1, b1, w2, b2)]
        y = ops.ProjVar.apply(*dl)
        (y * r.to(DEV)).sum().backward()
        out = {"y": y.detach().cpu().double()}
        out.update({n: t.grad.cpu().double() for n, t in zip(names, dl)})
        assert all(bool(torch.isfinite(v).all()) for v in out.values()), mode
        return out

    res = {m: run(m) for m in ("f32", "f22", "split")}
    print(f"\n[conv f22 vs float64, profile {profile}] relative L2 error: tensor  exact-f32  f22  split (f16 fwd / bf16 bwd)")
    for k in ref:
        e = {m: float((res[m][k] - ref[k]).norm() / ref[k].norm()) for m in res}
        print(f"  {k:6s} {e['f32']:.2e}  {e['f22']:.2e}  {e['split']:.2e}")
        assert e["f22"] <= 4.0 * e["f32"] + 2e-7, (k, e)
    if profile == "ragged_x":       # the quiet item: its own relative error, not the batch's
        q = {m: float((res[m]["y"][1] - ref["y"][1]).norm() / ref["y"][1].norm()) for m in res}
        qd = {m: float((res[m]["dx"][1] - ref["dx"][1]).norm() / ref["dx"][1].norm()) for m in res}
        print(f"  quiet item alone: y {q['f32']:.2e} / {q['f22']:.2e}, dx {qd['f32']:.2e} / {qd['f22']:.2e} (exact / f22)")
        assert q["f22"] <= 4.0 * q["f32"] + 2e-7 and qd["f22"] <= 4.0 * qd["f32"] + 2e-7, (q, qd)


@pytest.mark.parametrize("shape", ["k333_16", "k233_96"])
def test_wgrad_f22_activation_guard_keeps_finite_dw(shape, set_opt):
    """IDEE_CONV_MATH=f22 weight gradients (wgrad16_shift_kernel, wgrad_wide_split_kernel) at the far end of the range: activations
    of magnitude 3e36 in ONE batch item (the activation guard of the waves that meet it drops to k ~ -110; the other item is O(1))
    under gradients of magnitude 1e-18 (gradient scale at its cap 2^63): dW ~ 1e20 is finite and must come out finite and within 4x
    the exact kernels' error (+ 2e-7) of a float64 reference.  A guard that is reset at unit boundaries multiplies the persistent
    accumulators by 2^110 while they still carry the gradient scale 2^63: inf."""
    from idee_amd import _lib as L
    ops = _ops()
    if shape == "k333_16":
        # 2 rows per work unit: ~4,600 units per weight set, several per wave, in batch-item-major order — the waves that meet the
        # loud item 0 go on to quiet items with their accumulators still in registers
        set_opt("IDEE_CONV_HRUN", "2")
        mode, B, wsets, T, H, W, Cin, Cout, kt, pad = L.CONV_K333_REPLICATE, 12, 2, 8, 64, 72, 16, 16, 3, "replicate"
    else:
        mode, B, wsets, T, H, W, Cin, Cout, kt, pad = L.CONV_K233_S2_ZERO, 2, 1, 4, 12, 72, 96, 96, 2, "zeros"
    To = (T - 2) // 2 + 1 if kt == 2 else T
    g = torch.Generator().manual_seed(77)
    x = torch.randn((B, T, H, W, Cin), generator=g)
    x[0] *= 3e36
    dy = torch.randn((B, To, H, W, Cout), generator=g) * 1e-18
    w = torch.zeros((wsets, Cout, Cin, kt, 3, 3))
    # float64 reference: dW[s] = sum over the items of set s of the correlation of the padded input with dy
    import torch.nn.functional as F
    ref = torch.zeros_like(w, dtype=torch.float64)
    for b in range(B):
        xb = x[b].double().permute(3, 0, 1, 2)[None]                      # (1, Cin, T, H, W)
        if pad == "replicate":
            xb = F.pad(xb, (1, 1, 1, 1, 1, 1), mode="replicate")
            stride = (1, 1, 1)
        else:
            xb = F.pad(xb, (1, 1, 1, 1, 0, 0))
            stride = (2, 1, 1)
        wz = torch.zeros((Cout, Cin, kt, 3, 3), dtype=torch.float64, requires_grad=True)
        y = F.conv3d(xb, wz, stride=stride)
        y.backward(dy[b].double().permute(3, 0, 1, 2)[None])
        ref[b % wsets] += wz.grad
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 1e-6

    def run(math):
        set_opt("IDEE_CONV_MATH", math)
        conv = ops.ConvLayer(mode, B, wsets, T, H, W, Cin, Cout, ops._cl_strides(T, H, W, Cin), ops._cl_strides(To, H, W, Cout), True)
        dw, _ = conv.wgrad(dev(x), dev(dy), dev(w))
        return dw.cpu().double()

    err = {}
    for math in ("f32", "f22"):
        dw = run(math)
        assert bool(torch.isfinite(dw).all()), math
        err[math] = float((dw - ref).norm() / ref.norm())
    print(f"\n[wgrad {shape}, |x| 3e36 in one item, |dy| 1e-18] relative L2 error of dW: exact {err['f32']:.2e}  f22 {err['f22']:.2e}")
    assert err["f22"] <= 4.0 * err["f32"] + 2e-7, err


@pytest.mark.parametrize("profile", ["loss_scale", "ragged_rows", "huge_x", "ragged_x"])
def test_classifier_f22_error_against_float64(profile, set_opt):
    """The classifier on z_q (plain path) at V = 6: the 96 -> 96 joint head — conv_wide_split_kernel forward / data gradient and
    wgrad_wide_split_kernel (two 6 x 3 block halves, workgroup-wide row scale) — and the heads' 16 -> 16 (2,3,3) convs under
    IDEE_CONV_MATH=f22, against the oracle's classifier evaluated in FLOAT64, beside the exact-f32 kernels on the same inputs.
    `huge_x` / `ragged_x`: classifier inputs of magnitude 3e5, resp. one magnitude per (n, h) row over 1e-2 .. 1e7 — the activation
    range guard of the wide forward, of both weight-gradient kernels and of the heads' 16 -> 16 forward fires (common.h ActGuard).
    Asserted: relative L2 error of the logits, dx and every parameter gradient within 4x the exact kernels' (+ 2e-7).  Tie-free case."""
    from oracle import idee_oracle as O
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    N, V, E, T, H, W = 2, 6, 16, 8, 6, 72

    def build(seed):
        torch.manual_seed(seed)
        c = CNN_3D(in_var=V, embed_dim=E, dim=E, drop_rate=0.0)
        sd = {k: v.clone() for k, v in c.state_dict().items()}
        x = torch.randn(N, V, E, T, H, W)
        rz, ry = torch.randn(N, 1, H, W), torch.randn(V, N, 1, H, W)
        if profile == "loss_scale":
            rz, ry = rz * 1e-7, ry * 1e-7
        elif profile == "ragged_rows":
            mag = 10.0 ** (torch.rand(N, 1, H, 1) * 12.0 - 11.0)
            rz, ry = rz * mag, ry * mag
        elif profile == "huge_x":
            x = x * 3e5
        else:
            x = x * 10.0 ** (torch.rand(N, 1, 1, 1, H, 1) * 9.0 - 2.0)
        sdl = {k: v.double().requires_grad_() for k, v in sd.items()}
        xo = x.double().requires_grad_()
        with relu_record() as rec:
            zo, yo = O.classifier(xo, sdl, pre="")
        if profile == "ragged_x":
            _row_ties(rec, (1, 2, 4))         # (N, C, T, H, W) pre-activations: one scale per (n, h) row
        return (c, sd, x, rz, ry, sdl, xo, zo, yo), rec

    c, sd, x, rz, ry, sdl, xo, zo, yo = tie_free_case(build, range(900, 924), f"classifier f64 {profile}")
    assert zo.dtype == torch.float64
    ((zo * rz.double()).sum() + sum((yo[v] * ry[v].double()).sum() for v in range(V))).backward()
    ref = {"z": zo.detach(), "y": torch.stack(yo).detach(), "dx": xo.grad}
    ref.update({"d" + k: sdl[k].grad for k in sd})
    c = c.to(DEV)

    def run(mode):
        set_opt("IDEE_CONV_MATH", mode)
        for p_ in c.parameters():
            p_.grad = None
        xd = x.to(DEV).requires_grad_()
        z, y = c(xd)
        ((z * rz.to(DEV)).sum() + sum((y[v] * ry[v].to(DEV)).sum() for v in range(V))).backward()
        out = {"z": z.detach().cpu().double(), "y": torch.stack(y).detach().cpu().double(), "dx": xd.grad.cpu().double()}
        named = dict(c.named_parameters())
        for key, name, idx in c._key_map():
            g = named[name].grad
            out["d" + key] = (g if idx is None else g[idx]).detach().cpu().double().reshape(ref["d" + key].shape)
        assert all(bool(torch.isfinite(v).all()) for v in out.values()), mode
        return out

    res = {m: run(m) for m in ("f32", "f22")}
    worst = (0.0, None)
    print(f"\n[classifier f22 vs float64, profile {profile}] relative L2 error: tensor  exact-f32  f22")
    for k in ref:
        e = {m: float((res[m][k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-300)) for m in res}
        if k in ("z", "y", "dx") or e["f22"] > worst[0]:
            print(f"  {k:40s} {e['f32']:.2e}  {e['f22']:.2e}")
        worst = max(worst, (e["f22"], k))
        if ref[k].numel() == 1:      # conv3.bias: ONE atomically accumulated f32 sum over 12 decades of signed terms, computed by the
            assert e["f22"] <= 1e-5, (k, e)      # same exact streaming kernel in both modes — its error is summation-order noise
            continue
        assert e["f22"] <= 4.0 * e["f32"] + 2e-7, (k, e)
    print(f"  worst f22 tensor: {worst[1]} {worst[0]:.2e} ({len(ref)} tensors)")


@pytest.mark.parametrize("T,H,W", [(3, 5, 200), (8, 7, 40), (2, 4, 30), (2, 3, 31), (2, 3, 63)])
@variants('default', 'conv_split', 'conv_f22')
def test_conv_ln_unit_vs_oracle(T, H, W, kernel_variant):
    """CNN_3D encoder unit x + ReLU(LN(conv3^3 replicate(x))) twice at a frame-wide W (both edge columns, partial
    column blocks) against the oracle, strict tolerance on a tie-free case.  W = 30 / 31 / 63: the padded row W + 2 is exactly one
    32-pixel strip of the shifted-dy weight gradient (wgrad16_shift_kernel), one strip plus ONE padded pixel, two strips plus one."""
    from oracle import idee_oracle as O
    ops = _ops()
    N, V, C = 1, 2, 16
    keys = ("conv1.weight", "norm1.weight", "norm1.bias", "conv2.weight", "norm2.weight", "norm2.bias")

    def build(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((N, V, T, H, W, C), generator=g)
        r = torch.randn((N, V, T, H, W, C), generator=g)
        ps = {}
        for k in keys:
            ps[k] = (torch.randn((V, C, C, 3, 3, 3), generator=g) * 0.06 if "conv" in k else
                     (1 + 0.1 * torch.randn((V, C), generator=g) if k.endswith("weight") else 0.1 * torch.randn((V, C), generator=g)))
        xl = x.clone().requires_grad_()
        pl = {k: v.clone().requires_grad_() for k, v in ps.items()}
        with relu_record() as rec:
            yo = torch.stack([O.conv_block(xl[:, v].permute(0, 4, 1, 2, 3), {k: pl[k][v] for k in keys}, "", C, C)
                              .permute(0, 2, 3, 4, 1) for v in range(V)], 1)
        return (x, r, ps, xl, pl, yo), rec

    x, r, ps, xl, pl, yo = tie_free_case(build, range(40 + W, 48 + W), f"conv_ln_unit {T}x{H}x{W}")
    (yo * r).sum().backward()
    xd = x.to(DEV).requires_grad_()
    pd = {k: v.to(DEV).requires_grad_() for k, v in ps.items()}
    t = ops.ConvLNUnit.apply(xd, pd["conv1.weight"], pd["norm1.weight"], pd["norm1.bias"])
    t = ops.ConvLNUnit.apply(t, pd["conv2.weight"], pd["norm2.weight"], pd["norm2.bias"])
    close("y", t, yo)
    (t * r.to(DEV)).sum().backward()
    gclose("dx", xd.grad, xl.grad)
    for k in keys:
        gclose("d" + k, pd[k].grad, pl[k].grad)


def _lfq_module(sd, K):
    from idee_amd.models.codebook.LFQ import LFQ
    q = LFQ(dim=16, codebook_size=K, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    q.load_state_dict(sd, strict=True)
    return q.to(DEV)


@pytest.mark.parametrize("tag", ["k2_train", "k2_eval", "k4_train"])
def test_lfq_golden(tag):
    a, sd, grads = load_golden("lfq_" + tag)
    K, training = int(a["K"]), bool(int(a["training"]))
    q = _lfq_module(sd, K)
    q.train(training)
    x = dev(a["x"]).requires_grad_()
    zq, idx, aux = q(x)
    # tokens whose projection is within 1e-5 of the decision boundary may flip: exclude them (listed)
    s = a["x"] @ sd["project_in.weight"].t() + sd["project_in.bias"]
    safe = (s.abs() > 1e-5).all(-1)
    assert int((~safe).sum()) <= 8
    assert torch.equal(idx.cpu()[safe], a["indices"].long()[safe])
    close("z_q", zq.cpu()[safe], a["z_q"][safe])
    close("aux", aux, a["aux"], atol=1e-5, rtol=5e-4)
    close("code0", q.indices_to_codes(torch.tensor([0], device=DEV)), a["code0"])
    if training:
        ((zq * dev(a["r"])).sum() + 2.5 * aux).backward()
        gclose("dx", x.grad.cpu()[safe], a["dx"][safe])
        for n, p in q.named_parameters():
            gclose("d" + n, p.grad, grads[n])


@pytest.mark.parametrize("tag", ["k2_c2_train", "k4_c2_train", "k2_c4_eval"])
def test_lfq_several_codebooks_golden(tag):
    """LFQ(num_codebooks > 1) (LFQ.py:92-101, :215-300: per-codebook softmax, indices (b, n, c), mean-over-codebooks entropies) against
    the reference's own outputs and gradients; state-dict keys and shapes as the reference's."""
    from idee_amd.models.codebook.LFQ import LFQ
    a, sd, grads = load_golden("lfq_multi_" + tag)
    K, ncb, training = int(a["K"]), int(a["ncb"]), bool(int(a["training"]))
    q = LFQ(dim=16, codebook_size=K, num_codebooks=ncb, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    msd = q.state_dict()
    assert list(msd) == list(sd) and all(tuple(msd[k].shape) == tuple(sd[k].shape) for k in sd)
    q.load_state_dict(sd, strict=True)
    q = q.to(DEV)
    q.train(training)
    x = dev(a["x"]).requires_grad_()
    zq, idx, aux = q(x)
    assert tuple(idx.shape) == tuple(a["indices"].shape)
    s = a["x"] @ sd["project_in.weight"].t() + sd["project_in.bias"]
    safe = (s.abs() > 1e-5).all(-1)
    assert int((~safe).sum()) <= 8
    assert torch.equal(idx.cpu()[safe], a["indices"].long()[safe])
    close("z_q", zq.cpu()[safe], a["z_q"][safe])
    close("aux", aux, a["aux"], atol=1e-5, rtol=5e-4)
    close("codes", q.indices_to_codes(a["code_idx"].long().to(DEV)), a["codes"])
    if training:
        ((zq * dev(a["r"])).sum() + 2.5 * aux).backward()
        gclose("dx", x.grad.cpu()[safe], a["dx"][safe])
        for n, p in q.named_parameters():
            gclose("d" + n, p.grad, grads[n])


@pytest.mark.parametrize("tag", ["e16_k2_train", "e16_k4_train", "e32_k16_train", "e16_k16_eval"])
def test_lfq_without_projections_golden(tag):
    """LFQ with dim == num_codebooks * log2 K (LFQ.py:98-101: project_in / project_out = nn.Identity; the token's channels are
    the code bits) against the reference's own outputs and gradient; the state dict holds only `mask`, as the reference's."""
    from idee_amd.models.codebook.LFQ import LFQ
    a, sd, _ = load_golden("lfq_noproj_" + tag)
    K, dim, training = int(a["K"]), int(a["dim"]), bool(int(a["training"]))
    d = K.bit_length() - 1
    q = LFQ(dim=dim, codebook_size=K, num_codebooks=dim // d, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    assert not q.has_projections and list(q.state_dict()) == list(sd) == ["mask"]
    q.load_state_dict(sd, strict=True)
    q = q.to(DEV)
    q.train(training)
    x = dev(a["x"]).requires_grad_()
    zq, idx, aux = q(x)
    assert tuple(idx.shape) == tuple(a["indices"].shape) and idx.dtype == torch.int64
    assert torch.equal(idx.cpu(), a["indices"].long())          # no projection arithmetic: the bits are exact
    assert torch.equal(zq.cpu(), a["z_q"])
    close("aux", aux, a["aux"], atol=1e-5, rtol=5e-4)
    close("codes", q.indices_to_codes(a["code_idx"].long().to(DEV)), a["codes"])
    if training:
        ((zq * dev(a["r"])).sum() + 2.5 * aux).backward()
        gclose("dx", x.grad, a["dx"])


def test_lfq_without_projections_bf16_and_no_dense_consumer():
    """bf16 tokens (z_q and dz stay bf16; statistics f32 from the bf16-rounded values) and a backward with only the aux loss
    (no d z_q stream) against the f32 path on the same rounded tokens."""
    from idee_amd.models.codebook.LFQ import LFQ
    q = LFQ(dim=16, codebook_size=4, num_codebooks=8).to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 257, 16, generator=g) * 0.01).bfloat16()
    xf = dev(x.float()).requires_grad_()
    xb = dev(x).requires_grad_()
    zf, i_f, af = q(xf)
    zb, i_b, ab = q(xb)
    assert zb.dtype == torch.bfloat16 and torch.equal(i_f, i_b) and torch.equal(zf, zb.float())
    close("aux", ab, af, atol=1e-6, rtol=1e-5)
    af.backward()
    ab.backward()
    assert xb.grad.dtype == torch.bfloat16
    close("dx", xb.grad.float(), xf.grad, atol=1e-6, rtol=1e-2)


def _classifier(sd, V):
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    c = CNN_3D(in_var=V, embed_dim=16, dim=16, n_classes=sd["conv3.weight"].shape[0], drop_rate=0.0)
    c.load_state_dict(sd, strict=True)
    return c.to(DEV)


@pytest.mark.parametrize("tag", ["classifier_v2", "classifier_v2_k3"])       # _k3: n_classes = 3 (joint head only, as the reference)
@variants('default', 'conv_split', 'conv_f22', 'wgrad_wide_regs')
def test_classifier_golden(tag, kernel_variant):
    a, sd, grads = load_golden(tag)
    c = _classifier(sd, 2)
    x = dev(a["x"]).requires_grad_()
    z, y = c(x)
    close("z", z, a["z"])
    close("y", torch.stack(y), a["y"])
    ((z * dev(a["rz"])).sum() + sum((y[v] * dev(a["ry"][v])).sum() for v in range(2))).backward()
    gclose("dx", x.grad, a["dx"])
    got = {k: v for k, v in c.state_dict(keep_vars=True).items()}
    named = dict(c.named_parameters())
    m = {key: (name, idx) for key, name, idx in c._key_map()}
    for k, gref in grads.items():
        name, idx = m[k]
        g = named[name].grad
        gclose("d" + k, g if idx is None else g[idx], gref)


@pytest.mark.parametrize("V,H,W,E", [(3, 7, 72, 16), (6, 5, 40, 16), (1, 3, 3, 16), (5, 4, 20, 16), (6, 6, 200, 16), (2, 3, 200, 16),
                                     (6, 4, 24, 32), (3, 3, 20, 32), (2, 19, 40, 16), (3, 5, 24, 16), (2, 3, 31, 16), (2, 2, 30, 16)])
@variants('default', 'conv_split', 'conv_f22', 'wgrad_wide_regs')
def test_classifier_vs_oracle(V, H, W, E, set_opt, kernel_variant):
    """Plain path (convolutions on z_q).  W = 200: the frame-wide row-wrapping column-block numbering of conv_wide,
    its zero borders, the wide weight-gradient kernel and the heads' strip kernels at both frame edges.  E = 32 (config #4's
    width): the 192 -> 192 joint head (weight gradient as a grid of 6 x 6-block sub-problems) and the 32 -> 32 heads (strip
    weight gradient over (input block, output block) pairs)."""
    from oracle import idee_oracle as O
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    N, T = (1 if W == 200 else 2), (16 if (H, W, E) == (5, 24, 16) else 8)     # T = 16 (config #4): two output frames, the
    if H == 19:      # the heads' strip kernels in runs of 8, 8, 3 rows (run starts away from row 0)      # Cout = 1 convs read 4
        set_opt("IDEE_CONV_HRUN", "8")

    torch.manual_seed(V * 100 + W)
    c = CNN_3D(in_var=V, embed_dim=E, dim=E, drop_rate=0.0)
    sd = {k: v.clone() for k, v in c.state_dict().items()}
    x = torch.randn(N, V, E, T, H, W)
    oshape = (N, 1, H, W) if T == 8 else (N, 1, T // 8, H, W)
    rz, ry = torch.randn(oshape), torch.randn((V,) + oshape)
    runs = []
    for mode in (0, 1):                      # oracle backward with the ReLU derivative at tie units forced to 0 / to 1
        sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
        xo = x.clone().requires_grad_()
        with tie_relu(mode) as tr:
            zo, yo = O.classifier(xo, sdl, pre="")
        ((zo * rz).sum() + sum((yo[v] * ry[v]).sum() for v in range(V))).backward()
        runs.append((sdl, xo, zo, yo, tr))
    (sd0, x0, zo, yo, tr), (sd1, x1, _, _, _) = runs
    c = c.to(DEV)
    xd = x.to(DEV).requires_grad_()
    z, y = c(xd)
    close("z", z, zo)
    close("y", torch.stack(y), torch.stack(yo))
    ((z * rz.to(DEV)).sum() + sum((y[v] * ry[v].to(DEV)).sum() for v in range(V))).backward()
    affected = interval_close("dx", xd.grad, x0.grad, x1.grad)
    total = x.numel()
    named = dict(c.named_parameters())
    for key, name, idx in c._key_map():
        g = named[name].grad
        affected += interval_close("d" + key, g if idx is None else g[idx], sd0[key].grad, sd1[key].grad)
        total += sd0[key].numel()
    print(f"[relu ties] classifier V{V} {H}x{W}: {tr.n_ties} ties among {tr.n_units} units; gradient elements inside a tie "
          f"interval (all others compared strictly): {affected}/{total}")
    assert affected <= 0.01 * total


@pytest.mark.parametrize("V,D,H,W", [(3, 1, 7, 72), (6, 1, 5, 40), (1, 1, 3, 3), (5, 2, 4, 20), (8, 2, 3, 19), (6, 1, 6, 200),
                                     (2, 2, 3, 200), (6, 1, 19, 40)])
@variants('default', 'conv_split', 'conv_f22', 'dgrad_to16_wide')
def test_classifier_sign_field_vs_oracle(V, D, H, W, set_opt, kernel_variant):
    """The rank-d input path (ops.ClassifierHeadsSigns: convs on the sign field with folded weights) against
    the oracle's classifier applied to z_q = project_out(signs): outputs and every gradient (strict, tie-free case)."""
    _sign_field_case(V, D, H, W, set_opt, 1)


@pytest.mark.parametrize("V,D,H,W,ncls", [(2, 1, 5, 24, 3), (3, 2, 4, 20, 16)])
def test_classifier_sign_field_n_classes(V, D, H, W, ncls, set_opt):
    """n_classes > 1 (the joint head's conv3 has n_classes outputs, classifier/CNN_3D.py:85; the per-variable heads keep one):
    the sign-field path against the oracle, as above."""
    _sign_field_case(V, D, H, W, set_opt, ncls)


def _sign_field_case(V, D, H, W, set_opt, ncls):
    from oracle import idee_oracle as O
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    N, T, E = (1 if W == 200 else 2), 8, 16
    if H == 19:      # row ranges of at most 8 rows: the K-split data gradient of the sign-field convs starts away from row 0
        set_opt("IDEE_CONV_HRUN", "8")

    torch.manual_seed(V * 100 + W)
    c = CNN_3D(in_var=V, embed_dim=E, dim=16, n_classes=ncls, drop_rate=0.0)
    po = torch.nn.Linear(D, E)
    with torch.no_grad():
        po.bias.normal_(0, 0.3)
    sd = {k: v.clone() for k, v in c.state_dict().items()}
    signs = (torch.randint(0, 2, (N, V, T, H, W, D)).float() * 2 - 1)
    rz, ry = torch.randn(N, ncls, H, W), torch.randn(V, N, 1, H, W)
    runs = []
    for mode in (0, 1, 2):                   # 2: float64, plain ReLU (this case only: its head 5 has thousands of near-tie units)
        if mode == 2 and H != 19:
            continue
        dt = torch.float64 if mode == 2 else torch.float32
        sdl = {k: v.clone().to(dt).requires_grad_() for k, v in sd.items()}
        so = signs.clone().to(dt).requires_grad_()
        wo, bo = po.weight.detach().clone().to(dt).requires_grad_(), po.bias.detach().clone().to(dt).requires_grad_()
        zq = (so @ wo.t() + bo).permute(0, 1, 5, 2, 3, 4)
        if mode == 2:
            zo, yo = O.classifier(zq, sdl, pre="")
            tr = None
        else:
            with tie_relu(mode) as tr:
                zo, yo = O.classifier(zq, sdl, pre="")
        ((zo * rz.to(dt)).sum() + sum((yo[v] * ry[v].to(dt)).sum() for v in range(V))).backward()
        runs.append((sdl, so, wo, bo, zo, yo, tr))
    (sd0, so0, wo0, bo0, zo, yo, tr), (sd1, so1, wo1, bo1, _, _, _) = runs[:2]
    ex = runs[2] if len(runs) > 2 else None
    c, po = c.to(DEV), po.to(DEV)
    sg = signs.to(DEV).requires_grad_()
    z, y = c.forward_signs(sg, po)
    close("z", z, zo, atol=2e-5, rtol=2e-4)
    close("y", torch.stack(y), torch.stack(yo), atol=2e-5, rtol=2e-4)
    ((z * rz.to(DEV)).sum() + sum((y[v] * ry[v].to(DEV)).sum() for v in range(V))).backward()
    affected = interval_close("dsigns", sg.grad, so0.grad, so1.grad, exact=ex[1].grad if ex else None)
    affected += interval_close("dw_out", po.weight.grad, wo0.grad, wo1.grad, exact=ex[2].grad if ex else None)
    affected += interval_close("db_out", po.bias.grad, bo0.grad, bo1.grad, exact=ex[3].grad if ex else None)
    total = signs.numel() + wo0.numel() + bo0.numel()
    named = dict(c.named_parameters())
    for key, name, idx in c._key_map():
        g = named[name].grad
        affected += interval_close("d" + key, g if idx is None else g[idx], sd0[key].grad, sd1[key].grad,
                                   exact=ex[0][key].grad if ex else None)
        total += sd0[key].numel()
    print(f"[relu ties] sign-field V{V} D{D} {H}x{W}: {tr.n_ties} ties among {tr.n_units} units; gradient elements inside a tie "
          f"interval (all others compared strictly): {affected}/{total}")
    assert affected <= (0.5 if ex else 0.01) * total      # with the float64 reference every element is still pinned: by the interval or by it


@pytest.mark.parametrize("T,H,W", [(8, 10, 20), (9, 6, 17), (16, 5, 33)])
@variants('default', 'dgrad_to16_wide')
def test_sign_field_switch_matches_plain_path(monkeypatch, T, H, W, kernel_variant):
    """VQ_model with and without the rank-d classifier input: same outputs, same gradients (to f32 rounding);
    T = 9 leaves an unpaired last frame (never read by the stride-2 conv), T = 16 keeps two output frames.
    The case (weights seed) is one the oracle shows to be free of ReLU ties and of LFQ boundary tokens."""
    from oracle import idee_oracle as O
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    kw = dict(in_channels_dynamic=3, in_channels=1, encoder="Swin_3D")
    cfg = default_config(**kw)
    ocfg = O.default_cfg(**kw)
    torch.manual_seed(5)
    x = torch.randn(2, 3, 1, T, H, W)

    def build(seed):
        torch.manual_seed(seed)
        sd = {k: v.clone() for k, v in VQ_model(cfg).state_dict().items()}
        with torch.no_grad(), relu_record() as rec:
            O.vq_model_forward(x, sd, ocfg, training=True)
        return sd, rec

    sd = tie_free_case(build, range(11, 19), f"sign-field switch T{T} {H}x{W}")
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEE_SIGN_FIELD", flag)
        m = VQ_model(cfg)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        assert m.sign_field_classifier == (flag == "1")
        zc, y, an, zq, lq = m(x.to(DEV))
        (zc.sum() + sum(t.square().sum() for t in y) + zq.abs().sum() + lq.sum()).backward()
        outs.append((zc, torch.stack(y), an, zq, lq, {n: p.grad.clone() for n, p in m.named_parameters()}))
    a, b = outs
    close("z", a[0], b[0], atol=2e-5, rtol=2e-4)
    close("y", a[1], b[1], atol=2e-5, rtol=2e-4)
    assert torch.equal(a[2], b[2])
    close("zq", a[3], b[3], atol=0, rtol=0)
    close("loss", a[4], b[4], atol=0, rtol=0)
    for n in a[5]:
        gclose("d" + n, a[5][n], b[5][n])


def test_losses_golden():
    from idee_amd.models import losses as Lm
    a, _, _ = load_golden("losses")
    bce_s, bce_r = Lm.BCE_loss_synthetic(), Lm.BCE_loss()
    for tag in ("mixed", "zeros", "ones"):
        p = dev(a["bce_s/pred"]).requires_grad_()
        l = bce_s(p, dev(a[f"bce_s/{tag}/target"]))
        close(f"bce_s/{tag}", l, a[f"bce_s/{tag}/loss"], atol=1e-5, rtol=1e-4)
        l.backward()
        close(f"bce_s/{tag}/dpred", p.grad, a[f"bce_s/{tag}/dpred"], atol=1e-7, rtol=1e-3)
    p = dev(a["bce_r/pred"]).requires_grad_()
    l = bce_r(p, dev(a["bce_r/target"]), dev(a["bce_r/mask_valid"]))
    close("bce_r", l, a["bce_r/loss"], atol=1e-5, rtol=1e-4)
    l.backward()
    close("bce_r/dpred", p.grad, a["bce_r/dpred"], atol=1e-7, rtol=1e-3)
    N, V, C, T, H, W = a["an_s/z_q"].shape
    an_s = Lm.Anomaly_L1_loss_synthetic(n_dynamic=V, delta_t=T, dim=C)
    an_r = Lm.Anomaly_L1_loss(n_dynamic=V, delta_t=T, dim=C)
    zq = dev(a["an_s/z_q"]).requires_grad_()
    l = an_s(zq, dev(a["an_s/mask"]), dev(a["an_s/vq0"]))
    close("an_s", l, a["an_s/loss"], atol=1e-5, rtol=1e-4)
    l.backward()
    close("an_s/dz_q", zq.grad, a["an_s/dz_q"], atol=1e-9, rtol=1e-3)
    l1 = an_s(zq.detach(), torch.ones_like(dev(a["an_s/mask"])), dev(a["an_s/vq0"]))
    assert math.isnan(float(l1)) and math.isnan(float(a["an_s/all_extreme_loss"]))
    zq.grad = None
    l = an_r(zq, dev(a["an_s/mask"]), dev(a["an_r/mask_valid"]), dev(a["an_s/vq0"]))
    close("an_r", l, a["an_r/loss"], atol=1e-5, rtol=1e-4)
    l.backward()
    close("an_r/dz_q", zq.grad, a["an_r/dz_q"], atol=1e-9, rtol=1e-3)


@pytest.mark.parametrize("M", [12, 4000, 32768, 160000, 160004, 1000003])
@pytest.mark.parametrize("kind", ["binary", "all_equal", "real_valued"])
@pytest.mark.parametrize("masked", [False, True])
def test_bce_class_counts_vs_histc(M, kind, masked):
    """idee_bce_class_counts against torch.histc(target[valid], bins=2) (losses.py:80, :115) on the CPU: the one-workgroup kernel
    (small or unaligned M) and the many-workgroup kernel (M >= 32768, 16-byte loadable) — labels in {0, 1}, a constant target
    (histc widens its range to [min - 1, max + 1]), real-valued targets (data-derived range), with and without a validity mask.
    Counts are integers: exact."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + len(kind))
    if kind == "binary":
        t = (torch.rand(M, generator=g) < 0.3).float()
    elif kind == "all_equal":
        t = torch.full((M,), 1.0)
    else:
        t = torch.randn(M, generator=g) * 3.0 + 0.5
    v = (torch.rand(M, generator=g) < 0.8).float() if masked else None
    ref = torch.histc(t[v.bool()] if masked else t, bins=2)
    got = ops.bce_class_counts(dev(t), dev(v) if masked else None).cpu()
    assert torch.equal(got, ref), (got, ref)


@pytest.mark.parametrize("E,K,valid", [(16, 2, False), (16, 2, True), (16, 4, True), (32, 2, False), (32, 4, True)])
def test_anomaly_loss_on_code_bits_vs_oracle(E, K, valid):
    """The rank-d form of the anomaly loss (ops.AnomalyL1Signs: reads LFQ's code bits, returns gradients for the bits and
    project_out) against the oracle's Anomaly_L1_loss(_synthetic) applied to z_q of the oracle's LFQ: value, and the
    gradients that reach the LFQ input and all four LFQ parameters (losses.py:15-61, :127-168; LFQ.py:183-307)."""
    from oracle import idee_oracle as O
    ops = _ops()
    N, V, T, H, W = 2, 3, 4, 5, 7
    g = torch.Generator().manual_seed(E * 10 + K + valid)
    D = K.bit_length() - 1
    z = torch.randn((N, V, T, H, W, E), generator=g)
    sd = {"vq.project_in.weight": torch.randn((D, E), generator=g) * 0.5, "vq.project_in.bias": torch.randn((D,), generator=g) * 0.1,
          "vq.project_out.weight": torch.randn((E, D), generator=g) * 0.5, "vq.project_out.bias": torch.randn((E,), generator=g) * 0.1}
    me = (torch.rand((N, H, W), generator=g) < 0.3).float()
    mv = (torch.rand((N, T, H, W), generator=g) < 0.3).float() if valid else None
    cfg = O.default_cfg(codebook_dim=E, codebook_size=K)
    # oracle
    sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
    zo = z.clone().requires_grad_()
    zq_o, _, aux_o = O.lfq(zo.reshape(N, -1, E), sdl, cfg, True)
    zq6 = zq_o.reshape(N, V, T, H, W, E).permute(0, 1, 5, 2, 3, 4)
    vq0 = O.lfq_indices_to_codes(torch.tensor([0]), sdl, cfg).detach()
    lo = O.anomaly_l1(zq6, me, mv, vq0) if valid else O.anomaly_l1_synthetic(zq6, me, vq0)
    (lo * 1.7 + aux_o * 0.0).backward()
    # product: LFQ -> signs -> loss
    zd = z.to(DEV).requires_grad_()
    pd = {k: v.to(DEV).requires_grad_() for k, v in sd.items()}
    zq, idx, aux, signs = ops.LFQQuantize.apply(zd, pd["vq.project_in.weight"], pd["vq.project_in.bias"], pd["vq.project_out.weight"],
                                                pd["vq.project_out.bias"], K, True, cfg.lambda_entropy, cfg.diversity_gamma,
                                                cfg.lambda_commitment, True)
    l, den = ops.AnomalyL1Signs.apply(signs, pd["vq.project_out.weight"], pd["vq.project_out.bias"], me.to(DEV),
                                      mv.to(DEV) if valid else None, K)
    close("loss", l, lo, atol=1e-6, rtol=1e-5)
    (l * 1.7 + aux * 0.0).backward()
    gclose("dz", zd.grad, zo.grad)
    for k in sd:
        gclose("d" + k, pd[k].grad, sdl[k].grad)


def test_adam_flat_vs_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=0.003, betas=(0.9, 0.999))
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.003, step)
    close("adam", p, ref, atol=1e-6, rtol=1e-5)


# ---------------------------------------------------------------------------------------------
# end to end: VQ_model + losses + flat Adam against the golden training steps
# ---------------------------------------------------------------------------------------------
def _cfg_from(a):
    from idee_amd.config import default_config
    return default_config(in_channels_dynamic=int(a["cfg_V"]), in_channels=int(a["cfg_C"]), encoder=str(a["encoder"]))


def _safe_tokens(a, sd, cfg):
    """Tokens whose LFQ projection is at least 1e-5 away from the sign boundary (oracle, CPU)."""
    from oracle import idee_oracle as O
    ocfg = O.default_cfg(in_channels_dynamic=cfg.in_channels_dynamic, in_channels=cfg.in_channels, encoder=cfg.encoder)
    with torch.no_grad():
        z = (O.swin_encoder if cfg.encoder == "Swin_3D" else O.cnn_encoder)(a["x_d"], sd, ocfg)      # (N,V,E,T,H,W)
        s = torch.einsum("nvethw,de->nvthwd", z, sd["vq.project_in.weight"]) + sd["vq.project_in.bias"]
    return (s.abs() > 1e-5).all(-1)                                                                  # (N,V,T,H,W)


@pytest.mark.parametrize("tag", ["swin_v2_14x18", "swin_v6_8x8_c2", "cnn_v2_12x16"])
@variants('default', 'all_split', 'fwd22', 'all_f22')
def test_e2e_eval_forward_golden(tag, kernel_variant):
    from idee_amd.models.build import VQ_model
    a, sd, _ = load_golden("e2e_" + tag)
    cfg = _cfg_from(a)
    m = VQ_model(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        z, y, an, zq, lz = m(dev(a["x_d"]))
    safe = _safe_tokens(a, sd, cfg)
    assert int((~safe).sum()) <= 4, "too many boundary tokens"
    assert torch.equal(an.cpu()[safe], a["eval/anomaly"].long()[safe])
    if bool(safe.all()):
        close("z", z, a["eval/z"])
        close("y", torch.stack(y), a["eval/y"])
        close("z_q", zq, a["eval/z_q"])
    assert float(lz) == 0.0 and tuple(lz.shape) == (1,)
    assert zq.shape == a["eval/z_q"].shape and an.dtype == torch.int64


@pytest.mark.parametrize("tag", ["swin_v2_14x18", "cnn_v2_12x16", "era5_swin_v3_c2_10x12"])
@variants('default', 'all_split', 'fwd22', 'all_f22')
def test_e2e_train_steps_golden(tag, kernel_variant):
    """3 full steps of the train_synthetic.py / train_ERA5_Land.py sequence (fused step objects) vs the reference's
    outputs, losses, all gradients and the parameters after 3 Adam steps."""
    from idee_amd.models.build import VQ_model
    from idee_amd.train import SyntheticTrainStep, ERA5TrainStep
    a, sd, grads = load_golden("e2e_" + tag)
    cfg = _cfg_from(a)
    m = VQ_model(cfg)
    m.load_state_dict(sd, strict=True)
    era5 = "mask_drought" in a
    ts = (ERA5TrainStep if era5 else SyntheticTrainStep)(cfg, device=DEV, model=m)
    safe = _safe_tokens(a, sd, cfg)
    assert bool(safe.all()), "fixture has boundary tokens; regenerate with another seed"
    x = dev(a["x_d"])
    masks = tuple(dev(a[k]) for k in (("mask_drought", "mask_drought_loss", "mask_cold_surface", "mask_cold_surface_loss")
                                      if era5 else ("mask_extreme", "mask_extreme_loss")))
    losses = []
    for step in range(3):
        if step == 0:
            ts.opt.zero_grad()
            outs = ts.model(x)
            total, parts = ts.losses(outs, *masks)
            total.backward()
            close("z", outs[0], a["train/z"])
            close("y", torch.stack(outs[1]), a["train/y"])
            assert torch.equal(outs[2].cpu(), a["train/anomaly"].long())
            close("z_q", outs[3], a["train/z_q"])
            close("loss_z_q", outs[4], a["train/loss_z_q"], atol=1e-5, rtol=5e-4)
            close("l_ext", parts["extreme"], a["train/l_ext"], atol=1e-5, rtol=1e-4)
            close("l_anom", parts["anomaly"], a["train/l_anom"], atol=1e-6, rtol=1e-4)
            close("l_var", parts["var"], a["train/l_var"], atol=1e-5, rtol=1e-4)
            close("loss", total.reshape(()), a["train/loss"].reshape(()), atol=1e-4, rtol=1e-4)
            named = {}
            for mod_name, mod in (("encoder.", ts.model.encoder), ("cls.", ts.model.cls)):
                ps = dict(mod.named_parameters())
                for key, name, idx in mod._key_map():
                    if name.endswith("_index"):
                        continue
                    base = name[:-1] if name.endswith("_ds_w5") else name
                    g = ps[base].grad
                    g = g.view(*g.shape, 1, 1, 1) if name.endswith("_ds_w5") else g
                    named[mod_name + key] = g if idx is None else g[idx]
            for k, p in ts.model.vq.named_parameters():
                named["vq." + k] = p.grad
            assert set(named) == set(grads)
            for k, gref in grads.items():
                gclose("d" + k, named[k], gref, rtol=2e-3, floor=2e-4)
            from idee_amd import parallel
            parallel.sync_gradients(ts.opt)
            ts.opt.step()
            losses.append(float(total))
        else:
            total, _, _ = ts.step(x, *masks)
            losses.append(float(total))
    torch.testing.assert_close(torch.tensor(losses, dtype=torch.float64), a["train/loss_per_step"].double(), rtol=2e-3, atol=1e-3)
    after = ts.model.state_dict()
    for k, v in after.items():
        if v.is_floating_point():
            if _L().get_option("IDEE_CONV_MATH") == "split":
                # opt-in split-precision convolutions: a parameter whose gradient is at the f32 noise level can take an Adam
                # step of either sign (|step| <= lr per step): a handful of elements within 3 lr, everything else strict
                d = (v.detach().float().cpu() - a["sd_after3/" + k].float()).abs()
                loose = d > 1e-4 + 2e-3 * a["sd_after3/" + k].float().abs()
                assert float(d.max()) <= 3e-3 and int(loose.sum()) <= max(1, int(5e-4 * d.numel())), (k, float(d.max()), int(loose.sum()))
            else:
                close("after3/" + k, v, a["sd_after3/" + k], atol=1e-4, rtol=2e-3)


def test_drop_in_modules_match_fused_step():
    """Calling the reference-named loss modules one by one (as train_synthetic.py does) gives the
    same total as the fused step object."""
    from idee_amd.models.build import VQ_model
    from idee_amd.models import losses as Lm
    from idee_amd.train import SyntheticTrainStep
    a, sd, _ = load_golden("e2e_swin_v2_14x18")
    cfg = _cfg_from(a)
    m = VQ_model(cfg)
    m.load_state_dict(sd, strict=True)
    ts = SyntheticTrainStep(cfg, device=DEV, model=m)
    x, me, ml = dev(a["x_d"]), dev(a["mask_extreme"]), dev(a["mask_extreme_loss"])
    outs = ts.model(x)
    total, _ = ts.losses(outs, me, ml)
    crit, crit_an = Lm.BCE_loss_synthetic(), Lm.Anomaly_L1_loss_synthetic(n_dynamic=2, delta_t=8, dim=16)
    pred, pred_y, _, z_q, loss_z_q = outs
    tgt = me.unsqueeze(1)
    loss = crit(pred, tgt) + crit_an(z_q, ml, ts.model.vq.indices_to_codes(torch.tensor([0], device=DEV)).clone().detach()) \
        * cfg.lambda_anomaly + sum(crit(p, tgt) for p in pred_y) + loss_z_q
    close("total", loss.reshape(()), total.reshape(()), atol=1e-5, rtol=1e-5)
    close("ref total", loss.reshape(()), a["train/loss"].reshape(()), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("tag", ["k2_train", "k2_eval", "k8_train"])
def test_vq_nearest_golden(tag):
    from idee_amd.models.codebook.VQ import VQ
    a, sd, grads = load_golden("vq_" + tag)
    K, training = int(a["K"]), bool(int(a["training"]))
    q = VQ(dim=16, codebook_size=K, codebook_dim=16)
    q.load_state_dict(sd, strict=True)
    q = q.to(DEV)
    q.train(training)
    x = dev(a["x"]).requires_grad_()
    zq, idx, loss = q(x)
    assert torch.equal(idx.cpu(), a["indices"].long())
    close("z_q", zq, a["z_q"])
    close("loss", loss, a["loss"], atol=1e-6, rtol=1e-4)
    if training:
        ((zq * dev(a["r"])).sum() + 1.7 * loss.sum()).backward()
        gclose("dx", x.grad, a["dx"])
        gclose("dembed", q._codebook.embed.grad, grads["_codebook.embed"])


@variants('default', 'swin_split', 'swin_f22', 'conv32_generic', 'all_split', 'all_f22')
def test_small_config_c32_vs_oracle(kernel_variant):
    """Builder-defined "small" width: embed 32, head_dim 16, codebook/cls dim 32, C_in = 2,
    depths [2,2] — full model forward + one backward against the oracle (strict; tie-free, boundary-free case)."""
    from oracle import idee_oracle as O
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    kw = dict(encoder="Swin_3D", in_channels=2, in_channels_dynamic=2, en_embed_dim=[32, 32], en_depths=[2, 2],
              codebook_dim=32, cls_dim=32)
    cfg = default_config(**kw)
    ocfg = O.default_cfg(**kw)

    def build(seed):
        torch.manual_seed(seed)
        m = VQ_model(cfg)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1.0, float(p[0].numel() if p.ndim > 1 else 1)) ** 0.5) + (0.0 if p.ndim > 1 else 0.05))
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        x = torch.randn((1, 2, 2, 8, 12, 16), generator=g)
        sdl = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
        with relu_record() as rec:
            outs = O.vq_model_forward(x, sdl, ocfg, training=True)
        s_ = torch.einsum("nvethw,de->nvthwd", O.swin_encoder(x, sd, ocfg), sd["vq.project_in.weight"]) + sd["vq.project_in.bias"]
        if not bool((s_.abs() > 1e-5).all()):
            rec.inputs.append(torch.tensor([0.0, 1.0]))        # a boundary token disqualifies the seed like a tie does
        return (m, sd, x, sdl, outs, g), rec

    m, sd, x, sdl, (zo, yo, ano, zqo, lzo), g = tie_free_case(build, range(21, 29), "small config C=32")
    rz = torch.randn(zo.shape, generator=g)
    ((zo * rz).sum() + sum(t.sum() for t in yo) + lzo.sum()).backward()
    m = m.to(DEV).train()
    z, y, an, zq, lz = m(x.to(DEV))
    assert torch.equal(an.cpu(), ano)
    close("z", z, zo, atol=5e-5, rtol=5e-4)
    close("y", torch.stack(y), torch.stack(yo), atol=5e-5, rtol=5e-4)
    close("z_q", zq, zqo)
    close("loss_z_q", lz, lzo, atol=1e-5, rtol=1e-3)
    ((z * rz.to(DEV)).sum() + sum(t.sum() for t in y) + lz.sum()).backward()
    checked = 0
    for mod_name, mod in (("encoder.", m.encoder), ("cls.", m.cls)):
        ps = dict(mod.named_parameters())
        for key, name, idx in mod._key_map():
            if name.endswith("_index"):
                continue
            base = name[:-1] if name.endswith("_ds_w5") else name
            gg = ps[base].grad
            gg = gg.view(*gg.shape, 1, 1, 1) if name.endswith("_ds_w5") else gg
            gclose("d" + mod_name + key, gg if idx is None else gg[idx], sdl[mod_name + key].grad)
            checked += 1
    for k, p in m.vq.named_parameters():
        gclose("dvq." + k, p.grad, sdl["vq." + k].grad)
        checked += 1
    assert checked == sum(1 for v in sd.values() if v.is_floating_point())


def test_classifier_T16_shapes_and_T4_error():
    """Outside the reference's working envelope: T=16 yields a 2-step map, T=4 raises."""
    from oracle import idee_oracle as O
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    torch.manual_seed(31)
    c = CNN_3D(in_var=2, embed_dim=16, dim=16, drop_rate=0.0)
    sd = {k: v.clone() for k, v in c.state_dict().items()}
    x = torch.randn(1, 2, 16, 16, 6, 10)
    zo, yo = O.classifier(x, sd, pre="")
    c = c.to(DEV)
    z, y = c(x.to(DEV))
    assert tuple(z.shape) == tuple(zo.shape) == (1, 1, 2, 6, 10)
    close("z", z, zo)
    close("y", torch.stack(y), torch.stack(yo))
    with pytest.raises(RuntimeError):
        c(torch.randn(1, 2, 16, 4, 6, 10, device=DEV))


def test_harness_evaluators_collector_checkpoint(tmp_path):
    """Device-side confusion counts / majority vote vs the reference's numpy evaluators (golden), and the checkpoint dict."""
    import numpy as np
    from idee_amd import harness
    a, _, _ = load_golden("evaluators")
    ev = harness.ConfusionEvaluator(0.5, DEV)
    for i in range(3):
        ev.update(dev(a["syn/logits"][i]), dev(a["syn/gt"][i]))
    r = ev.results()
    assert [r[1]["correct"], r[1]["seen"], r[1]["iou_de"], r[1]["predicted"], r["seen_all"]] == a["syn/counts"].tolist()
    np.testing.assert_allclose([r[1]["precision"], r[1]["accuracy"], r[1]["F1"], r[1]["iou"]], a["syn/metrics"].numpy(), rtol=1e-12)
    er = harness.ConfusionEvaluator(0.35, DEV)
    for i in range(3):
        er.update(dev(a["real/logits"][i]), dev(a["real/gt"][i]), dev(a["real/mask"][i]))
    r = er.results()
    for l in (0, 1):
        assert [r[l]["correct"], r[l]["seen"], r[l]["iou_de"], r[l]["predicted"]] == [float(a[k][l]) for k in ("real/correct", "real/seen", "real/iou_de", "real/predicted")]
    np.testing.assert_allclose([r[0]["F1"], r[1]["F1"]], a["real/F1"].numpy(), rtol=1e-12)
    np.testing.assert_allclose(r["accuracy_all"], float(a["real/accuracy_all"]), rtol=1e-12)
    col = harness.AnomalyCollector(3, a["col/timesteps"].tolist(), 4, 5, 8, DEV)
    for i in range(6):
        col.update(dev(a["col/anomaly_batches"][i]).long(), a["col/timestep_batches"][i].tolist())
    out = col.majority_vote().cpu().numpy()
    assert np.array_equal(col.cnt.cpu().numpy(), a["col/counter"].numpy())
    assert np.array_equal(np.nan_to_num(out, nan=-1), np.nan_to_num(a["col/result"].numpy(), nan=-1))
    # checkpoint: reference dict keys and parameter keys
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd import optim
    cfg = default_config(encoder="Swin_3D", in_channels=1, in_channels_dynamic=2, dir_log=str(tmp_path), name="t")
    m = VQ_model(cfg).to(DEV)
    opt = optim.Adam(m.parameters(), lr=1e-3)
    path = harness.save_model(m, opt, 3, 1.5, 2.5, cfg, "loss")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "mean_loss_train", "mean_loss_validation", "model_state_dict", "optimizer_state_dict"}
    assert path.endswith("model_checkpoints/best_loss_model.pth")
    m2 = VQ_model(cfg)
    m2.load_state_dict({k.replace("module.", ""): v for k, v in ck["model_state_dict"].items()}, strict=True)


@pytest.mark.parametrize("tag", ["16_32", "32_16"])
@variants('default', 'swin_split', 'swin_f22')
def test_swin_hierarchical_encoder_golden(tag, kernel_variant):
    """Swin_3D with a width change between its stages (the reference re-embeds with PatchEmbed3D(in_dim -> dim) + LN,
    Swin_3D.py:414-419): outputs and every gradient against the reference's own."""
    from idee_amd.models.encoder.Swin_3D import Swin_3D
    a, sd, grads = load_golden("swin_hier_" + tag)
    enc = Swin_3D(in_vars=2, in_chans=1, embed_dim=[int(v) for v in a["embed_dim"]], depths=[2, 1], num_heads=[2, 2])
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(DEV).train()
    y = enc(dev(a["x"]))
    close("y", y, a["y"])
    (y * dev(a["r"])).sum().backward()
    ps = dict(enc.named_parameters())
    checked = 0
    for key, name, idx in enc._key_map():
        if name.endswith("_index"):
            continue
        base = name[:-1] if name.endswith("_ds_w5") else name
        g = ps[base].grad
        g = g.view(*g.shape, 1, 1, 1) if name.endswith("_ds_w5") else g
        gclose("d" + key, g if idx is None else g[idx], grads[key], rtol=2e-3, floor=2e-4)
        checked += 1
    assert checked == len(grads)


@pytest.mark.parametrize("enc_name", ["Swin_3D", "CNN_3D"])
def test_encoder_without_patch_embedding_golden(enc_name):
    """in_chans == embed_dim[0] with patch (1,1,1): the reference builds stage 0 (Swin_3D.py:417-420) / the first conv_block
    (encoder/CNN_3D.py:112-118) without a `downsample`; state-dict keys as the reference's, outputs, every parameter gradient and
    the INPUT gradient against the reference's own."""
    if enc_name == "Swin_3D":
        from idee_amd.models.encoder.Swin_3D import Swin_3D
        a, sd, grads = load_golden("swin_nopatch_c16")
        enc = Swin_3D(in_vars=2, in_chans=16, depths=[2, 1], num_heads=[2, 2])
    else:
        from idee_amd.models.encoder.CNN_3D import CNN_3D
        a, sd, grads = load_golden("cnn_nopatch_c16")
        enc = CNN_3D(in_vars=2, in_channels=16, out_channels=[16, 16])
    msd = enc.state_dict()
    assert list(msd) == list(sd) and all(tuple(msd[k].shape) == tuple(sd[k].shape) for k in sd)
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(DEV).train()
    x = dev(a["x"]).requires_grad_()
    y = enc(x)
    close("y", y, a["y"])
    (y * dev(a["r"])).sum().backward()
    gclose("dx", x.grad, a["dx"], rtol=2e-3, floor=2e-4)
    ps = dict(enc.named_parameters())
    checked = 0
    for key, name, idx in enc._key_map():
        if name.endswith("_index") or key not in grads:
            continue
        g = ps[name].grad
        g = g if idx is None else g[idx]
        gclose("d" + key, g.reshape(grads[key].shape), grads[key], rtol=2e-3, floor=2e-4)
        checked += 1
    assert checked == len(grads)


@pytest.mark.parametrize("tag", ["122", "222_c2"])
def test_swin_encoder_patch_size_golden(tag):
    """Swin_3D(patch_size != (1,1,1)) (en_patch_size; PatchEmbed3D with kernel = stride = patch incl. its zero padding,
    Swin_3D.py:473-491): state-dict keys / shapes as the reference's, outputs and every gradient against the reference's own."""
    from idee_amd.models.encoder.Swin_3D import Swin_3D
    a, sd, grads = load_golden("swin_patch_" + tag)
    enc = Swin_3D(in_vars=2, in_chans=int(a["in_chans"]), patch_size=tuple(int(v) for v in a["patch"]), depths=[2, 1], num_heads=[2, 2])
    msd = enc.state_dict()
    assert list(msd) == list(sd) and all(tuple(msd[k].shape) == tuple(sd[k].shape) for k in sd)
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(DEV).train()
    y = enc(dev(a["x"]))
    close("y", y, a["y"])
    (y * dev(a["r"])).sum().backward()
    ps = dict(enc.named_parameters())
    checked = 0
    for key, name, idx in enc._key_map():
        if name.endswith("_index"):
            continue
        base = name[:-1] if name.endswith("_ds_w5") else name
        g = ps[base].grad
        g = g if idx is None else g[idx]
        gclose("d" + key, g.reshape(grads[key].shape), grads[key], rtol=2e-3, floor=2e-4)
        checked += 1
    assert checked == len(grads)


@pytest.mark.parametrize("tag", ["16_32", "32_16"])
def test_cnn_hierarchical_encoder_golden(tag):
    """CNN_3D with a width change after its first block (conv_block's downsample = bias-free PatchEmbed3D(in -> out) + LN,
    encoder/CNN_3D.py:112-118; then two conv + LN + ReLU units and proj_var at the new width): outputs and every gradient
    against the reference's own."""
    from idee_amd.models.encoder.CNN_3D import CNN_3D
    a, sd, grads = load_golden("cnn_hier_" + tag)
    enc = CNN_3D(in_vars=2, in_channels=1, out_channels=[int(v) for v in a["out_channels"]])
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(DEV).train()
    y = enc(dev(a["x"]))
    close("y", y, a["y"])
    (y * dev(a["r"])).sum().backward()
    ps = dict(enc.named_parameters())
    checked = 0
    for key, name, idx in enc._key_map():
        base = name[:-1] if name.endswith("_ds_w5") else name
        g = ps[base].grad
        g = g.view(*g.shape, 1, 1, 1) if name.endswith("_ds_w5") else g
        gclose("d" + key, g if idx is None else g[idx], grads[key], rtol=2e-3, floor=2e-4)
        checked += 1
    assert checked == len(grads)


@pytest.mark.parametrize("tag", ["plain", "kmeans", "expire"])
def test_vq_ema_kmeans_expire_golden(tag):
    """VQ with the EMA codebook update, k-means initialisation and dead-code expiry against three training steps of the
    reference's VQ (its sample choice made deterministic: the first `num` batch samples) — outputs, dx and the codebook
    buffers after every step."""
    from idee_amd.models.codebook.VQ import VQ
    a, _, _ = load_golden("vq_ema")
    kw = dict(plain={}, kmeans=dict(kmeans_init=True, kmeans_iters=4), expire=dict(threshold_ema_dead_code=2))[tag]
    q = VQ(dim=16, codebook_size=8, codebook_dim=16, ema_update=True, learnable_codebook=False, decay=0.8, **kw)
    q.load_state_dict({k[len(tag) + 5:]: v for k, v in a.items() if k.startswith(f"{tag}/sd0/")}, strict=True)
    q.sample_indices = lambda n, num, device: torch.arange(num, device=device)
    q = q.to(DEV).train()
    for step in range(3):
        g = lambda k: a[f"{tag}/s{step}/{k}"]
        x = dev(g("x")).requires_grad_()
        zq, idx, loss = q(x)
        assert torch.equal(idx.cpu(), g("indices").long())
        close("z_q", zq, g("z_q"))
        close("loss", loss, g("loss"), atol=1e-6, rtol=1e-4)
        ((zq * dev(g("r"))).sum() + 1.7 * loss.sum()).backward()
        gclose("dx", x.grad, g("dx"))
        sd = q.state_dict()
        for k in ("_codebook.embed", "_codebook.cluster_size", "_codebook.embed_avg"):
            close(f"step {step} {k}", sd[k], g("sd/" + k), atol=2e-5, rtol=1e-4)
        assert bool(sd["_codebook.initted"])


@variants('default', 'swin_split', 'swin_f22')
def test_swin_block_qkv_bias_false_golden(kernel_variant):
    """SwinTransformerBlock3D built with qkv_bias=False (Swin_3D.py:137): the kernels get a zero, untrained bias."""
    ops = _ops()
    a, _, _ = load_golden("misc")
    sd = {k[10:]: v for k, v in a.items() if k.startswith("nobias/sd/")}
    grads = {k[12:]: v for k, v in a.items() if k.startswith("nobias/grad/")}
    assert "attn.qkv.bias" not in sd
    names = ["attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "mlp.fc1.weight", "mlp.fc1.bias",
             "mlp.fc2.weight", "mlp.fc2.bias", "attn.relative_position_bias_table"]
    params = [(dev(sd[n]).unsqueeze(0).contiguous().requires_grad_() if n in sd else torch.zeros(1, 48, device=DEV)) for n in names]
    x = dev(a["nobias/x"]).unsqueeze(1).contiguous().requires_grad_()
    y = ops.SwinBlock.apply(x, *params, dev(sd["attn.relative_position_index"]), 2, (2, 4, 4), (1, 2, 2), None)
    close("y", y[:, 0], a["nobias/y"])
    (y[:, 0] * dev(a["nobias/r"])).sum().backward()
    gclose("dx", x.grad[:, 0], a["nobias/dx"])
    for p_, n in zip(params, names):
        if n in grads:
            gclose("d" + n, p_.grad[0], grads[n])


@pytest.mark.parametrize("C", [16, 32])
@variants('default', 'swin_split', 'swin_f22')
def test_swin_block_drop_path_vs_oracle(C, kernel_variant):
    """Stochastic depth: the kernels take the per-sample DropPath factors (mask / keep_prob, incl. dropped samples = 0) of
    both residual branches; forward and every gradient against the oracle with the same factors.  C = 32 runs the exact-f32
    kernels, C = 16 the split-precision ones."""
    from oracle import idee_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(70 + C)
    N, V, D, H, W, NH = 3, 2, 8, 8, 12, 2
    x = torch.randn((N, V, D, H, W, C), generator=g)
    r = torch.randn((N, V, D, H, W, C), generator=g)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C), "attn.proj.bias": (C,),
              "mlp.fc1.weight": (4 * C, C), "mlp.fc1.bias": (4 * C,), "mlp.fc2.weight": (C, 4 * C), "mlp.fc2.bias": (C,),
              "attn.relative_position_bias_table": (147, 2)}
    sds = [{k: (torch.randn(s, generator=g) * (0.7 / s[-1] ** 0.5 if len(s) > 1 else 0.1)).requires_grad_() for k, s in shapes.items()} for _ in range(V)]
    ps = torch.tensor([[2.0, 0.0], [0.0, 2.0], [2.0, 2.0], [0.0, 0.0], [1.25, 1.25], [1.0, 1.0]])      # (N*V, 2), item b = n*V + v
    index = O.relative_position_index((2, 4, 4))
    xr = x.clone().requires_grad_()
    outs = []
    for v in range(V):
        sdv = dict(sds[v]); sdv["attn.relative_position_index"] = index
        outs.append(O.swin_block(xr[:, v], sdv, "", NH, (2, 4, 4), (1, 2, 2), path_scale=ps.reshape(N, V, 2)[:, v]))
    yo = torch.stack(outs, 1)
    (yo * r).sum().backward()
    names = list(shapes)
    params = [torch.stack([sds[v][n].detach() for v in range(V)]).to(DEV).requires_grad_() for n in names]
    xd = x.to(DEV).requires_grad_()
    y = ops.SwinBlock.apply(xd, *params, index.to(DEV), NH, (2, 4, 4), (1, 2, 2), None, ps.to(DEV))
    close("y", y, yo)
    assert torch.equal(y[1, 1].cpu(), x[1, 1]), "a sample with both branches dropped passes through unchanged"
    (y * r.to(DEV)).sum().backward()
    gclose("dx", xd.grad, xr.grad)
    for p_, n in zip(params, names):
        gclose("d" + n, p_.grad, torch.stack([sds[v][n].grad for v in range(V)]))


@pytest.mark.parametrize("C,window,shift,D,H,W", [(16, (2, 4, 4), (1, 2, 2), 4, 6, 10), (16, (2, 4, 4), (0, 0, 0), 4, 8, 8),
                                                   (16, (8, 1, 1), (0, 0, 0), 8, 3, 5), (32, (2, 4, 4), (1, 2, 2), 2, 8, 12),
                                                   (32, (8, 1, 1), (4, 0, 0), 16, 2, 3)])
@variants('default', 'swin_f22', 'swin_split')
def test_swin_block_elementwise_dropout_vs_oracle(C, window, shift, D, H, W, kernel_variant):
    """(Under IDEE_SWIN_MATH = f22 / split a block with a non-zero rate is routed to the exact-f32 kernels, forward and backward.)
    Element-wise dropout INSIDE the fused block (en_drop_rate: Mlp.drop after GELU and after fc2 + proj_drop; en_attn_drop_rate
    on the attention probabilities; Swin_3D.py:34-41, :138-140, :172, :176): output and every gradient against the oracle run
    with the SAME keep masks (oracle.swin_dropout_masks restates the kernels' counter-based generator; the reference's own
    masks come from torch's Philox stream, so parity is defined given the masks).  Shifted / padded / temporal windows, C = 16
    and 32; the backward regenerates the masks from the seed."""
    from oracle import idee_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(11 + C + D)
    N, V, NH = 2, 2, 2
    p_drop, p_attn, seed = 0.15, 0.25, 1234567 + C
    L_ = (2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C), "attn.proj.bias": (C,),
              "mlp.fc1.weight": (4 * C, C), "mlp.fc1.bias": (4 * C,), "mlp.fc2.weight": (C, 4 * C), "mlp.fc2.bias": (C,),
              "attn.relative_position_bias_table": (L_, NH)}
    x = torch.randn((N, V, D, H, W, C), generator=g)
    r = torch.randn((N, V, D, H, W, C), generator=g)
    sds = [{k: (torch.randn(s_, generator=g) * (0.3 if len(s_) > 1 else 0.1)).requires_grad_() for k, s_ in shapes.items()} for _ in range(V)]
    index = O.relative_position_index(window)
    # the kernels number tokens over the whole (N, V, D, H, W) tensor: build the masks for B = N * V items, item b = n * V + v
    masks = O.swin_dropout_masks(seed, p_drop, p_attn, N * V, D, H, W, C, NH, window, shift)
    ws, _ = O.get_window_size((D, H, W), window, shift)
    nwin = masks["attn"].shape[0] // (N * V)
    xr = x.clone().requires_grad_()
    outs = []
    for v in range(V):
        sdv = dict(sds[v]); sdv["attn.relative_position_index"] = index
        items = [n * V + v for n in range(N)]                      # rows of the (N*V)-item tensor that belong to variable v
        mv = dict(proj=masks["proj"][items], gelu=masks["gelu"][items], fc2=masks["fc2"][items],
                  attn=torch.cat([masks["attn"][b * nwin:(b + 1) * nwin] for b in items], 0))
        outs.append(O.swin_block(xr[:, v], sdv, "", NH, window, shift, dropout=mv))
    yo = torch.stack(outs, 1)
    (yo * r).sum().backward()
    names = list(shapes)
    params = [torch.stack([sds[v][n].detach() for v in range(V)]).to(DEV).requires_grad_() for n in names]
    xd = x.to(DEV).requires_grad_()
    y = ops.SwinBlock.apply(xd, *params, index.to(DEV), NH, window, shift, None, None, p_drop, p_attn, seed)
    drop_frac = float((masks["fc2"] == 0).float().mean())
    assert 0.1 < drop_frac < 0.2
    close("y", y, yo)
    (y * r.to(DEV)).sum().backward()
    gclose("dx", xd.grad, xr.grad)
    for p_, n in zip(params, names):
        gclose("d" + n, p_.grad, torch.stack([sds[v][n].grad for v in range(V)]))
    # without dropout the same call must differ (the masks are actually applied), and eval-mode rates of 0 reproduce the plain block
    y0 = ops.SwinBlock.apply(xd.detach(), *[p_.detach() for p_ in params], index.to(DEV), NH, window, shift, None, None)
    assert float((y0 - y.detach()).abs().max()) > 1e-3


def test_swin_encoder_dropout_module_flags(set_opt):
    """Swin_3D(drop_rate, attn_drop_rate): training forward / backward run with fresh masks per step (two steps differ), eval is
    deterministic and equals the rate-0 encoder; under IDEE_SWIN_MATH=f22 the dropout blocks run (on the exact kernels).  (bf16
    activation storage with a non-zero rate: tests/test_bf16_path.py.)"""
    from idee_amd.models.encoder.Swin_3D import Swin_3D
    from idee_amd import _lib as L
    torch.manual_seed(5)
    enc = Swin_3D(in_vars=2, in_chans=1, drop_rate=0.1, attn_drop_rate=0.1).to(DEV)
    ref = Swin_3D(in_vars=2, in_chans=1).to(DEV)
    ref.load_state_dict(enc.state_dict())
    x = torch.randn(1, 2, 1, 8, 8, 12, device=DEV)
    enc.train()
    y1 = enc(x); y2 = enc(x)
    assert float((y1 - y2).abs().max()) > 1e-4
    y1.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())
    enc.eval(); ref.eval()
    with torch.no_grad():
        torch.testing.assert_close(enc(x), ref(x), atol=0, rtol=0)
    enc.train()
    set_opt("IDEE_SWIN_MATH", "f22")
    yf = enc(x)
    yf.sum().backward()
    assert torch.isfinite(yf).all() and float((yf - y1).abs().max()) > 1e-4


@variants('default', 'swin_split', 'swin_f22')
def test_swin_encoder_drop_path_rate_module(kernel_variant):
    """Swin_3D(drop_path_rate > 0): identity in eval mode, stochastic in training (the decay rule gives block 0 rate 0)."""
    from idee_amd.models.encoder.Swin_3D import Swin_3D
    torch.manual_seed(3)
    enc = Swin_3D(in_vars=2, in_chans=1, drop_path_rate=0.5).to(DEV)
    x = torch.randn(4, 2, 1, 8, 8, 8, device=DEV)
    enc.eval()
    with torch.no_grad():
        y0, y1 = enc(x), enc(x)
    assert torch.equal(y0, y1)
    enc.train()
    ya, yb = enc(x), enc(x)
    assert not torch.equal(ya, yb) and torch.isfinite(ya).all()
    ya.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())


@variants('default', 'conv_split', 'conv_f22')
def test_conv_block_drop_path_vs_oracle(kernel_variant):
    """CNN_3D encoder unit with stochastic depth: y = x + s_b * ReLU(LN(conv(x))) with the per-item factor s_b (0 = dropped),
    forward and gradients against the oracle formula."""
    import torch.nn.functional as F
    ops = _ops()
    g = torch.Generator().manual_seed(91)
    N, V, T, H, W, C = 2, 2, 4, 6, 20, 16
    x = torch.randn((N, V, T, H, W, C), generator=g)
    r = torch.randn((N, V, T, H, W, C), generator=g)
    w = torch.randn((V, C, C, 3, 3, 3), generator=g) * 0.06
    gam, bet = 1 + 0.1 * torch.randn((V, C), generator=g), 0.1 * torch.randn((V, C), generator=g)
    ps = torch.tensor([2.0, 0.0, 1.0, 2.0])                       # item b = n*V + v
    leaves = [t.clone().requires_grad_() for t in (x, w, gam, bet)]
    outs = []
    for v in range(V):
        xv = leaves[0][:, v].permute(0, 4, 1, 2, 3)
        c = F.conv3d(F.pad(xv, (1,) * 6, mode="replicate"), leaves[1][v])
        u = F.relu(F.layer_norm(c.permute(0, 2, 3, 4, 1), (C,), leaves[2][v], leaves[3][v]))
        outs.append(leaves[0][:, v] + u * ps.reshape(N, V)[:, v].reshape(N, 1, 1, 1, 1))
    yo = torch.stack(outs, 1)
    (yo * r).sum().backward()
    dl = [t.to(DEV).requires_grad_() for t in (x, w, gam, bet)]
    y = ops.ConvLNUnit.apply(*dl, ps.to(DEV))
    close("y", y, yo)
    (y * r.to(DEV)).sum().backward()
    for n, a_, b_ in zip(("dx", "dw", "dgamma", "dbeta"), dl, leaves):
        gclose(n, a_.grad, b_.grad, rtol=2e-3, floor=2e-4)


def test_classifier_dropout_statistics_and_gradients():
    """cls_drop_rate > 0 (nn.Dropout after the first ReLU of every head, classifier/CNN_3D.py:41,53,93,133): identity in eval
    mode; repeatable under torch.manual_seed; the keep-mask generator drops a fraction p; and — with the second ReLU made
    inactive by a large positive bias, so that the head is LINEAR in the masked activation — the mean over many masks
    reproduces the eval output, which pins the 1/(1-p) scaling folded into conv2's weights."""
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    ops = _ops()
    t = torch.ones(1 << 20, device=DEV)
    ops._dropout_(t, 0.3, 12345)
    assert abs(float((t == 0).float().mean()) - 0.3) < 5e-3 and set(t.unique().tolist()) == {0.0, 1.0}
    t2 = torch.ones(1 << 20, device=DEV)
    ops._dropout_(t2, 0.3, 12345)
    assert torch.equal(t, t2)
    ops._dropout_(t2, 0.3, 999)
    assert abs(float((t2 == 0).float().mean()) - 0.51) < 1e-2          # an independent second mask: 1 - 0.7^2
    torch.manual_seed(5)
    V, N, T, H, W = 2, 2, 8, 12, 16
    c = CNN_3D(in_var=V, embed_dim=16, dim=16, drop_rate=0.5).to(DEV)
    with torch.no_grad():
        c.j_b2.fill_(50.0)
        c.h_b2.fill_(50.0)
    x = torch.randn(N, V, 16, T, H, W, device=DEV)
    c.eval()
    with torch.no_grad():
        z0, y0 = c(x)
        z1, _ = c(x)
    assert torch.equal(z0, z1)
    c.train()
    torch.manual_seed(1)
    za, ya = c(x)
    torch.manual_seed(1)
    zb, yb = c(x)
    assert torch.equal(za, zb), "same torch seed -> same dropout masks"
    zc, _ = c(x)
    assert not torch.equal(za, zc)
    acc, accy = torch.zeros_like(z0), torch.zeros_like(torch.stack(y0))
    n_draw = 200
    with torch.no_grad():
        for _ in range(n_draw):
            zz, yy = c(x)
            acc += zz
            accy += torch.stack(yy)
    for name, mean, ref in (("joint", acc / n_draw, z0), ("heads", accy / n_draw, torch.stack(y0))):
        signal = (ref - ref.mean()).abs().mean()
        err = (mean - ref).abs().mean() / signal
        assert float(err) < 0.05, (name, float(err))
    xg = x.clone().requires_grad_()
    z, y = c(xg)
    (z.sum() + sum(t_.sum() for t_ in y)).backward()
    assert torch.isfinite(xg.grad).all() and all(torch.isfinite(p_.grad).all() for p_ in c.parameters())
