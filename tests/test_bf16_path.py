"""-m gpu: the bf16 activation path (idee_hip.h IDEE_DTYPE_BF16; BASELINE config #4 "bf16 activations/GEMMs with fp32
accumulate and fp32 LFQ") against f32 references.

Tolerances.  A bf16 kernel receives bf16 operands, forms their products exactly, accumulates in f32 and rounds its output
to bf16 once (relative 2^-9 = 2e-3).  Kernel-level tests therefore feed the reference THE SAME bf16-rounded inputs and
weights and allow  |err| <= 2^-8 |ref| + 2^-8 * rms(ref)  (one output rounding + accumulation-order noise; stated per test).
Model-level tests compare with the f32 oracle and state the accumulated tolerance; LFQ sign flips (tokens whose projection
lies inside the bf16 noise band) are COUNTED and printed, never averaged in."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import variants

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def rb(t):
    """round to bf16 and back: the value a bf16 kernel actually reads"""
    return t.to(BF).float()


def close_bf16(name, got, ref, rel=2.0 ** -8, scale=2.0 ** -8):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite"
    rms = float(ref.pow(2).mean().sqrt())
    tol = rel * ref.abs() + scale * rms + 1e-30
    err = (got - ref).abs()
    bad = err > tol
    if bad.any():
        i = int((err - tol).argmax())
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} out of tolerance; worst got {float(got.flatten()[i]):.6g} "
                             f"ref {float(ref.flatten()[i]):.6g} (rms {rms:.3e})")


def ref_conv(mode, x, w, b):
    """x (B,Cin,T,H,W), w (Cout,Cin,kt,3,3) f32 on CPU -> conv of the given mode"""
    from idee_amd import _lib as L
    if mode == L.CONV_K333_REPLICATE:
        return F.conv3d(F.pad(x, (1, 1, 1, 1, 1, 1), mode="replicate"), w, b)
    if mode == L.CONV_K233_S2_ZERO:
        return F.conv3d(x, w, b, stride=(2, 1, 1), padding=(0, 1, 1))
    return F.conv3d(x, w, b, padding=(0, 1, 1))


CONV_CASES = [
    # mode name, wsets, N (items per weight set), Cin, Cout, T, H, W
    ("k333", 2, 1, 16, 16, 3, 5, 37),
    ("k333", 3, 2, 32, 32, 4, 6, 70),
    ("k333", 1, 1, 32, 32, 1, 1, 1),
    ("k333", 1, 1, 16, 32, 2, 3, 64),
    ("k333", 2, 1, 32, 32, 3, 4, 129),
    ("k333", 1, 1, 32, 32, 9, 7, 66),          # the LDS ring over many frames, a 2-pixel last tile, 3 idle rows
    ("k233", 1, 1, 32, 32, 10, 9, 64),
    ("k133", 1, 1, 32, 16, 5, 8, 64),
    ("k133", 2, 1, 32, 32, 3, 5, 65),          # (1,3,3) 32 -> 32 on the weight-stationary kernel
    ("k333", 1, 1, 32, 32, 2, 2, 63),          # both W edges in one tile, T = 2: every frame and row is a border
    ("k333", 1, 1, 32, 32, 1, 9, 130),         # T = 1: both temporal folds on the same frame
    ("k233", 2, 1, 32, 32, 7, 5, 64),          # odd T (the last frame has no output)
    ("k233", 1, 1, 64, 96, 4, 11, 50),         # wide input: the channel-chunked LDS-DMA kernel, 2-pixel last tile, 3 idle rows
    ("k233", 2, 1, 192, 192, 3, 9, 47),        # odd T through the chunked kernel (NT = 6), a tile one pixel short
    ("k133", 1, 2, 64, 16, 2, 17, 97),         # (1,3,3) wide -> 16
    ("k233", 3, 1, 32, 32, 4, 5, 40),
    ("k233", 1, 2, 96, 96, 4, 4, 33),
    ("k233", 1, 1, 192, 192, 2, 3, 20),
    ("k233", 2, 1, 16, 16, 6, 3, 65),
    ("k133", 1, 2, 16, 96, 3, 5, 40),
    ("k133", 1, 1, 16, 192, 2, 4, 70),
    ("k133", 1, 1, 16, 32, 2, 3, 17),
    ("k133", 1, 1, 96, 16, 2, 4, 40),
]


@variants("default", "conv_bf16_direct", "conv_bf16_lds")
@pytest.mark.parametrize("name,wsets,N,Cin,Cout,T,H,W", CONV_CASES)
def test_conv_bf16_fwd_dgrad_wgrad_vs_torch(name, wsets, N, Cin, Cout, T, H, W, kernel_variant):
    """Every pass of the bf16 convolution kernels (conv_bf16.hip) against torch conv3d on the same bf16-rounded operands:
    forward with bias + ReLU, data gradient with the ReLU mask, weight and bias gradient."""
    from idee_amd import ops
    from idee_amd import _lib as L
    mode = dict(k333=L.CONV_K333_REPLICATE, k233=L.CONV_K233_S2_ZERO, k133=L.CONV_K133_ZERO)[name]
    kt = dict(k333=3, k233=2, k133=1)[name]
    g = torch.Generator().manual_seed(hash((name, Cin, Cout, W)) & 0xffff)
    B = N * wsets
    x = rb(torch.randn((B, T, H, W, Cin), generator=g))
    w = rb(torch.randn((wsets, Cout, Cin, kt, 3, 3), generator=g) / math.sqrt(Cin * kt * 9))
    b = torch.randn((wsets, Cout), generator=g) * 0.1
    To = (T - 2) // 2 + 1 if name == "k233" else T
    r = rb(torch.randn((B, To, H, W, Cout), generator=g))
    # reference, item by item (item i uses weight set i % wsets)
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    br = b.clone().requires_grad_()
    ys = []
    for i in range(B):
        v = i % wsets
        ys.append(ref_conv(mode, xr[i:i + 1].permute(0, 4, 1, 2, 3), wr[v], br[v]).permute(0, 2, 3, 4, 1))
    pre = torch.cat(ys, 0)
    yref = F.relu(pre)
    (yref * r).sum().backward()
    # HIP
    conv = ops.ConvLayer(mode, B, wsets, T, H, W, Cin, Cout, ops._cl_strides(T, H, W, Cin), ops._cl_strides(To, H, W, Cout), True,
                         act=BF)
    wd = w.to(DEV)
    pf, pd = conv.pack(wd)
    assert pf.dtype == BF
    xd = x.to(DEV).to(BF)
    y = torch.empty((B, To, H, W, Cout), device=DEV, dtype=BF)
    conv.fwd(xd, pf, b.to(DEV), y)
    close_bf16("y", y, yref)
    # data gradient: dx = conv^T(dy) masked by ReLU'(relu_ref) — relu_ref has dx's layout (in the model it is the ReLU output
    # that IS this convolution's input); here the sign pattern of x itself
    dy = r.to(DEV).to(BF)
    dx = torch.empty_like(xd)
    conv.dgrad(dy, pd, xd, dx)
    xr2 = x.clone().requires_grad_()
    ys2 = []
    for i in range(B):
        v = i % wsets
        ys2.append(ref_conv(mode, xr2[i:i + 1].permute(0, 4, 1, 2, 3), w[v], None).permute(0, 2, 3, 4, 1))
    (torch.cat(ys2, 0) * r).sum().backward()
    close_bf16("dx", dx, xr2.grad * (x > 0).float())
    mask = (y.float().cpu() > 0).float()
    # weight / bias gradient of the masked upstream gradient (what the backward of conv + ReLU hands to wgrad)
    dym = rb(r * mask)
    wr3 = w.clone().requires_grad_()
    br3 = b.clone().requires_grad_()
    ys3 = []
    for i in range(B):
        v = i % wsets
        ys3.append(ref_conv(mode, x[i:i + 1].permute(0, 4, 1, 2, 3), wr3[v], br3[v]).permute(0, 2, 3, 4, 1))
    (torch.cat(ys3, 0) * dym).sum().backward()
    dw, db = conv.wgrad(xd, dym.to(DEV).to(BF), wd)
    # weight gradients are f32 sums of exact bf16 products: only summation order differs
    torch.testing.assert_close(dw.cpu(), wr3.grad, atol=2e-4 * float(wr3.grad.abs().max()) + 1e-6, rtol=2e-4)
    torch.testing.assert_close(db.cpu(), br3.grad, atol=2e-4 * float(br3.grad.abs().max()) + 1e-6, rtol=2e-4)


@variants("default", "conv_bf16_direct", "conv_bf16_lds")
def test_conv_bf16_addend_and_grouped_layouts(kernel_variant):
    """dgrad with accumulate (dx += ...) and the two-level channel-group addressing the classifier uses (the heads' conv1
    writes the V slices of its output; the dense joint head reads the V slices of z_q)."""
    from idee_amd import ops
    from idee_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    N, V, T, H, W, dim, CS = 2, 3, 4, 5, 40, 32, 16
    x = rb(torch.randn((N, T, H, W, CS), generator=g))
    w = rb(torch.randn((1, V * dim, CS, 1, 3, 3), generator=g) * 0.1)
    conv = ops.ConvLayer(L.CONV_K133_ZERO, N, 1, T, H, W, CS, V * dim, ops._cl_strides(T, H, W, CS),
                         (V * T * H * W * dim, H * W * dim, W * dim, dim, 16), False, y_units=(dim // 16, T * H * W * dim), act=BF)
    pf, pd = conv.pack(w.to(DEV))
    y = torch.empty((N, V, T, H, W, dim), device=DEV, dtype=BF)
    conv.fwd(x.to(DEV).to(BF), pf, None, y)
    yref = F.conv3d(x.permute(0, 4, 1, 2, 3), w[0], None, padding=(0, 1, 1))            # (N, V*dim, T, H, W)
    yref = yref.reshape(N, V, dim, T, H, W).permute(0, 1, 3, 4, 5, 2)
    close_bf16("y grouped", y, yref)
    # accumulate: dx = addend + conv^T(dy)
    dy = rb(torch.randn((N, V, T, H, W, dim), generator=g))
    add = rb(torch.randn((N, T, H, W, CS), generator=g))
    dx = add.to(DEV).to(BF).clone()
    conv.dgrad(dy.to(DEV).to(BF), pd, None, dx, accumulate=True)
    xr = x.clone().requires_grad_()
    yr = F.conv3d(xr.permute(0, 4, 1, 2, 3), w[0], None, padding=(0, 1, 1)).reshape(N, V, dim, T, H, W).permute(0, 1, 3, 4, 5, 2)
    (yr * dy).sum().backward()
    close_bf16("dx accumulate", dx, xr.grad + add)


@pytest.mark.parametrize("C,window,shift,D,H,W", [(32, (2, 4, 4), (1, 2, 2), 4, 10, 22), (32, (8, 1, 1), (4, 0, 0), 16, 3, 9),
                                                   (16, (2, 4, 4), (0, 0, 0), 4, 8, 16), (32, (2, 4, 4), (1, 2, 2), 2, 4, 424)])
def test_swin_block_bf16_vs_oracle(C, window, shift, D, H, W):
    """One Swin block on bf16 tokens (one-piece bf16 GEMM operands, f32 LayerNorm / softmax / GELU / scores) against the f32
    oracle on the same bf16-rounded input: outputs to 2^-6 of their scale (a chain of ~8 bf16-rounded GEMM operands), every
    gradient to 3 % of its largest magnitude.  Includes the config-#4 width W = 424 at C = 32."""
    from oracle import idee_oracle as O
    from idee_amd import ops
    g = torch.Generator().manual_seed(C + W)
    N, V, NH = 1, 2, 2
    L_ = (2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C), "attn.proj.bias": (C,),
              "mlp.fc1.weight": (4 * C, C), "mlp.fc1.bias": (4 * C,), "mlp.fc2.weight": (C, 4 * C), "mlp.fc2.bias": (C,),
              "attn.relative_position_bias_table": (L_, NH)}
    x = rb(torch.randn((N, V, D, H, W, C), generator=g))
    r = rb(torch.randn((N, V, D, H, W, C), generator=g))
    sds = [{k: (torch.randn(s, generator=g) * (0.5 / math.sqrt(s[-1]) if len(s) > 1 else 0.1)).requires_grad_() for k, s in shapes.items()}
           for _ in range(V)]
    index = O.relative_position_index(window)
    ws, ss = O.get_window_size((D, H, W), window, shift)
    xr = x.clone().requires_grad_()
    outs = []
    for v in range(V):
        sdv = dict(sds[v]); sdv["attn.relative_position_index"] = index
        outs.append(O.swin_block(xr[:, v], sdv, "", NH, window, shift))
    yo = torch.stack(outs, 1)
    (yo * r).sum().backward()
    names = list(shapes)
    params = [torch.stack([sds[v][n].detach() for v in range(V)]).to(DEV).requires_grad_() for n in names]
    xd = x.to(DEV).to(BF).requires_grad_()
    y = ops.SwinBlock.apply(xd, *params, index.to(DEV), NH, window, shift, None)
    assert y.dtype == BF
    close_bf16("y", y, yo, rel=2.0 ** -6, scale=2.0 ** -6)
    (y.float() * r.to(DEV)).sum().backward()
    def gclose(name, got, ref):
        got, ref = got.detach().float().cpu(), ref.detach()
        err = float((got - ref).abs().max())
        assert err <= 0.03 * float(ref.abs().max()) + 1e-6, f"{name}: max err {err:.3e} vs max |ref| {float(ref.abs().max()):.3e}"
    gclose("dx", xd.grad, xr.grad)
    for p_, n in zip(params, names):
        gclose("d" + n, p_.grad, torch.stack([sds[v][n].grad for v in range(V)]))


def _bf16_model(tag="swin_v2_14x18", **extra):
    from conftest import load_golden
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    a, sd, _ = load_golden("e2e_" + tag)
    kw = dict(encoder="Swin_3D", in_channels=int(a["cfg_C"]), in_channels_dynamic=int(a["cfg_V"]), act_dtype="bf16")
    kw.update(extra)
    cfg = default_config(**kw)
    m = VQ_model(cfg)
    return a, sd, cfg, m


def test_e2e_bf16_forward_vs_golden_with_flip_count():
    """VQ_model with bf16 activation storage against the reference's own f32 outputs (golden e2e fixture): LFQ code flips are
    counted (tokens whose f32 projection |s| lies inside the bf16 noise band may legitimately take the other code); logits are
    compared at pixels whose 7x7 neighbourhood holds no flipped token, at 5e-2 of the logit scale."""
    a, sd, cfg, m = _bf16_model()
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        z, y, an, zq, lz = m(a["x_d"].to(DEV))
    ref_an = a["eval/anomaly"].long()
    flips = (an.cpu() != ref_an)
    frac = float(flips.float().mean())
    print(f"[bf16 e2e] LFQ code flips vs the f32 reference: {int(flips.sum())}/{flips.numel()} ({100 * frac:.3f} %)")
    assert frac < 0.02
    assert z.dtype == torch.float32 and zq.dtype == BF
    bad_px = flips.any(dim=(1, 2)).float().unsqueeze(1)                          # (N,1,H,W)
    ok = (F.max_pool2d(bad_px, 7, 1, 3) == 0)
    print(f"[bf16 e2e] pixels with a flip-free 7x7 neighbourhood: {int(ok.sum())}/{ok.numel()}")
    assert float(ok.float().mean()) > 0.2          # (a 14 x 18 frame: each flipped token blanks up to 49 of its 252 pixels)
    zr = a["eval/z"]
    scale = float(zr.abs().max())
    err = ((z.cpu() - zr).abs() * ok).max()
    print(f"[bf16 e2e] joint logits: max err {float(err):.3e} at scale {scale:.3e} on {int(ok.sum())}/{ok.numel()} pixels")
    assert float(err) <= 5e-2 * scale + 1e-3
    safe = ~flips.unsqueeze(2).expand(-1, -1, zq.shape[2], -1, -1, -1)
    close_bf16("z_q", zq.float().cpu()[safe], a["eval/z_q"][safe], rel=2.0 ** -7, scale=2.0 ** -7)


@pytest.mark.parametrize("impl", ["LFQ", "VQ"])
def test_e2e_bf16_train_steps_reduce_the_loss(impl):
    """SyntheticTrainStep on the bf16 path (LFQ as wired by build.py, and the nearest-neighbour VQ codebook variant): finite
    losses and gradients for every parameter, f32 master parameters that move, a loss that falls over 8 steps on one batch."""
    from idee_amd.train import SyntheticTrainStep
    torch.manual_seed(0)          # the VQ variant starts from its own random codebook: make the trajectory repeatable
    a, sd, cfg, m = _bf16_model(codebook_impl=impl)
    if impl == "LFQ":
        m.load_state_dict(sd, strict=True)
    ts = SyntheticTrainStep(cfg, device=DEV, model=m)
    x = a["x_d"].to(DEV)
    g = torch.Generator().manual_seed(3)
    N, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    me = (torch.rand((N, H, W), generator=g) < 0.3).float().to(DEV)
    ml = (torch.rand((N, H, W), generator=g) < 0.4).float().to(DEV)
    before = ts.opt.flat_p.clone()
    losses = []
    for _ in range(8):
        total, parts, outs = ts.step(x, me, ml)
        assert torch.isfinite(total) and all(torch.isfinite(v) for v in parts.values())
        losses.append(float(total))
    assert torch.isfinite(ts.opt.flat_g).all() and ts.opt.flat_p.dtype == torch.float32
    assert float((ts.opt.flat_p != before).float().mean()) > 0.95
    print(f"[bf16 train {impl}] loss per step: {[round(v, 4) for v in losses]}")
    # LFQ (the golden model's trained-like weights): monotone enough that the last step is below the first.  VQ from a random
    # codebook: tokens switching between its two codes move the anomaly term by O(1) per switch, so only "some step improved"
    assert (losses[-1] < losses[0]) if impl == "LFQ" else (min(losses[1:]) < losses[0])


def test_bf16_storage_with_elementwise_dropout():
    """config.act_dtype = "bf16" with en_drop_rate / en_attn_drop_rate > 0 (reference: Swin_3D.py:34-41, 138-140, 172, 176): the
    dropout blocks run on the exact-f32 block kernels with their tokens widened for the block (models/encoder/Swin_3D.py).  Eval
    mode equals the rate-0 bf16 model bit for bit; training steps differ from step to step, every gradient is finite and the
    tokens handed on are bf16."""
    from idee_amd.train import SyntheticTrainStep
    a, sd, cfg, m = _bf16_model(en_drop_rate=0.1, en_attn_drop_rate=0.1)
    _, _, _, m0 = _bf16_model()
    m.load_state_dict(sd, strict=True); m0.load_state_dict(sd, strict=True)
    m, m0 = m.to(DEV), m0.to(DEV)
    x = a["x_d"].to(DEV)
    m.eval(); m0.eval()
    with torch.no_grad():
        o, o0 = m(x), m0(x)
    assert torch.equal(o[0], o0[0]) and torch.equal(o[2], o0[2]) and o[3].dtype == BF
    m.train()
    torch.manual_seed(1)
    z1 = m.encoder.forward_tokens(x); z2 = m.encoder.forward_tokens(x)
    assert z1.dtype == BF and float((z1.float() - z2.float()).abs().max()) > 1e-3
    ts = SyntheticTrainStep(cfg, device=DEV, model=m)
    g = torch.Generator().manual_seed(3)
    N, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    me = (torch.rand((N, H, W), generator=g) < 0.3).float().to(DEV)
    ml = (torch.rand((N, H, W), generator=g) < 0.4).float().to(DEV)
    total, parts, _ = ts.step(x, me, ml)
    assert torch.isfinite(total) and torch.isfinite(ts.opt.flat_g).all()


def test_bf16_storage_with_cnn3d_encoder():
    """config.act_dtype = "bf16" with encoder = CNN_3D (reference: models/encoder/CNN_3D.py:74-237): the encoder keeps f32 tokens
    inside (its LayerNorm / ReLU / residual kernels are f32-storage kernels) and hands bf16 tokens to the codebook and the classifier.
    Against the reference's own f32 outputs (golden e2e fixture): LFQ code flips counted, logits at flip-free pixels at 5e-2 of
    their scale; one train step with finite gradients for every parameter."""
    from conftest import load_golden
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd.train import SyntheticTrainStep
    a, sd, _ = load_golden("e2e_cnn_v2_12x16")
    cfg = default_config(encoder="CNN_3D", in_channels=int(a["cfg_C"]), in_channels_dynamic=int(a["cfg_V"]), act_dtype="bf16")
    m = VQ_model(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        z, y, an, zq, lz = m(a["x_d"].to(DEV))
    assert zq.dtype == BF and z.dtype == torch.float32
    flips = (an.cpu() != a["eval/anomaly"].long())
    frac = float(flips.float().mean())
    print(f"[bf16 cnn e2e] LFQ code flips vs the f32 reference: {int(flips.sum())}/{flips.numel()} ({100 * frac:.3f} %)")
    assert frac < 0.02
    ok = (F.max_pool2d(flips.any(dim=(1, 2)).float().unsqueeze(1), 7, 1, 3) == 0)
    assert float(ok.float().mean()) > 0.2
    zr = a["eval/z"]
    err = ((z.cpu() - zr).abs() * ok).max()
    assert float(err) <= 5e-2 * float(zr.abs().max()) + 1e-3
    m.train()
    ts = SyntheticTrainStep(cfg, device=DEV, model=m)
    x = a["x_d"].to(DEV)
    g = torch.Generator().manual_seed(3)
    N, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    me = (torch.rand((N, H, W), generator=g) < 0.3).float().to(DEV)
    ml = (torch.rand((N, H, W), generator=g) < 0.4).float().to(DEV)
    total, parts, _ = ts.step(x, me, ml)
    assert torch.isfinite(total) and torch.isfinite(ts.opt.flat_g).all()


def test_bf16_and_f32_gradients_agree_on_the_small_model():
    """The flat gradient of one train step with bf16 activation storage against the f32 path on the same parameters and batch:
    cosine similarity >= 0.99 and a relative l2 error <= 0.15 (the LFQ flips and bf16 rounding of ~40 chained units)."""
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd.train import SyntheticTrainStep
    a, sd, cfg, m = _bf16_model()
    m.load_state_dict(sd, strict=True)
    cfg32 = default_config(encoder="Swin_3D", in_channels=int(a["cfg_C"]), in_channels_dynamic=int(a["cfg_V"]))
    m32 = VQ_model(cfg32)
    m32.load_state_dict(sd, strict=True)
    x = a["x_d"].to(DEV)
    g = torch.Generator().manual_seed(3)
    N, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    me = (torch.rand((N, H, W), generator=g) < 0.3).float().to(DEV)
    ml = (torch.rand((N, H, W), generator=g) < 0.4).float().to(DEV)
    grads = []
    for cfg_, m_ in ((cfg, m), (cfg32, m32)):
        ts = SyntheticTrainStep(cfg_, device=DEV, model=m_)
        ts.opt.zero_grad(set_to_none=True)
        total, _ = ts.losses(ts.model(x), me, ml)
        total.backward()
        ts.opt.gather_grads()
        grads.append(ts.opt.flat_g.clone().double())
    gb, gf = grads
    cos = float((gb * gf).sum() / (gb.norm() * gf.norm()))
    rel = float((gb - gf).norm() / gf.norm())
    print(f"[bf16 vs f32] flat gradient: cosine {cos:.5f}, relative l2 error {rel:.4f}")
    assert cos >= 0.99 and rel <= 0.15
