"""GPU tests at BASELINE.json's full size (V=6, T=8, 200x200) through size-independent properties —
the CPU oracle cannot run these sizes in seconds, so parity is anchored by:
  * locality: the model's receptive field is bounded, so the full-size HIP outputs on an interior patch
    must equal the ORACLE run on a window-aligned crop (plus halo) of the same input;
  * linearity of the convolutions, roll-equivariance of the un-shifted Swin block;
  * LFQ code-book consistency, loss identities, run-to-run determinism of the forward."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(encoder="Swin_3D", V=6):
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    torch.manual_seed(11)
    cfg = default_config(encoder=encoder, in_channels=1, in_channels_dynamic=V)
    m = VQ_model(cfg)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():          # livelier weights than the N(0.02, 0.02) init so every term matters
        for n, p in m.named_parameters():
            if p.ndim >= 3 or n.endswith("_w") or "weight" in n:
                fan = p[0].numel() if p.ndim >= 2 else 1
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(max(fan // (1 if p.ndim < 3 or p.shape[0] > 6 else 1), 1))))
            elif "_norm" in n and n.endswith("_g"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return cfg, m


@pytest.mark.parametrize("encoder", ["Swin_3D", "CNN_3D"])
def test_fullsize_forward_matches_oracle_on_interior_crop(encoder):
    from oracle import idee_oracle as O
    cfg, m = _model(encoder)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(13)
    x = torch.randn((1, 6, 1, 8, 200, 200), generator=g).clamp_(-10, 10)
    with torch.no_grad():
        z, y, an, zq, lz = m(x.to(DEV))
        z2, _, an2, zq2, _ = m(x.to(DEV))
    assert torch.equal(z, z2) and torch.equal(an, an2) and torch.equal(zq, zq2), "forward must be deterministic"
    assert z.shape == (1, 1, 200, 200) and an.shape == (1, 6, 8, 200, 200) and zq.shape == (1, 6, 16, 8, 200, 200)
    assert torch.isfinite(z).all() and set(an.unique().tolist()) <= {0, 1}
    # crop [72:136) (window-aligned: 72 % 4 == 0), compare the central [96:112) patch: 24 pixels from the crop edge,
    # more than the receptive-field radius (Swin: 2 shifted 4x4 window blocks + temporal block; 2+3 conv layers)
    lo, hi, a0, a1 = 72, 136, 96, 112
    ocfg = O.default_cfg(encoder=encoder)
    with torch.no_grad():
        zo, yo, ano, zqo, _ = O.vq_model_forward(x[..., lo:hi, lo:hi], sd, ocfg, training=False)
    sl = slice(a0 - lo, a1 - lo)
    s_in = torch.einsum("nvethw,de->nvthwd", O.swin_encoder(x[..., lo:hi, lo:hi], sd, ocfg) if encoder == "Swin_3D"
                        else O.cnn_encoder(x[..., lo:hi, lo:hi], sd, ocfg), sd["vq.project_in.weight"]) + sd["vq.project_in.bias"]
    safe = (s_in.abs() > 1e-5).all(-1)[..., sl, sl]
    assert int((~safe).sum()) <= 8
    assert torch.equal(an.cpu()[..., a0:a1, a0:a1][safe], ano[..., sl, sl][safe])
    if bool(safe.all()):
        torch.testing.assert_close(zq.cpu()[..., a0:a1, a0:a1], zqo[..., sl, sl], atol=2e-5, rtol=2e-4)
    # the classifier sees a 3-pixel neighbourhood of z_q: compare where that neighbourhood holds no boundary token
    if bool((s_in.abs() > 1e-5).all()):
        torch.testing.assert_close(z.cpu()[..., a0:a1, a0:a1], zo[..., sl, sl], atol=5e-5, rtol=5e-4)
        torch.testing.assert_close(torch.stack(y).cpu()[..., a0:a1, a0:a1], torch.stack(yo)[..., sl, sl], atol=5e-5, rtol=5e-4)


def test_fullsize_conv_linearity_and_wgrad_identity():
    """conv(a x1 + b x2) = a conv(x1) + b conv(x2) (bias off) for the 3^3 replicate conv at full size, and
    <conv(x), r> = <x, conv^T(r)> = <w, wgrad(x, r)> (adjoint identities tie fwd, dgrad and wgrad together)."""
    from idee_amd import ops
    from idee_amd import _lib as L
    N, V, T, H, W, C = 2, 6, 8, 200, 200, 16
    g = torch.Generator(device=DEV).manual_seed(3)
    x1, x2, r = (torch.randn((N, V, T, H, W, C), device=DEV, generator=g) for _ in range(3))
    w = torch.randn((V, C, C, 3, 3, 3), device=DEV, generator=g) * 0.05
    s = ops._cl_strides(T, H, W, C)
    conv = ops.ConvLayer(L.CONV_K333_REPLICATE, N * V, V, T, H, W, C, C, s, s, False)
    pf, pd = conv.pack(w)
    y1, y2, y3 = torch.empty_like(x1), torch.empty_like(x1), torch.empty_like(x1)
    conv.fwd(x1, pf, None, y1)
    conv.fwd(x2, pf, None, y2)
    conv.fwd(0.5 * x1 - 2.0 * x2, pf, None, y3)
    torch.testing.assert_close(y3, 0.5 * y1 - 2.0 * y2, atol=2e-4, rtol=1e-4)
    dx = torch.zeros_like(x1)
    conv.dgrad(r, pd, None, dx)
    dw, _ = conv.wgrad(x1, r, w, want_bias=False)
    lhs = float((y1.double() * r.double()).sum())
    assert abs(lhs - float((x1.double() * dx.double()).sum())) <= 1e-5 * abs(lhs) + 1e-2
    assert abs(lhs - float((w.double() * dw.double()).sum())) <= 1e-4 * abs(lhs) + 1e-2


def test_fullsize_swin_block_roll_equivariance():
    """Un-shifted (2,4,4) windows on a grid that is a multiple of the window: rolling the input by whole
    windows rolls the output."""
    from idee_amd import ops
    N, V, D, H, W, C = 1, 6, 8, 200, 200, 16
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn((N, V, D, H, W, C), device=DEV, generator=g)
    mk = lambda *s, sc=0.2: torch.randn(s, device=DEV, generator=g) * sc
    params = [mk(V, 48, 16), mk(V, 48), mk(V, 16, 16), mk(V, 16), mk(V, 64, 16), mk(V, 64), mk(V, 16, 64), mk(V, 16), mk(V, 147, 2, sc=0.5)]
    from idee_amd.models.encoder.Swin_3D import relative_position_index
    idx = relative_position_index((2, 4, 4)).to(DEV)
    f = lambda t: ops.SwinBlock.apply(t, *params, idx, 2, (2, 4, 4), (0, 0, 0), None)
    y = f(x)
    yr = f(torch.roll(x, shifts=(2, 8, 12), dims=(2, 3, 4)).contiguous())
    assert torch.equal(yr, torch.roll(y, shifts=(2, 8, 12), dims=(2, 3, 4)))


def test_fullsize_lfq_and_losses_identities():
    from idee_amd.models.codebook.LFQ import LFQ
    from idee_amd.models import losses as Lm
    torch.manual_seed(5)
    q = LFQ(dim=16, codebook_size=2, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0).to(DEV).train()
    z = torch.randn((4, 6, 8, 200, 200, 16), device=DEV)
    zq, idx, aux = q.quantize_tokens(z)
    codes = q.indices_to_codes(torch.tensor([0, 1], device=DEV))
    assert torch.equal(zq, codes[idx])                       # z_q is exactly the projected code of its index
    assert torch.isfinite(aux)
    p1 = float(idx.float().mean())
    assert 0.0 < p1 < 1.0
    # anomaly loss: zero when z_q equals code 0 everywhere outside extremes; scales linearly with the distance
    an = Lm.Anomaly_L1_loss_synthetic(n_dynamic=6, delta_t=8, dim=16)
    mask = (torch.rand((4, 200, 200), device=DEV) < 0.2).float()
    z0 = codes[0].view(1, 1, 16, 1, 1, 1).expand(4, 6, 16, 8, 200, 200)
    assert float(an(z0.contiguous(), mask, codes[:1])) == 0.0
    d = torch.randn((4, 6, 16, 8, 200, 200), device=DEV)
    l1, l2 = float(an(z0 + d, mask, codes[:1])), float(an(z0 + 2 * d, mask, codes[:1]))
    assert abs(l2 - 2 * l1) <= 1e-4 * l2
    # BCE at zero logits = ln 2 * mean class weight
    tgt = (torch.rand((4, 1, 200, 200), device=DEV) < 0.1).float()
    n1 = float(tgt.sum()); n = tgt.numel()
    w0, w1 = math.log(((n - n1) / n) ** -0.5 + 1.1), math.log((n1 / n) ** -0.5 + 1.1)
    expect = math.log(2.0) * (w0 * (n - n1) + w1 * n1) / n
    got = float(Lm.BCE_loss_synthetic()(torch.zeros_like(tgt), tgt))
    assert abs(got - expect) <= 1e-5 * expect


def test_fullsize_train_step_runs_and_updates():
    """One full S200 step (batch 4): finite loss, every parameter receives a finite gradient and moves."""
    from idee_amd.config import default_config
    from idee_amd.train import SyntheticTrainStep
    torch.manual_seed(0)
    cfg = default_config(encoder="Swin_3D", in_channels=1)
    ts = SyntheticTrainStep(cfg, device=DEV)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((4, 6, 1, 8, 200, 200), generator=g).clamp_(-10, 10).to(DEV)
    me = (torch.rand((4, 200, 200), generator=g) < 0.10).float().to(DEV)
    ml = (torch.rand((4, 200, 200), generator=g) < 0.20).float().to(DEV)
    before = ts.opt.flat_p.clone()
    total, parts, _ = ts.step(x, me, ml)
    assert torch.isfinite(total) and all(torch.isfinite(v) for v in parts.values())
    assert torch.isfinite(ts.opt.flat_g).all() and float((ts.opt.flat_g != 0).float().mean()) > 0.99
    assert float((ts.opt.flat_p != before).float().mean()) > 0.99
