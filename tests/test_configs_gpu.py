"""-m gpu: BASELINE.json's configurations that the headline bench does not run, each against the oracle.

  #1  Synthetic_dataset 64x64, CNN_3D encoder, batch 2 — the full train step at T=8 (the reference's classifier cannot
      run T=4: three stride-2 temporal kernels), and at T=4 the encoder + LFQ alone plus the classifier's error;
  #4  ERA5-Land-shaped 412x424, T=16, V=6, C=2, Swin "small" (embed [32,32], depths [2,2], head_dim 16, codebook/cls
      dim 32), f32: forward against the oracle on crops incl. the true frame corners, the (N,1,2,H,W) output shape,
      and one ERA5 train step (finite, every parameter moves)."""
import os

import pytest
import torch

from conftest import tie_relu, interval_close, adam_params_close, variants, KERNEL_VARIANTS

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(name, got, ref, atol=2e-5, rtol=2e-4):
    torch.testing.assert_close(got.detach().float().cpu(), ref.detach().float().cpu(), atol=atol, rtol=rtol, msg=lambda m: f"{name}: {m}")


def _named_grads(model):
    out = {}
    for mod_name, mod in (("encoder.", model.encoder), ("cls.", model.cls)):
        ps = dict(mod.named_parameters())
        for key, name, idx in mod._key_map():
            if name.endswith("_index"):
                continue
            base = name[:-1] if name.endswith("_ds_w5") else name
            g = ps[base].grad
            g = g.view(*g.shape, 1, 1, 1) if name.endswith("_ds_w5") else g
            out[mod_name + key] = g if idx is None else g[idx]
    for k, p in model.vq.named_parameters():
        out["vq." + k] = p.grad
    return out


def _lively(model, seed, proj_scale):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2:
                fan = p[0].numel() if p.ndim == 2 else p[0][0].numel() if p.shape[0] <= 9 and p.ndim > 3 else p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / max(fan, 1) ** 0.5))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for n, p in model.named_parameters():
            if n.endswith("_g") or "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
        model.vq.project_in.weight.mul_(proj_scale)


@variants('default', 'all_f22')
def test_config1_cnn_64x64_batch2_train_step_vs_oracle(kernel_variant):
    from oracle import idee_oracle as O
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd.train import SyntheticTrainStep
    kw = dict(encoder="CNN_3D", in_channels=1, in_channels_dynamic=6)
    cfg, ocfg = default_config(**kw), O.default_cfg(**kw)
    torch.manual_seed(1)
    m = VQ_model(cfg)
    _lively(m, 2, 16.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    N, T, H, W = 2, 8, 64, 64
    for seed in range(3, 23):          # an input without an LFQ boundary token (|s| < 1e-5 flips z_q discretely)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((N, 6, 1, T, H, W), generator=g).clamp_(-10, 10)
        with torch.no_grad():
            s_in = torch.einsum("nvethw,de->nvthwd", O.cnn_encoder(x, sd, ocfg), sd["vq.project_in.weight"]) + sd["vq.project_in.bias"]
        if bool((s_in.abs() > 1e-5).all()):
            break
    else:
        raise AssertionError("no boundary-free input among 20 seeds")
    print(f"[config 1] input seed {seed}: no LFQ boundary token among {s_in.numel()}")
    me = (torch.rand((N, H, W), generator=g) < 0.10).float()
    ml = (torch.rand((N, H, W), generator=g) < 0.20).float()
    # the oracle twice: ReLU derivative at tie units forced to 0 / to 1
    runs = []
    for mode in (0, 1):
        sdl = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
        with tie_relu(mode) as tr:
            outs = O.vq_model_forward(x, sdl, ocfg, training=True)
            total, parts = O.train_losses_synthetic(outs, me, ml, sdl, ocfg)
        total.backward()
        runs.append((outs, total, parts, {k: v.grad for k, v in sdl.items() if v.is_floating_point()}, tr))
    (outs_o, total_o, parts_o, g0, tr), (_, _, _, g1, _) = runs
    ts = SyntheticTrainStep(cfg, device=DEV, model=m)
    ts.opt.zero_grad()
    outs = ts.model(x.to(DEV))
    total, parts = ts.losses(outs, me.to(DEV), ml.to(DEV))
    total.backward()
    _close("z", outs[0], outs_o[0], atol=5e-5, rtol=5e-4)
    _close("y", torch.stack(outs[1]), torch.stack(outs_o[1]), atol=5e-5, rtol=5e-4)
    assert torch.equal(outs[2].cpu(), outs_o[2])
    _close("z_q", outs[3], outs_o[3])
    _close("loss_z_q", outs[4], outs_o[4], atol=1e-5, rtol=5e-4)
    for k in ("extreme", "anomaly", "var"):
        _close("loss " + k, parts[k], parts_o[k], atol=1e-5, rtol=2e-4)
    _close("total", total.reshape(()), total_o.reshape(()), atol=1e-4, rtol=2e-4)
    named = _named_grads(ts.model)
    assert set(named) == set(g0)
    affected = sum(interval_close("d" + k, named[k], g0[k], g1[k], rtol=2e-3, floor=2e-4) for k in g0)
    total_el = sum(v.numel() for v in g0.values())
    print(f"[config 1] ReLU ties: {tr.n_ties} of {tr.n_units} units; gradient elements inside a tie interval: {affected}/{total_el}")
    assert affected <= 0.02 * total_el
    from idee_amd import parallel
    parallel.sync_gradients(ts.opt)
    ts.opt.step()
    assert torch.isfinite(ts.opt.flat_p).all()


def test_config1_T4_encoder_lfq_only_and_classifier_error():
    """BASELINE config #1 as written (T=4): the reference's classifier raises (kernel larger than input); the encoder and
    LFQ run, and match the oracle."""
    from oracle import idee_oracle as O
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    kw = dict(encoder="CNN_3D", in_channels=1, in_channels_dynamic=6)
    cfg, ocfg = default_config(**kw), O.default_cfg(**kw)
    torch.manual_seed(4)
    m = VQ_model(cfg)
    _lively(m, 5, 4.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn((2, 6, 1, 4, 64, 64), generator=torch.Generator().manual_seed(6)).clamp_(-10, 10)
    m = m.to(DEV).train()
    z = m.encoder.forward_tokens(x.to(DEV))
    zo = O.cnn_encoder(x, sd, ocfg)
    _close("encoder", z.permute(0, 1, 5, 2, 3, 4), zo, atol=3e-5, rtol=3e-4)
    z_q, idx, aux = m.vq.quantize_tokens(z)
    N, V, C, T, H, W = zo.shape
    zq_o, idx_o, aux_o = O.lfq(zo.permute(0, 2, 1, 3, 4, 5).reshape(N, C, -1).permute(0, 2, 1), sd, ocfg, True)
    s_in = torch.einsum("nvethw,de->nvthwd", zo, sd["vq.project_in.weight"]) + sd["vq.project_in.bias"]
    safe = (s_in.abs() > 1e-5).all(-1)
    assert float(safe.float().mean()) > 0.9999
    assert torch.equal(idx.cpu()[safe], idx_o.reshape(N, V, T, H, W)[safe])
    _close("aux", aux, aux_o, atol=1e-5, rtol=5e-4)
    with pytest.raises(RuntimeError):
        m(x.to(DEV))


C4 = dict(encoder="Swin_3D", in_channels=2, in_channels_dynamic=6, en_embed_dim=[32, 32], en_depths=[2, 2], codebook_dim=32,
          cls_dim=32, delta_t=16)
C4_CROPS = {     # (rows, cols) of the crop, then of the compared region; 412 = 103*4, 424 = 106*4
    "interior": ((172, 236), (180, 244), (196, 212), (204, 220)),
    "corner_bottom_right": ((348, 412), (360, 424), (372, 412), (384, 424)),
    "corner_top_left": ((0, 64), (0, 64), (0, 40), (0, 40)),
}


@pytest.fixture(scope="module", params=["default", "all_f22"])
def c4_forward(request):
    """The config-#4 forward, once on the exact-f32 kernels and once on the f16-piece f32-level kernels (the library reads its
    switches at every launch; a module-scoped fixture sets them around its own launches)."""
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd import _lib
    opts = {"IDEE_SWIN_MATH": "", "IDEE_CONV_MATH": ""}
    opts.update(KERNEL_VARIANTS[request.param])
    with _lib.options(**opts):
        return _c4_forward(default_config, VQ_model)


def _c4_forward(default_config, VQ_model):
    cfg = default_config(**C4)
    torch.manual_seed(7)
    m = VQ_model(cfg)
    _lively(m, 8, 12.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn((1, 6, 2, 16, 412, 424), generator=torch.Generator().manual_seed(9)).clamp_(-10, 10)
    m = m.to(DEV).eval()
    with torch.no_grad():
        z, y, an, zq, lz = m(x.to(DEV))
        enc = m.encoder(x.to(DEV))
    out = dict(sd=sd, x=x, z=z.cpu(), y=torch.stack(y).cpu(), an=an.cpu(), zq=zq.cpu(), enc=enc.cpu())
    del m, z, y, an, zq, enc
    torch.cuda.empty_cache()
    return out


def test_config4_shapes(c4_forward):
    f = c4_forward
    assert f["z"].shape == (1, 1, 2, 412, 424) and f["y"].shape == (6, 1, 1, 2, 412, 424)       # T=16: two temporal maps
    assert f["an"].shape == (1, 6, 16, 412, 424) and f["an"].dtype == torch.int64
    assert f["zq"].shape == (1, 6, 32, 16, 412, 424)
    assert all(bool(torch.isfinite(f[k]).all()) for k in ("z", "y", "zq", "enc"))
    p1 = float(f["an"].float().mean())
    assert 0.02 < p1 < 0.98, f"degenerate code usage {p1}"


@pytest.mark.parametrize("crop", list(C4_CROPS))
def test_config4_forward_matches_oracle_on_crop(c4_forward, crop):
    import torch.nn.functional as F
    from oracle import idee_oracle as O
    f = c4_forward
    ocfg = O.default_cfg(**{k: v for k, v in C4.items() if k != "delta_t"})
    (r0, r1), (c0, c1), (a0, a1), (b0, b1) = C4_CROPS[crop]
    xc = f["x"][..., r0:r1, c0:c1]
    with torch.no_grad():
        zo, yo, ano, zqo, _ = O.vq_model_forward(xc, f["sd"], ocfg, training=False)
        enco = O.swin_encoder(xc, f["sd"], ocfg)
    rs, cs = slice(a0 - r0, a1 - r0), slice(b0 - c0, b1 - c0)
    R = lambda t: t[..., a0:a1, b0:b1]
    _close("encoder", R(f["enc"]), enco[..., rs, cs], atol=3e-5, rtol=3e-4)
    s_in = torch.einsum("nvethw,de->nvthwd", enco, f["sd"]["vq.project_in.weight"]) + f["sd"]["vq.project_in.bias"]
    unsafe = ~(s_in.abs() > 1e-5).all(-1)
    safe = ~unsafe[..., rs, cs]
    print(f"[config 4/{crop}] boundary tokens in region {int((~safe).sum())}/{safe.numel()}, in crop {int(unsafe.sum())}")
    assert float(safe.float().mean()) >= 0.999
    assert torch.equal(R(f["an"])[safe], ano[..., rs, cs][safe])
    ok = safe.unsqueeze(2).expand(-1, -1, 32, -1, -1, -1)
    _close("z_q", R(f["zq"])[ok], zqo[..., rs, cs][ok])
    bad_px = unsafe.any(dim=(1, 2)).float().unsqueeze(1)
    ok_px = (F.max_pool2d(bad_px, 7, 1, 3) == 0)[..., rs, cs]                 # (N,1,h,w)
    frac = float(ok_px.float().mean())
    print(f"[config 4/{crop}] classifier pixels compared {int(ok_px.sum())}/{ok_px.numel()}")
    assert frac >= 0.90
    zk = ok_px.unsqueeze(2).expand(-1, -1, 2, -1, -1)
    _close("z", R(f["z"])[zk], zo[..., rs, cs][zk], atol=5e-5, rtol=5e-4)
    yk = zk.unsqueeze(0).expand(6, -1, -1, -1, -1, -1)
    _close("y", R(f["y"])[yk], torch.stack(yo)[..., rs, cs][yk], atol=5e-5, rtol=5e-4)


def test_config4_era5_train_step_runs():
    from idee_amd.config import default_config
    from idee_amd.train import ERA5TrainStep
    cfg = default_config(**C4)
    torch.manual_seed(0)
    ts = ERA5TrainStep(cfg, device=DEV)
    g = torch.Generator().manual_seed(1)
    N, T, H, W = 1, 16, 412, 424
    x = torch.randn((N, 6, 2, T, H, W), generator=g).clamp_(-10, 10).to(DEV)
    me = (torch.rand((N, H, W), generator=g) < 0.10).float().to(DEV)
    ml = (torch.rand((N, H, W), generator=g) < 0.20).float().to(DEV)
    cold = (torch.rand((N, H, W), generator=g) < 0.05).float().to(DEV)
    cold_t = (torch.rand((N, T, H, W), generator=g) < 0.05).float().to(DEV)
    before = ts.opt.flat_p.clone()
    total, parts, outs = ts.step(x, me, ml, cold, cold_t)
    assert outs[0].shape == (1, 1, 2, H, W)
    assert torch.isfinite(total) and all(torch.isfinite(v) for v in parts.values())
    assert torch.isfinite(ts.opt.flat_g).all() and float((ts.opt.flat_g != 0).float().mean()) > 0.99
    assert float((ts.opt.flat_p != before).float().mean()) > 0.99
    print(f"[config 4] peak HBM allocated {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")


def test_checkpoint_resume_continues_the_same_trajectory(tmp_path):
    """harness.save_model / load_model (the reference's checkpoint file, utils_train.py:562-584): a run interrupted after
    two steps and resumed in a fresh process state (new model, new optimizer) takes the same next two steps as the
    uninterrupted run — parameters, Adam moments, step count (bias correction) and learning rate all travel."""
    from idee_amd import harness
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd.train import SyntheticTrainStep
    kw = dict(encoder="Swin_3D", in_channels=1, in_channels_dynamic=2, dir_log=str(tmp_path), name="resume")
    N, T, H, W = 2, 8, 16, 24

    def batch(k):
        g = torch.Generator().manual_seed(50 + k)
        x = torch.randn((N, 2, 1, T, H, W), generator=g).clamp_(-10, 10)
        return [t.to(DEV) for t in (x, (torch.rand((N, H, W), generator=g) < 0.1).float(), (torch.rand((N, H, W), generator=g) < 0.2).float())]

    def make():
        torch.manual_seed(7)
        st = SyntheticTrainStep(default_config(**kw), model=VQ_model(default_config(**kw)))
        with torch.no_grad():
            st.model.vq.project_in.weight.mul_(20.0)
        return st

    full = make()
    for k in range(4):
        full.step(*batch(k))
    first = make()
    for k in range(2):
        first.step(*batch(k))
    first.opt.param_groups[0]["lr"] = 7e-4            # a scheduler moved it: must travel too
    full_lr = make()
    for k in range(2):
        full_lr.step(*batch(k))
    full_lr.opt.param_groups[0]["lr"] = 7e-4
    g_full = []
    for k in range(2, 4):
        full_lr.step(*batch(k))
        g_full.append(full_lr.opt.flat_g.detach().clone())
    path = harness.save_model(first.model, first.opt, 1, 0.5, 0.6, first.config, "train")
    torch.manual_seed(999)
    resumed = SyntheticTrainStep(default_config(**kw), model=VQ_model(default_config(**kw)))
    epoch, ltr, lva = harness.load_model(path, resumed.model, resumed.opt)
    assert (epoch, ltr, lva) == (1, 0.5, 0.6) and resumed.opt.step_count == 2 and resumed.opt.param_groups[0]["lr"] == 7e-4
    g_res = []
    for k in range(2, 4):
        resumed.step(*batch(k))
        g_res.append(resumed.opt.flat_g.detach().clone())
    # the two runs differ only by float-atomic summation order in steps 3 and 4: compared where that noise cannot decide a step
    adam_params_close("resume", resumed.opt.flat_p.detach(), full_lr.opt.flat_p.detach(), g_res, g_full, lr=7e-4)
    assert float((full.opt.flat_p - full_lr.opt.flat_p).abs().max()) > 1e-5      # the learning-rate change is visible at all


# ---------------------------------------------------------------------------------------------
# BASELINE config #4 AS SPECIFIED: bf16 activations / GEMMs with f32 accumulate, f32 LFQ (config.act_dtype = "bf16")
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c4_forward_bf16(c4_forward):
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    cfg = default_config(act_dtype="bf16", **C4)
    m = VQ_model(cfg)
    m.load_state_dict(c4_forward["sd"], strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        z, y, an, zq, lz = m(c4_forward["x"].to(DEV))
        enc = m.encoder(c4_forward["x"].to(DEV))
    out = dict(z=z.cpu(), y=torch.stack(y).cpu(), an=an.cpu(), zq=zq.float().cpu(), enc=enc.float().cpu(), zq_dtype=zq.dtype)
    del m, z, y, an, zq, enc
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("crop", list(C4_CROPS))
def test_config4_bf16_forward_matches_oracle_on_crop(c4_forward, c4_forward_bf16, crop):
    """The bf16 path at config #4's full size (412 x 424, T = 16, E = 32) against the f32 ORACLE on three crops.
    Stated bf16 tolerances: encoder output within 3e-2 of its rms (4 Swin blocks + 2 convolutions of bf16-rounded operands and
    bf16-stored activations); LFQ code flips COUNTED — a token may flip only where the oracle's projection |s| is below
    0.05 * rms(s) (inside the bf16 noise band), every other token must keep its code, and at most 1 % may flip; z_q compared
    on unflipped tokens to bf16 rounding; logits compared where the 7x7 neighbourhood holds no flipped token, at 5e-2 of scale."""
    import torch.nn.functional as F
    from oracle import idee_oracle as O
    f, fb = c4_forward, c4_forward_bf16
    assert fb["zq_dtype"] == torch.bfloat16
    ocfg = O.default_cfg(**{k: v for k, v in C4.items() if k != "delta_t"})
    (r0, r1), (c0, c1), (a0, a1), (b0, b1) = C4_CROPS[crop]
    xc = f["x"][..., r0:r1, c0:c1]
    with torch.no_grad():
        zo, yo, ano, zqo, _ = O.vq_model_forward(xc, f["sd"], ocfg, training=False)
        enco = O.swin_encoder(xc, f["sd"], ocfg)
    rs, cs = slice(a0 - r0, a1 - r0), slice(b0 - c0, b1 - c0)
    R = lambda t: t[..., a0:a1, b0:b1]
    e_ref = enco[..., rs, cs]
    e_err = (R(fb["enc"]) - e_ref).abs()
    rms = float(e_ref.pow(2).mean().sqrt())
    print(f"[config 4 bf16/{crop}] encoder: max err {float(e_err.max()):.3e}, rms err {float(e_err.pow(2).mean().sqrt()):.3e}, rms {rms:.3e}")
    assert float(e_err.pow(2).mean().sqrt()) <= 1e-2 * rms and float(e_err.max()) <= 1e-1 * rms
    s_in = torch.einsum("nvethw,de->nvthwd", enco, f["sd"]["vq.project_in.weight"]) + f["sd"]["vq.project_in.bias"]
    s_reg = s_in[..., rs, cs, :]
    band = 0.05 * float(s_in.pow(2).mean().sqrt())
    flips = R(fb["an"]) != ano[..., rs, cs]
    inside = (s_reg.abs() < band).any(-1)
    print(f"[config 4 bf16/{crop}] LFQ code flips {int(flips.sum())}/{flips.numel()} ({100 * float(flips.float().mean()):.3f} %), "
          f"tokens inside the noise band |s| < {band:.3e}: {int(inside.sum())}")
    assert not bool((flips & ~inside).any()), "a token outside the bf16 noise band changed its code"
    assert float(flips.float().mean()) <= 0.01
    keep = (~flips).unsqueeze(2).expand(-1, -1, 32, -1, -1, -1)
    zq_err = (R(fb["zq"]) - zqo[..., rs, cs]).abs()[keep]
    assert float(zq_err.max()) <= 2.0 ** -7 * float(zqo.abs().max()) + 1e-6
    # classifier: with ~0.2 % of the 6 x 16 tokens of a pixel column flipped, no 7x7 neighbourhood is flip-free, so the logits are
    # compared against the ORACLE's classifier applied to the codes THIS run chose (codes -> project_out -> classifier, f32):
    # that isolates the classifier's own bf16 error from the (counted) flips upstream.  Crop margins exceed its 3-pixel radius.
    an_crop = fb["an"][..., r0:r1, c0:c1]                                            # our codes on the whole crop (N,V,T,h,w)
    zq_ours = O.lfq_indices_to_codes(an_crop.reshape(-1), f["sd"], ocfg).reshape(*an_crop.shape, 32).permute(0, 1, 5, 2, 3, 4)
    with torch.no_grad():
        zo2, yo2 = O.classifier(zq_ours, f["sd"])
    zr = zo2[..., rs, cs]
    scale = float(zr.abs().max())
    zerr = float((R(fb["z"]) - zr).abs().max())
    yr = torch.stack(yo2)[..., rs, cs]
    yerr = float((R(fb["y"]) - yr).abs().max())
    print(f"[config 4 bf16/{crop}] logits vs oracle classifier on the same codes: joint max err {zerr:.3e} (scale {scale:.3e}), "
          f"heads max err {yerr:.3e} (scale {float(yr.abs().max()):.3e})")
    assert zerr <= 2e-2 * scale + 1e-3
    assert yerr <= 2e-2 * float(yr.abs().max()) + 1e-3


@pytest.mark.parametrize("impl", ["LFQ", "VQ"])
def test_config4_bf16_era5_train_step_runs(impl):
    """One ERA5 train step of config #4 with bf16 activation storage, LFQ and the nearest-neighbour VQ codebook variant: shapes,
    finite losses, a gradient for every parameter, f32 master parameters that move; peak memory printed."""
    from idee_amd.config import default_config
    from idee_amd.train import ERA5TrainStep
    cfg = default_config(act_dtype="bf16", codebook_impl=impl, **C4)
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    ts = ERA5TrainStep(cfg, device=DEV)
    g = torch.Generator().manual_seed(1)
    N, T, H, W = 1, 16, 412, 424
    x = torch.randn((N, 6, 2, T, H, W), generator=g).clamp_(-10, 10).to(DEV)
    me = (torch.rand((N, H, W), generator=g) < 0.10).float().to(DEV)
    ml = (torch.rand((N, H, W), generator=g) < 0.20).float().to(DEV)
    cold = (torch.rand((N, H, W), generator=g) < 0.05).float().to(DEV)
    cold_t = (torch.rand((N, T, H, W), generator=g) < 0.05).float().to(DEV)
    before = ts.opt.flat_p.clone()
    total, parts, outs = ts.step(x, me, ml, cold, cold_t)
    assert outs[0].shape == (1, 1, 2, H, W) and outs[0].dtype == torch.float32
    assert outs[3].shape == (1, 6, 32, 16, H, W)
    assert torch.isfinite(total) and all(torch.isfinite(v) for v in parts.values())
    assert torch.isfinite(ts.opt.flat_g).all() and float((ts.opt.flat_g != 0).float().mean()) > 0.99
    assert ts.opt.flat_p.dtype == torch.float32 and float((ts.opt.flat_p != before).float().mean()) > 0.99
    print(f"[config 4 bf16 {impl}] loss {float(total):.4f}; peak HBM allocated {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
