"""GPU parity: the HIP path (through the C-ABI, via idee_amd.ops / idee_amd.models) against
(a) the golden vectors generated from the real reference and (b) the CPU oracle on seeded
random inputs at sizes that exercise multi-tile paths.  Run with `-m gpu` on an MI355X.

Tolerances (fp32 path): activations 2e-5 abs + 2e-4 rel; gradients 1e-3 rel with an absolute
floor of 1e-4 x the largest reference magnitude of that tensor (atomically accumulated sums);
`anomaly` indices bit-exact except tokens whose projection |s| < 1e-5 (listed, not averaged)."""
import math

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from idee_amd import ops
    return ops


def close(name, got, ref, atol=2e-5, rtol=2e-4):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        i = int(err.argmax())
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} out of tol; max abs err {float(err.max()):.3e} "
                             f"(got {float(got.flatten()[i]):.6g} ref {float(ref.flatten()[i]):.6g}), ref max {float(ref.abs().max()):.3e}")


def gclose(name, got, ref, rtol=1e-3, floor=1e-4):
    close(name, got, ref, atol=max(1e-6, floor * float(ref.abs().max())), rtol=rtol)


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


@pytest.mark.parametrize("tag", ["w244_noshift", "w244_shift", "w244_shift_pad", "w244_noshift_pad", "w811",
                                 "w811_T16_shift", "w244_c32"])
def test_swin_block_golden(tag):
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


def test_swin_block_two_variables_vs_oracle():
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


def test_proj_var_golden():
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


@pytest.mark.parametrize("T,H,W", [(8, 9, 72), (3, 5, 200), (1, 1, 1), (2, 2, 17)])
def test_proj_var_vs_oracle(T, H, W):
    """Multi-tile widths (72 = 3+2 column blocks, 200 = 4+3+3+3), thin extents (clamp-everything cases)."""
    from oracle import idee_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(T * 1000 + W)
    N, V, C = 2, 2, 16
    x = torch.randn((N, V, T, H, W, C), generator=g)
    r = torch.randn((N, V, T, H, W, C), generator=g)
    w1, w2 = (torch.randn((V, C, C, 3, 3, 3), generator=g) * 0.05 for _ in range(2))
    b1, b2 = (torch.randn((V, C), generator=g) * 0.1 for _ in range(2))
    leaves = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
    outs = []
    for v in range(V):
        sd = {"0.weight": leaves[1][v], "0.bias": leaves[2][v], "2.weight": leaves[3][v], "2.bias": leaves[4][v]}
        outs.append(O.proj_var(leaves[0][:, v].permute(0, 4, 1, 2, 3), sd, "").permute(0, 2, 3, 4, 1))
    yo = torch.stack(outs, 1)
    (yo * r).sum().backward()
    dl = [t.to(DEV).requires_grad_() for t in (x, w1, b1, w2, b2)]
    y = ops.ProjVar.apply(*dl)
    close("y", y, yo)
    (y * r.to(DEV)).sum().backward()
    for n, a_, b_ in zip(("dx", "dw1", "db1", "dw2", "db2"), dl, leaves):
        gclose(n, a_.grad, b_.grad)


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


def _classifier(sd, V):
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    c = CNN_3D(in_var=V, embed_dim=16, dim=16, drop_rate=0.0)
    c.load_state_dict(sd, strict=True)
    return c.to(DEV)


def test_classifier_golden():
    a, sd, grads = load_golden("classifier_v2")
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


@pytest.mark.parametrize("V,H,W", [(3, 7, 72), (6, 5, 40), (1, 3, 3), (5, 4, 20)])
def test_classifier_vs_oracle(V, H, W):
    from oracle import idee_oracle as O
    from idee_amd.models.classifier.CNN_3D import CNN_3D
    torch.manual_seed(V * 100 + W)
    c = CNN_3D(in_var=V, embed_dim=16, dim=16, drop_rate=0.0)
    sd = {k: v.clone() for k, v in c.state_dict().items()}
    N, T = 2, 8
    x = torch.randn(N, V, 16, T, H, W)
    sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
    xo = x.clone().requires_grad_()
    zo, yo = O.classifier(xo, sdl, pre="")
    rz, ry = torch.randn_like(zo), torch.randn(V, *yo[0].shape)
    ((zo * rz).sum() + sum((yo[v] * ry[v]).sum() for v in range(V))).backward()
    c = c.to(DEV)
    xd = x.to(DEV).requires_grad_()
    z, y = c(xd)
    close("z", z, zo)
    close("y", torch.stack(y), torch.stack(yo))
    ((z * rz.to(DEV)).sum() + sum((y[v] * ry[v].to(DEV)).sum() for v in range(V))).backward()
    gclose("dx", xd.grad, xo.grad)
    named = dict(c.named_parameters())
    for key, name, idx in c._key_map():
        g = named[name].grad
        gclose("d" + key, g if idx is None else g[idx], sdl[key].grad)


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
