"""Pin oracle/idee_oracle.py (the CPU restatement) to the golden vectors generated from the real
reference (tests/golden/make_fixtures.py).  CPU only."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import idee_oracle as O

ATOL, RTOL = 1e-5, 1e-4


def close(a, b, atol=ATOL, rtol=RTOL):
    torch.testing.assert_close(a.float(), b.float(), atol=atol, rtol=rtol, equal_nan=True)


def leaf(sd):
    return {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}


def check_grads(sd, grads, atol=1e-5, rtol=1e-3):
    for k, g in grads.items():
        assert sd[k].grad is not None, k
        close(sd[k].grad, g, atol, rtol)


@pytest.mark.parametrize("tag", ["w244_noshift", "w244_shift", "w244_shift_pad", "w244_noshift_pad", "w811",
                                 "w811_T16_shift", "w244_c32"])
def test_swin_block(tag):
    a, sd, grads = load_golden("swin_block_" + tag)
    sd = leaf(sd)
    x = a["x"].clone().requires_grad_()
    y = O.swin_block(x, sd, "", int(a["heads"]), tuple(a["window"].tolist()), tuple(a["shift"].tolist()))
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    close(x.grad, a["dx"], 1e-5, 1e-3)
    check_grads(sd, grads)


def test_mask_closed_form():
    a, _, _ = load_golden("swin_block_w244_shift_pad")
    m = O.compute_mask(8, 8, 12, (2, 4, 4), (1, 2, 2))
    assert torch.equal(m, a["mask"])
    a, _, _ = load_golden("swin_block_w811_T16_shift")
    assert torch.equal(O.compute_mask(16, 4, 6, (8, 1, 1), (4, 0, 0)), a["mask"])


@pytest.mark.parametrize("tag", ["swin_c1", "swin_c2", "cnn_c1"])
def test_patch_embed(tag):
    a, sd, grads = load_golden("patch_embed_" + tag)
    sd = leaf(sd)
    y = O.patch_embed(a["x"], sd, "")
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    check_grads(sd, grads)


@pytest.mark.parametrize("tag", ["c16_even", "c16_odd", "c16_d1", "c32_even"])
def test_patch_merging(tag):
    a, sd, grads = load_golden("patch_merging_" + tag)
    sd = leaf(sd)
    x = a["x"].clone().requires_grad_()
    y = O.patch_merging(x, sd)
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    close(x.grad, a["dx"], 1e-5, 1e-3)
    check_grads(sd, grads)


@pytest.mark.parametrize("tag,cin", [("1_16", 1), ("16_16", 16)])
def test_conv_block(tag, cin):
    a, sd, grads = load_golden("conv_block_" + tag)
    sd = leaf(sd)
    x = a["x"].clone().requires_grad_()
    y = O.conv_block(x, sd, "", cin, 16)
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    close(x.grad, a["dx"], 1e-5, 1e-3)
    check_grads(sd, grads)


def test_proj_var():
    a, sd, grads = load_golden("proj_var")
    sd = leaf(sd)
    x = a["x"].clone().requires_grad_()
    y = O.proj_var(x, sd, "")
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    close(x.grad, a["dx"], 1e-5, 1e-3)
    check_grads(sd, grads)


@pytest.mark.parametrize("tag", ["k2_train", "k2_eval", "k4_train"])
def test_lfq(tag):
    a, sd, grads = load_golden("lfq_" + tag)
    sd = leaf(sd)
    cfg = O.default_cfg(codebook_size=int(a["K"]))
    training = bool(int(a["training"]))
    x = a["x"].clone().requires_grad_()
    zq, idx, aux = O.lfq(x, sd, cfg, training, pre="")
    close(zq, a["z_q"])
    assert torch.equal(idx, a["indices"].long())
    close(aux, a["aux"])
    close(O.lfq_indices_to_codes(torch.tensor([0]), sd, cfg, pre=""), a["code0"])
    if training:
        ((zq * a["r"]).sum() + 2.5 * aux).backward()
        close(x.grad, a["dx"], 1e-5, 1e-3)
        check_grads(sd, grads)


@pytest.mark.parametrize("tag", ["k2_c2_train", "k4_c2_train", "k2_c4_eval"])
def test_lfq_several_codebooks(tag):
    """LFQ(num_codebooks > 1) (LFQ.py:92-101, :215-300) in the oracle against the reference's outputs and gradients."""
    a, sd, grads = load_golden("lfq_multi_" + tag)
    sd = leaf(sd)
    K, ncb, training = int(a["K"]), int(a["ncb"]), bool(int(a["training"]))
    x = a["x"].clone().requires_grad_()
    zq, idx, aux = O.lfq_multi(x, sd, K, ncb, training, 0.1, 0.1, 3.0)
    close(zq, a["z_q"])
    assert torch.equal(idx, a["indices"].long())
    close(aux, a["aux"])
    close(O.lfq_multi_indices_to_codes(a["code_idx"].long(), sd, K), a["codes"])
    if training:
        ((zq * a["r"]).sum() + 2.5 * aux).backward()
        close(x.grad, a["dx"], 1e-5, 1e-3)
        check_grads(sd, grads)


@pytest.mark.parametrize("tag", ["e16_k2_train", "e16_k4_train", "e32_k16_train", "e16_k16_eval"])
def test_lfq_without_projections(tag):
    """LFQ with dim == num_codebooks * log2 K (LFQ.py:98-101: Identity projections) in the oracle against the reference."""
    a, sd, _ = load_golden("lfq_noproj_" + tag)
    assert list(sd) == ["mask"]
    K, dim, training = int(a["K"]), int(a["dim"]), bool(int(a["training"]))
    ncb = dim // (K.bit_length() - 1)
    x = a["x"].clone().requires_grad_()
    zq, idx, aux = O.lfq_multi(x, sd, K, ncb, training, 0.1, 0.1, 3.0)
    close(zq, a["z_q"])
    assert torch.equal(idx, a["indices"].long())
    close(aux, a["aux"])
    close(O.lfq_multi_indices_to_codes(a["code_idx"].long(), sd, K), a["codes"])
    if training:
        ((zq * a["r"]).sum() + 2.5 * aux).backward()
        close(x.grad, a["dx"], 1e-5, 1e-3)


@pytest.mark.parametrize("tag", ["classifier_v2", "classifier_v2_k3"])       # _k3: n_classes = 3
def test_classifier(tag):
    a, sd, grads = load_golden(tag)
    sd = leaf(sd)
    x = a["x"].clone().requires_grad_()
    z, y = O.classifier(x, sd, pre="")
    close(z, a["z"])
    close(torch.stack(y), a["y"])
    ((z * a["rz"]).sum() + sum((y[v] * a["ry"][v]).sum() for v in range(len(y)))).backward()
    close(x.grad, a["dx"], 1e-5, 1e-3)
    check_grads(sd, grads)


def test_losses():
    a, _, _ = load_golden("losses")
    for tag in ("mixed", "zeros", "ones"):
        p = a["bce_s/pred"].clone().requires_grad_()
        l = O.bce_loss_synthetic(p, a[f"bce_s/{tag}/target"])
        close(l, a[f"bce_s/{tag}/loss"])
        l.backward()
        close(p.grad, a[f"bce_s/{tag}/dpred"])
    p = a["bce_r/pred"].clone().requires_grad_()
    l = O.bce_loss(p, a["bce_r/target"], a["bce_r/mask_valid"])
    close(l, a["bce_r/loss"])
    l.backward()
    close(p.grad, a["bce_r/dpred"])
    zq = a["an_s/z_q"].clone().requires_grad_()
    l = O.anomaly_l1_synthetic(zq, a["an_s/mask"], a["an_s/vq0"])
    close(l, a["an_s/loss"])
    l.backward()
    close(zq.grad, a["an_s/dz_q"])
    assert math.isnan(float(a["an_s/all_extreme_loss"]))
    assert math.isnan(float(O.anomaly_l1_synthetic(zq.detach(), torch.ones_like(a["an_s/mask"]), a["an_s/vq0"])))
    zq.grad = None
    l = O.anomaly_l1(zq, a["an_s/mask"], a["an_r/mask_valid"], a["an_s/vq0"])
    close(l, a["an_r/loss"])
    l.backward()
    close(zq.grad, a["an_r/dz_q"])


def _cfg_from(a):
    return O.default_cfg(in_channels_dynamic=int(a["cfg_V"]), in_channels=int(a["cfg_C"]), encoder=str(a["encoder"]))


@pytest.mark.parametrize("tag", ["swin_v2_14x18", "cnn_v2_12x16", "swin_v6_8x8_c2"])
def test_e2e_forward(tag):
    a, sd, _ = load_golden("e2e_" + tag)
    cfg = _cfg_from(a)
    with torch.no_grad():
        z, y, an, zq, lz = O.vq_model_forward(a["x_d"], sd, cfg, training=False)
    close(z, a["eval/z"])
    close(torch.stack(y), a["eval/y"])
    assert torch.equal(an, a["eval/anomaly"].long())
    close(zq, a["eval/z_q"])
    close(lz, a["eval/loss_z_q"])


def _losses(a, outs, sd, cfg):
    if "mask_drought" in a:      # train_ERA5_Land.py sequence
        return O.train_losses_era5(outs, a["mask_drought"], a["mask_drought_loss"], a["mask_cold_surface"],
                                   a["mask_cold_surface_loss"], sd, cfg)
    return O.train_losses_synthetic(outs, a["mask_extreme"], a["mask_extreme_loss"], sd, cfg)


@pytest.mark.parametrize("tag", ["swin_v2_14x18", "cnn_v2_12x16", "era5_swin_v3_c2_10x12"])
def test_e2e_train_steps(tag):
    a, sd, grads = load_golden("e2e_" + tag)
    cfg = _cfg_from(a)
    sd = leaf(sd)
    names = [k for k in grads]
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    losses = []
    for step in range(1, 4):
        outs = O.vq_model_forward(a["x_d"], sd, cfg, training=True)
        total, parts = _losses(a, outs, sd, cfg)
        for k in names:
            sd[k].grad = None
        total.backward()
        if step == 1:
            close(outs[0], a["train/z"])
            close(torch.stack(outs[1]), a["train/y"])
            assert torch.equal(outs[2], a["train/anomaly"].long())
            close(outs[3], a["train/z_q"])
            close(outs[4], a["train/loss_z_q"])
            close(parts["extreme"], a["train/l_ext"])
            close(parts["anomaly"], a["train/l_anom"])
            close(parts["var"], a["train/l_var"])
            close(total.reshape(()), a["train/loss"].reshape(()))
            check_grads(sd, grads, atol=2e-5, rtol=1e-3)
        losses.append(float(total))
        with torch.no_grad():
            for k in names:
                p, m, v = O.adam_step(sd[k].detach(), sd[k].grad, state[k][0], state[k][1], step, cfg.lr, cfg.beta1,
                                      cfg.beta2, 1e-8, cfg.weight_decay)
                state[k] = (m, v)
                sd[k] = p.requires_grad_()
    torch.testing.assert_close(torch.tensor(losses, dtype=torch.float64), a["train/loss_per_step"].double(),
                               rtol=1e-3, atol=1e-4)
    for k in names:
        close(sd[k].detach(), a["sd_after3/" + k], atol=5e-5, rtol=1e-3)


@pytest.mark.parametrize("tag", ["k2_train", "k2_eval", "k8_train"])
def test_vq_nearest(tag):
    a, sd, grads = load_golden("vq_" + tag)
    training = bool(int(a["training"]))
    emb = sd["_codebook.embed"][0].clone().requires_grad_()
    x = a["x"].clone().requires_grad_()
    zq, idx, loss = O.vq_nearest(x, emb, training)
    close(zq, a["z_q"])
    assert torch.equal(idx, a["indices"].long())
    close(loss, a["loss"])
    if training:
        ((zq * a["r"]).sum() + 1.7 * loss.sum()).backward()
        close(x.grad, a["dx"], 1e-6, 1e-3)
        close(emb.grad, grads["_codebook.embed"][0], 1e-6, 1e-3)


def test_evaluators_and_collector():
    import numpy as np
    a, _, _ = load_golden("evaluators")
    tot = sum(O.confusion_counts(a["syn/logits"][i], a["syn/gt"][i], 0.5) for i in range(3))
    assert tot[4:8].tolist() == a["syn/counts"][:4].tolist() and tot[8] == a["syn/counts"][4]
    tot = sum(O.confusion_counts(a["real/logits"][i], a["real/gt"][i], 0.35, a["real/mask"][i]) for i in range(3))
    for l in (0, 1):
        assert tot[4 * l:4 * l + 4].tolist() == [a["real/correct"][l], a["real/seen"][l], a["real/iou_de"][l], a["real/predicted"][l]]
    batches = [(a["col/anomaly_batches"][i].numpy(), a["col/timestep_batches"][i].numpy()) for i in range(6)]
    out, cnt = O.anomaly_majority_vote(batches, a["col/timesteps"].numpy(), 8, tuple(a["col/result"].shape))
    assert np.array_equal(cnt, a["col/counter"].numpy())
    assert np.array_equal(np.nan_to_num(out, nan=-1), np.nan_to_num(a["col/result"].numpy(), nan=-1))


@pytest.mark.parametrize("tag", ["16_32", "32_16"])
def test_swin_hierarchical_encoder(tag):
    """Width change between Swin stages (PatchEmbed3D(in_dim -> dim) + LN, Swin_3D.py:414-419) in the oracle."""
    a, sd, grads = load_golden("swin_hier_" + tag)
    cfg = O.default_cfg(in_channels_dynamic=2, en_embed_dim=[int(v) for v in a["embed_dim"]])
    sdl = leaf({"encoder." + k: v for k, v in sd.items()})
    y = O.swin_encoder(a["x"], sdl, cfg)
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    check_grads(sdl, {"encoder." + k: v for k, v in grads.items()}, atol=2e-5, rtol=1e-3)


def test_encoders_without_patch_embedding():
    """in_chans == embed_dim[0] with patch (1,1,1): the reference builds stage 0 (Swin_3D.py:417-420) / the first conv_block
    (encoder/CNN_3D.py:112-118) WITHOUT a downsample — oracle against the reference's outputs, parameter and input gradients."""
    for name, enc_fn, kw in (("swin_nopatch_c16", O.swin_encoder, dict()), ("cnn_nopatch_c16", O.cnn_encoder, dict(encoder="CNN_3D"))):
        a, sd, grads = load_golden(name)
        assert not any("layers_var.0.0.downsample" in k for k in sd)
        cfg = O.default_cfg(in_channels_dynamic=2, in_channels=16, **kw)
        sdl = leaf({"encoder." + k: v for k, v in sd.items()})
        x = a["x"].clone().requires_grad_()
        y = enc_fn(x, sdl, cfg)
        close(y, a["y"])
        (y * a["r"]).sum().backward()
        check_grads(sdl, {"encoder." + k: v for k, v in grads.items()}, atol=2e-5, rtol=1e-3)
        torch.testing.assert_close(x.grad, a["dx"], atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("tag", ["122", "222_c2"])
def test_swin_encoder_patch_size(tag):
    """en_patch_size != (1,1,1): stage-0 PatchEmbed3D with kernel = stride = patch and zero padding to a multiple of the patch
    (Swin_3D.py:473-491), the encoder on the reduced grid — oracle against the reference's outputs and gradients."""
    a, sd, grads = load_golden("swin_patch_" + tag)
    cfg = O.default_cfg(in_channels_dynamic=2, in_channels=int(a["in_chans"]), en_patch_size=tuple(int(v) for v in a["patch"]))
    sdl = leaf({"encoder." + k: v for k, v in sd.items()})
    y = O.swin_encoder(a["x"], sdl, cfg)
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    check_grads(sdl, {"encoder." + k: v for k, v in grads.items()}, atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("tag", ["16_32", "32_16"])
def test_cnn_hierarchical_encoder(tag):
    """Width change after the first CNN_3D block (conv_block's bias-free PatchEmbed3D(in -> out) + LN downsample,
    encoder/CNN_3D.py:112-118) in the oracle, against the reference's outputs and gradients."""
    a, sd, grads = load_golden("cnn_hier_" + tag)
    cfg = O.default_cfg(encoder="CNN_3D", in_channels_dynamic=2, en_embed_dim=[int(v) for v in a["out_channels"]])
    sdl = leaf({"encoder." + k: v for k, v in sd.items()})
    y = O.cnn_encoder(a["x"], sdl, cfg)
    close(y, a["y"])
    (y * a["r"]).sum().backward()
    check_grads(sdl, {"encoder." + k: v for k, v in grads.items()}, atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("tag", ["plain", "kmeans", "expire"])
def test_vq_ema_kmeans_expire(tag):
    """EuclideanCodebook's EMA update, k-means initialisation and dead-code expiry (VQ.py:213-253, :354-377, :449-475,
    :519-545) restated in the oracle, three training steps against the reference's buffers."""
    a, _, _ = load_golden("vq_ema")
    g = lambda k: a[f"{tag}/{k}"]
    embed, cs, avg = g("sd0/_codebook.embed")[0], g("sd0/_codebook.cluster_size")[0], g("sd0/_codebook.embed_avg")[0]
    initted = bool(g("sd0/_codebook.initted"))
    for step in range(3):
        x = g(f"s{step}/x").clone().requires_grad_()
        x2 = x.detach().reshape(-1, 16)
        if not initted:
            embed, bins = O.vq_kmeans(x2, x2[:8].clone(), 4)
            avg, cs, initted = embed * bins.unsqueeze(1), bins, True
        zq, idx, loss = O.vq_nearest(x, embed, True)
        close(zq, g(f"s{step}/z_q"))
        assert torch.equal(idx, g(f"s{step}/indices").long())
        close(loss, g(f"s{step}/loss"))
        ((zq * g(f"s{step}/r")).sum() + 1.7 * loss.sum()).backward()
        close(x.grad, g(f"s{step}/dx"), 1e-6, 1e-3)
        embed, cs, avg = O.vq_ema_update(x2, idx.reshape(-1), embed, cs, avg, 0.8, 1e-5, 2 if tag == "expire" else 0,
                                         torch.arange(8))
        close(embed, g(f"s{step}/sd/_codebook.embed")[0], 1e-6, 1e-5)
        close(cs, g(f"s{step}/sd/_codebook.cluster_size")[0], 1e-6, 1e-5)
        close(avg, g(f"s{step}/sd/_codebook.embed_avg")[0], 1e-5, 1e-5)
