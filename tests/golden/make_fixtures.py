#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by importing the REAL reference
(/root/reference, read-only) on CPU.  Runs only in the authoring container — the
reference never travels to the GPU box; only the .npz data files produced here do.

    python tests/golden/make_fixtures.py

What is imported from the reference: models.build.VQ_model and the modules it pulls in
(Swin_3D, CNN_3D encoder, LFQ, classifier CNN_3D) plus models.losses.  `timm` is absent
from this image; the two symbols the path uses (DropPath, trunc_normal_) are provided by
an in-memory stub.  Parameters are re-randomised after construction
(seeded) so that every bias / table / affine term is exercised — parity is by
state_dict, so the reference's own init RNG is irrelevant.

Each fixture is inputs + expected outputs (+ gradients / post-Adam parameters): data
only, no reference source.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _install_timm_stub():
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            keep = 1 - self.p
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * m / keep

    tl.DropPath = DropPath
    tl.trunc_normal_ = nn.init.trunc_normal_
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})


_install_timm_stub()
sys.path.insert(0, os.path.dirname(HERE))
from conftest import relu_record  # noqa: E402  (ReLU-tie listing: tests/conftest.py)
sys.path.insert(0, "/root/reference")
import config as ref_config  # noqa: E402
from models.build import VQ_model  # noqa: E402
from models import losses as ref_losses  # noqa: E402
from models.encoder import Swin_3D as ref_swin  # noqa: E402
from models.encoder import CNN_3D as ref_cnn  # noqa: E402
from models.codebook.LFQ import LFQ as RefLFQ  # noqa: E402
from models.classifier.CNN_3D import CNN_3D as RefCls  # noqa: E402


def ref_cfg(**kw):
    cfg = ref_config.add_all_arguments(argparse.ArgumentParser()).parse_args([])
    cfg.encoder = "Swin_3D"
    cfg.in_channels = 1
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def rerandomise(module, seed):
    """Seeded, well-scaled values for every parameter (weights ~ 1/sqrt(fan_in), biases 0.1,
    rel-pos tables 0.5, LN affine 1 +- 0.1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "relative_position_bias_table" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / np.sqrt(fan_in)))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def sd_arrays(module, prefix="sd/"):
    return {prefix + k: v.detach().cpu().clone().numpy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1e6:.2f} MB  ({len(arrs)} arrays)")


def rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


# ----------------------------------------------------------------------------
# module-level fixtures
# ----------------------------------------------------------------------------


def fx_swin_block(tag, window, shift, D, H, W, C=16, heads=2, B=2, seed=0):
    blk = ref_swin.SwinTransformerBlock3D(C, heads, window_size=window, shift_size=shift, qkv_bias=True)
    rerandomise(blk, seed)
    x = rand((B, D, H, W, C), seed + 1).requires_grad_()
    ws, ss = ref_swin.get_window_size((D, H, W), window, shift)
    Dp, Hp, Wp = (int(np.ceil(a / b)) * b for a, b in zip((D, H, W), ws))
    mask = ref_swin.compute_mask(Dp, Hp, Wp, ws, ss, x.device)
    y = blk(x, mask)
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x.detach(), y=y.detach(), r=r, dx=x.grad, window=np.array(window), shift=np.array(shift),
               heads=np.array(heads), mask=mask)
    out.update(sd_arrays(blk))
    out.update({"grad/" + k: p.grad for k, p in blk.named_parameters()})
    save("swin_block_" + tag, **out)


def fx_swin_hier(tag, embed_dim, seed):
    """A hierarchical Swin_3D (embed_dim changes between stages: BasicLayer re-embeds with PatchEmbed3D(in_dim -> dim) + LN,
    Swin_3D.py:414-419): whole encoder, forward + every gradient."""
    enc = ref_swin.Swin_3D(in_vars=2, in_chans=1, embed_dim=list(embed_dim), depths=[2, 1], num_heads=[2, 2])
    rerandomise(enc, seed)
    x = rand((1, 2, 1, 8, 8, 12), seed + 1)
    with relu_record() as rec:
        y = enc(x)
    assert rec.ties() == 0, f"swin_hier_{tag}: seed {seed} has a ReLU tie; pick another seed"
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x, y=y.detach(), r=r, embed_dim=np.array(embed_dim))
    out.update(sd_arrays(enc))
    out.update({"grad/" + k: p.grad for k, p in enc.named_parameters()})
    save("swin_hier_" + tag, **out)


if __name__ == "__main__" and "--hier-only" in sys.argv:
    fx_swin_hier("16_32", (16, 32), 1200)
    fx_swin_hier("32_16", (32, 16), 1210)


def fx_cnn_hier(tag, out_channels, seed):
    """A CNN_3D encoder whose width changes after the first block (conv_block builds a bias-free PatchEmbed3D(in -> out) + LN as
    its `downsample`, encoder/CNN_3D.py:112-118): whole encoder, forward + every gradient."""
    enc = ref_cnn.CNN_3D(in_vars=2, in_channels=1, out_channels=list(out_channels))
    rerandomise(enc, seed)
    x = rand((1, 2, 1, 4, 6, 10), seed + 1)
    with relu_record() as rec:
        y = enc(x)
    assert rec.ties() == 0, f"cnn_hier_{tag}: seed {seed} has a ReLU tie; pick another seed"
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x, y=y.detach(), r=r, out_channels=np.array(out_channels))
    out.update(sd_arrays(enc))
    out.update({"grad/" + k: p.grad for k, p in enc.named_parameters()})
    save("cnn_hier_" + tag, **out)


def fx_swin_patch(tag, patch, in_chans, shape, seed):
    """Swin_3D with en_patch_size != (1,1,1): stage 0 embeds with Conv3d(kernel = stride = patch) after zero padding to a
    multiple of the patch (PatchEmbed3D.forward, Swin_3D.py:473-491); whole encoder on the reduced grid, forward + gradients."""
    enc = ref_swin.Swin_3D(in_vars=2, in_chans=in_chans, patch_size=tuple(patch), depths=[2, 1], num_heads=[2, 2])
    rerandomise(enc, seed)
    x = rand((1, 2, in_chans) + tuple(shape), seed + 1)
    with relu_record() as rec:
        y = enc(x)
    assert rec.ties() == 0, f"swin_patch_{tag}: seed {seed} has a ReLU tie; pick another seed"
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x, y=y.detach(), r=r, patch=np.array(patch), in_chans=np.array(in_chans))
    out.update(sd_arrays(enc))
    out.update({"grad/" + k: p.grad for k, p in enc.named_parameters()})
    save("swin_patch_" + tag, **out)


def fx_swin_nopatch(seed):
    """Swin_3D with in_chans == embed_dim[0] and patch (1,1,1): BasicLayer builds stage 0 WITHOUT a downsample (Swin_3D.py:417-420),
    the blocks read the input channels directly; whole encoder, forward + every gradient (incl. the input's)."""
    enc = ref_swin.Swin_3D(in_vars=2, in_chans=16, depths=[2, 1], num_heads=[2, 2])
    assert enc.layers_var[0][0].downsample is None
    rerandomise(enc, seed)
    x = rand((1, 2, 16, 8, 8, 12), seed + 1).requires_grad_()
    with relu_record() as rec:
        y = enc(x)
    assert rec.ties() == 0, f"swin_nopatch: seed {seed} has a ReLU tie; pick another seed"
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x.detach(), y=y.detach(), r=r, dx=x.grad)
    out.update(sd_arrays(enc))
    out.update({"grad/" + k: p.grad for k, p in enc.named_parameters()})
    save("swin_nopatch_c16", **out)


def fx_cnn_nopatch(seed):
    """CNN_3D encoder with in_channels == out_channels[0]: the first conv_block has no downsample (encoder/CNN_3D.py:112-118)."""
    enc = ref_cnn.CNN_3D(in_vars=2, in_channels=16, out_channels=[16, 16])
    assert enc.layers_var[0][0].downsample is None
    rerandomise(enc, seed)
    x = rand((1, 2, 16, 4, 6, 10), seed + 1).requires_grad_()
    with relu_record() as rec:
        y = enc(x)
    assert rec.ties() == 0, f"cnn_nopatch: seed {seed} has a ReLU tie; pick another seed"
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x.detach(), y=y.detach(), r=r, dx=x.grad)
    out.update(sd_arrays(enc))
    out.update({"grad/" + k: p.grad for k, p in enc.named_parameters()})
    save("cnn_nopatch_c16", **out)


if __name__ == "__main__" and "--nopatch-only" in sys.argv:
    fx_swin_nopatch(1800)
    fx_cnn_nopatch(1810)


if __name__ == "__main__" and "--patch-only" in sys.argv:
    fx_swin_patch("122", (1, 2, 2), 1, (8, 9, 14), 1600)
    fx_swin_patch("222_c2", (2, 2, 2), 2, (16, 8, 10), 1610)


if __name__ == "__main__" and "--cnnhier-only" in sys.argv:
    fx_cnn_hier("16_32", (16, 32), 1500)
    fx_cnn_hier("32_16", (32, 16), 1510)


def fx_patch_embed(tag, cls, cin, seed):
    pe = cls(patch_size=(1, 1, 1), in_chans=cin, embed_dim=16, norm_layer=nn.LayerNorm)
    rerandomise(pe, seed)
    x = rand((2, cin, 8, 6, 10), seed + 1)
    y = pe(x)
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x, y=y.detach(), r=r)
    out.update(sd_arrays(pe))
    out.update({"grad/" + k: p.grad for k, p in pe.named_parameters()})
    save("patch_embed_" + tag, **out)


def fx_conv_block(tag, cin, cout, seed):
    cb = ref_cnn.conv_block(cin, cout, (3, 3, 3), 0.0, 0.0)
    rerandomise(cb, seed)
    x = rand((2, cin, 8, 6, 10), seed + 1).requires_grad_()
    with relu_record() as rec:
        y = cb(x)
    assert rec.ties() == 0, f"conv_block_{tag}: seed {seed} has a ReLU tie; pick another seed"
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x.detach(), y=y.detach(), r=r, dx=x.grad)
    out.update(sd_arrays(cb))
    out.update({"grad/" + k: p.grad for k, p in cb.named_parameters()})
    save("conv_block_" + tag, **out)


def fx_proj_var(seed):
    enc = ref_swin.Swin_3D(in_vars=1)
    pv = enc.proj_var[0]
    rerandomise(pv, seed)
    x = rand((2, 16, 8, 6, 10), seed + 1).requires_grad_()
    y = pv(x)
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x.detach(), y=y.detach(), r=r, dx=x.grad)
    out.update(sd_arrays(pv))
    out.update({"grad/" + k: p.grad for k, p in pv.named_parameters()})
    save("proj_var", **out)


def fx_lfq(tag, K, training, seed):
    q = RefLFQ(dim=16, codebook_size=K, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    rerandomise(q, seed)
    q.train(training)
    x = rand((2, 300, 16), seed + 1)
    x[0, :7] = 0.0           # exact-zero tokens
    with torch.no_grad():    # make s == 0 exactly reachable: zero bias on the first code dim
        q.project_in.bias[0] = 0.0
    x.requires_grad_()
    zq, idx, aux = q(x)
    r = rand(zq.shape, seed + 2)
    out = dict(x=x.detach(), z_q=zq.detach(), indices=idx, aux=aux.detach(), r=r, K=np.array(K),
               training=np.array(int(training)), code0=q.indices_to_codes(torch.tensor([0])).detach())
    if training:
        ((zq * r).sum() + 2.5 * aux).backward()
        out["dx"] = x.grad
        out.update({"grad/" + k: p.grad for k, p in q.named_parameters()})
    out.update(sd_arrays(q))
    save("lfq_" + tag, **out)


def fx_lfq_multi(tag, K, ncb, training, seed):
    """LFQ with SEVERAL codebooks (num_codebooks > 1, LFQ.py:92-101: the projected token is split 'b n (c d) -> b n c d', every
    codebook has its own softmax over its K codes; indices come back as (b, n, c), :293-298), forward + every gradient."""
    q = RefLFQ(dim=16, codebook_size=K, num_codebooks=ncb, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    assert q.has_projections and q.keep_num_codebooks_dim
    rerandomise(q, seed)
    q.train(training)
    x = rand((2, 300, 16), seed + 1)
    x[0, :7] = 0.0
    with torch.no_grad():
        q.project_in.bias[0] = 0.0
    x.requires_grad_()
    zq, idx, aux = q(x)
    r = rand(zq.shape, seed + 2)
    code_idx = torch.tensor([[0] * ncb, [K - 1] * ncb, list(range(ncb))]) % K
    out = dict(x=x.detach(), z_q=zq.detach(), indices=idx, aux=aux.detach(), r=r, K=np.array(K), ncb=np.array(ncb),
               training=np.array(int(training)), code_idx=code_idx, codes=q.indices_to_codes(code_idx).detach())
    if training:
        ((zq * r).sum() + 2.5 * aux).backward()
        out["dx"] = x.grad
        out.update({"grad/" + k: p.grad for k, p in q.named_parameters()})
    out.update(sd_arrays(q))
    save("lfq_multi_" + tag, **out)


if __name__ == "__main__" and "--lfqmulti-only" in sys.argv:
    fx_lfq_multi("k2_c2_train", 2, 2, True, 230)
    fx_lfq_multi("k4_c2_train", 4, 2, True, 240)
    fx_lfq_multi("k2_c4_eval", 2, 4, False, 250)


def fx_lfq_noproj(tag, dim, K, training, seed):
    """LFQ WITHOUT projections (dim == num_codebooks * log2 K, LFQ.py:98-101: project_in / project_out are nn.Identity): the
    token's channels are the code bits.  Half the tokens are small (|200 z| ~ 1: soft probabilities, live entropy gradients)."""
    d = int(np.log2(K))
    q = RefLFQ(dim=dim, codebook_size=K, num_codebooks=dim // d, entropy_loss_weight=0.1, diversity_gamma=0.1,
               commitment_loss_weight=3.0)
    assert not q.has_projections
    q.train(training)
    x = rand((2, 300, dim), seed + 1)
    x[:, ::2] *= 0.01
    x[0, :7] = 0.0
    x.requires_grad_()
    zq, idx, aux = q(x)
    r = rand(zq.shape, seed + 2)
    ncb = dim // d
    code_idx = torch.tensor([[0] * ncb, [K - 1] * ncb, list(range(ncb))]) % K
    out = dict(x=x.detach(), z_q=zq.detach(), indices=idx, aux=aux.detach(), r=r, K=np.array(K), dim=np.array(dim),
               training=np.array(int(training)), code_idx=code_idx, codes=q.indices_to_codes(code_idx).detach())
    if training:
        ((zq * r).sum() + 2.5 * aux).backward()
        out["dx"] = x.grad
    out.update(sd_arrays(q))
    save("lfq_noproj_" + tag, **out)


if __name__ == "__main__" and "--lfqnoproj-only" in sys.argv:
    fx_lfq_noproj("e16_k2_train", 16, 2, True, 260)
    fx_lfq_noproj("e16_k4_train", 16, 4, True, 270)
    fx_lfq_noproj("e32_k16_train", 32, 16, True, 280)
    fx_lfq_noproj("e16_k16_eval", 16, 16, False, 290)


def fx_classifier(seed, V=2, n_classes=1):
    c = RefCls(in_var=V, embed_dim=16, dim=16, n_classes=n_classes, drop_rate=0.0)
    rerandomise(c, seed)
    x = rand((2, V, 16, 8, 6, 10), seed + 1).requires_grad_()
    z, y = c(x)
    rz = rand(z.shape, seed + 2)
    ry = rand((V,) + tuple(y[0].shape), seed + 3)
    ((z * rz).sum() + sum((y[v] * ry[v]).sum() for v in range(V))).backward()
    out = dict(x=x.detach(), z=z.detach(), y=torch.stack([t.detach() for t in y]), rz=rz, ry=ry, dx=x.grad)
    out.update(sd_arrays(c))
    out.update({"grad/" + k: p.grad for k, p in c.named_parameters()})
    save("classifier_v%d" % V + ("" if n_classes == 1 else "_k%d" % n_classes), **out)


if __name__ == "__main__" and "--clsk-only" in sys.argv:
    fx_classifier(1700, V=2, n_classes=3)


def fx_losses(seed):
    g = torch.Generator().manual_seed(seed)
    N, V, C, T, H, W = 2, 3, 16, 8, 6, 10
    out = {}
    bce_s, bce_r = ref_losses.BCE_loss_synthetic(), ref_losses.BCE_loss()
    an_s = ref_losses.Anomaly_L1_loss_synthetic(n_dynamic=V, delta_t=T, dim=C)
    an_r = ref_losses.Anomaly_L1_loss(n_dynamic=V, delta_t=T, dim=C)
    pred = torch.randn((N, 1, H, W), generator=g).requires_grad_()
    for tag, tgt in (("mixed", (torch.rand((N, 1, H, W), generator=g) < 0.3).float()),
                     ("zeros", torch.zeros(N, 1, H, W)), ("ones", torch.ones(N, 1, H, W))):
        pred.grad = None
        l = bce_s(pred, tgt)
        l.backward()
        out.update({f"bce_s/{tag}/target": tgt, f"bce_s/{tag}/loss": l.detach(), f"bce_s/{tag}/dpred": pred.grad.clone()})
    out["bce_s/pred"] = pred.detach()
    # real-data BCE (pred (N,H,W), target (N,H,W), mask_valid (N,H,W))
    p3 = torch.randn((N, H, W), generator=g).requires_grad_()
    tgt3 = (torch.rand((N, H, W), generator=g) < 0.3).float()
    mv = (torch.rand((N, H, W), generator=g) < 0.8).float()
    l = bce_r(p3, tgt3, mv)
    l.backward()
    out.update({"bce_r/pred": p3.detach(), "bce_r/target": tgt3, "bce_r/mask_valid": mv, "bce_r/loss": l.detach(),
                "bce_r/dpred": p3.grad})
    # anomaly losses
    zq = torch.randn((N, V, C, T, H, W), generator=g).requires_grad_()
    vq0 = torch.randn((1, C), generator=g)
    me = (torch.rand((N, H, W), generator=g) < 0.2).float()
    l = an_s(zq, me, vq0.clone())
    l.backward()
    out.update({"an_s/z_q": zq.detach(), "an_s/vq0": vq0, "an_s/mask": me, "an_s/loss": l.detach(), "an_s/dz_q": zq.grad.clone()})
    l1 = an_s(zq.detach(), torch.ones(N, H, W), vq0.clone())
    out["an_s/all_extreme_loss"] = l1.detach()       # 0/0 -> nan in the reference
    zq.grad = None
    mvalid = (torch.rand((N, T, H, W), generator=g) < 0.1).float()   # "mask_valid" of Anomaly_L1_loss: (N,T,H,W)
    l = an_r(zq, me, mvalid, vq0.clone())
    l.backward()
    out.update({"an_r/mask_valid": mvalid, "an_r/loss": l.detach(), "an_r/dz_q": zq.grad.clone()})
    save("losses", **out)


# ----------------------------------------------------------------------------
# end-to-end fixtures (train_synthetic.py:175-205 call sequence)
# ----------------------------------------------------------------------------


def fx_e2e(tag, cfg, N, T, H, W, seed, n_steps=3, with_grads=True):
    torch.manual_seed(seed)
    model = VQ_model(cfg)
    rerandomise(model, seed)
    model.train()
    V, C = cfg.in_channels_dynamic, cfg.in_channels
    g = torch.Generator().manual_seed(seed + 1)
    x_d = torch.randn((N, V, C, T, H, W), generator=g).clamp_(-10, 10)
    mask_extreme = (torch.rand((N, H, W), generator=g) < 0.3).float()
    mask_loss = (torch.rand((N, H, W), generator=g) < 0.4).float()
    out = dict(x_d=x_d, mask_extreme=mask_extreme, mask_extreme_loss=mask_loss,
               cfg_V=np.array(V), cfg_C=np.array(C), encoder=np.array(cfg.encoder))
    out.update(sd_arrays(model))
    crit = ref_losses.BCE_loss_synthetic()
    crit_an = ref_losses.Anomaly_L1_loss_synthetic(n_dynamic=V, delta_t=cfg.delta_t, dim=cfg.en_embed_dim[-1])
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay, betas=(cfg.beta1, cfg.beta2))

    # eval-mode forward
    model.eval()
    with torch.no_grad():
        z, y, an, zq, lz = model(x_d)
    out.update({"eval/z": z, "eval/y": torch.stack(y), "eval/anomaly": an, "eval/z_q": zq, "eval/loss_z_q": lz})
    model.train()

    losses = []
    for step in range(n_steps if with_grads else 0):
        opt.zero_grad(set_to_none=True)
        with relu_record() as rec:
            pred, pred_y, anomaly, z_q, loss_z_q = model(x_d)
        assert step > 0 or rec.ties() == 0, f"e2e_{tag}: seed {seed} has a ReLU tie in the compared step; pick another seed"
        tgt = mask_extreme.unsqueeze(1)
        l_ext = crit(pred, tgt)
        l_an = crit_an(z_q, mask_loss, model.vq.indices_to_codes(torch.Tensor([0]).long()).clone().detach())
        l_var = 0
        for k in range(V):
            l_var = l_var + crit(pred_y[k], tgt)
        loss = l_ext + l_an * cfg.lambda_anomaly + l_var + loss_z_q
        loss.backward()
        if step == 0:
            out.update({"train/z": pred.detach(), "train/y": torch.stack([t.detach() for t in pred_y]),
                        "train/anomaly": anomaly, "train/z_q": z_q.detach(), "train/loss_z_q": loss_z_q.detach(),
                        "train/l_ext": l_ext.detach(), "train/l_anom": l_an.detach(), "train/l_var": l_var.detach(),
                        "train/loss": loss.detach()})
            out.update({"grad/" + k: p.grad.clone() for k, p in model.named_parameters()})
        opt.step()
        losses.append(float(loss))
    if with_grads:
        out["train/loss_per_step"] = np.array(losses)
        out.update(sd_arrays(model, prefix=f"sd_after{n_steps}/"))
    save("e2e_" + tag, **out)


def fx_e2e_era5(tag, cfg, N, T, H, W, seed, n_steps=3):
    """The train_ERA5_Land.py:168-198 call sequence on random tensors shaped like the ERA5-Land dataset's items
    (dataset/ERA5_Land_dataset.py:521-625: C = 2 channels mean/std; drought, cold-surface masks)."""
    torch.manual_seed(seed)
    model = VQ_model(cfg)
    rerandomise(model, seed)
    model.train()
    V, C = cfg.in_channels_dynamic, cfg.in_channels
    g = torch.Generator().manual_seed(seed + 1)
    x_d = torch.randn((N, V, C, T, H, W), generator=g).clamp_(-10, 10)
    drought = (torch.rand((N, H, W), generator=g) < 0.3).float()
    drought_loss = (torch.rand((N, H, W), generator=g) < 0.4).float()
    cold = (torch.rand((N, H, W), generator=g) < 0.15).float()
    cold_loss = (torch.rand((N, T, H, W), generator=g) < 0.15).float()
    out = dict(x_d=x_d, mask_drought=drought, mask_drought_loss=drought_loss, mask_cold_surface=cold,
               mask_cold_surface_loss=cold_loss, cfg_V=np.array(V), cfg_C=np.array(C), encoder=np.array(cfg.encoder))
    out.update(sd_arrays(model))
    crit = ref_losses.BCE_loss()
    crit_an = ref_losses.Anomaly_L1_loss(n_dynamic=V, delta_t=cfg.delta_t, dim=cfg.en_embed_dim[-1])
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay, betas=(cfg.beta1, cfg.beta2))
    losses = []
    for step in range(n_steps):
        opt.zero_grad(set_to_none=True)
        mask_valid = 1 - cold                                   # train_ERA5_Land.py:176-177
        mask_valid[mask_valid < 0] = 0
        with relu_record() as rec:
            pred, pred_y, anomaly, z_q, loss_z_q = model(x_d)
        assert step > 0 or rec.ties() == 0, f"e2e_{tag}: seed {seed} has a ReLU tie in the compared step; pick another seed"
        loss = crit(pred[:, 0, ...], drought, mask_valid)
        l_ext = loss.detach().clone()
        l_an = crit_an(z_q, drought_loss.float(), cold_loss, model.vq.indices_to_codes(torch.Tensor([0]).long()).detach())
        l_var = 0
        for k in range(V):
            l_var = l_var + crit(pred_y[k][:, 0, ...], drought.float(), mask_valid)
        loss = loss + l_an * cfg.lambda_anomaly + l_var + loss_z_q
        loss.backward()
        if step == 0:
            out.update({"train/z": pred.detach(), "train/y": torch.stack([t.detach() for t in pred_y]),
                        "train/anomaly": anomaly, "train/z_q": z_q.detach(), "train/loss_z_q": loss_z_q.detach(),
                        "train/l_ext": l_ext, "train/l_anom": l_an.detach(), "train/l_var": l_var.detach(),
                        "train/loss": loss.detach()})
            out.update({"grad/" + k: p.grad.clone() for k, p in model.named_parameters()})
        opt.step()
        losses.append(float(loss.detach()))
    out["train/loss_per_step"] = np.array(losses)
    out.update(sd_arrays(model, prefix=f"sd_after{n_steps}/"))
    save("e2e_" + tag, **out)


if __name__ == "__main__" and "--era5-only" in sys.argv:
    fx_e2e_era5("era5_swin_v3_c2_10x12", ref_cfg(in_channels_dynamic=3, in_channels=2), N=2, T=8, H=10, W=12, seed=1100)


def main():
    fx_swin_block("w244_noshift", (2, 4, 4), (0, 0, 0), 8, 8, 12, seed=10)
    fx_swin_block("w244_shift", (2, 4, 4), (1, 2, 2), 8, 8, 12, seed=20)
    fx_swin_block("w244_shift_pad", (2, 4, 4), (1, 2, 2), 8, 6, 10, seed=30)      # pads to 8x12
    fx_swin_block("w244_noshift_pad", (2, 4, 4), (0, 0, 0), 8, 7, 9, seed=35)
    fx_swin_block("w811", (8, 1, 1), (4, 0, 0), 8, 6, 10, seed=40)                # shift clamped off (D == window)
    fx_swin_block("w811_T16_shift", (8, 1, 1), (4, 0, 0), 16, 4, 6, seed=50)      # genuinely shifted temporal window
    fx_swin_block("w244_c32", (2, 4, 4), (1, 2, 2), 8, 8, 8, C=32, heads=2, seed=60)
    fx_patch_embed("swin_c1", ref_swin.PatchEmbed3D, 1, 100)
    fx_patch_embed("swin_c2", ref_swin.PatchEmbed3D, 2, 110)
    fx_patch_embed("cnn_c1", ref_cnn.PatchEmbed3D, 1, 120)
    fx_conv_block("1_16", 1, 16, 131)    # 130 has a ReLU tie
    fx_conv_block("16_16", 16, 16, 140)
    fx_proj_var(150)
    fx_lfq("k2_train", 2, True, 200)
    fx_lfq("k2_eval", 2, False, 210)
    fx_lfq("k4_train", 4, True, 220)
    fx_classifier(300, V=2)
    fx_losses(400)
    fx_e2e("swin_v2_14x18", ref_cfg(in_channels_dynamic=2), N=2, T=8, H=14, W=18, seed=500)
    fx_e2e("cnn_v2_12x16", ref_cfg(in_channels_dynamic=2, encoder="CNN_3D"), N=2, T=8, H=12, W=16, seed=601)  # 600: tie
    fx_e2e("swin_v6_8x8_c2", ref_cfg(in_channels=2), N=1, T=8, H=8, W=8, seed=700, with_grads=False)
    fx_config_defaults()


def fx_config_defaults():
    """Names and defaults of the reference's flag surface (config.py:18-189) as data."""
    import json
    cfg = vars(ref_config.add_all_arguments(argparse.ArgumentParser()).parse_args([]))
    with open(os.path.join(HERE, "config_defaults.json"), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True, default=list)
    print("config_defaults.json", len(cfg), "flags")


if __name__ == "__main__" and "--config-only" in sys.argv:
    fx_config_defaults()


def fx_vq(tag, K, training, seed):
    """Nearest-neighbour VQ at the reference's defaults (Euclidean codebook, learnable, no EMA, STE, commitment mse):
    models/codebook/VQ.py, class VQ called directly (build.py does not wire it)."""
    from models.codebook.VQ import VQ as RefVQ
    torch.manual_seed(seed)
    q = RefVQ(dim=16, codebook_size=K, codebook_dim=16)
    with torch.no_grad():
        q._codebook.embed.copy_(torch.randn(q._codebook.embed.shape, generator=torch.Generator().manual_seed(seed)) * 0.7)
    q.train(training)
    x = rand((2, 300, 16), seed + 1).requires_grad_()
    zq, idx, loss = q(x)
    r = rand(zq.shape, seed + 2)
    out = dict(x=x.detach(), z_q=zq.detach(), indices=idx, loss=loss.detach(), r=r, K=np.array(K), training=np.array(int(training)))
    if training:
        ((zq * r).sum() + 1.7 * loss.sum()).backward()
        out["dx"] = x.grad
        out.update({"grad/" + k: p.grad for k, p in q.named_parameters()})
    out.update(sd_arrays(q))
    save("vq_" + tag, **out)


def fx_misc(seed=1300):
    """Small API corners: LFQ.indices_to_codes on a video-shaped index map (LFQ.py:152-181) and a Swin block built with
    qkv_bias=False (WindowAttention3D, Swin_3D.py:137)."""
    out = {}
    q = RefLFQ(dim=16, codebook_size=4, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    rerandomise(q, seed)
    idx = torch.randint(0, 4, (2, 3, 4, 5), generator=torch.Generator().manual_seed(seed))
    out.update({"lfq/indices": idx, "lfq/codes": q.indices_to_codes(idx).detach(), "lfq/codes_flat": q.indices_to_codes(idx[0, 0, 0]).detach()})
    out.update(sd_arrays(q, prefix="lfq/sd/"))
    blk = ref_swin.SwinTransformerBlock3D(16, 2, window_size=(2, 4, 4), shift_size=(1, 2, 2), qkv_bias=False)
    rerandomise(blk, seed + 1)
    x = rand((1, 8, 8, 8, 16), seed + 2).requires_grad_()
    mask = ref_swin.compute_mask(8, 8, 8, (2, 4, 4), (1, 2, 2), x.device)
    y = blk(x, mask)
    r = rand(y.shape, seed + 3)
    (y * r).sum().backward()
    out.update({"nobias/x": x.detach(), "nobias/y": y.detach(), "nobias/r": r, "nobias/dx": x.grad})
    out.update(sd_arrays(blk, prefix="nobias/sd/"))
    out.update({"nobias/grad/" + k: p.grad for k, p in blk.named_parameters()})
    save("misc", **out)


if __name__ == "__main__" and "--misc-only" in sys.argv:
    fx_misc()


def fx_vq_ema(seed=830):
    """The reference VQ with the EMA codebook update (ema_update=True, learnable_codebook=False), k-means initialisation and
    dead-code expiry, its sample choices made deterministic by substituting `sample_fn` (the first `num` batch samples) —
    models/codebook/VQ.py:213-253, :354-377, :449-475, :519-545."""
    from models.codebook.VQ import VQ as RefVQ
    first = lambda samples, num: samples[:, :num]
    out = {}
    for tag, kw in (("plain", dict()), ("kmeans", dict(kmeans_init=True, kmeans_iters=4)), ("expire", dict(threshold_ema_dead_code=2))):
        torch.manual_seed(seed)
        q = RefVQ(dim=16, codebook_size=8, codebook_dim=16, ema_update=True, learnable_codebook=False, decay=0.8, **kw)
        q._codebook.sample_fn = first
        if not kw.get("kmeans_init"):
            with torch.no_grad():
                q._codebook.embed.copy_(torch.randn(q._codebook.embed.shape, generator=torch.Generator().manual_seed(seed)) * 0.7)
                q._codebook.embed_avg.copy_(q._codebook.embed)
        out.update({f"{tag}/sd0/" + k: v.clone() for k, v in q.state_dict().items()})
        q.train()
        for step in range(3):
            x = rand((2, 150, 16), seed + 10 * step + 1).requires_grad_()
            zq, idx, loss = q(x)
            r = rand(zq.shape, seed + 10 * step + 2)
            ((zq * r).sum() + 1.7 * loss.sum()).backward()
            out.update({f"{tag}/s{step}/x": x.detach(), f"{tag}/s{step}/r": r, f"{tag}/s{step}/z_q": zq.detach(), f"{tag}/s{step}/indices": idx,
                        f"{tag}/s{step}/loss": loss.detach(), f"{tag}/s{step}/dx": x.grad})
            out.update({f"{tag}/s{step}/sd/" + k: v.clone() for k, v in q.state_dict().items()})
    save("vq_ema", **out)


if __name__ == "__main__" and "--vqema-only" in sys.argv:
    fx_vq_ema()


if __name__ == "__main__" and "--vq-only" in sys.argv:
    fx_vq("k2_train", 2, True, 800)
    fx_vq("k2_eval", 2, False, 810)
    fx_vq("k8_train", 8, True, 820)






def fx_evaluators(seed=900):
    """Confusion-count evaluators of the training drivers (utils/utils_train.py:175-348) on random batches.
    utils_train imports timm.scheduler (absent): stubbed in memory, the schedulers are not used here."""
    sched = types.ModuleType("timm.scheduler")
    for n in ("step_lr", "cosine_lr"):
        m = types.ModuleType("timm.scheduler." + n)
        setattr(m, "StepLRScheduler" if n == "step_lr" else "CosineLRScheduler", object)
        sys.modules["timm.scheduler." + n] = m
    sys.modules["timm.scheduler"] = sched
    import logging
    from utils import utils_train as U
    logger = logging.getLogger("fx")
    g = torch.Generator().manual_seed(seed)
    out = {}
    ev = U.evaluator_synthetic(logger, "Training")
    logits, gts = [], []
    for _ in range(3):
        lg = torch.randn((2, 1, 12, 16), generator=g) * 2
        gt = (torch.rand((2, 1, 12, 16), generator=g) < 0.3).float()
        pred = torch.sigmoid(lg)
        pred_c = pred.clone()
        pred_c[pred > 0.50] = 1
        pred_c[pred <= 0.50] = 0                        # train_synthetic.py:209-212
        ev(pred_c.numpy(), gt.numpy())
        logits.append(lg); gts.append(gt)
    ev.get_results(0.0, 0.0)
    out.update({"syn/logits": torch.stack(logits), "syn/gt": torch.stack(gts),
                "syn/counts": np.array([ev.correct[0], ev.seen[0], ev.iou_de[0], ev.predicted[0], ev.seen_all], dtype=np.float64),
                "syn/metrics": np.array([ev.precision[0], ev.accuracy[0], ev.F1[0], ev.iou[0]], dtype=np.float64)})
    # real-data evaluator: two classes, valid mask, threshold 0.35 (train_ERA5_Land.py:205-211)
    cfg = ref_cfg()
    er = U.evaluator(logger, "Training", cfg)
    logits, gts, mvs = [], [], []
    for _ in range(3):
        lg = torch.randn((2, 12, 16), generator=g) * 2
        gt = (torch.rand((2, 12, 16), generator=g) < 0.3).float()
        mv = (torch.rand((2, 12, 16), generator=g) < 0.8).float()
        pred = torch.sigmoid(lg)
        pred_c = pred.clone()
        pred_c[pred > 0.35] = 1
        pred_c[pred <= 0.35] = 0
        er(pred_c.numpy(), gt.numpy(), mv.numpy())
        logits.append(lg); gts.append(gt); mvs.append(mv)
    er.get_results(0.0, 0.0)
    out.update({"real/logits": torch.stack(logits), "real/gt": torch.stack(gts), "real/mask": torch.stack(mvs),
                "real/correct": np.array(er.correct_label_all, dtype=np.float64), "real/seen": np.array(er.seen_label_all, dtype=np.float64),
                "real/iou_de": np.array(er.iou_de_label_all, dtype=np.float64), "real/predicted": np.array(er.predicted_label_all, dtype=np.float64),
                "real/F1": np.array(er.F1, dtype=np.float64), "real/iou": np.array(er.iou, dtype=np.float64),
                "real/accuracy_all": np.array(er.accuracy_all, dtype=np.float64)})
    # majority-vote anomaly collector (utils_train.py:529-554)
    V, TT, H, W, dt = 3, 20, 4, 5, 8
    cfg.delta_t = dt
    timesteps = np.arange(100, 100 + TT)
    col = U.anomaly_collector(np.zeros((V, TT, H, W)), timesteps, cfg)
    batches = []
    for n in range(6):
        an = (torch.rand((2, V, dt, H, W), generator=g) < 0.4).float().numpy()
        ts = np.array([timesteps[dt - 1 + 2 * n], timesteps[dt + 2 * n]])
        col(an, ts)
        batches.append((an, ts))
    with np.errstate(invalid="ignore", divide="ignore"):
        col.majority_vote()
    out.update({"col/anomaly_batches": np.stack([b[0] for b in batches]), "col/timestep_batches": np.stack([b[1] for b in batches]),
                "col/timesteps": timesteps, "col/result": col.anomaly, "col/counter": col.anomaly_counter})
    save("evaluators", **out)


if __name__ == "__main__" and "--eval-only" in sys.argv:
    fx_evaluators()


def fx_patch_merging(tag, C, B, D, H, W, seed):
    """PatchMerging (Swin_3D.py:290-335) called directly (the reference's configs never instantiate it)."""
    pm = ref_swin.PatchMerging(C)
    rerandomise(pm, seed)
    x = rand((B, D, H, W, C), seed + 1).requires_grad_()
    y = pm(x)
    r = rand(y.shape, seed + 2)
    (y * r).sum().backward()
    out = dict(x=x.detach(), y=y.detach(), r=r, dx=x.grad)
    out.update(sd_arrays(pm))
    out.update({"grad/" + k: p.grad for k, p in pm.named_parameters()})
    save("patch_merging_" + tag, **out)


def fx_patch_mergings():
    fx_patch_merging("c16_even", 16, 2, 4, 6, 8, 950)
    fx_patch_merging("c16_odd", 16, 2, 3, 5, 7, 960)
    fx_patch_merging("c16_d1", 16, 1, 1, 6, 5, 970)
    fx_patch_merging("c32_even", 32, 1, 2, 4, 6, 980)


if __name__ == "__main__" and "--pm-only" in sys.argv:
    fx_patch_mergings()


def fx_dataset_items(seed=1000):
    """Synthetic_Dataset.__getitem__ (dataset/Synthetic_dataset.py:310-376) on an in-memory cube.  The module imports
    xarray (absent here) at its top and the constructor reads NetCDF files; __getitem__ itself touches neither, so the
    class is imported with an empty in-memory `xarray` module object and an instance is made WITHOUT running __init__
    (object.__new__), with only the attributes __getitem__ / __len__ read set by hand."""
    sys.modules.setdefault("xarray", types.ModuleType("xarray"))
    from dataset.Synthetic_dataset import Synthetic_Dataset
    rng = np.random.default_rng(seed)
    V, Tt, H, W, Vs, dt = 3, 14, 6, 9, 2, 8
    cube = dict(dynamic=np.clip(rng.standard_normal((V, Tt, H, W)).astype(np.float32) * 4, -10, 10),
                static=rng.standard_normal((Vs, H, W)).astype(np.float32),
                time=((np.arange(Tt) + 47) % 52).astype(np.float32), timestep=(np.arange(Tt) + 100).astype(np.int64),
                extreme=rng.integers(0, 3, (Tt, H, W)).astype(np.uint8),        # 0 none, 1 extreme, 2 = "other" class
                anomaly=(rng.random((V, Tt, H, W)) < 0.2).astype(np.uint8))
    out = {"cube/" + k: v for k, v in cube.items()}
    out["delta_t"] = np.array(dt)
    names = ("dynamic", "static", "t", "mask_extreme", "mask_extreme_loss", "mask_anomaly", "tstep")
    for is_aug in (False, True):
        ds = object.__new__(Synthetic_Dataset)
        ds._datacube_dynamic, ds._datacube_static = cube["dynamic"], cube["static"]
        ds._datacube_time, ds._datacube_timestep = cube["time"], cube["timestep"]
        ds._extreme, ds._anomaly = cube["extreme"], cube["anomaly"]
        ds.delta_t, ds.is_aug = dt, is_aug
        out["len"] = np.array(len(ds))
        for index in (0, 3, len(ds) - 1):
            for rs in ((0,) if not is_aug else (0, 1, 2, 3, 4, 5)):
                np.random.seed(1234 + rs)
                item = ds[index]
                for n, v in zip(names, item):
                    out[f"item/aug{int(is_aug)}/i{index}/s{rs}/{n}"] = np.ascontiguousarray(v)
    save("dataset_items", **out)


def fx_era5_items(seed=1400):
    """ERA5_Land_Dataset.__getitem__ and its __generate_mask (dataset/ERA5_Land_dataset.py:417-484, :521-625) on in-memory
    arrays: the module's file access is replaced IN MEMORY (an `xarray` stand-in whose load_dataset returns the arrays of this
    fixture, the instance's __load_datacube replaced by an array lookup); __getitem__ / __generate_mask themselves run
    unmodified.  The instance is made without __init__ (which walks directories), with the attributes those two methods read."""
    rng = np.random.default_rng(seed)
    V, Tt, H, W, dt = 3, 12, 6, 9, 4
    store = {}

    class _Var:
        def __init__(self, a): self.values = a
        def isel(self, **kw): return self
    xr_fake = types.ModuleType("xarray")
    xr_fake.load_dataset = lambda path: {k: _Var(v) for k, v in store[path].items()}
    sys.modules["xarray"] = xr_fake
    for m in ("dask", "matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(m, types.ModuleType(m))
    if not hasattr(sys.modules["dask"], "config"):      # the module sets dask's scheduler at import (ERA5_Land_dataset.py:24)
        sys.modules["dask"].config = types.SimpleNamespace(set=lambda **kw: None)
    import importlib
    E = importlib.import_module("dataset.ERA5_Land_dataset")
    E.xr = xr_fake
    cls = E.ERA5_Land_Dataset
    dynamic = (rng.standard_normal((V, 2, Tt, H, W)) * 3 + 1).astype(np.float32)
    dynamic[0, 0, 2, 1, 1] = np.nan
    dynamic[1, 1, 5, 2, 3] = np.inf
    nfiles = [1 + (w % 2) for w in range(Tt)]
    vci = [rng.uniform(0, 100, (n, H, W)).astype(np.float32) for n in nfiles]
    tci = [rng.uniform(0, 100, (n, H, W)).astype(np.float32) for n in nfiles]
    vci[3][0, 0, 0] = np.nan
    cold = [(rng.random((n, H, W)) < 0.2).astype(np.float32) for n in nfiles]
    water = (rng.random((H, W)) < 0.1).astype(np.float32)
    noveg = (rng.random((H, W)) < 0.1).astype(np.float32)
    mean_var = rng.standard_normal(V).astype(np.float32)
    std_var = (1 + rng.random(V)).astype(np.float32)
    for w in range(Tt):
        for f in range(nfiles[w]):
            store[f"noaa/{w}/{f}"] = dict(VCI=vci[w][f], TCI=tci[w][f], mask_cold_surface=cold[w][f])
    out = dict(dynamic=dynamic, mask_water=water, mask_no_vegetation=noveg, mean_var=mean_var, std_var=std_var,
               delta_t=np.array(dt), nfiles=np.array(nfiles))
    for w in range(Tt):
        out[f"vci/{w}"], out[f"tci/{w}"], out[f"cold/{w}"] = vci[w], tci[w], cold[w]
    names = ("dynamic", "mask_drought", "mask_drought_loss", "mask_cold_surface", "mask_cold_surface_loss", "mask_sea", "mask_no_vegetation")
    for is_aug in (False, True):
        ds = object.__new__(cls)
        ds.delta_t, ds.n_lat_window, ds.n_lon_window, ds.threshold, ds.alpha = dt, H, W, 26.0, 0.5
        ds.is_reduce, ds.is_norm, ds.is_clima_scale, ds.is_aug, ds.nan_fill, ds.var_n_dynamic = False, True, False, is_aug, 0.0, V
        ds.x_min, ds.x_max, ds.y_min, ds.y_max, ds.region, ds.CORDEX_regions = 0, W, 0, H, "EUR-11", {"EUR-11": (H, W)}
        ds._ERA5_Land_Dataset__mask_water, ds._ERA5_Land_Dataset__mask_no_vegetation = water, noveg
        ds._ERA5_Land_Dataset__mean_var, ds._ERA5_Land_Dataset__std_var = mean_var, std_var
        ds.files = []
        for idx in range(Tt - dt + 1):
            weeks = [idx + dt - 1 - k for k in range(dt)]
            ds.files.append(([f"era5/{w}xxxxxxx.nc"[:0] + f"week{w:07d}.nc" for w in weeks],
                             [[f"noaa/{w}/{f}" for f in range(nfiles[w])] for w in weeks], np.array(weeks, dtype=np.float32)))
        ds._ERA5_Land_Dataset__load_datacube = lambda files: dynamic[:, :, [int(f[4:11]) for f in files]].copy()
        for index in (0, 3, Tt - dt):
            for rs in ((0,) if not is_aug else (0, 1, 2, 3, 4, 5)):
                np.random.seed(2234 + rs)
                item = ds[index]
                for n, v in zip(names, item[:7]):
                    out[f"item/aug{int(is_aug)}/i{index}/s{rs}/{n}"] = np.ascontiguousarray(v)
    save("era5_items", **out)


if __name__ == "__main__" and "--era5items-only" in sys.argv:
    fx_era5_items()


if __name__ == "__main__" and "--data-only" in sys.argv:
    fx_dataset_items()


if __name__ == "__main__" and not any(a in sys.argv for a in ("--config-only", "--vq-only", "--eval-only", "--pm-only", "--data-only",
                                                                "--era5-only", "--hier-only", "--vqema-only", "--misc-only", "--era5items-only",
                                                                "--cnnhier-only", "--patch-only", "--clsk-only", "--nopatch-only", "--lfqmulti-only", "--lfqnoproj-only")):
    main()
    fx_classifier(1700, V=2, n_classes=3)
    fx_swin_hier("16_32", (16, 32), 1200)
    fx_swin_hier("32_16", (32, 16), 1210)
    fx_cnn_hier("16_32", (16, 32), 1500)
    fx_cnn_hier("32_16", (32, 16), 1510)
    fx_swin_patch("122", (1, 2, 2), 1, (8, 9, 14), 1600)
    fx_swin_patch("222_c2", (2, 2, 2), 2, (16, 8, 10), 1610)
    fx_swin_nopatch(1800)
    fx_cnn_nopatch(1810)
    fx_lfq_multi("k2_c2_train", 2, 2, True, 230)
    fx_lfq_multi("k4_c2_train", 4, 2, True, 240)
    fx_lfq_multi("k2_c4_eval", 2, 4, False, 250)
    fx_e2e_era5("era5_swin_v3_c2_10x12", ref_cfg(in_channels_dynamic=3, in_channels=2), N=2, T=8, H=10, W=12, seed=1100)
    fx_patch_mergings()
    fx_dataset_items()
    fx_era5_items()
    fx_vq("k2_train", 2, True, 800)
    fx_vq("k2_eval", 2, False, 810)
    fx_vq("k8_train", 8, True, 820)
    fx_vq_ema()
    fx_misc()
    fx_evaluators()
