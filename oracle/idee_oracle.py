"""
oracle/idee_oracle.py — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch, CPU, fp32 restatement of the reference's hot path
(encoder -> LFQ -> classifier, the four losses, Adam), written from the
reference's algorithm and citing the reference file:line each function follows
(paths relative to /root/reference).  It is the *checker* for the HIP path:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it.  Nothing under idee_amd/ imports this file; the product path fails loudly
when the HIP library is missing.

Pinning: every function here is checked against golden vectors produced by
importing the real reference in the authoring container
(tests/golden/make_fixtures.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

Parameters are taken from a *reference-keyed* state dict (`sd`), exactly the keys
`VQ_model.state_dict()` yields, so the oracle doubles as the
state-dict compatibility check.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------


def default_cfg(**kw):
    """Model-relevant defaults of config.py:40-132 (encoder forced to Swin_3D, C_in=1)."""
    cfg = SimpleNamespace(
        encoder="Swin_3D", in_channels_dynamic=6, in_channels=1,
        en_embed_dim=[16, 16], en_depths=[2, 1], en_patch_size=(1, 1, 1),
        en_window_size=[(2, 4, 4), (8, 1, 1)], en_mlp_ratio=4.0, en_n_heads=[2, 2],
        en_qkv_bias=True, en_qk_scale=None,
        codebook_size=2, codebook_dim=16, cls_dim=16,
        lambda_entropy=0.1, diversity_gamma=0.1, lambda_commitment=3.0, lambda_anomaly=100.0,
        lr=1e-3, weight_decay=0.003, beta1=0.9, beta2=0.999, delta_t=8,
    )
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


# ----------------------------------------------------------------------------
# Video-Swin pieces  (models/encoder/Swin_3D.py)
# ----------------------------------------------------------------------------


def get_window_size(x_size, window_size, shift_size=None):
    """Clamp window (and zero the shift) where the extent is <= the window. Swin_3D.py:77-90."""
    ws = list(window_size)
    ss = list(shift_size) if shift_size is not None else None
    for i, n in enumerate(x_size):
        if n <= window_size[i]:
            ws[i] = n
            if ss is not None:
                ss[i] = 0
    return tuple(ws) if ss is None else (tuple(ws), tuple(ss))


def relative_position_index(ws):
    """(Nw, Nw) int64 index into the (2wd-1)(2wh-1)(2ww-1) bias table. Swin_3D.py:115-135."""
    wd, wh, ww = ws
    d, h, w = torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")
    c = torch.stack([d.reshape(-1), h.reshape(-1), w.reshape(-1)])  # 3, Nw
    rel = c[:, :, None] - c[:, None, :]  # 3, Nw, Nw
    return ((rel[0] + wd - 1) * ((2 * wh - 1) * (2 * ww - 1))
            + (rel[1] + wh - 1) * (2 * ww - 1) + (rel[2] + ww - 1))


def region_ids(Dp, Hp, Wp, ws, ss):
    """Region label of every position of the rolled (Dp,Hp,Wp) frame: the 27-slice loop of
    compute_mask (Swin_3D.py:340-348) in closed form.  A dim with shift 0 contributes a
    constant (its third slice `slice(-0, None)` overwrites the whole axis)."""
    def axis(P, w, s):
        p = torch.arange(P)
        if s == 0:
            return torch.full((P,), 2, dtype=torch.long)
        return (p >= P - w).long() + (p >= P - s).long()
    a, b, c = axis(Dp, ws[0], ss[0]), axis(Hp, ws[1], ss[1]), axis(Wp, ws[2], ss[2])
    return a[:, None, None] * 9 + b[None, :, None] * 3 + c[None, None, :]


def _partition(x, ws):
    """(B,D,H,W,C) -> (B*nW, Nw, C). Swin_3D.py:45-57."""
    B, D, H, W, C = x.shape
    x = x.reshape(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], C)


def _reverse(win, ws, B, D, H, W):
    """(B*nW, Nw, C) -> (B,D,H,W,C). Swin_3D.py:60-74."""
    x = win.reshape(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def compute_mask(Dp, Hp, Wp, ws, ss):
    """(nW, Nw, Nw) 0 / -100 mask. Swin_3D.py:340-352."""
    rid = region_ids(Dp, Hp, Wp, ws, ss).reshape(1, Dp, Hp, Wp, 1).float()
    r = _partition(rid, ws).squeeze(-1)
    diff = r[:, None, :] - r[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def _hash32(x):
    """lowbias32 on uint32 arrays (numpy): the counter-based generator of the HIP Swin kernels' element-wise dropout
    (idee_amd/csrc/swin_mfma.hip hash32).  Dropout masks are implementation-defined randomness — the reference draws them from
    torch's Philox stream — so parity is checked GIVEN the masks: swin_block(..., dropout=swin_dropout_masks(...))."""
    import numpy as np
    x = x.astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m
    x ^= x >> np.uint64(16)
    return x.astype(np.uint64)


def window_token_ids(B, D, H, W, ws, ss):
    """(B_, Nw) int64: global token index ((b D + d) H + h) W + w of every window position after pad / roll / partition as
    swin_block does them; -1 at padded positions."""
    pd, pb, pr = (-D) % ws[0], (-H) % ws[1], (-W) % ws[2]
    ids = torch.arange(B * D * H * W, dtype=torch.float64).reshape(B, D, H, W, 1)
    ids = F.pad(ids + 1, (0, 0, 0, pr, 0, pb, 0, pd)) - 1          # padding -> -1
    if any(s > 0 for s in ss):
        ids = torch.roll(ids, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    return _partition(ids, ws).squeeze(-1).long()


def swin_dropout_masks(seed, drop, attn_drop, B, D, H, W, C, num_heads, window_size, shift_size, hidden=None):
    """The keep-scale masks (0 or 1/(1-p)) the HIP block applies for `seed`: dict(attn (B_, heads, Nw, Nw) [query, key], proj
    (B,D,H,W,C), gelu (B,D,H,W,hidden), fc2 (B,D,H,W,C)); element ids and salts as in swin_mfma.hip."""
    import numpy as np
    hidden = hidden or 4 * C
    ws, ss = get_window_size((D, H, W), window_size, shift_size)
    Nw = ws[0] * ws[1] * ws[2]
    salt = lambda site: _hash32(np.array([(int(seed) + 0x9E3779B9 * site) & 0xFFFFFFFF], dtype=np.uint64))[0]

    def keep(ids, site, p):
        thr = np.uint64(int(float(np.float32(p).astype(np.float64)) * 4294967296.0))
        h = _hash32((ids.astype(np.uint64) & np.uint64(0xFFFFFFFF)) ^ salt(site))
        inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        return torch.from_numpy(np.where(h >= thr, inv, np.float32(0.0)).astype(np.float32))
    ntok = B * D * H * W
    tok = np.arange(ntok, dtype=np.uint64)
    out = dict(proj=keep((tok[:, None] * C + np.arange(C, dtype=np.uint64)[None]), 2, drop).reshape(B, D, H, W, C),
               gelu=keep((tok[:, None] * hidden + np.arange(hidden, dtype=np.uint64)[None]), 3, drop).reshape(B, D, H, W, hidden),
               fc2=keep((tok[:, None] * C + np.arange(C, dtype=np.uint64)[None]), 4, drop).reshape(B, D, H, W, C))
    wid = window_token_ids(B, D, H, W, ws, ss).numpy()                       # (B_, Nw) query token ids, -1 = padding
    q = np.where(wid < 0, 0, wid).astype(np.uint64)
    ids = ((q[:, None, :, None] * num_heads + np.arange(num_heads, dtype=np.uint64)[None, :, None, None]) * Nw
           + np.arange(Nw, dtype=np.uint64)[None, None, None, :])           # (B_, heads, Nw query, Nw key)
    out["attn"] = keep(ids, 1, attn_drop)
    return out


def window_attention(xw, sd, pre, num_heads, mask=None, qk_scale=None, attn_keep=None, proj_keep=None):
    """WindowAttention3D.forward, Swin_3D.py:145-178.  xw (B_, Nw, C).  attn_keep (B_, heads, Nw, Nw) / proj_keep (B_, Nw, C):
    keep-scale masks of attn_drop (:172) and proj_drop (:176), or None (rates 0 / eval)."""
    B_, Nw, C = xw.shape
    hd = C // num_heads
    scale = qk_scale or hd ** -0.5
    qkv = F.linear(xw, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias"))
    qkv = qkv.reshape(B_, Nw, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * scale, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = sd[pre + "relative_position_index"][:Nw, :Nw].reshape(-1)
    bias = sd[pre + "relative_position_bias_table"][idx].reshape(Nw, Nw, num_heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.reshape(B_ // nW, nW, num_heads, Nw, Nw) + mask[None, :, None]
        attn = attn.reshape(-1, num_heads, Nw, Nw)
    attn = attn.softmax(-1)
    if attn_keep is not None:
        attn = attn * attn_keep
    out = (attn @ v).transpose(1, 2).reshape(B_, Nw, C)
    out = F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    return out * proj_keep if proj_keep is not None else out


def swin_block(x, sd, pre, num_heads, window_size, shift_size, qk_scale=None, path_scale=None, dropout=None):
    """SwinTransformerBlock3D.forward, Swin_3D.py:224-287.  x (B,D,H,W,C).
    dropout: None (rates 0 / eval) or the dict of keep-scale masks of swin_dropout_masks — attn_drop on the probabilities
    (:172), proj_drop (:176; applied here in token order after the window reverse, which is the same thing for an
    element-wise mask), Mlp.drop after the activation and after fc2 (:39, :41).
    path_scale (B, 2) | None: the DropPath factors mask / keep_prob timm's DropPath (:16) would draw for the attention
    branch (:280) and the MLP branch (:265, :285) of every sample."""
    B, D, H, W, C = x.shape
    ws, ss = get_window_size((D, H, W), window_size, shift_size)
    h = F.layer_norm(x, (C,))                                  # norm1, non-affine :229-231
    pd, pb, pr = (-D) % ws[0], (-H) % ws[1], (-W) % ws[2]      # zero pad AFTER the norm :234-238
    h = F.pad(h, (0, 0, 0, pr, 0, pb, 0, pd))
    Dp, Hp, Wp = D + pd, H + pb, W + pr
    shifted = any(s > 0 for s in ss)
    mask = None
    if shifted:
        h = torch.roll(h, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        mask = compute_mask(Dp, Hp, Wp, ws, ss)
    a = window_attention(_partition(h, ws), sd, pre + "attn.", num_heads, mask, qk_scale,
                         attn_keep=dropout["attn"] if dropout is not None else None)
    h = _reverse(a, ws, B, Dp, Hp, Wp)
    if shifted:
        h = torch.roll(h, shifts=ss, dims=(1, 2, 3))
    h = h[:, :D, :H, :W]
    if dropout is not None:
        h = h * dropout["proj"]
    if path_scale is not None:
        h = h * path_scale[:, 0].reshape(B, 1, 1, 1, 1)
    x = x + h                                                   # :280
    m = F.layer_norm(x, (C,))                                   # norm2 non-affine
    m = F.linear(m, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    m = F.gelu(m)                                               # exact erf GELU, Swin_3D.py:31
    if dropout is not None:
        m = m * dropout["gelu"]
    m = F.linear(m, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    if dropout is not None:
        m = m * dropout["fc2"]
    if path_scale is not None:
        m = m * path_scale[:, 1].reshape(B, 1, 1, 1, 1)
    return x + m                                                # :285


def patch_embed(x, sd, pre, patch_size=(1, 1, 1)):
    """PatchEmbed3D.forward: Conv3d(k=s=patch) then non-affine LN over channels.
    Swin_3D.py:473-491 (bias) / encoder/CNN_3D.py:44-71 (no bias).  x (B,C,D,H,W)."""
    _, _, D, H, W = x.shape
    pt, ph, pw = patch_size
    x = F.pad(x, (0, (-W) % pw, 0, (-H) % ph, 0, (-D) % pt))
    x = F.conv3d(x, sd[pre + "proj.weight"], sd.get(pre + "proj.bias"), stride=patch_size)
    E = x.shape[1]
    return F.layer_norm(x.permute(0, 2, 3, 4, 1), (E,)).permute(0, 4, 1, 2, 3)


def patch_merging(x, sd, pre=""):
    """PatchMerging.forward, Swin_3D.py:303-335 (never instantiated by the reference's configs, :578).
    x (B,D,H,W,C) -> (B,D',H/2,W/2,2C).  Note the sampling the reference writes for D > 1 (:323-326):
    x1 and x3 take ODD d together with odd h."""
    B, D, H, W, C = x.shape
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    if D % 2 == 1 and D != 1:
        x = F.pad(x, (0, 0, 0, 0, 0, 0, 0, D % 2))
    if D == 1:
        parts = [x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]]
    else:
        parts = [x[:, 0::2, 0::2, 0::2], x[:, 1::2, 1::2, 0::2], x[:, 0::2, 0::2, 1::2], x[:, 1::2, 1::2, 1::2]]
    x = torch.cat(parts, -1)
    x = F.layer_norm(x, (4 * C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return F.linear(x, sd[pre + "reduction.weight"])


def proj_var(x, sd, pre):
    """Conv3d 3^3 replicate -> ReLU -> Conv3d 3^3 replicate. Swin_3D.py:586-592."""
    x = F.conv3d(F.pad(x, (1,) * 6, mode="replicate"), sd[pre + "0.weight"], sd[pre + "0.bias"])
    x = F.relu(x)
    return F.conv3d(F.pad(x, (1,) * 6, mode="replicate"), sd[pre + "2.weight"], sd[pre + "2.bias"])


def swin_encoder(x, sd, cfg, pre="encoder."):
    """Swin_3D.forward, Swin_3D.py:616-636; BasicLayer.forward :422-446.
    x (N,V,C,T,H,W) -> (N,V,E,T,H,W)."""
    outs = []
    for v in range(cfg.in_channels_dynamic):
        xv = x[:, v]
        for s, depth in enumerate(cfg.en_depths):
            lp = f"{pre}layers_var.{v}.{s}."
            in_dim = cfg.in_channels if s == 0 else cfg.en_embed_dim[s - 1]
            psz = tuple(cfg.en_patch_size) if s == 0 else (1, 1, 1)
            if in_dim != cfg.en_embed_dim[s] or psz != (1, 1, 1):      # :414-419
                xv = patch_embed(xv, sd, lp + "downsample.", psz)
            t = xv.permute(0, 2, 3, 4, 1)
            wsz = tuple(cfg.en_window_size[s])
            for b in range(depth):
                shift = (0, 0, 0) if b % 2 == 0 else tuple(i // 2 for i in wsz)  # :393,405
                t = swin_block(t, sd, f"{lp}blocks.{b}.", cfg.en_n_heads[s], wsz, shift, cfg.en_qk_scale)
            xv = t.permute(0, 4, 1, 2, 3)
        outs.append(proj_var(xv, sd, f"{pre}proj_var.{v}."))
    return torch.stack(outs, dim=1)


# ----------------------------------------------------------------------------
# CNN_3D encoder  (models/encoder/CNN_3D.py)
# ----------------------------------------------------------------------------


def conv_block(x, sd, pre, in_ch, out_ch):
    """conv_block.forward, encoder/CNN_3D.py:120-144: x = sc + ReLU(LN(conv(x))) twice;
    convs 3^3 replicate, no bias; LN affine over channels."""
    if in_ch != out_ch:
        x = patch_embed(x, sd, pre + "downsample.")

    def unit(t, conv, norm):
        y = F.conv3d(F.pad(t, (1,) * 6, mode="replicate"), sd[pre + conv + ".weight"])
        y = F.layer_norm(y.permute(0, 2, 3, 4, 1), (out_ch,), sd[pre + norm + ".weight"], sd[pre + norm + ".bias"])
        return t + F.relu(y.permute(0, 4, 1, 2, 3))
    return unit(unit(x, "conv1", "norm1"), "conv2", "norm2")


def cnn_encoder(x, sd, cfg, pre="encoder."):
    """CNN_3D.forward, encoder/CNN_3D.py:215-237."""
    outs = []
    chans = [cfg.in_channels] + list(cfg.en_embed_dim[:-1])
    for v in range(cfg.in_channels_dynamic):
        xv = x[:, v]
        for l, oc in enumerate(cfg.en_embed_dim):
            xv = conv_block(xv, sd, f"{pre}layers_var.{v}.{l}.", chans[l], oc)
        outs.append(proj_var(xv, sd, f"{pre}proj_var.{v}."))
    return torch.stack(outs, dim=1)


# ----------------------------------------------------------------------------
# LFQ  (models/codebook/LFQ.py)
# ----------------------------------------------------------------------------


def _log_eps(t, eps=1e-5):
    return t.clamp(min=eps).log()       # LFQ.py:52-53


def _entropy(p):
    return (-p * _log_eps(p)).sum(-1)   # LFQ.py:55-56


def lfq_codebook(codebook_size):
    """(K, d) table of +-1 codes, d = log2 K.  LFQ.py:140-146."""
    d = int(math.log2(codebook_size))
    mask = 2 ** torch.arange(d - 1, -1, -1)
    bits = ((torch.arange(codebook_size)[:, None] & mask) != 0).float()
    return bits * 2 - 1, mask


def lfq(z, sd, cfg, training, pre="vq."):
    """LFQ.forward, LFQ.py:183-307, for (b, n, dim) input with projections on.
    Returns z_q (b,n,dim), indices int64 (b,n), aux loss scalar."""
    K = cfg.codebook_size
    codebook, bitmask = lfq_codebook(K)
    x = F.linear(z.float(), sd[pre + "project_in.weight"], sd[pre + "project_in.bias"])   # :211
    q = torch.where(x > 0, torch.ones_like(x), -torch.ones_like(x))                       # :221-222
    xs = x + (q - x).detach() if training else q                                          # :226-230
    indices = ((xs > 0).long() * bitmask).sum(-1)                                         # :234
    if training:
        prob = (2.0 * x @ codebook.t() * 100.0).softmax(-1).reshape(-1, K)                # :239-246
        per_sample = _entropy(prob).mean()                                                # :258
        batch = _entropy(prob.mean(0))                                                    # :260-261
        ent_aux = cfg.lambda_entropy * per_sample - cfg.diversity_gamma * batch           # :262
        commit = ((x - q.detach()) ** 2).mean()                                           # :270-274
    else:
        ent_aux = commit = torch.zeros(())
    out = F.linear(xs, sd[pre + "project_out.weight"], sd[pre + "project_out.bias"])      # :284
    return out, indices, commit * cfg.lambda_commitment + ent_aux                         # :300


def lfq_multi(z, sd, K, ncb, training, w_entropy, gamma, w_commit, pre=""):
    """LFQ.forward with num_codebooks = ncb > 1 (LFQ.py:92-101, :215-300): the projected token is split into ncb groups of
    d = log2 K bits; every codebook has its own softmax over its K codes; per-sample entropy = mean over (tokens, codebooks),
    codebook entropy = mean over codebooks of H(mean prob) (:258-262); commit = mean over every bit (:270-274).
    Returns z_q (b,n,dim), indices int64 (b,n,ncb), aux loss scalar."""
    codebook, bitmask = lfq_codebook(K)
    d = codebook.shape[1]
    proj = (pre + "project_in.weight") in sd       # dim == ncb * d: project_in / project_out are nn.Identity (:98-101)
    x = F.linear(z.float(), sd[pre + "project_in.weight"], sd[pre + "project_in.bias"]) if proj else z.float()
    x = x.reshape(*x.shape[:-1], ncb, d)                                                  # :215
    q = torch.where(x > 0, torch.ones_like(x), -torch.ones_like(x))
    xs = x + (q - x).detach() if training else q
    indices = ((xs > 0).long() * bitmask).sum(-1)                                         # (b, n, c)
    if training:
        prob = (2.0 * torch.einsum("...id,jd->...ij", x, codebook) * 100.0).softmax(-1)   # (b, n, c, K)
        prob = prob.reshape(-1, ncb, K)                                                   # 'b n ... -> (b n) ...'
        per_sample = _entropy(prob).mean()
        batch = _entropy(prob.mean(0)).mean()                                             # avg over tokens per codebook, then mean
        ent_aux = w_entropy * per_sample - gamma * batch
        commit = ((x - q.detach()) ** 2).mean()
    else:
        ent_aux = commit = torch.zeros(())
    out = xs.reshape(*xs.shape[:-2], ncb * d)
    if proj:
        out = F.linear(out, sd[pre + "project_out.weight"], sd[pre + "project_out.bias"])
    return out, indices, commit * w_commit + ent_aux


def lfq_multi_indices_to_codes(indices, sd, K, pre=""):
    """LFQ.indices_to_codes for (..., ncb) index tensors with project_out (LFQ.py:152-181)."""
    codebook, _ = lfq_codebook(K)
    codes = codebook[indices]
    codes = codes.reshape(*codes.shape[:-2], -1)
    if (pre + "project_out.weight") not in sd:
        return codes
    return F.linear(codes, sd[pre + "project_out.weight"], sd[pre + "project_out.bias"])


def lfq_indices_to_codes(indices, sd, cfg, pre="vq."):
    """LFQ.indices_to_codes with project_out, for a 1-D index tensor. LFQ.py:152-181."""
    codebook, _ = lfq_codebook(cfg.codebook_size)
    return F.linear(codebook[indices], sd[pre + "project_out.weight"], sd[pre + "project_out.bias"])


# ----------------------------------------------------------------------------
# classifier  (models/classifier/CNN_3D.py)
# ----------------------------------------------------------------------------


def _cls_head(x, sd, pre):
    """conv(2,3,3) s(2,1,1) p(0,1,1) -> ReLU -> conv -> ReLU -> conv -> squeeze(2).
    classifier/CNN_3D.py:43-58 and :131-137 (drop rate 0)."""
    for i in (1, 2, 3):
        x = F.conv3d(x, sd[f"{pre}conv{i}.weight"], sd[f"{pre}conv{i}.bias"], stride=(2, 1, 1), padding=(0, 1, 1))
        if i < 3:
            x = F.relu(x)
    return x.squeeze(2)


def classifier(zq, sd, pre="cls."):
    """CNN_3D.forward, classifier/CNN_3D.py:112-139.  zq (N,V,C,T,H,W)."""
    N, V, C, T, H, W = zq.shape
    y = [_cls_head(zq[:, v], sd, f"{pre}layers.{v}.") for v in range(V)]
    z = _cls_head(zq.reshape(N, V * C, T, H, W), sd, pre)
    return z, y


# ----------------------------------------------------------------------------
# model assembly  (models/build.py)
# ----------------------------------------------------------------------------


def vq_model_forward(x_d, sd, cfg, training=True):
    """VQ_model.forward, models/build.py:130-159."""
    enc = swin_encoder if cfg.encoder == "Swin_3D" else cnn_encoder
    z = enc(x_d, sd, cfg)
    N, V, C, T, H, W = z.shape
    zt = z.permute(0, 2, 1, 3, 4, 5).reshape(N, C, V * T * H * W).permute(0, 2, 1)        # :150
    z_q, idx, loss_z_q = lfq(zt, sd, cfg, training)
    z_q = z_q.permute(0, 2, 1).reshape(N, C, V, T, H, W).permute(0, 2, 1, 3, 4, 5)        # :153
    anomaly = idx.reshape(N, V, T, H, W)
    zc, y = classifier(z_q, sd)
    return zc, y, anomaly, z_q, loss_z_q.unsqueeze(0)


# ----------------------------------------------------------------------------
# losses  (models/losses.py)
# ----------------------------------------------------------------------------


def _class_weights(values):
    """histc(bins=2) -> log(1.1 + freq^-0.5).  losses.py:115-119 / :80-84."""
    w = torch.histc(values.float(), bins=2)
    w = (w / w.sum()) ** -0.5
    return torch.log(w + 1.1)


def bce_loss_synthetic(pred, target):
    """BCE_loss_synthetic.forward, losses.py:105-124."""
    w = _class_weights(target)[target.long()]
    return (F.binary_cross_entropy_with_logits(pred, target, reduction="none") * w).mean()


def bce_loss(pred, target, mask_valid):
    """BCE_loss.forward, losses.py:71-95."""
    w = _class_weights(target[mask_valid.bool()])[target.long()].clone()
    w[mask_valid == 0] = 0
    l = F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none") * w
    return l.sum() / mask_valid.sum()


def anomaly_l1_synthetic(z_q, mask_extreme, vq_0):
    """Anomaly_L1_loss_synthetic.forward, losses.py:145-168.
    z_q (N,V,C,T,H,W); mask_extreme (N,H,W) in {0,1}; vq_0 (1,C)."""
    N, V, C, T, H, W = z_q.shape
    m = mask_extreme[:, None, None, None].expand(N, V, C, T, H, W)
    wgt = 1 - m
    tgt = vq_0.reshape(1, 1, C, 1, 1, 1).expand_as(z_q)
    tgt = torch.where(m == 1, z_q.detach(), tgt)
    return ((z_q - tgt).abs() * wgt).sum() / wgt.sum()


def anomaly_l1(z_q, mask_extreme, mask_valid, vq_0):
    """Anomaly_L1_loss.forward, losses.py:36-61.  mask_valid (N,T,H,W) is added to the
    extreme mask and clipped to 1 (:49-52)."""
    N, V, C, T, H, W = z_q.shape
    m = mask_extreme[:, None, None, None].expand(N, V, C, T, H, W) + mask_valid[:, None, None].expand(N, V, C, T, H, W)
    m = m.clamp(max=1)
    wgt = 1 - m
    tgt = vq_0.reshape(1, 1, C, 1, 1, 1).expand_as(z_q)
    tgt = torch.where(m == 1, z_q.detach(), tgt)
    return ((z_q - tgt).abs() * wgt).sum() / wgt.sum()


def train_losses_synthetic(outs, mask_extreme, mask_extreme_loss, sd, cfg):
    """The loss assembly of train_synthetic.py:178-201.  mask_extreme (N,H,W), mask_extreme_loss (N,H,W)."""
    pred, pred_y, _anom, z_q, loss_z_q = outs
    tgt = mask_extreme.unsqueeze(1).float()
    l_ext = bce_loss_synthetic(pred, tgt)
    vq0 = lfq_indices_to_codes(torch.tensor([0]), sd, cfg).detach()
    l_anom = anomaly_l1_synthetic(z_q, mask_extreme_loss.float(), vq0)
    l_var = sum(bce_loss_synthetic(p, tgt) for p in pred_y)
    total = l_ext + l_anom * cfg.lambda_anomaly + l_var + loss_z_q
    return total, dict(extreme=l_ext, anomaly=l_anom, var=l_var, z_q=loss_z_q.reshape(()))


def train_losses_era5(outs, mask_drought, mask_drought_loss, mask_cold_surface, mask_cold_surface_loss, sd, cfg):
    """The loss assembly of train_ERA5_Land.py:176-198: mask_valid = max(1 - mask_cold_surface, 0) (:176-177);
    BCE_loss on pred[:, 0] and every pred_y[k][:, 0] (:181, :196); Anomaly_L1_loss with the per-time-step cold-surface
    mask (:184-192).  Masks: (N,H,W) except mask_cold_surface_loss (N,T,H,W)."""
    pred, pred_y, _anom, z_q, loss_z_q = outs
    mask_valid = (1 - mask_cold_surface.float()).clamp(min=0)
    tgt = mask_drought.float()
    l_ext = bce_loss(pred[:, 0], tgt, mask_valid)
    vq0 = lfq_indices_to_codes(torch.tensor([0]), sd, cfg).detach()
    l_anom = anomaly_l1(z_q, mask_drought_loss.float(), mask_cold_surface_loss.float(), vq0)
    l_var = sum(bce_loss(p[:, 0], tgt, mask_valid) for p in pred_y)
    total = l_ext + l_anom * cfg.lambda_anomaly + l_var + loss_z_q
    return total, dict(extreme=l_ext, anomaly=l_anom, var=l_var, z_q=loss_z_q.reshape(()))


# ----------------------------------------------------------------------------
# Adam  (torch.optim.Adam as constructed at train_synthetic.py:127-129)
# ----------------------------------------------------------------------------


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """One torch.optim.Adam update (L2 weight decay folded into the gradient, not AdamW).
    Returns new (p, m, v); `step` is the 1-based step count."""
    g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p = p - (lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps)
    return p, m, v


# ----------------------------------------------------------------------------
# VQ nearest-neighbour op (secondary; models/codebook/VQ.py)
# ----------------------------------------------------------------------------


def vq_nearest(x, embed, training, commitment_weight=1.0):
    """Euclidean nearest-neighbour quantise + straight-through + commitment loss, at the reference VQ's defaults
    (learnable codebook, no EMA, no projections, one head).  VQ.py:44-48 (cdist), :477-555
    (EuclideanCodebook.forward: argmax of -dist, one-hot x embed), :986 (straight-through),
    :1056 (commit = mse(quantize, x), quantize NOT detached because the codebook is learnable), :1083 (loss (1,)).
    x (b,n,d); embed (K,d).  Returns quantised (b,n,d), idx (b,n), loss (1,)."""
    x2 = (x * x).sum(-1, keepdim=True)
    e2 = (embed * embed).sum(-1)
    dist = -(x2 + e2[None, None] - 2 * x @ embed.t()).clamp(min=0).sqrt()
    idx = dist.argmax(-1)
    q = embed[idx]
    loss = torch.zeros(1)
    if training:
        loss = loss + F.mse_loss(q, x) * commitment_weight
        q = x + (q - x).detach()
    return q, idx, loss


def vq_kmeans(samples, means, num_iters):
    """kmeans, VQ.py:213-253, Euclidean, one codebook, from given initial means (the reference samples them).
    samples (M,d), means (K,d) -> (means, bins)."""
    K = means.shape[0]
    bins = None
    for _ in range(num_iters):
        dist = -torch.cdist(samples[None], means[None])[0] if False else -(
            (samples * samples).sum(-1, keepdim=True) + (means * means).sum(-1)[None] - 2 * samples @ means.t()).clamp(min=0).sqrt()
        buckets = dist.argmax(-1)
        bins = torch.bincount(buckets, minlength=K).to(samples.dtype)
        sums = torch.zeros_like(means).index_add_(0, buckets, samples)
        new_means = sums / bins.clamp(min=1).unsqueeze(1)
        means = torch.where((bins == 0).unsqueeze(1), means, new_means)
    return means, bins


def vq_ema_update(x, idx, embed, cluster_size, embed_avg, decay=0.8, eps=1e-5, threshold_dead=0, candidates=None):
    """The codebook update of EuclideanCodebook.forward, VQ.py:519-545 (+ expire_codes_ / replace :449-475 with the
    replacement samples x[candidates[r]] for the r-th dead code).  x (M,d), idx (M) -> new (embed, cluster_size, embed_avg)."""
    K = embed.shape[0]
    onehot = F.one_hot(idx, K).to(x.dtype)
    cluster_size = torch.lerp(cluster_size, onehot.sum(0), 1 - decay)
    embed_avg = torch.lerp(embed_avg, onehot.t() @ x, 1 - decay)
    smoothed = (cluster_size + eps) / (cluster_size.sum() + K * eps) * cluster_size.sum()
    embed = embed_avg / smoothed.unsqueeze(1)
    if threshold_dead > 0:
        dead = cluster_size < threshold_dead
        if bool(dead.any()):
            sampled = x[candidates[: int(dead.sum())]]
            embed, cluster_size, embed_avg = embed.clone(), cluster_size.clone(), embed_avg.clone()
            embed[dead] = sampled
            cluster_size[dead] = float(threshold_dead)
            embed_avg[dead] = sampled * float(threshold_dead)
    return embed, cluster_size, embed_avg


# ----------------------------------------------------------------------------
# training-driver bookkeeping (utils/utils_train.py) — SURVEY.md §8(f) rows 1-2
# ----------------------------------------------------------------------------


def confusion_counts(logits, gt, threshold, mask_valid=None):
    """Counts of evaluator.__call__ (utils_train.py:247-267) / evaluator_synthetic.__call__ (:339-348) after the drivers'
    thresholding (train_synthetic.py:209-212).  -> float64 (9,): [label*4 + (correct, seen, union, predicted)], [8] = pixels."""
    pred = torch.sigmoid(logits)
    pc = torch.where(pred > threshold, torch.ones_like(pred), torch.zeros_like(pred)).flatten()
    g = gt.flatten().float()
    if mask_valid is not None:
        keep = mask_valid.flatten() > 0
        pc, g = pc[keep], g[keep]
    out = torch.zeros(9, dtype=torch.float64)
    for l in (0, 1):
        p, t = pc == l, g == l
        out[4 * l + 0] = (p & t).sum(); out[4 * l + 1] = t.sum(); out[4 * l + 2] = (p | t).sum(); out[4 * l + 3] = p.sum()
    out[8] = g.numel()
    return out


def anomaly_majority_vote(batches, timesteps, delta_t, shape):
    """anomaly_collector (utils_train.py:529-554): batches = [(anomaly (n,V,dt,H,W), last time steps (n,))]."""
    import numpy as np
    acc, cnt = np.zeros(shape), np.zeros(shape)
    for anomaly, ts in batches:
        for n in range(len(ts)):
            idx = int(np.argwhere(ts[n] == timesteps).item())
            acc[:, idx - delta_t + 1: idx + 1] += np.flip(np.asarray(anomaly[n]), axis=1)
            cnt[:, idx - delta_t + 1: idx + 1] += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        r = acc / cnt
    out = r.copy()
    out[r >= 0.5] = 1
    out[r < 0.5] = 0
    return out, cnt
