"""Video-Swin-3D encoder on MI355X — drop-in for the reference's models/encoder/Swin_3D.py.

Same class name, module path (models.encoder.Swin_3D.Swin_3D, see models/build.py:17-20 of the
reference) and constructor keywords as the reference (Swin_3D.py:494-512).  Differences are
internal: the V per-variable stacks are one set of V-stacked parameters, activations are
channels-last tokens (N,V,T,H,W,C), and each SwinTransformerBlock3D is one fused HIP kernel
(idee_swin_block_fwd/bwd), PatchEmbed3D+LN one (idee_patch_embed_ln_*), proj_var two MFMA
convolutions (idee_conv_*).  state_dict()/load_state_dict() speak the reference's keys.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _stacked
from ... import ops


def relative_position_index(ws):
    """(Nw, Nw) int64 index into the bias table.  Reference: Swin_3D.py:115-135."""
    wd, wh, ww = ws
    d, h, w = torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")
    c = torch.stack([d.reshape(-1), h.reshape(-1), w.reshape(-1)])
    rel = c[:, :, None] - c[:, None, :]
    return ((rel[0] + wd - 1) * ((2 * wh - 1) * (2 * ww - 1)) + (rel[1] + wh - 1) * (2 * ww - 1) + (rel[2] + ww - 1)).contiguous()


def drop_path_factors(n_items, rate, device, generator=None):
    """timm DropPath (Swin_3D.py:16) as per-sample factors: bernoulli(keep) / keep, two independent draws per item (the
    attention branch and the MLP branch of a block each call the block's DropPath once, :280-285)."""
    keep = 1.0 - float(rate)
    m = (torch.rand((n_items, 2), device=device, generator=generator) < keep).float()
    return m / keep if keep > 0 else m


class PatchMerging(nn.Module):
    """Patch merging layer — drop-in for the reference's PatchMerging (Swin_3D.py:290-335): same constructor,
    same parameters (`reduction.weight` (2*dim, 4*dim), `norm.{weight,bias}` (4*dim)), x (B,D,H,W,C) channels-last in,
    (B,D',H/2,W/2,2C) out; one HIP kernel per direction (idee_patch_merging_*).  The reference's own encoder never
    instantiates it (Swin_3D.py:578 passes downsample=PatchEmbed3D); it is provided for hierarchical configurations."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("idee_amd PatchMerging: only nn.LayerNorm is built")
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        if x.shape[-1] != self.dim:
            raise ValueError(f"PatchMerging: expected (B,D,H,W,{self.dim}), got {tuple(x.shape)}")
        return ops.PatchMergingFn.apply(x, self.norm.weight, self.norm.bias, self.reduction.weight)


class Swin_3D(_stacked.StackedModule):
    def __init__(self, in_vars: int = 6, in_chans: int = 1, embed_dim: list = None, window_size: list = None,
                 depths: list = None, num_heads: list = None, mlp_ratio: int = 4, drop_rate: float = 0.,
                 attn_drop_rate: float = 0., drop_path_rate: float = 0., qkv_bias: bool = True, qk_scale: float = None,
                 patch_size: tuple = (1, 1, 1), patch_norm: bool = False, use_checkpoint: bool = False):
        super().__init__()
        self.in_vars, self.in_chans = in_vars, in_chans
        self.embed_dim = list(embed_dim) if embed_dim is not None else [16, 16]
        self.num_layers = len(self.embed_dim)
        self.window_size = [tuple(w) for w in (window_size if window_size is not None else [(2, 4, 4), (8, 1, 1)])]
        self.depths = list(depths) if depths is not None else [2, 1]
        self.num_heads = list(num_heads) if num_heads is not None else [2, 2]
        self.mlp_ratio = mlp_ratio
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        self.qkv_bias, self.qk_scale = qkv_bias, qk_scale
        # storage of the token tensors between the fused units: torch.float32 (the reference's arithmetic) or torch.bfloat16
        # (BASELINE config #4: bf16 activations / GEMMs, f32 accumulate); set by VQ_model from config.act_dtype
        self.act_dtype = torch.float32
        self.patch_size = tuple(patch_size)
        self.patch_norm = patch_norm            # no-op in the reference too (Swin_3D.py:417-418)
        self.use_checkpoint = use_checkpoint    # backward always recomputes the block from its input
        if len(self.patch_size) != 3 or any(int(p) < 1 for p in self.patch_size):
            raise ValueError(f"Swin_3D: patch_size must be three positive integers, got {self.patch_size}")
        self._pk = in_chans * self.patch_size[0] * self.patch_size[1] * self.patch_size[2]      # input values per stage-0 token
        if self._pk > 128:
            raise NotImplementedError(f"idee_amd Swin_3D: in_chans x patch volume = {self._pk} > 128 is not built")
        V = in_vars
        self._stages = []
        for s in range(self.num_layers):
            C = self.embed_dim[s]
            in_dim = self.embed_dim[s - 1] if s > 0 else in_chans
            # downsample iff the width changes or (stage 0) the patch is not (1,1,1) (Swin_3D.py:414-419)
            has_ds = in_dim != C or (s == 0 and self.patch_size != (1, 1, 1))
            hid = int(C * mlp_ratio)
            wd, wh, ww = self.window_size[s]
            L = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1)
            if has_ds:
                # stage 0: the Conv3d(kernel = stride = patch) weight (C, in, pt, ph, pw) flattened over its input side
                self.register_parameter(f"s{s}_ds_w", nn.Parameter(torch.empty(V, C, self._pk if s == 0 else in_dim)))
                self.register_parameter(f"s{s}_ds_b", nn.Parameter(torch.zeros(V, C)))
            self.register_buffer(f"s{s}_index", relative_position_index((wd, wh, ww)), persistent=False)
            for b in range(self.depths[s]):
                p = f"s{s}_b{b}_"
                for name, shape in (("table", (V, L, self.num_heads[s])), ("qkv_w", (V, 3 * C, C)), ("qkv_b", (V, 3 * C)),
                                    ("proj_w", (V, C, C)), ("proj_b", (V, C)), ("fc1_w", (V, hid, C)), ("fc1_b", (V, hid)),
                                    ("fc2_w", (V, C, hid)), ("fc2_b", (V, C))):
                    if name == "qkv_b" and not qkv_bias:     # nn.Linear(dim, 3 dim, bias=False), Swin_3D.py:137: a zero, untrained bias
                        self.register_buffer(p + name, torch.zeros(shape), persistent=False)
                        continue
                    self.register_parameter(p + name, nn.Parameter(torch.zeros(shape)))
            self._stages.append((s, C, in_dim, has_ds))
        E = self.embed_dim[-1]
        self.pv_w1 = nn.Parameter(torch.empty(V, E, E, 3, 3, 3))
        self.pv_b1 = nn.Parameter(torch.zeros(V, E))
        self.pv_w2 = nn.Parameter(torch.empty(V, E, E, 3, 3, 3))
        self.pv_b2 = nn.Parameter(torch.zeros(V, E))
        self.init_weights()

    # ---- initialisation (Swin_3D.py:596-613: trunc-normal weights, zero biases) ----
    def weight_names(self):
        """-> (matrix-like weights, biases, relative-position tables)."""
        ws, bs, ts = [], [], []
        for n, _ in self.named_parameters(recurse=False):
            if n.endswith("table"):
                ts.append(n)
            elif n.endswith("_w") or n.endswith("_w1") or n.endswith("_w2"):
                ws.append(n)
            else:
                bs.append(n)
        return ws, bs, ts

    def init_weights(self):
        ws, bs, ts = self.weight_names()
        with torch.no_grad():
            for n in ws + ts:
                nn.init.trunc_normal_(getattr(self, n), std=.02)
            for n in bs:
                getattr(self, n).zero_()

    # ---- reference state-dict keys ----
    def _key_map(self):
        m = []
        for v in range(self.in_vars):
            for s, C, in_dim, has_ds in self._stages:
                base = f"layers_var.{v}.{s}."
                for b in range(self.depths[s]):
                    k, p = f"{base}blocks.{b}.", f"s{s}_b{b}_"
                    m += [(k + "attn.relative_position_bias_table", p + "table", v),
                          (k + "attn.relative_position_index", f"s{s}_index", None),
                          (k + "attn.qkv.weight", p + "qkv_w", v)] + ([(k + "attn.qkv.bias", p + "qkv_b", v)] if self.qkv_bias else []) + [
                          (k + "attn.proj.weight", p + "proj_w", v), (k + "attn.proj.bias", p + "proj_b", v),
                          (k + "mlp.fc1.weight", p + "fc1_w", v), (k + "mlp.fc1.bias", p + "fc1_b", v),
                          (k + "mlp.fc2.weight", p + "fc2_w", v), (k + "mlp.fc2.bias", p + "fc2_b", v)]
                if has_ds:
                    m += [(base + "downsample.proj.weight", f"s{s}_ds_w5", v), (base + "downsample.proj.bias", f"s{s}_ds_b", v)]
        for v in range(self.in_vars):        # the reference registers layers_var (all variables) before proj_var: its key order
            m += [(f"proj_var.{v}.0.weight", "pv_w1", v), (f"proj_var.{v}.0.bias", "pv_b1", v),
                  (f"proj_var.{v}.2.weight", "pv_w2", v), (f"proj_var.{v}.2.bias", "pv_b2", v)]
        return m

    def __getattr__(self, name):
        # "<stage>_ds_w5": the (V, C, Cin, 1, 1, 1) view of the 1x1x1 conv weight, for the state dict
        if name.endswith("_ds_w5"):
            w = super().__getattr__(name[:-1])
            if name.startswith("s0_") and self.patch_size != (1, 1, 1):      # (V, C, Cin, pt, ph, pw)
                return w.view(w.shape[0], w.shape[1], self.in_chans, *self.patch_size)
            return w.view(*w.shape, 1, 1, 1)
        return super().__getattr__(name)

    # ---- forward ----
    def _gather_patches(self, x):
        """x (N,V,C,T,H,W) -> (N,V,C pt ph pw,T',H',W'): PatchEmbed3D's zero padding to a multiple of the patch (Swin_3D.py:476-482)
        and the per-patch gather of its strided convolution, in the weight's (C, pt, ph, pw) order — a view for patch (1,1,1),
        one strided copy otherwise (data movement only; the convolution + LayerNorm is idee_patch_embed_ln_*)."""
        pt, ph, pw = self.patch_size
        if (pt, ph, pw) == (1, 1, 1):
            return x
        N, V, C, T, H, W = x.shape
        x = torch.nn.functional.pad(x, (0, (-W) % pw, 0, (-H) % ph, 0, (-T) % pt))
        T2, H2, W2 = x.shape[3] // pt, x.shape[4] // ph, x.shape[5] // pw
        x = x.reshape(N, V, C, T2, pt, H2, ph, W2, pw).permute(0, 1, 2, 4, 6, 8, 3, 5, 7)
        return x.reshape(N, V, C * pt * ph * pw, T2, H2, W2).contiguous()

    def forward_tokens(self, x):
        """x (N,V,C,T,H,W) -> channels-last tokens (N,V,T,H,W,E)."""
        # element-wise dropout (Mlp.drop, proj_drop: drop_rate; attn_drop: attn_drop_rate; Swin_3D.py:34-41, :138-140) runs inside
        # the fused block from a counter-based generator; one fresh seed per block and step from torch's default generator
        use_drop = self.training and (self.drop_rate > 0.0 or self.attn_drop_rate > 0.0)
        # stochastic depth decay rule: block i of sum(depths) gets linspace(0, drop_path_rate)[i] (Swin_3D.py:562)
        dpr = torch.linspace(0, self.drop_path_rate, sum(self.depths)).tolist()
        blk = 0
        if x.shape[1] != self.in_vars or x.shape[2] != self.in_chans:
            raise ValueError(f"Swin_3D: expected (N,{self.in_vars},{self.in_chans},T,H,W), got {tuple(x.shape)}")
        t = None
        for s, C, in_dim, has_ds in self._stages:
            if has_ds:
                if s == 0:
                    t = ops.PatchEmbedLN.apply(self._gather_patches(x), getattr(self, f"s{s}_ds_w"), getattr(self, f"s{s}_ds_b"),
                                               self.act_dtype)
                else:   # hierarchical configuration: the reference re-embeds with PatchEmbed3D(in_dim -> dim) + LN (:414-419)
                    # (an f32 kernel: under bf16 activation storage the tokens are widened for this one stage boundary)
                    t = ops.StageEmbedLN.apply(t.float(), getattr(self, f"s{s}_ds_w"), getattr(self, f"s{s}_ds_b")).to(self.act_dtype)
            elif s == 0:
                # in_chans == embed_dim[0] with patch (1,1,1): the reference builds stage 0 without a `downsample` (Swin_3D.py:417-420)
                # and its blocks read the input channels directly; here: the input in the kernels' channels-last token layout
                # (data movement only)
                t = x.permute(0, 1, 3, 4, 5, 2).contiguous().to(self.act_dtype)
            wsz = self.window_size[s]
            shift = tuple(i // 2 for i in wsz)
            index = getattr(self, f"s{s}_index")
            for b in range(self.depths[s]):
                p = f"s{s}_b{b}_"
                ps = None
                if self.training and dpr[blk] > 0.0:
                    ps = drop_path_factors(t.shape[0] * t.shape[1], dpr[blk], t.device)
                blk += 1
                # (dropout blocks run on the exact-f32 kernels: under bf16 storage the tokens are widened for the block)
                widen = use_drop and t.dtype != torch.float32
                t = ops.SwinBlock.apply(t.float() if widen else t,
                                        *(getattr(self, p + n) for n in ("qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w",
                                                                         "fc1_b", "fc2_w", "fc2_b", "table")),
                                        index, self.num_heads[s], wsz, (0, 0, 0) if b % 2 == 0 else shift, self.qk_scale, ps,
                                        self.drop_rate if use_drop else 0.0, self.attn_drop_rate if use_drop else 0.0,
                                        int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if use_drop else 0)
                if widen:
                    t = t.to(self.act_dtype)
        return ops.ProjVar.apply(t, self.pv_w1, self.pv_b1, self.pv_w2, self.pv_b2)

    def forward(self, x):
        """x (N,V,C,T,H,W) -> (N,V,E,T,H,W), as Swin_3D.forward (Swin_3D.py:616-636)."""
        return self.forward_tokens(x).permute(0, 1, 5, 2, 3, 4)
