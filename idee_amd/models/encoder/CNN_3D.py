"""3-D CNN encoder on MI355X — drop-in for the reference's models/encoder/CNN_3D.py (class CNN_3D
:147-237, conv_block :74-144, PatchEmbed3D :17-71): same class name, module path and constructor
keywords; V per-variable stacks as V-stacked parameters; channels-last tokens; every conv an MFMA
implicit GEMM (idee_conv_*), the LayerNorm+ReLU+residual tails one streaming kernel each
(idee_ln_relu_res_*).  state_dict()/load_state_dict() speak the reference's keys."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _stacked
from ... import ops


class CNN_3D(_stacked.StackedModule):
    def __init__(self, in_vars: int = 6, in_channels: int = 1, out_channels: list = None, drop_path_rate: float = 0.,
                 drop_rate: float = 0.):
        super().__init__()
        self.in_vars = in_vars
        self.out_channels = list(out_channels) if out_channels is not None else [16, 16]
        self.n_layers = len(self.out_channels)
        self.in_channels = [in_channels] + self.out_channels[:-1]
        self.drop_path_rate, self.drop_rate = drop_path_rate, drop_rate
        V = in_vars
        self._layers = []
        for l, (ci, co) in enumerate(zip(self.in_channels, self.out_channels)):
            has_ds = ci != co                    # encoder/CNN_3D.py:112-118
            if has_ds:
                if l != 0 and not (ci in (16, 32) and co in (16, 32)):
                    raise NotImplementedError("idee_amd CNN_3D: a width change after the first block is built for widths 16 and 32")
                self.register_parameter(f"l{l}_ds_w", nn.Parameter(torch.empty(V, co, ci)))
            for u in (1, 2):
                self.register_parameter(f"l{l}_conv{u}_w", nn.Parameter(torch.empty(V, co, co, 3, 3, 3)))
                self.register_parameter(f"l{l}_norm{u}_g", nn.Parameter(torch.ones(V, co)))
                self.register_parameter(f"l{l}_norm{u}_b", nn.Parameter(torch.zeros(V, co)))
            self._layers.append((l, ci, co, has_ds))
        E = self.out_channels[-1]
        self.pv_w1 = nn.Parameter(torch.empty(V, E, E, 3, 3, 3))
        self.pv_b1 = nn.Parameter(torch.zeros(V, E))
        self.pv_w2 = nn.Parameter(torch.empty(V, E, E, 3, 3, 3))
        self.pv_b2 = nn.Parameter(torch.zeros(V, E))
        self.init_weights()

    def weight_names(self):
        ws, bs = [], []
        for n, _ in self.named_parameters(recurse=False):
            if "_norm" in n:
                continue
            (ws if n.endswith(("_w", "_w1", "_w2")) else bs).append(n)
        return ws, bs, []

    def norm_names(self):
        return ([n for n, _ in self.named_parameters(recurse=False) if n.endswith("_g") and "_norm" in n],
                [n for n, _ in self.named_parameters(recurse=False) if n.endswith("_b") and "_norm" in n])

    def init_weights(self):
        """trunc-normal(0.02) conv weights, zero biases, LayerNorm (1, 0) — encoder/CNN_3D.py:195-213."""
        ws, bs, _ = self.weight_names()
        gs, nbs = self.norm_names()
        with torch.no_grad():
            for n in ws:
                nn.init.trunc_normal_(getattr(self, n), std=.02)
            for n in bs + nbs:
                getattr(self, n).zero_()
            for n in gs:
                getattr(self, n).fill_(1.0)

    def _key_map(self):
        m = []
        for v in range(self.in_vars):
            for l, ci, co, has_ds in self._layers:
                base = f"layers_var.{v}.{l}."
                for u in (1, 2):
                    m += [(f"{base}conv{u}.weight", f"l{l}_conv{u}_w", v), (f"{base}norm{u}.weight", f"l{l}_norm{u}_g", v),
                          (f"{base}norm{u}.bias", f"l{l}_norm{u}_b", v)]
                if has_ds:
                    m.append((base + "downsample.proj.weight", f"l{l}_ds_w5", v))
        for v in range(self.in_vars):        # the reference registers layers_var (all variables) before proj_var: its key order
            m += [(f"proj_var.{v}.0.weight", "pv_w1", v), (f"proj_var.{v}.0.bias", "pv_b1", v),
                  (f"proj_var.{v}.2.weight", "pv_w2", v), (f"proj_var.{v}.2.bias", "pv_b2", v)]
        return m

    def __getattr__(self, name):
        if name.endswith("_ds_w5"):              # (V, C, Cin, 1, 1, 1) view of the 1x1x1 conv weight
            w = super().__getattr__(name[:-1])
            return w.view(*w.shape, 1, 1, 1)
        return super().__getattr__(name)

    def forward_tokens(self, x):
        """x (N,V,C,T,H,W) -> channels-last tokens (N,V,T,H,W,E)."""
        # drop_rate is stored but never applied by the reference's conv_block.forward (encoder/CNN_3D.py:96, :120-144): a no-op
        # here too.  drop_path_rate: every block's two units call DropPath(drop_path_rate) (:97, :140, :142).
        if x.shape[1] != self.in_vars or x.shape[2] != self.in_channels[0]:
            raise ValueError(f"CNN_3D: expected (N,{self.in_vars},{self.in_channels[0]},T,H,W), got {tuple(x.shape)}")
        t = None
        for l, ci, co, has_ds in self._layers:
            if has_ds and l == 0:
                t = ops.PatchEmbedLN.apply(x, getattr(self, f"l{l}_ds_w"), None)
            elif has_ds:      # a width change between blocks: the same bias-free 1x1x1 conv + non-affine LN on tokens (:112-118)
                t = ops.StageEmbedLN.apply(t, getattr(self, f"l{l}_ds_w"), None)
            elif l == 0:
                # in_channels == out_channels[0]: the reference's first conv_block has no `downsample` (encoder/CNN_3D.py:112-118);
                # the input in the kernels' channels-last token layout (data movement only)
                t = x.permute(0, 1, 3, 4, 5, 2).contiguous()
            for u in (1, 2):
                ps = None
                if self.training and self.drop_path_rate > 0:
                    from .Swin_3D import drop_path_factors
                    ps = drop_path_factors(t.shape[0] * t.shape[1], self.drop_path_rate, t.device)[:, 0].contiguous()
                t = ops.ConvLNUnit.apply(t, getattr(self, f"l{l}_conv{u}_w"), getattr(self, f"l{l}_norm{u}_g"),
                                         getattr(self, f"l{l}_norm{u}_b"), ps)
        return ops.ProjVar.apply(t, self.pv_w1, self.pv_b1, self.pv_w2, self.pv_b2)

    def forward(self, x):
        """x (N,V,C,T,H,W) -> (N,V,E,T,H,W), as CNN_3D.forward (encoder/CNN_3D.py:215-237)."""
        return self.forward_tokens(x).permute(0, 1, 5, 2, 3, 4)
