"""3-D CNN classifier on MI355X — drop-in for the reference's models/classifier/CNN_3D.py
(class CNN_3D :61-139, Layer_v :17-58).  The joint head and the V per-variable heads are six
MFMA implicit-GEMM convolutions (idee_conv_*) run from one autograd node; the V heads' weight
sets are V-stacked.  state_dict()/load_state_dict() speak the reference's keys
(conv{1,2,3}.*, layers.{v}.conv{1,2,3}.*)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _stacked
from ... import ops



class HeadLogits(list):
    """The per-variable heads' logits as the reference returns them — a list of V tensors (N,1,H,W) — made of VIEWS of the one
    (N,V,...) tensor the kernels write (`.stacked`).  torch.split instead of V slices: its backward is one concatenation (V
    slices cost V zero-fills of the whole tensor, V copies and V - 1 adds per step); a consumer that wants all heads at once
    (train.py's loss rows) reads `.stacked` and skips V reshape copies."""

    def __init__(self, y):
        super().__init__(torch.split(y, 1, dim=1))
        self.stacked = y

class CNN_3D(_stacked.StackedModule):
    def __init__(self, in_var: int = 6, embed_dim: int = 16, dim: int = 16, n_classes: int = 1, drop_rate: int = 0.2):
        super().__init__()
        if not 1 <= n_classes <= 16:
            raise NotImplementedError("idee_amd classifier: n_classes must be 1..16 (one output channel block of the joint head's conv3)")
        self.in_var, self.embed_dim, self.dim = in_var, embed_dim * in_var, dim * in_var
        self.n_classes, self.drop_rate = n_classes, drop_rate
        V, E, D, Dj, Ej = in_var, embed_dim, dim, dim * in_var, embed_dim * in_var
        k = (2, 3, 3)
        self.j_w1 = nn.Parameter(torch.empty(Dj, Ej, *k)); self.j_b1 = nn.Parameter(torch.zeros(Dj))
        self.j_w2 = nn.Parameter(torch.empty(Dj, Dj, *k)); self.j_b2 = nn.Parameter(torch.zeros(Dj))
        # the joint head ends in n_classes logits; the per-variable heads always in one (classifier/CNN_3D.py:85, 90)
        self.j_w3 = nn.Parameter(torch.empty(n_classes, Dj, *k)); self.j_b3 = nn.Parameter(torch.zeros(n_classes))
        self.h_w1 = nn.Parameter(torch.empty(V, D, E, *k)); self.h_b1 = nn.Parameter(torch.zeros(V, D))
        self.h_w2 = nn.Parameter(torch.empty(V, D, D, *k)); self.h_b2 = nn.Parameter(torch.zeros(V, D))
        self.h_w3 = nn.Parameter(torch.empty(V, 1, D, *k)); self.h_b3 = nn.Parameter(torch.zeros(V, 1))
        self.reset_parameters()

    def weight_names(self):
        ws = ["j_w1", "j_w2", "j_w3", "h_w1", "h_w2", "h_w3"]
        return ws, [n.replace("_w", "_b") for n in ws], []

    def reset_parameters(self):
        """nn.Conv3d's default init (kaiming-uniform, a=sqrt(5)) — the reference never calls its
        own init_weights (classifier/CNN_3D.py:95-110); VQ_model._init_weights overrides anyway."""
        with torch.no_grad():
            for n in self.weight_names()[0]:
                w = getattr(self, n)
                fan_in = w[0].numel() if w.ndim == 5 else w[0, 0].numel()
                bound = 1.0 / fan_in ** 0.5
                w.uniform_(-bound, bound)
                getattr(self, n.replace("_w", "_b")).uniform_(-bound, bound)

    def init_weights(self):
        with torch.no_grad():
            for n in self.weight_names()[0]:
                nn.init.trunc_normal_(getattr(self, n), std=.04)
                getattr(self, n.replace("_w", "_b")).zero_()

    def _key_map(self):
        m = []
        for i in (1, 2, 3):
            m += [(f"conv{i}.weight", f"j_w{i}", None), (f"conv{i}.bias", f"j_b{i}", None)]
        for v in range(self.in_var):
            for i in (1, 2, 3):
                m += [(f"layers.{v}.conv{i}.weight", f"h_w{i}", v), (f"layers.{v}.conv{i}.bias", f"h_b{i}", v)]
        return m

    def _dropout_args(self):
        """(p, seed) of nn.Dropout(drop_rate) after the first ReLU of the joint head and of every per-variable head
        (classifier/CNN_3D.py:41,53,93,133): active in training only; the seed of the counter-based mask generator is drawn
        from torch's default generator, so torch.manual_seed makes runs repeatable."""
        if self.training and self.drop_rate and self.drop_rate > 0:
            if self.drop_rate >= 1:
                raise ValueError("classifier: drop_rate must be < 1")
            return float(self.drop_rate), int(torch.randint(0, 2 ** 62, (1,)).item())
        return 0.0, 0

    def forward_tokens(self, zq):
        """zq (N,V,T,H,W,C) channels-last -> (z (N,1,H,W), [y_v (N,1,H,W)] * V)."""
        z, y = ops.ClassifierHeads.apply(zq, self.j_w1, self.j_b1, self.j_w2, self.j_b2, *self._joint_out(), self.h_w1,
                                         self.h_b1, self.h_w2, self.h_b2, self.h_w3, self.h_b3, *self._dropout_args())
        return self._joint_logits(z), HeadLogits(y)      # (N,1,H,W) each; (N,1,T/8,H,W) when T > 8, as the reference

    def _joint_out(self):
        """conv3 of the joint head as the kernels take it: one class -> the streaming Cout = 1 kernels; n_classes > 1 -> zero-padded
        to one block of 16 output channels (differentiable padding: the extra rows receive zero gradient)."""
        if self.n_classes == 1:
            return self.j_w3, self.j_b3
        pad = 16 - self.n_classes
        return nn.functional.pad(self.j_w3, (0, 0) * 4 + (0, pad)), nn.functional.pad(self.j_b3, (0, pad))

    def _joint_logits(self, z):
        """(N,T3,H,W,16) padded logits -> (N,n_classes,H,W) (T3 = 1, squeezed as classifier/CNN_3D.py:137) or (N,n_classes,T3,H,W)"""
        if self.n_classes == 1:
            return z
        z = z[..., :self.n_classes].float().permute(0, 4, 1, 2, 3)
        return z.squeeze(2) if z.shape[2] == 1 else z

    def forward_signs(self, signs, project_out, act_dtype=torch.float32):
        """The same outputs as forward_tokens(z_q) for z_q = project_out(signs): LFQ's output has rank
        codebook_dim per variable, so conv1 of every head runs on the (N,T,H,W,16)-channel sign field with
        weights folded through project_out (ops.ClassifierHeadsSigns).  signs (N,V,T,H,W,d)."""
        z, y = ops.ClassifierHeadsSigns.apply(signs, project_out.weight, project_out.bias, self.j_w1, self.j_b1, self.j_w2,
                                              self.j_b2, *self._joint_out(), self.h_w1, self.h_b1, self.h_w2, self.h_b2,
                                              self.h_w3, self.h_b3, *self._dropout_args(), act_dtype)
        return self._joint_logits(z), HeadLogits(y)

    def forward(self, x):
        """x (N,V,C,T,H,W) as CNN_3D.forward (classifier/CNN_3D.py:112-139)."""
        return self.forward_tokens(x.permute(0, 1, 3, 4, 5, 2).contiguous())
