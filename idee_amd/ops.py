"""torch.autograd.Function wrappers over the C-ABI (idee_amd/_lib.py -> libidee_hip.so).

PyTorch is used here for device memory (torch.empty/zeros through the caching allocator),
the current HIP stream, and the autograd graph; every numerical operation of the path is a
call into libidee_hip.so.  Internal activation layout is channels-last tokens
(N, V, T, H, W, C); parameters are V-stacked (leading V axis).
"""
from __future__ import annotations

import ctypes as C
import threading

import torch

from . import _lib as L
from ._lib import ptr


class ZeroArena:
    """One zero-filled f32 buffer per training step for every accumulate-into gradient / statistics buffer of the path
    (~30 of them per step: each used to be its own fill kernel).  The first step under an arena records the sequence of
    requested sizes; every later step takes ONE torch.zeros of the total and hands out 16-byte-aligned slices in the same
    order (256-byte aligned: float atomics of different buffers must not share a memory line).  A request that does not match the recorded sequence falls back to its own torch.zeros (and the arena re-records
    on the next step), so a change of shapes or of control flow is safe.  Used as `with arena:` around forward + backward
    (idee_amd/train.py); without an active arena `_zeros` is plain torch.zeros.

    Re-entrancy: the active arena is looked up BY HIP STREAM (autograd runs a backward node on the stream of its forward), so
    two host threads stepping concurrently on their own streams each see their own arena: a slice is only ever handed to a
    caller whose current stream is the one the buffer was zero-filled on, and `take` is serialised by a lock."""
    _active = {}                      # stream handle -> arena
    _lock = threading.Lock()

    def __init__(self):
        self.plan, self.recording, self.buf, self.pos, self.off, self.ok = None, [], None, 0, 0, True
        self.stream = None

    @staticmethod
    def _stream_key():
        return torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0

    @classmethod
    def current(cls):
        if not cls._active:
            return None
        return cls._active.get(cls._stream_key())

    def __enter__(self):
        self.recording, self.pos, self.off, self.ok = [], 0, 0, True
        self.buf = None
        if self.plan is not None:
            self.buf = torch.zeros(self.plan[1], device=self.plan[2], dtype=torch.float32)     # filled on the current stream
        self.stream = self._stream_key()
        with ZeroArena._lock:
            ZeroArena._active[self.stream] = self
        return self

    def __exit__(self, *exc):
        with ZeroArena._lock:
            if ZeroArena._active.get(self.stream) is self:
                del ZeroArena._active[self.stream]
        if self.buf is None or not self.ok or self.pos != len(self.plan[0]):
            total = sum(n for n, _ in self.recording)
            dev = self.recording[0][1] if self.recording else None
            self.plan = ([n for n, _ in self.recording], total, dev) if self.recording else None
        self.buf = None
        return False

    def take(self, numel, device):
        n = (numel + 63) // 64 * 64        # 256-byte slots: small accumulate-into buffers on separate lines
        with ZeroArena._lock:
            self.recording.append((n, device))
            if self.buf is not None and self.ok and self.pos < len(self.plan[0]) and self.plan[0][self.pos] == n and device == self.plan[2]:
                out = self.buf[self.off:self.off + numel]
                self.pos += 1
                self.off += n
                return out
            self.ok = False
        return torch.zeros(numel, device=device, dtype=torch.float32)


def _zeros(shape, device):
    """Zero-filled f32 tensor: a slice of the ZeroArena active on the CURRENT stream when there is one, else torch.zeros."""
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    numel = 1
    for d in shape:
        numel *= int(d)
    a = ZeroArena.current()
    if a is None or numel == 0:
        return torch.zeros(shape, device=device, dtype=torch.float32)
    return a.take(numel, torch.device(device)).view(shape)


def _zeros_like(t):
    return _zeros(t.shape, t.device)


def _zeros_like_many(*ts):
    """Zero-filled f32 tensors shaped like `ts`, carved (16-byte aligned) out of ONE buffer: one fill kernel instead of len(ts)."""
    sizes = [(t.numel() + 63) // 64 * 64 for t in ts]      # 256-byte slots (float atomics of different buffers on separate lines)
    flat = _zeros(sum(sizes), ts[0].device)
    out, o = [], 0
    for t, n in zip(ts, sizes):
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += n
    return out


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _act(t):
    """Contiguous ACTIVATION tensor in its storage dtype: bf16 stays bf16 (the bf16 activation path, idee_hip.h
    IDEE_DTYPE_BF16), anything else becomes f32."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    return t.contiguous()


def _like(t, ref):
    """t as a contiguous tensor of ref's activation dtype (gradients autograd hands in may be f32 views)."""
    if t.dtype != ref.dtype:
        t = t.to(ref.dtype)
    return t.contiguous()


# ----------------------------------------------------------------------------------------
# PatchEmbed3D(k=1) + LN
# ----------------------------------------------------------------------------------------
class PatchEmbedLN(torch.autograd.Function):
    """x (N,V,Cin,T,H,W) [reference layout], w (V,E,Cin), b (V,E)|None -> (N,V,T,H,W,E).
    Reference: PatchEmbed3D.forward, models/encoder/Swin_3D.py:473-491, encoder/CNN_3D.py:44-71."""

    @staticmethod
    def forward(ctx, x, w, b, act_dtype=torch.float32):
        """act_dtype: storage of the token tensor this produces (torch.float32 | torch.bfloat16)."""
        x, w = _f32c(x), _f32c(w)
        b = _f32c(b) if b is not None else None
        N, V, Cin, T, H, W = x.shape
        E = w.shape[1]
        y = torch.empty((N, V, T, H, W, E), device=x.device, dtype=act_dtype)
        L.call("idee_patch_embed_ln_fwd", ptr(x), ptr(w), ptr(b), ptr(y), N, V, Cin, T * H * W, E, L.act_dtype(y), L.stream())
        ctx.save_for_backward(x, w, b)
        ctx.act = act_dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        N, V, Cin, T, H, W = x.shape
        E = w.shape[1]
        dy = dy.to(ctx.act).contiguous()
        dw = _zeros_like(w)
        db = _zeros_like(b) if b is not None else None
        L.call("idee_patch_embed_ln_bwd", ptr(x), ptr(w), ptr(b), ptr(dy), ptr(dw), ptr(db), N, V, Cin, T * H * W, E,
               L.act_dtype(dy), L.stream())
        return None, dw, db, None


class StageEmbedLN(torch.autograd.Function):
    """PatchEmbed3D (1,1,1) + non-affine LN between two Swin stages of different width, on channels-last tokens:
    x (N,V,T,H,W,Cin), w (V,E,Cin), b (V,E) -> (N,V,T,H,W,E).  Reference: BasicLayer.forward's downsample,
    models/encoder/Swin_3D.py:414-419, :422-428, PatchEmbed3D.forward :473-491."""

    @staticmethod
    def forward(ctx, x, w, b):
        """b may be None: the CNN_3D encoder's downsample has no bias (encoder/CNN_3D.py:24-38, :112-118)."""
        x, w = _f32c(x), _f32c(w)
        b = _f32c(b) if b is not None else None
        N, V, T, H, W, Cin = x.shape
        E = w.shape[1]
        y = torch.empty((N, V, T, H, W, E), device=x.device, dtype=torch.float32)
        L.call("idee_stage_embed_ln_fwd", ptr(x), ptr(w), ptr(b), ptr(y), N * V, V, Cin, T * H * W, E, L.stream())
        ctx.save_for_backward(x, w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        N, V, T, H, W, Cin = x.shape
        E = w.shape[1]
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        if b is not None:
            dw, db = _zeros_like_many(w, b)
        else:
            dw, db = _zeros_like(w), None
        L.call("idee_stage_embed_ln_bwd", ptr(x), ptr(w), ptr(b), ptr(dy), ptr(dx), ptr(dw), ptr(db), N * V, V, Cin, T * H * W, E,
               L.stream())
        return dx, dw, db


# ----------------------------------------------------------------------------------------
# fused Swin block
# ----------------------------------------------------------------------------------------
class PatchMergingFn(torch.autograd.Function):
    """x (B,D,H,W,C) -> LayerNorm(4C)(gather of 4 neighbours) @ w^T, (B,D',ceil(H/2),ceil(W/2),2C).
    Reference: PatchMerging.forward, Swin_3D.py:303-335."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w):
        x, gamma, beta, w = (_f32c(t) for t in (x, gamma, beta, w))
        B, D, H, W, Cc = x.shape
        D2 = 1 if D == 1 else (D + 1) // 2
        y = torch.empty((B, D2, (H + 1) // 2, (W + 1) // 2, 2 * Cc), device=x.device, dtype=torch.float32)
        L.call("idee_patch_merging_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(w), ptr(y), B, D, H, W, Cc, L.stream())
        ctx.save_for_backward(x, gamma, beta, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, w = ctx.saved_tensors
        B, D, H, W, Cc = x.shape
        dy = _f32c(dy)
        dx = torch.zeros_like(x)                  # positions the gather never reads keep a zero gradient
        dg, db, dw = torch.zeros_like(gamma), torch.zeros_like(beta), torch.zeros_like(w)
        L.call("idee_patch_merging_bwd", ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(w), ptr(dx), ptr(dg), ptr(db), ptr(dw),
               B, D, H, W, Cc, L.stream())
        return dx, dg, db, dw


def clamp_window(extent, window, shift):
    """get_window_size, Swin_3D.py:77-90."""
    ws, ss = list(window), list(shift)
    for i, n in enumerate(extent):
        if n <= window[i]:
            ws[i] = n
            ss[i] = 0
    return tuple(ws), tuple(ss)


class SwinBlock(torch.autograd.Function):
    """One SwinTransformerBlock3D on (N,V,D,H,W,C).  Reference: Swin_3D.py:224-287."""

    @staticmethod
    def forward(ctx, x, qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b, table, index, heads, window, shift,
                qk_scale, path_scale=None, drop=0.0, attn_drop=0.0, drop_seed=0):
        """path_scale (N*V, 2) | None: DropPath factors (mask / keep_prob) of the attention and MLP branch per batch item.
        x may be bf16 (the bf16 activation path): y, the saved state and all token gradients then are bf16 too."""
        x = _act(x)
        params = [_f32c(t) for t in (qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b)]
        table = _f32c(table)
        N, V, D, H, W, Cc = x.shape
        ws, ss = clamp_window((D, H, W), window, shift)
        Nw = ws[0] * ws[1] * ws[2]
        hd = Cc // heads
        geom = L.SwinGeom(N, V, D, H, W, Cc, heads, fc1_w.shape[1], ws[0], ws[1], ws[2], ss[0], ss[1], ss[2],
                          float(qk_scale) if qk_scale else float(hd) ** -0.5, L.act_dtype(x), float(drop), float(attn_drop),
                          int(drop_seed) & 0xFFFFFFFF)
        # drop / attn_drop: element-wise dropout rates of the block's four Dropout modules (Mlp.drop x2 + proj_drop; attn_drop),
        # TRAINING ONLY (pass 0 in eval mode); the masks are a function of drop_seed, which the backward reuses through ctx.geom
        idx = index[:Nw, :Nw].contiguous()
        dense = torch.empty((V, heads, Nw, Nw), device=x.device, dtype=torch.float32)
        L.call("idee_relpos_bias_gather", ptr(table), ptr(idx), ptr(dense), V, table.shape[1], heads, Nw, L.stream())
        y = torch.empty_like(x)
        need_bwd = any(ctx.needs_input_grad)
        hs = torch.empty_like(x) if need_bwd else None      # block state before the MLP, for the backward
        ps = _f32c(path_scale) if path_scale is not None else None
        if ps is not None and tuple(ps.shape) != (N * V, 2):
            raise ValueError(f"SwinBlock: path_scale must be ({N * V}, 2), got {tuple(ps.shape)}")
        sp = L.SwinParams(*[ptr(t) for t in params], ptr(dense), ptr(ps))
        L.call("idee_swin_block_fwd", ptr(x), ptr(y), ptr(hs), C.byref(sp), C.byref(geom), L.stream())
        if need_bwd:
            ctx.save_for_backward(x, *params, table, idx, dense, hs)
        ctx.geom, ctx.path_scale = geom, ps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *rest = ctx.saved_tensors
        params, table, idx, dense, hs = rest[:8], rest[8], rest[9], rest[10], rest[11]
        geom = ctx.geom
        dy = _like(dy, x)
        dx = torch.empty_like(x)
        dh = torch.empty_like(x)
        zs = _zeros_like_many(*params, dense, table)     # one fill for all accumulating gradient buffers
        grads, ddense, dtable = list(zs[:8]), zs[8], zs[9]
        sp = L.SwinParams(*[ptr(t) for t in params], ptr(dense), ptr(ctx.path_scale))
        sg = L.SwinGrads(*[ptr(t) for t in grads], ptr(ddense))
        L.call("idee_swin_block_bwd", ptr(x), ptr(hs), ptr(dy), ptr(dx), ptr(dh), C.byref(sp), C.byref(sg), C.byref(geom),
               L.stream())
        V, heads, Nw, _ = dense.shape
        L.call("idee_relpos_bias_scatter", ptr(ddense), ptr(idx), ptr(dtable), V, table.shape[1], heads, Nw, L.stream())
        return (dx, *grads, dtable, None, None, None, None, None, None, None, None, None)


# ----------------------------------------------------------------------------------------
# convolution layers (not autograd by themselves: composed by the Functions below)
# ----------------------------------------------------------------------------------------
class ConvLayer:
    """One convolution of the path, described once (idee_conv_desc) and run fwd / dgrad / wgrad."""

    def __init__(self, mode, B, wsets, T, H, W, Cin, Cout, xs, ys, relu, x_units=None, y_units=None, act=torch.float32):
        """xs / ys: (sB, sT, sH, sW, sG) element strides.  x_units / y_units = (groups per unit, unit stride): two-level
        channel groups (the joint head reads the V slices of z_q; the folded head conv writes the V slices of its output).
        act: activation storage (torch.float32 | torch.bfloat16) of x, y and their gradients (a Cout == 1 output stays f32)."""
        gpu, sU = x_units if x_units is not None else (0, 0)
        ygpu, ysU = y_units if y_units is not None else (0, 0)
        self.act = act
        self.desc = L.ConvDesc(mode, B, wsets, T, H, W, Cin, Cout, *xs, *ys, int(relu), gpu, ygpu, sU, ysU,
                               L.DTYPE_BF16 if act == torch.bfloat16 else L.DTYPE_F32)
        self.mode, self.B, self.wsets, self.Cin, self.Cout = mode, B, wsets, Cin, Cout
        self.ntaps = {L.CONV_K333_REPLICATE: 27, L.CONV_K233_S2_ZERO: 18, L.CONV_K133_ZERO: 9}[mode]
        self.packed_floats = L.load().idee_conv_packed_floats(C.byref(self.desc))
        self.wgrad_floats = L.load().idee_conv_wgrad_floats(C.byref(self.desc))

    def pack(self, w, want_fwd=True, want_dgrad=True):
        dev = w.device
        pdt = torch.bfloat16 if (self.act == torch.bfloat16 and self.Cout != 1) else torch.float32    # bf16 fragments (idee_hip.h)
        pf = torch.empty(self.packed_floats, device=dev, dtype=pdt) if want_fwd else None
        pd = torch.empty(self.packed_floats, device=dev, dtype=pdt) if want_dgrad else None
        L.call("idee_conv_pack_weights", C.byref(self.desc), ptr(w), ptr(pf), ptr(pd), L.stream())
        return pf, pd

    def fwd(self, x, pf, bias, y):
        L.call("idee_conv_fwd", C.byref(self.desc), ptr(x), ptr(pf), ptr(bias), ptr(y), L.stream())

    def dgrad(self, dy, pd, relu_ref, dx, accumulate=False):
        L.call("idee_conv_dgrad", C.byref(self.desc), ptr(dy), ptr(pd), ptr(relu_ref), ptr(dx), int(accumulate), L.stream())

    def dgrad_add(self, dy, pd, addend, dx):
        """dx = addend + conv^T(dy) (no copy of the addend)."""
        L.call("idee_conv_dgrad_add", C.byref(self.desc), ptr(dy), ptr(pd), None, ptr(addend), ptr(dx), L.stream())

    def wgrad(self, x, dy, w_like, want_bias=True):
        """-> (dw in the reference layout of w_like, dbias (wsets, Cout) | None)."""
        dev = x.device
        cpad = ((self.Cout + 15) // 16) * 16
        nw = (self.wgrad_floats + 3) // 4 * 4
        flat = _zeros(nw + (self.wsets * cpad if want_bias else 0), dev)   # one fill (a slice of the step's arena)
        dwp = flat[:self.wgrad_floats]
        dbp = flat[nw:].view(self.wsets, cpad) if want_bias else None
        L.call("idee_conv_wgrad", C.byref(self.desc), ptr(x), ptr(dy), ptr(dwp), ptr(dbp), L.stream())
        dw = torch.empty_like(w_like)
        L.call("idee_conv_unpack_wgrad", C.byref(self.desc), ptr(dwp), ptr(dw), L.stream())
        db = dbp[:, : self.Cout].contiguous() if want_bias else None
        return dw, db


def _cl_strides(T, H, W, Cc):
    """(sB, sT, sH, sW, sG) of a contiguous (B, T, H, W, Cc) tensor with 16-channel groups."""
    return (T * H * W * Cc, H * W * Cc, W * Cc, Cc, 16)


class ProjVar(torch.autograd.Function):
    """Conv3d 3^3 replicate -> ReLU -> Conv3d 3^3 replicate, per variable.
    x (N,V,T,H,W,C); w1, w2 (V,C,C,3,3,3); b1, b2 (V,C).  Reference: Swin_3D.py:586-592."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = _act(x)
        w1, b1, w2, b2 = (_f32c(t) for t in (w1, b1, w2, b2))
        N, V, T, H, W, Cc = x.shape
        s = _cl_strides(T, H, W, Cc)
        c1 = ConvLayer(L.CONV_K333_REPLICATE, N * V, V, T, H, W, Cc, Cc, s, s, True, act=x.dtype)
        c2 = ConvLayer(L.CONV_K333_REPLICATE, N * V, V, T, H, W, Cc, Cc, s, s, False, act=x.dtype)
        p1f, p1d = c1.pack(w1)
        p2f, p2d = c2.pack(w2)
        a1 = torch.empty_like(x)
        y = torch.empty_like(x)
        c1.fwd(x, p1f, b1, a1)
        c2.fwd(a1, p2f, b2, y)
        ctx.save_for_backward(x, a1, w1, w2, p1d, p2d)
        ctx.convs = (c1, c2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a1, w1, w2, p1d, p2d = ctx.saved_tensors
        c1, c2 = ctx.convs
        dy = _like(dy, x)
        dw2, db2 = c2.wgrad(a1, dy, w2)
        da1 = torch.empty_like(a1)
        c2.dgrad(dy, p2d, a1, da1)              # masked by ReLU'(a1)
        dw1, db1 = c1.wgrad(x, da1, w1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            c1.dgrad(da1, p1d, None, dx)
        return dx, dw1, db1, dw2, db2


def _dropout_(t, p, seed):
    """nn.Dropout's mask applied in place to a ReLU output (the 1/(1-p) factor is folded into the next conv's weights)."""
    L.call("idee_dropout_mask_", ptr(t), t.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, L.act_dtype(t), L.stream())


class ClassifierHeads(torch.autograd.Function):
    """The whole classifier: joint head on the V*C channels + V per-variable heads, each
    conv(2,3,3)s(2,1,1) -> ReLU -> conv -> ReLU -> conv -> squeeze.  Reference:
    classifier/CNN_3D.py:112-139, Layer_v.forward :43-58.
      zq (N,V,T,H,W,C)
      joint: jw1 (D,V*C,2,3,3) jb1 (D) jw2 (D,D,2,3,3) jb2 (D) jw3 (1,D,2,3,3) jb3 (1),  D = V*dim
      heads: hw1 (V,dim,C,2,3,3) hb1 (V,dim) hw2 (V,dim,dim,..) hb2 hw3 (V,1,dim,..) hb3 (V,1)
    -> z (N,1,H,W), y (N,V,H,W)"""

    @staticmethod
    def forward(ctx, zq, jw1, jb1, jw2, jb2, jw3, jb3, hw1, hb1, hw2, hb2, hw3, hb3, drop_p=0.0, drop_seed=0):
        """drop_p, drop_seed: nn.Dropout(drop_rate) after the first ReLU of every head (classifier/CNN_3D.py:41,53,93,133) in
        training: keep-mask in place on the activation, 1/(1-p) folded into conv2's weights."""
        zq = _act(zq)
        act = zq.dtype
        ts = [_f32c(t) for t in (jw1, jb1, jw2, jb2, jw3, jb3, hw1, hb1, hw2, hb2, hw3, hb3)]
        jw1, jb1, jw2, jb2, jw3, jb3, hw1, hb1, hw2, hb2, hw3, hb3 = ts
        inv = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        if drop_p > 0:
            jw2, hw2 = jw2 * inv, hw2 * inv
        N, V, T, H, W, Cc = zq.shape
        dim = hw1.shape[1]
        Dj = jw1.shape[0]
        if T < 8:
            raise RuntimeError(f"classifier: T={T} is smaller than the three stride-2 temporal kernels need (the reference "
                               "raises too: kernel size can't be greater than actual input size)")
        T1 = (T - 2) // 2 + 1
        T2 = (T1 - 2) // 2 + 1
        T3 = (T2 - 2) // 2 + 1          # 1 at T = 8 (squeezed); 2 at T = 16: the reference then returns (N,1,2,H,W)
        dev = zq.device
        K = L.CONV_K233_S2_ZERO
        # joint head: channel groups of conv1's input are the V slices of zq
        j1 = ConvLayer(K, N, 1, T, H, W, V * Cc, Dj, (V * T * H * W * Cc, H * W * Cc, W * Cc, Cc, 16),
                       _cl_strides(T1, H, W, Dj), True, x_units=(Cc // 16, T * H * W * Cc), act=act)
        j2 = ConvLayer(K, N, 1, T1, H, W, Dj, Dj, _cl_strides(T1, H, W, Dj), _cl_strides(T2, H, W, Dj), True, act=act)
        J3 = jw3.shape[0]            # 1, or 16: n_classes > 1 padded to one output channel block by the module (z is then (N,T3,H,W,16))
        j3 = ConvLayer(K, N, 1, T2, H, W, Dj, J3, _cl_strides(T2, H, W, Dj),
                       (T3 * H * W, H * W, W, 1, 1) if J3 == 1 else _cl_strides(T3, H, W, J3), False, act=act)
        h1 = ConvLayer(K, N * V, V, T, H, W, Cc, dim, _cl_strides(T, H, W, Cc), _cl_strides(T1, H, W, dim), True, act=act)
        h2 = ConvLayer(K, N * V, V, T1, H, W, dim, dim, _cl_strides(T1, H, W, dim), _cl_strides(T2, H, W, dim), True, act=act)
        h3 = ConvLayer(K, N * V, V, T2, H, W, dim, 1, _cl_strides(T2, H, W, dim), (T3 * H * W, H * W, W, 1, 1), False, act=act)
        convs = (j1, j2, j3, h1, h2, h3)
        weights = (jw1, jw2, jw3, hw1, hw2, hw3)
        packs = [c.pack(w) for c, w in zip(convs, weights)]
        ja1 = torch.empty((N, T1, H, W, Dj), device=dev, dtype=act)
        ja2 = torch.empty((N, T2, H, W, Dj), device=dev, dtype=act)
        z = (torch.empty((N, 1, H, W) if T3 == 1 else (N, 1, T3, H, W), device=dev, dtype=torch.float32) if J3 == 1 else
             torch.empty((N, T3, H, W, J3), device=dev, dtype=act))
        ha1 = torch.empty((N, V, T1, H, W, dim), device=dev, dtype=act)
        ha2 = torch.empty((N, V, T2, H, W, dim), device=dev, dtype=act)
        y = torch.empty((N, V, H, W) if T3 == 1 else (N, V, T3, H, W), device=dev, dtype=torch.float32)
        j1.fwd(zq, packs[0][0], jb1, ja1)
        if drop_p > 0:
            _dropout_(ja1, drop_p, drop_seed)
        j2.fwd(ja1, packs[1][0], jb2, ja2)
        j3.fwd(ja2, packs[2][0], jb3, z)
        h1.fwd(zq, packs[3][0], hb1, ha1)
        if drop_p > 0:
            _dropout_(ha1, drop_p, drop_seed + 1)
        h2.fwd(ha1, packs[4][0], hb2, ha2)
        h3.fwd(ha2, packs[5][0], hb3, y)
        ctx.save_for_backward(zq, ja1, ja2, ha1, ha2, *weights, *[p[1] for p in packs])
        ctx.convs, ctx.inv = convs, inv
        return z, y

    @staticmethod
    def backward(ctx, dz, dy):
        saved = ctx.saved_tensors
        zq, ja1, ja2, ha1, ha2 = saved[:5]
        jw1, jw2, jw3, hw1, hw2, hw3 = saved[5:11]
        pd = saved[11:17]
        j1, j2, j3, h1, h2, h3 = ctx.convs
        dz, dy = (_f32c(dz) if jw3.shape[0] == 1 else dz.to(ja2.dtype).contiguous()), _f32c(dy)
        # joint head
        djw3, djb3 = j3.wgrad(ja2, dz, jw3)
        dja2 = torch.empty_like(ja2)
        j3.dgrad(dz, pd[2], ja2, dja2)
        djw2, djb2 = j2.wgrad(ja1, dja2, jw2)
        dja1 = torch.empty_like(ja1)
        j2.dgrad(dja2, pd[1], ja1, dja1)
        djw1, djb1 = j1.wgrad(zq, dja1, jw1)
        # per-variable heads
        dhw3, dhb3 = h3.wgrad(ha2, dy, hw3)
        dha2 = torch.empty_like(ha2)
        h3.dgrad(dy, pd[5], ha2, dha2)
        dhw2, dhb2 = h2.wgrad(ha1, dha2, hw2)
        dha1 = torch.empty_like(ha1)
        h2.dgrad(dha2, pd[4], ha1, dha1)
        dhw1, dhb1 = h1.wgrad(zq, dha1, hw1)
        dzq = None
        if ctx.needs_input_grad[0]:
            dzq = torch.empty_like(zq)
            j1.dgrad(dja1, pd[0], None, dzq)
            h1.dgrad(dha1, pd[3], None, dzq, accumulate=True)
        if ctx.inv != 1.0:      # conv2 ran with weights * 1/(1-p)
            djw2, dhw2 = djw2 * ctx.inv, dhw2 * ctx.inv
        return (dzq, djw1, djb1.reshape(-1), djw2, djb2.reshape(-1), djw3, djb3.reshape(-1), dhw1, dhb1, dhw2, dhb2, dhw3,
                dhb3, None, None)


class ClassifierHeadsSigns(torch.autograd.Function):
    """ClassifierHeads fed by LFQ's sign bits instead of z_q (exact refactoring, see
    idee_hip.h "Rank-d classifier input"): z_q = b_out + signs @ w_out^T has rank d per variable, so both first
    convolutions run as (1,3,3) convolutions on the paired (N,T/2,H,W,CP) sign field with weights folded through
    [w_out | b_out] — K = 9 taps x 16 channels instead of 18 x 96 for the default model.  Layers 2 and 3 are those of
    ClassifierHeads.
      signs (N,V,T,H,W,D) +-1;  w_out (C,D), b_out (C): LFQ.project_out;  remaining arguments as ClassifierHeads."""

    @staticmethod
    def forward(ctx, signs, w_out, b_out, jw1, jb1, jw2, jb2, jw3, jb3, hw1, hb1, hw2, hb2, hw3, hb3, drop_p=0.0, drop_seed=0,
                act=torch.float32):
        """act: storage of the sign field and of the hidden activations (torch.float32 | torch.bfloat16); logits are f32."""
        ts = [_f32c(t) for t in (signs, w_out, b_out, jw1, jb1, jw2, jb2, jw3, jb3, hw1, hb1, hw2, hb2, hw3, hb3)]
        signs, w_out, b_out, jw1, jb1, jw2, jb2, jw3, jb3, hw1, hb1, hw2, hb2, hw3, hb3 = ts
        inv = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0      # nn.Dropout after the first ReLU, as ClassifierHeads
        if drop_p > 0:
            jw2, hw2 = jw2 * inv, hw2 * inv
        N, V, T, H, W, D = signs.shape
        Cc = w_out.shape[0]
        dim = hw1.shape[1]
        Dj = jw1.shape[0]
        if T < 8:
            raise RuntimeError(f"classifier: T={T} is smaller than the three stride-2 temporal kernels need (the reference "
                               "raises too: kernel size can't be greater than actual input size)")
        if dim % 16 or Dj % 16:
            raise RuntimeError("classifier (sign-field path): cls_dim must be a multiple of 16")
        T1 = (T - 2) // 2 + 1
        T2 = (T1 - 2) // 2 + 1
        T3 = (T2 - 2) // 2 + 1
        CS = ((2 * (V * D + 1) + 15) // 16) * 16        # both frames of a temporal pair side by side in the channels
        dev = signs.device
        K, KP = L.CONV_K233_S2_ZERO, L.CONV_K133_ZERO
        st = L.stream()
        sx = torch.empty((N, T1, H, W, CS), device=dev, dtype=act)
        L.call("idee_lfq_sign_field_fwd", ptr(signs), ptr(sx), N, V, T, H * W, D, CS, L.act_dtype(sx), st)
        jwe = torch.empty((Dj, CS, 1, 3, 3), device=dev, dtype=torch.float32)
        hwe = torch.empty((V * dim, CS, 1, 3, 3), device=dev, dtype=torch.float32)
        L.call("idee_lfq_fold_weights_fwd", ptr(jw1), ptr(w_out), ptr(b_out), ptr(jwe), Dj, V, Cc, D, CS, 0, st)
        L.call("idee_lfq_fold_weights_fwd", ptr(hw1), ptr(w_out), ptr(b_out), ptr(hwe), dim, V, Cc, D, CS, 1, st)
        sxs = _cl_strides(T1, H, W, CS)
        j1 = ConvLayer(KP, N, 1, T1, H, W, CS, Dj, sxs, _cl_strides(T1, H, W, Dj), True, act=act)
        # the heads' conv1 writes variable v's `dim` channels into slice v of ha1 (N,V,T1,H,W,dim)
        h1 = ConvLayer(KP, N, 1, T1, H, W, CS, V * dim, sxs, (V * T1 * H * W * dim, H * W * dim, W * dim, dim, 16), True,
                       y_units=(dim // 16, T1 * H * W * dim), act=act)
        j2 = ConvLayer(K, N, 1, T1, H, W, Dj, Dj, _cl_strides(T1, H, W, Dj), _cl_strides(T2, H, W, Dj), True, act=act)
        J3 = jw3.shape[0]            # 1, or 16 (n_classes > 1 padded by the module: see ClassifierHeads)
        j3 = ConvLayer(K, N, 1, T2, H, W, Dj, J3, _cl_strides(T2, H, W, Dj),
                       (T3 * H * W, H * W, W, 1, 1) if J3 == 1 else _cl_strides(T3, H, W, J3), False, act=act)
        h2 = ConvLayer(K, N * V, V, T1, H, W, dim, dim, _cl_strides(T1, H, W, dim), _cl_strides(T2, H, W, dim), True, act=act)
        h3 = ConvLayer(K, N * V, V, T2, H, W, dim, 1, _cl_strides(T2, H, W, dim), (T3 * H * W, H * W, W, 1, 1), False, act=act)
        convs = (j1, j2, j3, h1, h2, h3)
        weights = (jwe, jw2, jw3, hwe, hw2, hw3)
        need_dx = ctx.needs_input_grad[0]
        packs = [c.pack(w, want_dgrad=(need_dx or i not in (0, 3))) for i, (c, w) in enumerate(zip(convs, weights))]
        ja1 = torch.empty((N, T1, H, W, Dj), device=dev, dtype=act)
        ja2 = torch.empty((N, T2, H, W, Dj), device=dev, dtype=act)
        z = (torch.empty((N, 1, H, W) if T3 == 1 else (N, 1, T3, H, W), device=dev, dtype=torch.float32) if J3 == 1 else
             torch.empty((N, T3, H, W, J3), device=dev, dtype=act))
        ha1 = torch.empty((N, V, T1, H, W, dim), device=dev, dtype=act)
        ha2 = torch.empty((N, V, T2, H, W, dim), device=dev, dtype=act)
        y = torch.empty((N, V, H, W) if T3 == 1 else (N, V, T3, H, W), device=dev, dtype=torch.float32)
        j1.fwd(sx, packs[0][0], jb1, ja1)
        if drop_p > 0:
            _dropout_(ja1, drop_p, drop_seed)
        j2.fwd(ja1, packs[1][0], jb2, ja2)
        j3.fwd(ja2, packs[2][0], jb3, z)
        h1.fwd(sx, packs[3][0], hb1.reshape(-1), ha1)
        if drop_p > 0:
            _dropout_(ha1, drop_p, drop_seed + 1)
        h2.fwd(ha1, packs[4][0], hb2, ha2)
        h3.fwd(ha2, packs[5][0], hb3, y)
        pd = [p[1] if p[1] is not None else torch.empty(0, device=dev) for p in packs]
        ctx.save_for_backward(sx, ja1, ja2, ha1, ha2, jw1, hw1, w_out, b_out, *weights, *pd)
        ctx.convs, ctx.inv = convs, inv
        ctx.geom = (N, V, T, H, W, D, CS, Cc, dim, Dj)
        return z, y

    @staticmethod
    def backward(ctx, dz, dy):
        saved = ctx.saved_tensors
        sx, ja1, ja2, ha1, ha2, jw1, hw1, w_out, b_out = saved[:9]
        jwe, jw2, jw3, hwe, hw2, hw3 = saved[9:15]
        pd = saved[15:21]
        j1, j2, j3, h1, h2, h3 = ctx.convs
        N, V, T, H, W, D, CS, Cc, dim, Dj = ctx.geom
        dz, dy = (_f32c(dz) if jw3.shape[0] == 1 else dz.to(ja2.dtype).contiguous()), _f32c(dy)
        st = L.stream()
        dev = sx.device
        djw3, djb3 = j3.wgrad(ja2, dz, jw3)
        dja2 = torch.empty_like(ja2)
        j3.dgrad(dz, pd[2], ja2, dja2)
        djw2, djb2 = j2.wgrad(ja1, dja2, jw2)
        dja1 = torch.empty_like(ja1)
        j2.dgrad(dja2, pd[1], ja1, dja1)
        djwe, djb1 = j1.wgrad(sx, dja1, jwe)
        dhw3, dhb3 = h3.wgrad(ha2, dy, hw3)
        dha2 = torch.empty_like(ha2)
        h3.dgrad(dy, pd[5], ha2, dha2)
        dhw2, dhb2 = h2.wgrad(ha1, dha2, hw2)
        dha1 = torch.empty_like(ha1)
        h2.dgrad(dha2, pd[4], ha1, dha1)
        dhwe, dhb1 = h1.wgrad(sx, dha1, hwe)
        # unfold the weight gradients through [w_out | b_out]
        djw1, dhw1 = torch.empty_like(jw1), torch.empty_like(hw1)
        dw_out, db_out = _zeros_like_many(w_out, b_out)
        L.call("idee_lfq_fold_weights_bwd", ptr(djwe), ptr(jw1), ptr(w_out), ptr(b_out), ptr(djw1), ptr(dw_out), ptr(db_out),
               Dj, V, Cc, D, CS, 0, st)
        L.call("idee_lfq_fold_weights_bwd", ptr(dhwe), ptr(hw1), ptr(w_out), ptr(b_out), ptr(dhw1), ptr(dw_out), ptr(db_out),
               dim, V, Cc, D, CS, 1, st)
        dsigns = None
        if ctx.needs_input_grad[0]:
            dsx = torch.empty_like(sx)
            j1.dgrad(dja1, pd[0], None, dsx)
            h1.dgrad(dha1, pd[3], None, dsx, accumulate=True)
            dsigns = torch.empty((N, V, T, H, W, D), device=dev, dtype=torch.float32)
            L.call("idee_lfq_sign_field_bwd", ptr(dsx), ptr(dsigns), N, V, T, H * W, D, CS, L.act_dtype(dsx), st)
        if ctx.inv != 1.0:      # conv2 ran with weights * 1/(1-p)
            djw2, dhw2 = djw2 * ctx.inv, dhw2 * ctx.inv
        return (dsigns, dw_out, db_out, djw1, djb1.reshape(-1), djw2, djb2.reshape(-1), djw3, djb3.reshape(-1), dhw1,
                dhb1.reshape(V, dim), dhw2, dhb2, dhw3, dhb3, None, None, None)


# ----------------------------------------------------------------------------------------
# LFQ
# ----------------------------------------------------------------------------------------
class LFQQuantize(torch.autograd.Function):
    """z (..., E) tokens -> z_q (..., E), indices int64 (...), aux loss (), signs (..., D) | None.  Reference:
    LFQ.forward, models/codebook/LFQ.py:183-307.  `signs` (want_signs) are the +-1 code bits x_ste of LFQ.py:232-240
    BEFORE project_out, with the same straight-through gradient: consumers may read them instead of z_q."""

    @staticmethod
    def forward(ctx, z, w_in, b_in, w_out, b_out, codebook_size, training, w_entropy, gamma, w_commit, want_signs=False,
                num_codebooks=1):
        """num_codebooks > 1 (LFQ.py:92-101, :215): `codebook_size` codes PER codebook, w_in (num_codebooks * log2 K, E); `idx` then
        packs the per-codebook indices, most significant codebook first (LFQ.split_indices unpacks them)."""
        z = _act(z)       # bf16 tokens stay bf16 (z_q and dz then are bf16 too); projections, bits and statistics are f32
        w_in, b_in, w_out, b_out = (_f32c(t) for t in (w_in, b_in, w_out, b_out))
        E = z.shape[-1]
        M = z.numel() // E
        dev = z.device
        zq = torch.empty_like(z)
        idx = torch.empty(z.shape[:-1], device=dev, dtype=torch.int64)
        stats = _zeros(4 + num_codebooks * codebook_size, dev)
        ws = torch.empty(L.load().idee_lfq_workspace_floats(M, codebook_size, num_codebooks), device=dev, dtype=torch.float32)
        D = max(1, codebook_size.bit_length() - 1) * num_codebooks
        signs = torch.empty(z.shape[:-1] + (D,), device=dev, dtype=torch.float32) if want_signs else None
        L.call("idee_lfq_fwd", ptr(z), ptr(w_in), ptr(b_in), ptr(w_out), ptr(b_out), ptr(zq), ptr(idx), ptr(stats), ptr(ws),
               ptr(signs), M, E, codebook_size, num_codebooks, int(training), w_entropy, gamma, w_commit, L.act_dtype(z), L.stream())
        ctx.save_for_backward(z, w_in, b_in, w_out, stats)
        ctx.cfg = (M, E, codebook_size, num_codebooks, w_entropy, gamma, w_commit)
        ctx.mark_non_differentiable(idx)
        # an output nobody differentiates (z_q in a train step whose classifier and anomaly loss read the code bits) must reach
        # backward as None, not as a materialised zero tensor: that is what lets idee_lfq_bwd skip the dense d z_q stream
        ctx.set_materialize_grads(False)
        return zq, idx, stats[0], signs

    @staticmethod
    def backward(ctx, dzq, _didx, dloss, dsigns=None):
        z, w_in, b_in, w_out, stats = ctx.saved_tensors
        M, E, K, ncb, w_entropy, gamma, w_commit = ctx.cfg
        dzq = _like(dzq, z) if dzq is not None else None   # None: no dense consumer of z_q (its users read the signs)
        gl = _f32c(dloss).reshape(1) if dloss is not None else torch.zeros(1, device=z.device)
        dz = torch.empty_like(z)
        dw_in, db_in, dw_out = _zeros_like_many(w_in, b_in, w_out)
        db_out = _zeros(E, z.device)
        dsigns = _f32c(dsigns) if dsigns is not None else None
        L.call("idee_lfq_bwd", ptr(z), ptr(dzq), ptr(w_in), ptr(b_in), ptr(w_out), ptr(stats), ptr(gl), ptr(dsigns), ptr(dz),
               ptr(dw_in), ptr(db_in), ptr(dw_out), ptr(db_out), M, E, K, ncb, w_entropy, gamma, w_commit, L.act_dtype(z), L.stream())
        return dz, dw_in, db_in, dw_out, db_out, None, None, None, None, None, None, None


class LFQQuantizeNoProj(torch.autograd.Function):
    """LFQ without projections (LFQ.py:98-101, dim == num_codebooks * log2 K): z (..., E) -> z_q = sign(z) (..., E) with the
    straight-through gradient, indices int64 (...) packing the E bits (channel 0 most significant: LFQ.split_indices unpacks the
    per-codebook indices), aux loss ()."""

    @staticmethod
    def forward(ctx, z, codebook_size, training, w_entropy, gamma, w_commit):
        z = _act(z)
        E = z.shape[-1]
        M = z.numel() // E
        dev = z.device
        zq = torch.empty_like(z)
        idx = torch.empty(z.shape[:-1], device=dev, dtype=torch.int64)
        ncb = E // max(1, codebook_size.bit_length() - 1)
        stats = _zeros(4 + ncb * codebook_size, dev)
        ws = torch.empty(L.load().idee_lfq_noproj_workspace_floats(M, E, codebook_size), device=dev, dtype=torch.float32)
        L.call("idee_lfq_noproj_fwd", ptr(z), ptr(zq), ptr(idx), ptr(stats), ptr(ws), M, E, codebook_size, int(training),
               w_entropy, gamma, w_commit, L.act_dtype(z), L.stream())
        ctx.save_for_backward(z, stats)
        ctx.cfg = (M, E, codebook_size, w_entropy, gamma, w_commit)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return zq, idx, stats[0]

    @staticmethod
    def backward(ctx, dzq, _didx, dloss):
        z, stats = ctx.saved_tensors
        M, E, K, w_entropy, gamma, w_commit = ctx.cfg
        dzq = _like(dzq, z) if dzq is not None else None
        gl = _f32c(dloss).reshape(1) if dloss is not None else torch.zeros(1, device=z.device)
        dz = torch.empty_like(z)
        L.call("idee_lfq_noproj_bwd", ptr(z), ptr(dzq), ptr(stats), ptr(gl), None, ptr(dz), M, E, K, w_entropy, gamma, w_commit,
               L.act_dtype(z), L.stream())
        return dz, None, None, None, None, None


# ----------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------
def bce_class_counts(target, valid=None):
    """torch.histc(target[valid], bins=2) as a device tensor (2,) — models/losses.py:80,115."""
    target = _f32c(target)
    valid = _f32c(valid) if valid is not None else None
    counts = torch.empty(2, device=target.device, dtype=torch.float32)
    L.call("idee_bce_class_counts", ptr(target), ptr(valid), target.numel(), ptr(counts), L.stream())
    return counts


class WeightedBCE(torch.autograd.Function):
    """pred (P, M...) logits sharing one target (M...) -> (P,) losses.  valid (M...)|None.
    Reference: BCE_loss_synthetic / BCE_loss, models/losses.py:64-124."""

    @staticmethod
    def forward(ctx, pred, target, valid, counts):
        pred, target = _f32c(pred), _f32c(target)
        valid = _f32c(valid) if valid is not None else None
        P = pred.shape[0]
        M = target.numel()
        assert pred.numel() == P * M
        out = torch.empty(P + 1, device=pred.device, dtype=torch.float32)
        ws = torch.empty(L.load().idee_bce_workspace_floats(P, M), device=pred.device, dtype=torch.float32)
        L.call("idee_bce_weighted_fwd", ptr(pred), ptr(target), ptr(valid), ptr(counts), ptr(out), ptr(ws), P, M, L.stream())
        ctx.save_for_backward(pred, target, valid, counts, out)
        return out[:P]

    @staticmethod
    def backward(ctx, gout):
        pred, target, valid, counts, out = ctx.saved_tensors
        P, M = pred.shape[0], target.numel()
        gout = _f32c(gout)
        dpred = torch.empty_like(pred)
        denom = out[P:]
        L.call("idee_bce_weighted_bwd", ptr(pred), ptr(target), ptr(valid), ptr(counts), ptr(gout), ptr(denom), ptr(dpred), P,
               M, L.stream())
        return dpred, None, None, None


class AnomalyL1(torch.autograd.Function):
    """Masked L1 of z_q (N,V,T,H,W,E) against the code-0 vector outside extremes -> (num, den)
    Reference: Anomaly_L1_loss(_synthetic), models/losses.py:15-61, :127-168.
    Returns (loss = num/den, den); den lets data-parallel ranks renormalise by the global
    denominator (parallel.global_anomaly_scale)."""

    @staticmethod
    def forward(ctx, zq, mask_extreme, mask_valid, vq0):
        zq, mask_extreme, vq0 = _f32c(zq), _f32c(mask_extreme), _f32c(vq0).reshape(-1)
        mask_valid = _f32c(mask_valid) if mask_valid is not None else None
        N, V, T, H, W, E = zq.shape
        out = torch.empty(2, device=zq.device, dtype=torch.float32)
        ws = torch.empty(L.load().idee_anomaly_workspace_floats(N * V * T * H * W), device=zq.device, dtype=torch.float32)
        L.call("idee_anomaly_l1_fwd", ptr(zq), ptr(mask_extreme), ptr(mask_valid), ptr(vq0), ptr(out), ptr(ws), N, V, T, H * W,
               E, L.stream())
        ctx.save_for_backward(zq, mask_extreme, mask_valid, vq0, out)
        den = out[1]
        ctx.mark_non_differentiable(den)
        return out[0], den

    @staticmethod
    def backward(ctx, g, _gden):
        zq, mask_extreme, mask_valid, vq0, out = ctx.saved_tensors
        N, V, T, H, W, E = zq.shape
        g = _f32c(g).reshape(1)
        dzq = torch.empty_like(zq)
        L.call("idee_anomaly_l1_bwd", ptr(zq), ptr(mask_extreme), ptr(mask_valid), ptr(vq0), ptr(g), ptr(out[1:]), ptr(dzq), N,
               V, T, H * W, E, L.stream())
        return dzq, None, None, None


class AnomalyL1Signs(torch.autograd.Function):
    """AnomalyL1 read from LFQ's code bits: signs (N,V,T,H,W,D) (LFQQuantize's 4th output), project_out weight (E,D) and bias
    (E) -> (loss, den).  z_q = project_out(signs) is recomputed per token (rank-d form); the backward returns the
    gradient of the signs (straight-through into LFQQuantize) and of project_out, so no dense d z_q is written or read."""

    @staticmethod
    def forward(ctx, signs, w_out, b_out, mask_extreme, mask_valid, codebook_size):
        signs, w_out, b_out, mask_extreme = _f32c(signs), _f32c(w_out), _f32c(b_out), _f32c(mask_extreme)
        mask_valid = _f32c(mask_valid) if mask_valid is not None else None
        N, V, T, H, W, D = signs.shape
        E = w_out.shape[0]
        dev = signs.device
        code0 = torch.zeros(1, dtype=torch.long, device=dev)
        vq0 = lfq_code_vectors(code0, w_out, b_out, codebook_size).reshape(-1)      # detached target (losses.py:29,141)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        ws = torch.empty(L.load().idee_anomaly_signs_workspace_floats(N * V * T * H * W, E, codebook_size), device=dev,
                         dtype=torch.float32)
        L.call("idee_anomaly_l1_signs_fwd", ptr(signs), ptr(mask_extreme), ptr(mask_valid), ptr(w_out), ptr(b_out), ptr(vq0),
               ptr(out), ptr(ws), N, V, T, H * W, E, codebook_size, L.stream())
        ctx.save_for_backward(signs, mask_extreme, mask_valid, w_out, b_out, vq0, out)
        ctx.codebook_size = codebook_size
        den = out[1]
        ctx.mark_non_differentiable(den)
        return out[0], den

    @staticmethod
    def backward(ctx, g, _gden):
        signs, mask_extreme, mask_valid, w_out, b_out, vq0, out = ctx.saved_tensors
        N, V, T, H, W, D = signs.shape
        E = w_out.shape[0]
        g = _f32c(g).reshape(1)
        dsigns = torch.empty_like(signs)
        dwb = torch.empty(E * (D + 1), device=signs.device, dtype=torch.float32)
        ws = torch.empty(L.load().idee_anomaly_signs_workspace_floats(N * V * T * H * W, E, ctx.codebook_size),
                         device=signs.device, dtype=torch.float32)
        L.call("idee_anomaly_l1_signs_bwd", ptr(signs), ptr(mask_extreme), ptr(mask_valid), ptr(w_out), ptr(b_out), ptr(vq0),
               ptr(g), ptr(out[1:]), ptr(dsigns), ptr(dwb), dwb.data_ptr() + 4 * E * D, ptr(ws), N, V, T, H * W, E,
               ctx.codebook_size, L.stream())
        return dsigns, dwb[:E * D].view(E, D), dwb[E * D:], None, None, None


def lfq_code_vectors(indices, w_out, b_out, codebook_size):
    """LFQ.indices_to_codes(indices) with project_out for a 1-D index tensor, no autograd (LFQ.py:152-181) -> (n, E)."""
    indices, w_out, b_out = indices.contiguous(), _f32c(w_out.detach()), _f32c(b_out.detach())
    n, E = indices.numel(), w_out.shape[0]
    out = torch.empty((n, E), device=w_out.device, dtype=torch.float32)
    L.call("idee_lfq_code_vectors", ptr(indices), ptr(w_out), ptr(b_out), ptr(out), n, E, int(codebook_size), L.stream())
    return out


class CombineLosses(torch.autograd.Function):
    """total = mean_t bce[0] + lam * l_anom + sum_p mean_t bce[p] + loss_z_q in one kernel per direction.
    bce (P * T3,) rows of idee_bce_weighted_fwd (joint prediction first, T3 temporal slices each) -> (5,) =
    [total, extreme, anomaly, var, z_q]; only out[0] carries a gradient."""

    @staticmethod
    def forward(ctx, bce, l_anom, loss_zq, lam, T3):
        bce, l_anom, loss_zq = _f32c(bce), _f32c(l_anom).reshape(1), _f32c(loss_zq).reshape(1)
        P = bce.numel() // T3
        out = torch.empty(5, device=bce.device, dtype=torch.float32)
        L.call("idee_loss_combine_fwd", ptr(bce), P, T3, ptr(l_anom), float(lam), ptr(loss_zq), ptr(out), L.stream())
        ctx.cfg = (P, T3, float(lam), l_anom.shape, loss_zq.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        P, T3, lam, sa, sz = ctx.cfg
        g0 = _f32c(g)[:1].contiguous()
        dbce = torch.empty(P * T3, device=g.device, dtype=torch.float32)
        da, dz = torch.empty(1, device=g.device, dtype=torch.float32), torch.empty(1, device=g.device, dtype=torch.float32)
        L.call("idee_loss_combine_bwd", ptr(g0), P, T3, lam, ptr(dbce), ptr(da), ptr(dz), L.stream())
        return dbce, da.reshape(()), dz.reshape(()), None, None


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """In-place torch.optim.Adam update of the flat buffers (train_synthetic.py:127-129)."""
    L.call("idee_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, int(step),
           grad_scale, L.stream())


class ConvLNUnit(torch.autograd.Function):
    """One unit of the CNN_3D encoder's conv_block: y = x + ReLU(LayerNorm_affine(conv3x3x3_replicate(x))),
    per variable, conv without bias.  x (N,V,T,H,W,C); w (V,C,C,3,3,3); gamma, beta (V,C).
    Reference: conv_block.forward, models/encoder/CNN_3D.py:131-142."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, item_scale=None):
        """item_scale (N*V,) | None: DropPath factors mask / keep_prob of the ReLU(LN(conv)) branch per batch item."""
        x, w, gamma, beta = (_f32c(t) for t in (x, w, gamma, beta))
        item_scale = _f32c(item_scale) if item_scale is not None else None
        N, V, T, H, W, Cc = x.shape
        s = _cl_strides(T, H, W, Cc)
        conv = ConvLayer(L.CONV_K333_REPLICATE, N * V, V, T, H, W, Cc, Cc, s, s, False)
        pf, pd = conv.pack(w)
        c = torch.empty_like(x)
        conv.fwd(x, pf, None, c)
        y = torch.empty_like(x)
        L.call("idee_ln_relu_res_fwd", ptr(c), ptr(gamma), ptr(beta), ptr(x), ptr(y), N * V, V, T * H * W, Cc, ptr(item_scale),
               L.stream())
        ctx.save_for_backward(x, c, w, gamma, beta, pd)
        ctx.conv, ctx.item_scale = conv, item_scale
        return y

    @staticmethod
    def backward(ctx, dy):
        x, c, w, gamma, beta, pd = ctx.saved_tensors
        conv = ctx.conv
        N, V, T, H, W, Cc = x.shape
        dy = _f32c(dy)
        dc = torch.empty_like(c)
        dgamma, dbeta = _zeros_like_many(gamma, beta)
        L.call("idee_ln_relu_res_bwd", ptr(c), ptr(gamma), ptr(beta), ptr(dy), ptr(dc), ptr(dgamma), ptr(dbeta), N * V, V,
               T * H * W, Cc, ptr(ctx.item_scale), L.stream())
        dw, _ = conv.wgrad(x, dc, w, want_bias=False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)            # shortcut gradient (dy) + the conv's data gradient, without a copy of dy
            conv.dgrad_add(dc, pd, dy, dx)
        return dx, dw, dgamma, dbeta, None


class VQNearest(torch.autograd.Function):
    """Euclidean nearest-neighbour quantiser with straight-through gradient and commitment loss.
    x (..., E), embed (K, E) -> (quantised (..., E), indices int64 (...), loss (1,)).
    Reference: models/codebook/VQ.py:44-48, :477-555, :986, :1056 (defaults: learnable codebook, no EMA)."""

    @staticmethod
    def forward(ctx, x, embed, training, w_commit):
        x, embed = _f32c(x), _f32c(embed)
        E = x.shape[-1]
        M = x.numel() // E
        K = embed.shape[0]
        dev = x.device
        q = torch.empty_like(x)
        idx = torch.empty(x.shape[:-1], device=dev, dtype=torch.int64)
        out = torch.zeros(2, device=dev, dtype=torch.float32)
        ws = torch.empty(max(L.load().idee_vq_workspace_floats(M), 2), device=dev, dtype=torch.float32)
        L.call("idee_vq_nn_fwd", ptr(x), ptr(embed), ptr(q), ptr(idx), ptr(out), ptr(ws), M, E, K, int(training), w_commit,
               L.stream())
        # out[0] = sum / (number of blocks) after the ratio finaliser; undo the block count, apply the mean and the weight
        loss = (out[0] * out[1] * (w_commit / (M * E))).reshape(1) if training else torch.zeros(1, device=dev)
        ctx.save_for_backward(x, embed, idx)
        ctx.cfg = (M, E, K, w_commit)
        ctx.mark_non_differentiable(idx)
        return q, idx, loss

    @staticmethod
    def backward(ctx, dq, _didx, dloss):
        x, embed, idx = ctx.saved_tensors
        M, E, K, w_commit = ctx.cfg
        dq = _f32c(dq) if dq is not None else torch.zeros_like(x)
        gl = _f32c(dloss).reshape(1) if dloss is not None else torch.zeros(1, device=x.device)
        dx = torch.empty_like(x)
        dembed = torch.zeros_like(embed)
        L.call("idee_vq_nn_bwd", ptr(x), ptr(embed), ptr(idx), ptr(dq), ptr(gl), ptr(dx), ptr(dembed), M, E, K, w_commit,
               L.stream())
        # a codebook that is not learnable (EMA update) enters detached: commit_quantize = quantize.detach(), VQ.py:983-985
        return dx, (dembed if ctx.needs_input_grad[1] else None), None, None
