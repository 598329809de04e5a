"""Adam over one flat parameter / gradient buffer (idee_adam_step), with torch.optim.Adam's math
(train_synthetic.py:127-129: L2 weight decay folded into the gradient, not AdamW).

The parameters of the model are re-pointed to views of one flat f32 buffer and their `.grad`s to
views of one flat gradient buffer, so (a) the optimizer step is ONE kernel instead of ~40 tiny ones
and (b) the data-parallel gradient exchange is ONE all-reduce of that buffer (parallel.py)."""
from __future__ import annotations

import torch

from . import ops


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, reference_slices=None):
        """reference_slices: models._stacked.reference_param_slices(model) — when given, state_dict() / load_state_dict() speak
        the REFERENCE model's parameter indexing (one entry per per-variable parameter of the reference, in its
        model.parameters() order) instead of this model's V-stacked parameters, so `optimizer_state_dict` interchanges with
        the reference's torch.optim.Adam (utils_train.py:562-584)."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                off += k
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_count = 0
        self.grad_scale = 1.0      # parallel.py sets 1/world after a SUM all-reduce
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.reference_slices = None
        if reference_slices is not None:
            where = {id(p): (a, b, shape) for p, (a, b, shape) in zip(self.params, self._spans())}
            self.reference_slices = [(key,) + where[id(p)] + (idx, rshape) for key, p, idx, rshape in reference_slices if id(p) in where]
            covered = sum((b - a) // (shape[0] if idx is not None else 1) for _, a, b, shape, idx, _r in self.reference_slices)
            if covered != n:
                raise ValueError(f"idee_amd.optim.Adam: reference_slices cover {covered} of {n} optimised elements")

    def zero_grad(self, set_to_none: bool = True):
        """set_to_none=True (what train_synthetic.py:175 passes): drop the gradients, so autograd hands each parameter
        its gradient tensor as is (no `grad += new` kernel per parameter, no fill); `gather_grads` then copies them into
        the flat buffer with one multi-tensor copy.  set_to_none=False: zero the flat buffer and keep the parameters'
        .grad as views of it (autograd accumulates in place)."""
        if set_to_none:
            for p in self.params:
                p.grad = None
            self._gathered = False
            return
        self.flat_g.zero_()
        for p, g in zip(self.params, self._grad_views()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g
        self._gathered = True

    @torch.no_grad()
    def gather_grads(self):
        """Bring every parameter's gradient into the flat buffer (idempotent); afterwards p.grad is the flat view."""
        if getattr(self, "_gathered", True):
            return
        views = list(self._grad_views())
        src, dst = [], []
        for p, v in zip(self.params, views):
            if p.grad is None:
                # torch.optim.Adam skips parameters without a gradient (no weight decay, no moment update); the flat
                # kernel cannot skip a span, so an unused parameter is an error here rather than a silent drift
                raise RuntimeError("idee_amd.optim.Adam: a parameter received no gradient in this step (torch.optim.Adam "
                                   "would skip it; the flat-buffer step cannot) — freeze it with requires_grad_(False) "
                                   "before constructing the optimizer")
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad if p.grad.dtype == torch.float32 else p.grad.float())
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, views):
            p.grad = v
        self._gathered = True

    def _grad_views(self):
        off = 0
        for p in self.params:
            k = p.numel()
            yield self.flat_g[off:off + k].view(p.shape)
            off += k

    @torch.no_grad()
    def step(self):
        self.gather_grads()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, float(g["lr"]), float(g["betas"][0]),
                      float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.step_count, float(self.grad_scale))

    def _spans(self):
        off = 0
        for p in self.params:
            yield off, off + p.numel(), p.shape
            off += p.numel()

    def _moment_views(self):
        """[(exp_avg view, exp_avg_sq view)] per state-dict entry: reference-indexed when reference_slices was given."""
        if self.reference_slices is None:
            return [(self.exp_avg[a:b].view(shape), self.exp_avg_sq[a:b].view(shape)) for a, b, shape in self._spans()]
        out = []
        for _key, a, b, shape, idx, rshape in self.reference_slices:
            m, v = self.exp_avg[a:b].view(shape), self.exp_avg_sq[a:b].view(shape)
            out.append((m.view(rshape), v.view(rshape)) if idx is None else (m[idx].view(rshape), v[idx].view(rshape)))
        return out

    def state_dict(self):
        """torch.optim.Adam's layout — {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{..., "params":
        [0..n-1]}]}.  With reference_slices (what idee_amd.train builds): entry i is the REFERENCE model's i-th parameter
        (e.g. `encoder.layers_var.3.0.blocks.1.attn.qkv.weight` (48,16), cut from the V-stacked moment buffers), so the
        reference's `optimizer.load_state_dict(checkpoint["optimizer_state_dict"])` accepts a checkpoint written here and
        vice versa.  Without them: one entry per parameter of this optimizer, in the order given."""
        views = self._moment_views()
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(views):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=m.clone(), exp_avg_sq=v.clone())
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                 params=list(range(len(views))))
        return dict(state=state, param_groups=[g])

    def load_state_dict(self, sd):
        views = self._moment_views()
        n_ckpt = len(sd["param_groups"][0]["params"])
        if n_ckpt != len(views):
            raise ValueError(f"idee_amd.optim.Adam: the checkpoint's optimizer indexes {n_ckpt} parameters, this optimizer "
                             f"{len(views)} ({'reference' if self.reference_slices is not None else 'own'} indexing); it was written "
                             f"for another model or parameter layout")
        for i, (m, v) in enumerate(views):      # validate everything before touching the moments
            st = sd["state"].get(i)
            if st is not None and (tuple(st["exp_avg"].shape) != tuple(m.shape) or tuple(st["exp_avg_sq"].shape) != tuple(v.shape)):
                key = self.reference_slices[i][0] if self.reference_slices is not None else f"parameter {i}"
                raise ValueError(f"idee_amd.optim.Adam: moment shape {tuple(st['exp_avg'].shape)} of checkpoint entry {i} does not "
                                 f"match {key} {tuple(m.shape)}")
        steps = set()
        with torch.no_grad():
            for i, (m, v) in enumerate(views):
                st = sd["state"].get(i)
                if st is None:
                    m.zero_()
                    v.zero_()
                    continue
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"idee_amd.optim.Adam: per-parameter step counts differ ({sorted(steps)}); the flat step has one")
        self.step_count = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.param_groups[0].update({k: g[k] for k in ("lr", "betas", "eps", "weight_decay") if k in g})
