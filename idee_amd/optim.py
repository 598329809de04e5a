"""Adam over one flat parameter / gradient buffer (idee_adam_step), with torch.optim.Adam's math
(train_synthetic.py:127-129: L2 weight decay folded into the gradient, not AdamW).

The parameters of the model are re-pointed to views of one flat f32 buffer and their `.grad`s to
views of one flat gradient buffer, so (a) the optimizer step is ONE kernel instead of ~40 tiny ones
and (b) the data-parallel gradient exchange is ONE all-reduce of that buffer (parallel.py)."""
from __future__ import annotations

import torch

from . import ops


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
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

    def state_dict(self):
        """torch.optim.Adam's layout — {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{..., "params":
        [0..n-1]}]} with per-parameter tensors cut from the flat moment buffers — so the reference's
        `optimizer.load_state_dict(checkpoint["optimizer_state_dict"])` accepts a checkpoint written here and vice versa
        (parameter order = model.parameters() order of the module the optimizer was built from)."""
        state = {}
        if self.step_count > 0:
            for i, (a, b, shape) in enumerate(self._spans()):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[a:b].view(shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[a:b].view(shape).clone())
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                 params=list(range(len(self.params))))
        return dict(state=state, param_groups=[g])

    def load_state_dict(self, sd):
        steps = set()
        for i, (a, b, shape) in enumerate(self._spans()):
            st = sd["state"].get(i)
            if st is None:
                self.exp_avg[a:b].zero_()
                self.exp_avg_sq[a:b].zero_()
                continue
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"idee_amd.optim.Adam: per-parameter step counts differ ({sorted(steps)}); the flat step has one")
        self.step_count = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.param_groups[0].update({k: g[k] for k in ("lr", "betas", "eps", "weight_decay") if k in g})
