"""The per-iteration call sequence of the reference's training loop, train_synthetic.py:175-205
(zero_grad -> forward -> BCE(joint) + lambda * anomaly-L1 + sum_v BCE(head_v) + loss_z_q ->
backward -> Adam), as one object.  The 1 + V BCE terms share one target, so they run as ONE
weighted-BCE kernel over the stacked logits; nothing else differs from calling the drop-in
modules one by one (tests assert both give the same numbers)."""
from __future__ import annotations

import torch

from . import ops, optim, parallel
from .models.build import VQ_model


class SyntheticTrainStep:
    def __init__(self, config, device="cuda", model=None):
        self.config = config
        self.model = (model if model is not None else VQ_model(config)).to(device)
        self.model.train()
        self.opt = optim.Adam(self.model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                              betas=(config.beta1, config.beta2))
        parallel.broadcast_parameters(self.opt.flat_p)
        self.code0 = torch.zeros(1, dtype=torch.long, device=device)

    def losses(self, outs, mask_extreme, mask_extreme_loss):
        """outs of VQ_model.forward; mask_extreme (N,H,W) -> total loss and its parts."""
        pred, pred_y, _anomaly, z_q, loss_z_q = outs
        target = mask_extreme.float().reshape(-1)
        counts = parallel.all_reduce_sum_(ops.bce_class_counts(target))
        logits = torch.cat([pred.reshape(1, -1)] + [p.reshape(1, -1) for p in pred_y], 0)
        bce = ops.WeightedBCE.apply(logits, target, None, counts)
        l_ext, l_var = bce[0], bce[1:].sum()
        vq0 = self.model.vq.indices_to_codes(self.code0).detach()
        zq_tokens = z_q.permute(0, 1, 3, 4, 5, 2)
        l_anom, den = ops.AnomalyL1.apply(zq_tokens if zq_tokens.is_contiguous() else zq_tokens.contiguous(),
                                          mask_extreme_loss.float(), None, vq0)
        if parallel.world_size() > 1:      # global-batch normalisation
            l_anom = l_anom * parallel.global_anomaly_scale(den)
        total = l_ext + l_anom * self.config.lambda_anomaly + l_var + loss_z_q.reshape(())
        return total, dict(extreme=l_ext, anomaly=l_anom, var=l_var, z_q=loss_z_q.reshape(()))

    def step(self, x_d, mask_extreme, mask_extreme_loss):
        self.opt.zero_grad(set_to_none=True)
        outs = self.model(x_d)
        total, parts = self.losses(outs, mask_extreme, mask_extreme_loss)
        total.backward()
        parallel.sync_gradients(self.opt)
        self.opt.step()
        return total.detach(), parts, outs
