"""The per-iteration call sequences of the reference's training loops as step objects:

  SyntheticTrainStep — train_synthetic.py:175-205 (zero_grad -> forward -> BCE(joint) + lambda * anomaly-L1
      + sum_v BCE(head_v) + loss_z_q -> backward -> Adam)
  ERA5TrainStep      — train_ERA5_Land.py:168-198 (the same with BCE_loss / Anomaly_L1_loss: `mask_valid` built from the
      cold-surface mask, `pred[:, 0]`, the per-time-step cold-surface mask added to the anomaly loss' extreme mask)

The 1 + V BCE terms share one target, so they run as ONE weighted-BCE kernel over the stacked logits; nothing else
differs from calling the drop-in loss modules one by one (tests assert both give the same numbers).

Data parallel (world > 1, idee_amd/parallel.py): every loss is the GLOBAL-batch loss of the reference (which computes
its losses on DataParallel-gathered outputs): class counts are all-reduced before the BCE weights are formed, and a
rank's ratio-type losses num_r / den_r are rescaled by den_r * world / sum_r den_r so that the averaged gradient is the
gradient of sum_r num_r / sum_r den_r.  Class counts and both denominators depend on the step's masks only: they travel
in ONE 4-float all-reduce issued before the forward pass (`loss_stats`, parallel.LossStats), so a step has exactly two
collectives — that one and the gradient bucket."""
from __future__ import annotations

import torch

from . import ops, optim, parallel
from .models.build import VQ_model
from .models._stacked import reference_param_slices


class _TrainStep:
    def __init__(self, config, device="cuda", model=None):
        self.config = config
        self.model = (model if model is not None else VQ_model(config)).to(device)
        self.model.train()
        # state_dict() / load_state_dict() of this optimizer index the REFERENCE model's ~240 per-variable parameters
        self.opt = optim.Adam(self.model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                              betas=(config.beta1, config.beta2), reference_slices=reference_param_slices(self.model))
        parallel.broadcast_parameters(self.opt.flat_p)
        self.code0 = torch.zeros(1, dtype=torch.long, device=device)
        self.arena = ops.ZeroArena()

    @staticmethod
    def _stack_logits(pred, pred_y):
        """joint (N,1,H,W) | (N,1,T3,H,W) and V heads of the same shape -> ((1+V)*T3, N*H*W) rows sharing one target.
        T3 > 1 only at T = 16, where the reference's own loss is ill-defined: each temporal
        slice is then scored against the same target and the slices are averaged."""
        if pred.dim() == 4:
            y = getattr(pred_y, "stacked", None)       # classifier.HeadLogits: all V heads as one (N,V,H,W) tensor
            if y is not None and y.dim() == 4 and pred.shape[1] == 1:
                return torch.cat([pred.reshape(1, -1), y.permute(1, 0, 2, 3).reshape(y.shape[1], -1)], 0), 1
            return torch.cat([pred.reshape(1, -1)] + [p.reshape(1, -1) for p in pred_y], 0), 1
        T3 = pred.shape[2]
        rows = [p[:, 0].permute(1, 0, 2, 3).reshape(T3, -1) for p in [pred] + list(pred_y)]
        return torch.cat(rows, 0), T3

    def _loss_stats(self, target, valid, mask_loss, mask_valid_t):
        """[n_class0, n_class1, n_valid, anomaly weight sum] of this rank's masks -> parallel.LossStats (one async all-reduce
        when world > 1).  The anomaly weight sum is over (n[, t], h, w) of 1 - min(mask_loss + mask_valid_t, 1)
        (models/losses.py:49-52, :158-163); the loss' own denominator is that times V * E [* T], the same factor on every rank."""
        counts = ops.bce_class_counts(target.float().reshape(-1), valid.float().reshape(-1) if valid is not None else None)
        if parallel.world_size() == 1:
            return parallel.LossStats(counts)          # (only counts() is read with one rank)
        nvalid = valid.float().sum() if valid is not None else counts.sum()
        if mask_valid_t is not None:
            wsum = (1.0 - (mask_loss.float().unsqueeze(1) + mask_valid_t.float()).clamp_max_(1.0)).sum()
        else:
            wsum = (1.0 - mask_loss.float()).sum()
        return parallel.LossStats(torch.cat([counts, nvalid.reshape(1), wsum.reshape(1)]))

    def _bce_terms(self, outs, target, valid, stats):
        pred, pred_y = outs[0], outs[1]
        target = target.float().reshape(-1)
        valid = valid.float().reshape(-1) if valid is not None else None
        logits, T3 = self._stack_logits(pred, pred_y)
        bce = ops.WeightedBCE.apply(logits, target, valid, stats.counts())
        if valid is not None and parallel.world_size() > 1:       # denominators (valid pixels) differ between shards
            bce = bce * stats.valid_scale()
        return bce, T3        # ((1 + V) * T3,) rows, joint prediction first

    def _anomaly_term(self, z_q, mask_loss, mask_valid_t, stats):
        vq = self.model.vq      # indices_to_codes(LongTensor([0])).detach() of train_synthetic.py:193-194 as one tiny kernel
        signs = getattr(self.model, "last_signs", None)
        if signs is not None:    # rank-d form: the loss reads the code bits the classifier already reads, not the z_q tensor
            self.model.last_signs = None
            l_anom, _den = ops.AnomalyL1Signs.apply(signs, vq.project_out.weight, vq.project_out.bias, mask_loss.float(),
                                                    mask_valid_t.float() if mask_valid_t is not None else None, vq.codebook_size)
            if parallel.world_size() > 1:
                l_anom = l_anom * stats.anomaly_scale()
            return l_anom
        if getattr(self.model, "codebook_impl", "LFQ") == "VQ":       # nearest-neighbour variant: code 0 of the codebook itself
            vq0 = vq.indices_to_codes(self.code0).detach().reshape(-1).contiguous()
        else:
            vq0 = ops.lfq_code_vectors(self.code0, vq.project_out.weight, vq.project_out.bias, vq.codebook_size)
        zq_tokens = z_q.permute(0, 1, 3, 4, 5, 2)
        if zq_tokens.dtype != torch.float32:       # the dense loss kernel reads f32 (bf16 storage + IDEE_SIGN_FIELD=0 only)
            zq_tokens = zq_tokens.float()
        l_anom, _den = ops.AnomalyL1.apply(zq_tokens if zq_tokens.is_contiguous() else zq_tokens.contiguous(),
                                           mask_loss.float(), mask_valid_t.float() if mask_valid_t is not None else None, vq0)
        if parallel.world_size() > 1:      # global-batch normalisation
            l_anom = l_anom * stats.anomaly_scale()
        return l_anom

    def _finish(self, bce_rows, T3, l_anom, loss_z_q):
        out = ops.CombineLosses.apply(bce_rows, l_anom, loss_z_q.reshape(()), float(self.config.lambda_anomaly), T3)
        parts = out.detach()
        return out[0], dict(extreme=parts[1], anomaly=parts[2], var=parts[3], z_q=parts[4])

    def _step(self, x_d, *masks):
        self.opt.zero_grad(set_to_none=True)
        stats = self.loss_stats(*masks)      # mask-only statistics: their ONE all-reduce runs under the forward pass
        with self.arena:        # one fill for all accumulate-into buffers of forward + backward
            outs = self.model(x_d)
            total, parts = self.losses(outs, *masks, stats=stats)
            total.backward()
        parallel.sync_gradients(self.opt)
        self.opt.step()
        return total.detach(), parts, outs


class SyntheticTrainStep(_TrainStep):
    def loss_stats(self, mask_extreme, mask_extreme_loss):
        return self._loss_stats(mask_extreme, None, mask_extreme_loss, None)

    def losses(self, outs, mask_extreme, mask_extreme_loss, stats=None):
        """outs of VQ_model.forward; mask_extreme (N,H,W) -> total loss and its parts.  `stats`: loss_stats(...) of the same
        masks when the caller issued it ahead of the forward pass (train step); computed here otherwise."""
        stats = stats if stats is not None else self.loss_stats(mask_extreme, mask_extreme_loss)
        bce, T3 = self._bce_terms(outs, mask_extreme, None, stats)
        l_anom = self._anomaly_term(outs[3], mask_extreme_loss, None, stats)
        return self._finish(bce, T3, l_anom, outs[4])

    def step(self, x_d, mask_extreme, mask_extreme_loss):
        return self._step(x_d, mask_extreme, mask_extreme_loss)


class ERA5TrainStep(_TrainStep):
    """train_ERA5_Land.py:168-198.  Batch as the ERA5-Land dataset yields it (dataset/ERA5_Land_dataset.py:521-625):
    x_d (N,V,2,T,H,W); mask_drought (N,H,W); mask_drought_loss (N,H,W); mask_cold_surface (N,H,W);
    mask_cold_surface_loss (N,T,H,W).  mask_valid = max(1 - mask_cold_surface, 0) (:176-177)."""

    threshold = 0.35          # sigmoid threshold of the real-data drivers (train_ERA5_Land.py:205-208)

    def loss_stats(self, mask_drought, mask_drought_loss, mask_cold_surface, mask_cold_surface_loss):
        mask_valid = (1.0 - mask_cold_surface.float()).clamp_min_(0.0)
        return self._loss_stats(mask_drought, mask_valid, mask_drought_loss, mask_cold_surface_loss)

    def losses(self, outs, mask_drought, mask_drought_loss, mask_cold_surface, mask_cold_surface_loss, stats=None):
        mask_valid = (1.0 - mask_cold_surface.float()).clamp_min_(0.0)
        stats = stats if stats is not None else self.loss_stats(mask_drought, mask_drought_loss, mask_cold_surface, mask_cold_surface_loss)
        bce, T3 = self._bce_terms(outs, mask_drought, mask_valid, stats)
        l_anom = self._anomaly_term(outs[3], mask_drought_loss, mask_cold_surface_loss, stats)
        return self._finish(bce, T3, l_anom, outs[4])

    def step(self, x_d, mask_drought, mask_drought_loss, mask_cold_surface, mask_cold_surface_loss):
        return self._step(x_d, mask_drought, mask_drought_loss, mask_cold_surface, mask_cold_surface_loss)
