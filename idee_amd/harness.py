"""The caller side of the hot path, restated for MI355X: the per-epoch learning-rate
schedule, the confusion-count evaluators, the majority-vote anomaly collector and the checkpoint files of the
reference's training drivers — with the per-iteration `.cpu()` syncs of train_synthetic.py:207-215 replaced by
device-side accumulation and ONE device-to-host read per epoch.

Reference: train_synthetic.py:162-331, train_ERA5_Land.py:160-313, utils/utils_train.py:143-172, :175-348, :529-584."""
from __future__ import annotations

import math
import os

import torch

from . import _lib as L
from ._lib import ptr


# ----------------------------------------------------------------------------------------
# learning-rate schedule
# ----------------------------------------------------------------------------------------
class CosineLRScheduler:
    """The schedule `utils.get_learning_scheduler` builds (utils_train.py:156-167): timm's CosineLRScheduler
    (third-party, timm==1.0.11 in the reference's requirements.txt:65, absent from this image — PARITY UNPINNED,
    restated from its published algorithm) with t_initial=n_epochs, cycle_mul=1, cycle_limit=1, lr_min, linear warm-up
    from warmup_lr_init over warmup_t steps, warmup_prefix=False.  The drivers call `step_update(epoch)` once per epoch
    (train_synthetic.py:331).  As in timm, construction already sets the optimizer to `warmup_lr_init`."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0, cycle_limit=1):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init, self.cycle_limit = t_initial, lr_min, warmup_t, warmup_lr_init, cycle_limit
        self.base_values = [g["lr"] for g in optimizer.param_groups]
        if warmup_t:
            self._set([warmup_lr_init for _ in self.base_values])

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (v - self.warmup_lr_init) / self.warmup_t for v in self.base_values]
        i = t // self.t_initial
        t_curr = t - self.t_initial * i
        if i < self.cycle_limit:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t_curr / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def step_update(self, num_updates):
        self._set(self.get_lr(num_updates))

    @classmethod
    def from_config(cls, optimizer, config):
        if config.lr_scheduler != "cosine":
            return get_learning_scheduler(optimizer, config)
        return cls(optimizer, t_initial=config.n_epochs, lr_min=config.lr_min, warmup_t=config.lr_warmup_epochs,
                   warmup_lr_init=config.lr_warmup)


class StepLRScheduler:
    """The `lr_scheduler == 'step'` branch of `utils.get_learning_scheduler` (utils_train.py:146-154): timm's
    StepLRScheduler (timm==1.0.11, absent from this image — PARITY UNPINNED, restated from its published algorithm) with
    decay_t, decay_rate, linear warm-up from warmup_lr_init over warmup_t, t_in_epochs=True:
        lr(t) = warmup_lr_init + t * (base - warmup_lr_init) / warmup_t      for t < warmup_t
              = base * decay_rate ** (t // decay_t)                          otherwise.
    timm applies a schedule on `step(epoch)` when t_in_epochs is True and on `step_update(n)` when it is False, and
    ignores the other call.  The reference builds this scheduler with t_in_epochs=True but its drivers only ever call
    `step_update(epoch)` (train_synthetic.py:331, train_ERA5_Land.py:313) — so under the reference the learning rate
    stays at the value the constructor set (warmup_lr_init).  Both calls are provided with timm's semantics;
    `step_update` is therefore a no-op here too, and a caller who wants the decay calls `step(epoch)`."""

    def __init__(self, optimizer, decay_t, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0, t_in_epochs=True):
        self.optimizer = optimizer
        self.decay_t, self.decay_rate, self.warmup_t, self.warmup_lr_init, self.t_in_epochs = decay_t, decay_rate, warmup_t, warmup_lr_init, t_in_epochs
        self.base_values = [g["lr"] for g in optimizer.param_groups]
        if warmup_t:
            self._set([warmup_lr_init for _ in self.base_values])

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (v - self.warmup_lr_init) / self.warmup_t for v in self.base_values]
        return [v * (self.decay_rate ** (t // self.decay_t)) for v in self.base_values]

    def step(self, epoch):
        if self.t_in_epochs:
            self._set(self.get_lr(epoch))

    def step_update(self, num_updates):
        if not self.t_in_epochs:
            self._set(self.get_lr(num_updates))


def get_learning_scheduler(optimizer, config):
    """utils.get_learning_scheduler, utils_train.py:143-172 (same branches, same error text)."""
    if config.lr_scheduler == "step":
        return StepLRScheduler(optimizer, decay_t=config.lr_decay_step, decay_rate=config.lr_decay_rate,
                               warmup_lr_init=config.lr_warmup, warmup_t=config.lr_warmup_epochs, t_in_epochs=True)
    if config.lr_scheduler == "cosine":
        return CosineLRScheduler(optimizer, t_initial=config.n_epochs, lr_min=config.lr_min, warmup_t=config.lr_warmup_epochs,
                                 warmup_lr_init=config.lr_warmup)
    raise ValueError("Unexpected optimizer {}, supported scheduler is step or cosine".format(config.optimizer))


# ----------------------------------------------------------------------------------------
# evaluators
# ----------------------------------------------------------------------------------------
class ConfusionEvaluator:
    """Counts of evaluator / evaluator_synthetic (utils_train.py:175-348) accumulated ON DEVICE over an epoch.
    `update(logits, gt[, mask_valid])` thresholds sigmoid(logits) (0.5 synthetic — train_synthetic.py:209-212; 0.35 real
    data — train_ERA5_Land.py:205-208) inside the kernel; `results()` is the one D2H read."""

    def __init__(self, threshold=0.5, device="cuda"):
        self.threshold = float(threshold)
        self.counts = torch.zeros(9, dtype=torch.int64, device=device)

    def reset(self):
        self.counts.zero_()

    def update(self, logits, gt, mask_valid=None):
        logits, gt = logits.detach().float().contiguous(), gt.float().contiguous()
        mv = mask_valid.float().contiguous() if mask_valid is not None else None
        L.call("idee_confusion_counts", ptr(logits), ptr(gt), ptr(mv), logits.numel(), self.threshold, ptr(self.counts), L.stream())

    def results(self):
        """-> dict per label: precision, accuracy (recall), F1, IoU exactly as get_results computes them, plus raw counts."""
        c = self.counts.cpu().double().tolist()
        out = dict(seen_all=c[8], correct_all=c[0] + c[4])
        for l in (0, 1):
            correct, seen, union, predicted = c[4 * l:4 * l + 4]
            precision = correct / predicted if predicted else float("nan")
            accuracy = correct / (seen + 1e-6)
            f1 = 2 * precision * accuracy / (accuracy + precision) if (accuracy + precision) else float("nan")
            iou = correct / union if union else float("nan")
            out[l] = dict(correct=correct, seen=seen, iou_de=union, predicted=predicted, precision=precision, accuracy=accuracy, F1=f1, iou=iou)
        out["accuracy_all"] = out["correct_all"] / c[8] if c[8] else float("nan")
        return out


class AnomalyCollector:
    """Majority vote over the overlapping clips that cover each time step (anomaly_collector, utils_train.py:529-554):
    votes are accumulated on device, time-flipped as the reference does (`np.flip(anomaly[n], axis=1)`)."""

    def __init__(self, n_vars, timesteps, height, width, delta_t, device="cuda"):
        self.timesteps = [int(t) for t in timesteps]
        self.pos = {t: i for i, t in enumerate(self.timesteps)}
        self.V, self.TT, self.H, self.W, self.dt = n_vars, len(self.timesteps), height, width, delta_t
        self.acc = torch.zeros((n_vars, self.TT, height, width), device=device)
        self.cnt = torch.zeros_like(self.acc)

    def reset(self):
        self.acc.zero_()
        self.cnt.zero_()

    def update(self, anomaly, timestep):
        """anomaly int64 (n, V, dt, H, W) from VQ_model.forward; timestep: the n clips' last time steps."""
        idx = torch.tensor([self.pos[int(t)] for t in timestep], dtype=torch.int32, device=self.acc.device)
        a = anomaly.contiguous()
        L.call("idee_anomaly_vote_accumulate", ptr(a), ptr(idx), ptr(self.acc), ptr(self.cnt), a.shape[0], self.V, self.dt,
               self.H * self.W, self.TT, L.stream())

    def majority_vote(self):
        out = torch.empty_like(self.acc)
        L.call("idee_anomaly_vote_finalize", ptr(self.acc), ptr(self.cnt), ptr(out), out.numel(), L.stream())
        return out


# ----------------------------------------------------------------------------------------
# checkpoints
# ----------------------------------------------------------------------------------------
def save_model(model, optimizer, epoch, mean_loss_train, mean_loss_val, config, metric):
    """The reference's checkpoint file (utils_train.py:562-584): same path, same dict keys; `model_state_dict` holds the
    reference's parameter keys, so either code base can load it (models/build.py:120-125); `optimizer_state_dict` is in
    torch.optim.Adam's {state, param_groups} layout (idee_amd.optim.Adam.state_dict)."""
    ckpt_dir = os.path.join(config.dir_log, config.name, "model_checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    name = {"loss": "best_loss_model.pth", "F1": "best_F1_model.pth", "train": "best_train_model.pth"}[metric]
    state = {"epoch": epoch, "mean_loss_train": mean_loss_train, "mean_loss_validation": mean_loss_val,
             "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
             "optimizer_state_dict": optimizer.state_dict()}
    path = os.path.join(ckpt_dir, name)
    torch.save(state, path)
    return path


def load_model(path, model, optimizer=None, map_location="cpu"):
    """Resume from a checkpoint written by save_model or by the reference (utils_train.py:562-584): parameters (with or
    without DataParallel's `module.` prefix, models/build.py:120-125) and, when given, the optimizer's moments, step count and
    hyper-parameters.  Returns (epoch, mean_loss_train, mean_loss_validation)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in ck["model_state_dict"].items()}
    model.load_state_dict(sd, strict=True)
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    return ck.get("epoch"), ck.get("mean_loss_train"), ck.get("mean_loss_validation")


# ----------------------------------------------------------------------------------------
# epoch loop (train_synthetic.py:162-233) with deferred metric reads
# ----------------------------------------------------------------------------------------
def train_epoch(step, batches, evaluator=None, collector=None):
    """step: idee_amd.train.SyntheticTrainStep with batches (x_d, mask_extreme, mask_extreme_loss, timestep), or
    idee_amd.train.ERA5TrainStep with batches (x_d, mask_drought, mask_drought_loss, mask_cold_surface,
    mask_cold_surface_loss, timestep | None) — train_ERA5_Land.py:168-211 (evaluator threshold 0.35, `mask_valid`).
    Returns (mean loss, evaluator results | None).  The loss values stay on the device until the epoch ends."""
    if evaluator is not None:
        evaluator.reset()
    losses = []
    for batch in batches:
        x_d, masks, timestep = batch[0], batch[1:-1], batch[-1]
        total, _parts, outs = step.step(x_d, *masks)
        losses.append(total)
        if evaluator is not None:
            if len(masks) == 4:          # real data: valid = max(1 - cold surface, 0), joint logits' first map
                pred = outs[0][:, 0]     # (N,H,W), or (N,T3,H,W) at T = 16 where the classifier keeps T3 = 2 temporal maps
                if pred.dim() == 4:      # the loss scores every slice against the same target and averages (train.py); so does
                    pred = pred.mean(1)  # the metric: the mean logit over the T3 slices
                evaluator.update(pred, masks[0], (1.0 - masks[2].float()).clamp_min_(0.0))
            else:
                evaluator.update(outs[0], masks[0].unsqueeze(1))
        if collector is not None and timestep is not None:
            collector.update(outs[2], timestep)
    mean_loss = float(torch.stack(losses).mean()) if losses else float("nan")     # the single sync of the epoch
    return mean_loss, (evaluator.results() if evaluator is not None else None)
