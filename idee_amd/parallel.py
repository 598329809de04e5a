"""Batch-sharded data parallelism over RCCL/xGMI: one process per GPU, launched by torchrun,
replacing the reference's nn.DataParallel wrapper (train_synthetic.py:133-137).

Per step exactly TWO collectives:
  1. the loss statistics, ONE vector of 4 floats (SUM): the BCE target's two class counts (-> global-batch class weights),
     the valid-pixel count (-> BCE normalisation on real data) and the anomaly loss' weight sum (-> its normalisation).  All
     four are functions of the step's MASKS only, so the all-reduce is issued asynchronously BEFORE the forward pass and
     waited for when the losses are formed: it is never on the critical path between forward and backward (`LossStats`);
  2. the flat gradient buffer                 (2.14 MB,  SUM; the optimizer divides by world)
The 535,892 parameters are one flat bucket: on a fully connected 8-GPU xGMI mesh this is a
latency-bound exchange, so it is issued once, un-bucketed, after backward.  Backend "nccl" is RCCL
on ROCm; the same code runs on gloo/CPU tensors for the tests.  `loss_z_q` (LFQ aux loss) is the
mean over ranks of the per-shard value (each DataParallel replica would compute its own too)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world, local_rank); a no-op returning (0, 1, 0) without WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal knobs (one-GPU boxes): IDEE_DIST_BACKEND=gloo runs the same collectives through gloo on device tensors,
    # IDEE_FORCE_DEVICE=0 puts every rank on one device (RCCL itself refuses two ranks per device)
    backend = os.environ.get("IDEE_DIST_BACKEND", backend)
    if "IDEE_FORCE_DEVICE" in os.environ:
        local = int(os.environ["IDEE_FORCE_DEVICE"])
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_(t):
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_parameters(flat_p, src=0):
    """Identical replicas: rank `src`'s flat parameter buffer to everyone (once, at start)."""
    if world_size() > 1:
        dist.broadcast(flat_p, src=src)
    return flat_p


def sync_gradients(optimizer):
    """SUM all-reduce of the flat gradient bucket; the Adam kernel applies the 1/world."""
    w = world_size()
    if hasattr(optimizer, "gather_grads"):      # flat-buffer optimizers that let autograd own the per-parameter grads
        optimizer.gather_grads()
    if w > 1:
        dist.all_reduce(optimizer.flat_g, op=dist.ReduceOp.SUM)
    optimizer.grad_scale = 1.0 / w


def shard_batch(n_global, rank=None, world=None):
    """[start, stop) of this rank's clips of a global batch (equal shards required by the
    mean-of-means identities used for the BCE terms)."""
    world = world_size() if world is None else world
    rank = (dist.get_rank() if world > 1 else 0) if rank is None else rank
    if n_global % world:
        raise ValueError(f"global batch {n_global} is not divisible by world size {world}")
    per = n_global // world
    return rank * per, (rank + 1) * per


class LossStats:
    """The step's loss statistics, all-reduced in ONE collective (see the module docstring).

    local / total: (4,) f32 = [n_class0, n_class1, n_valid_pixels, anomaly_weight_sum]; with one rank `total is local`.
    counts(): global class counts for the BCE weights (models/losses.py:80,115 on the gathered batch).
    valid_scale() / anomaly_scale(): den_r * world / sum_r den_r — the factor that turns a rank's ratio-type loss num_r / den_r into
    its share of the global-batch loss sum_r num_r / sum_r den_r under gradient AVERAGING (1.0 with one rank)."""

    def __init__(self, local):
        self.local = local
        self.total = local
        self.work = None
        w = world_size()
        if w > 1:
            self.total = local.clone()
            self.work = dist.all_reduce(self.total, op=dist.ReduceOp.SUM, async_op=True)

    def _wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def counts(self):
        self._wait()
        return self.total[:2]

    def _scale(self, i):
        w = world_size()
        if w == 1:
            return None
        self._wait()
        return self.local[i] * w / self.total[i]

    def valid_scale(self):
        return self._scale(2)

    def anomaly_scale(self):
        return self._scale(3)


def global_anomaly_scale(den_local):
    """Factor that turns a rank's local anomaly loss num_r/den_r into its share of the
    global-batch loss under gradient averaging: den_r * world / sum_r den_r."""
    w = world_size()
    if w == 1:
        return torch.ones_like(den_local)
    den = den_local.clone()
    dist.all_reduce(den, op=dist.ReduceOp.SUM)
    return den_local * w / den
