"""world_size-2 gloo (CPU) tests of the data-parallel path: the two collectives per step of
idee_amd/parallel.py and the loss algebra that makes a 2-rank x N=2 step equal a 1-rank x N=4 step
(checked with the CPU oracle standing in for the kernels: the collectives are device-agnostic)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from idee_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    return parallel


class _FakeOpt:
    def __init__(self, g):
        self.flat_g, self.grad_scale = g, 1.0


def _worker_collectives(rank, world, port, q):
    P = _init(rank, world, port)
    try:
        flat_p = torch.full((1000,), float(rank + 1))
        P.broadcast_parameters(flat_p)
        assert torch.equal(flat_p, torch.ones(1000))
        opt = _FakeOpt(torch.arange(10.0) * (rank + 1))
        P.sync_gradients(opt)
        assert torch.equal(opt.flat_g, torch.arange(10.0) * 3) and opt.grad_scale == 0.5
        counts = P.all_reduce_sum_(torch.tensor([3.0 + rank, 1.0]))
        assert torch.equal(counts, torch.tensor([7.0, 2.0]))
        assert P.shard_batch(8) == (4 * rank, 4 * rank + 4)
        with pytest.raises(ValueError):
            P.shard_batch(7)
        den = torch.tensor(10.0 if rank == 0 else 30.0)
        assert torch.allclose(P.global_anomaly_scale(den), den * 2 / 40.0)
        # the loss statistics of a step: ONE (asynchronous) all-reduce of [n0, n1, n_valid, anomaly weight sum]
        calls, real = [], dist.all_reduce

        def counting(t, *a, **k):
            calls.append(int(t.numel()))
            return real(t, *a, **k)
        dist.all_reduce = counting
        try:
            st = P.LossStats(torch.tensor([3.0 + rank, 1.0, 4.0 + rank, 10.0 if rank == 0 else 30.0]))
            assert torch.equal(st.counts(), torch.tensor([7.0, 2.0]))
            assert torch.allclose(st.valid_scale(), torch.tensor((4.0 + rank) * 2 / 9.0))
            assert torch.allclose(st.anomaly_scale(), den * 2 / 40.0)
        finally:
            dist.all_reduce = real
        assert calls == [4], calls
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_step_equivalence(rank, world, port, q):
    P = _init(rank, world, port)
    try:
        torch.set_num_threads(2)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import load_golden
        from oracle import idee_oracle as O
        a, sd, _ = load_golden("e2e_swin_v2_14x18")
        cfg = O.default_cfg(in_channels_dynamic=2)
        g = torch.Generator().manual_seed(5)
        x = torch.cat([a["x_d"], torch.randn(a["x_d"].shape, generator=g)], 0)[:, :, :, :, :8, :8]      # 4 clips
        me = (torch.rand((4, 8, 8), generator=g) < 0.3).float()
        ml = (torch.rand((4, 8, 8), generator=g) < 0.4).float()
        names = [k for k, v in sd.items() if v.is_floating_point() and not k.startswith("vq.")]

        def grads(xs, mes, mls, counts=None, den_scale=None):
            s = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
            outs = O.vq_model_forward(xs, s, cfg, training=True)
            pred, pred_y, _, z_q, loss_z_q = outs
            tgt = mes.unsqueeze(1)
            w = torch.log(((counts if counts is not None else torch.histc(tgt, bins=2)) / (counts.sum() if counts is not None else tgt.numel())) ** -0.5 + 1.1)
            bce = lambda p: (torch.nn.functional.binary_cross_entropy_with_logits(p, tgt, reduction="none") * w[tgt.long()]).mean()
            vq0 = O.lfq_indices_to_codes(torch.tensor([0]), s, cfg).detach()
            l_an = O.anomaly_l1_synthetic(z_q, mls, vq0)
            if den_scale is not None:
                l_an = l_an * den_scale
            total = bce(pred) + sum(bce(p) for p in pred_y) + cfg.lambda_anomaly * l_an      # loss_z_q is per-shard by definition
            total.backward()
            return torch.cat([s[k].grad.reshape(-1) for k in names])

        lo, hi = P.shard_batch(4)
        calls, real = [], dist.all_reduce

        def counting(t, *a, **k):
            calls.append(int(t.numel()))
            return real(t, *a, **k)
        dist.all_reduce = counting
        # the step's collectives as idee_amd/train.py issues them: the mask-only statistics in one vector (before the forward
        # pass), then the gradient bucket
        local_counts = torch.histc(me[lo:hi].unsqueeze(1), bins=2)
        stats = P.LossStats(torch.cat([local_counts, local_counts.sum().reshape(1), (1 - ml[lo:hi]).sum().reshape(1)]))
        flat = grads(x[lo:hi], me[lo:hi], ml[lo:hi], stats.counts(), stats.anomaly_scale())
        opt = _FakeOpt(flat)
        P.sync_gradients(opt)
        dist.all_reduce = real
        assert calls == [4, flat.numel()], calls      # exactly two collectives per step
        ddp = opt.flat_g * opt.grad_scale
        ref = grads(x, me, ml)                      # the same 4 clips in one process
        torch.testing.assert_close(ddp, ref, rtol=2e-4, atol=2e-6)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(worker, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_collectives_world2():
    _run(_worker_collectives, 29611)


def test_two_ranks_equal_one_rank_global_batch():
    _run(_worker_step_equivalence, 29612)
