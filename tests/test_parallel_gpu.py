"""-m gpu: the product's world > 1 branch (idee_amd/train.py + idee_amd/parallel.py) on the GPU.  Two child ranks share
device 0 and talk through gloo (RCCL refuses two ranks per device; the collectives are backend-agnostic), each running
SyntheticTrainStep / ERA5TrainStep on 2 clips; a third child runs the same 4 clips in one rank.  The averaged flat
gradient, the parameters after one hand-made step and after a second `.step()` must agree.
Replaces nn.DataParallel, train_synthetic.py:133-137,175-205 / train_ERA5_Land.py:142-198."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, adam_params_close

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(kind, world, out):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    if world == 1:
        procs = [subprocess.Popen([sys.executable, WORKER, kind, out], env=base)]
    else:
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, WORKER, kind, out],
                                  env=dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(port), IDEE_DIST_BACKEND="gloo", IDEE_FORCE_DEVICE="0"))
                 for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0, f"{kind}: a rank of the world-{world} run failed"
    return torch.load(out, weights_only=False)


@pytest.mark.parametrize("kind", ["synthetic", "era5"])
def test_two_ranks_product_step_equals_one_rank(kind, tmp_path):
    two = _run_world(kind, 2, str(tmp_path / "w2.pt"))
    one = _run_world(kind, 1, str(tmp_path / "w1.pt"))
    assert two["world"] == 2 and one["world"] == 1 and two["replicas_equal"]
    # exactly two collectives per step — the 4-float loss statistics and the flat gradient bucket
    assert len(two["step_collectives"]) == 2 and two["step_collectives"][0] == 4 and two["step_collectives"][1] > 1000, two["step_collectives"]
    assert one["step_collectives"] == []
    assert torch.equal(two["p0"], one["p0"]), "rank 0's replica must have been broadcast"
    g2, g1 = two["grad"], one["grad"]
    tol = 2e-4 * float(g1.abs().max())
    assert float((g2 - g1).abs().max()) <= tol, f"flat gradient: max err {float((g2 - g1).abs().max()):.3e} > {tol:.3e}"
    # Parameters after Adam are compared where the comparison means something: on every element whose effective gradient
    # (g + weight_decay * p) exceeds a stated noise floor — 100 x the largest difference observed between the 2-shard and the
    # 1-shard gradient (summation order; float atomics).  There Adam's update lr * m_hat / (sqrt(v_hat) + eps) changes by at most
    # ~1 % of lr per step under that noise, and the replicas must agree that tightly.  Below the floor the update's SIGN is
    # decided by rounding noise (|step| = lr whatever |g| is), so those elements only get the trivial bound of 2 lr per step.
    lr, wd = one["lr"], one["weight_decay"]
    geff = g1 + wd * one["p0"]
    noise = float((g2 - g1).abs().max())
    firm = geff.abs() > 100.0 * max(noise, 1e-12)
    print(f"[dp {kind}] gradient noise {noise:.3e} (max |g| {float(g1.abs().max()):.3e}); elements above 100 x noise: "
          f"{int(firm.sum())}/{firm.numel()}")
    assert int(firm.sum()) >= 1000
    d1 = (two["p1"] - one["p1"]).abs()
    assert float(d1[firm].max()) <= 0.01 * lr, f"p1 on firm elements: {float(d1[firm].max()):.3e}"
    assert float(d1.max()) <= 2.0 * lr + 1e-7
    d2 = (two["p2"] - one["p2"]).abs()
    assert float(d2[firm].max()) <= 0.03 * lr, f"p2 on firm elements: {float(d2[firm].max()):.3e}"
    assert float(d2.max()) <= 4.0 * lr + 1e-7
    # losses: the mean over ranks of the per-rank (rescaled) shares is the global-batch loss of the one-rank run
    for k in one["parts"]:
        assert two["parts"][k] == pytest.approx(one["parts"][k], rel=1e-4, abs=1e-6), (k, two["parts"][k], one["parts"][k])
    assert two["total1"] == pytest.approx(one["total1"], rel=1e-4)
    assert two["total2"] == pytest.approx(one["total2"], rel=1e-4)


def test_two_ranks_vq_ema_kmeans_expiry_equals_one_rank(tmp_path):
    """VQ(ema_update=True, kmeans_init=True, threshold_ema_dead_code=2) on two ranks (2 clips each) against one rank on the
    4 clips: k-means seeds broadcast from rank 0, bins / sums / cluster sizes all-reduced, expiry replacements broadcast —
    replicas identical after every step and equal to the one-rank codebook; code indices equal token for token.
    Reference: models/codebook/VQ.py:164-202, 233-244, 528-539."""
    two = _run_world("vq_ema", 2, str(tmp_path / "vq2.pt"))
    one = _run_world("vq_ema", 1, str(tmp_path / "vq1.pt"))
    assert two["world"] == 2 and one["world"] == 1
    for step, (a, b) in enumerate(zip(two["steps"], one["steps"])):
        assert a["replicas_equal"], f"step {step}: replicas differ"
        assert torch.equal(a["idx"], b["idx"]), f"step {step}: code indices differ"
        torch.testing.assert_close(a["zq0"], b["zq0"], atol=1e-6, rtol=1e-5)
        for k in ("_codebook.embed", "_codebook.cluster_size", "_codebook.embed_avg"):
            torch.testing.assert_close(a["sd"][k], b["sd"][k], atol=2e-5, rtol=1e-5, msg=lambda m: f"step {step} {k}: {m}")
        assert bool(a["sd"]["_codebook.initted"])
    # the scenario must actually exercise expiry: some code of the one-rank run was replaced (cluster size reset to 2)
    assert any(float((s["sd"]["_codebook.cluster_size"] == 2.0).sum()) > 0 for s in one["steps"])


def test_two_host_threads_step_concurrently_on_their_own_streams():
    """The library is re-entrant (forward on the caller's thread, backward on autograd engine threads, one
    thread per replica under DataParallel): two host threads, each with its own model, optimizer and HIP stream, run train
    steps at the same time on one device; each must reproduce what it computes alone (the only run-to-run difference the
    kernels allow is the order of float atomics in the weight-gradient flushes)."""
    import threading
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd.train import SyntheticTrainStep

    kw = dict(encoder="Swin_3D", in_channels=1, in_channels_dynamic=3)
    N, T, H, W = 2, 8, 24, 40

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((N, 3, 1, T, H, W), generator=g).clamp_(-10, 10)
        me = (torch.rand((N, H, W), generator=g) < 0.10).float()
        ml = (torch.rand((N, H, W), generator=g) < 0.20).float()
        return [t.to("cuda") for t in (x, me, ml)]

    def make(seed):            # on the main thread: model initialisation draws from the process-wide CPU generator
        torch.manual_seed(100 + seed)
        st = SyntheticTrainStep(default_config(**kw), model=VQ_model(default_config(**kw)))
        with torch.no_grad():
            st.model.vq.project_in.weight.mul_(20.0)     # keep tokens off the LFQ sign boundary (parameters are views of flat_p)
        return st, [batch(seed * 10 + k) for k in range(3)]

    def run(seed, st, batches, stream, out):
        try:
            with torch.cuda.stream(stream):
                losses, grads = [], []
                for b in batches:
                    total, parts, _ = st.step(*b)
                    losses.append(torch.stack([total.reshape(())] + [parts[n].reshape(()) for n in ("extreme", "anomaly", "var", "z_q")]))
                    grads.append(st.opt.flat_g.detach().clone())          # stays on the device: no host sync inside the loop
                stream.synchronize()
                out[seed] = (torch.stack(losses).float().cpu(), st.opt.flat_p.detach().float().cpu().clone(),
                             [g_.float().cpu() for g_ in grads])
        except BaseException as e:          # surfaced by the main thread
            out[seed] = e

    alone, together = {}, {}
    for seed in (1, 2):
        st, bs = make(seed)
        torch.cuda.synchronize()
        run(seed, st, bs, torch.cuda.Stream(), alone)
    jobs = [(seed,) + make(seed) for seed in (1, 2)]
    torch.cuda.synchronize()
    threads = [threading.Thread(target=run, args=(seed, st, bs, torch.cuda.Stream(), together)) for seed, st, bs in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for seed in (1, 2):
        for res in (alone[seed], together[seed]):
            if isinstance(res, BaseException):
                raise res
        la, pa, ga = alone[seed]
        lt, pt, gt = together[seed]
        assert torch.isfinite(lt).all()
        torch.testing.assert_close(lt, la, rtol=2e-4, atol=1e-6)
        adam_params_close(f"threads seed {seed}", pt, pa, gt, ga, lr=1e-3)
