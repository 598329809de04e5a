"""Input pipeline (idee_amd/data.py) against items produced by the reference's own
Synthetic_Dataset.__getitem__ (tests/golden/dataset_items.npz, made by tests/golden/make_fixtures.py --data-only)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from idee_amd.data import ClipWindowDataset, PrefetchLoader, synthetic_cube

NAMES = ("dynamic", "static", "t", "mask_extreme", "mask_extreme_loss", "mask_anomaly", "tstep")


def _np(v):
    return v.numpy() if isinstance(v, torch.Tensor) else v


def _cube():
    a, _, _ = load_golden("dataset_items")
    cube = {k[5:]: _np(v) for k, v in a.items() if k.startswith("cube/")}
    return a, cube, int(a["delta_t"])


@pytest.mark.parametrize("is_aug", [False, True])
def test_items_match_reference_getitem(is_aug):
    a, cube, dt = _cube()
    ds = ClipWindowDataset(cube["dynamic"], cube["static"], cube["time"], cube["timestep"], cube["extreme"], cube["anomaly"],
                           delta_t=dt, is_aug=is_aug)
    assert len(ds) == int(a["len"])
    keys = [k for k in a if k.startswith(f"item/aug{int(is_aug)}/") and k.endswith("/dynamic")]
    assert keys
    seen_variants = set()
    for k in keys:
        _, _, itag, stag, _ = k.split("/")
        index, rs = int(itag[1:]), int(stag[1:])
        np.random.seed(1234 + rs)          # the reference draws its augmentation from numpy's global generator
        item = ds[index]
        for n, v in zip(NAMES, item):
            want = _np(a[f"item/aug{int(is_aug)}/{itag}/{stag}/{n}"])
            assert v.shape == want.shape and v.dtype == want.dtype, (n, v.shape, want.shape, v.dtype, want.dtype)
            assert np.array_equal(v, want), (k, n)
        seen_variants.add(item[0].tobytes())
    if is_aug:
        assert len(seen_variants) > len({k.split("/")[2] for k in keys})     # the seeds exercised different augmentations


def test_item_contract_shapes():
    c = synthetic_cube(V=3, T_total=12, H=10, W=14, V_static=2, seed=5)
    assert c["dynamic"].min() >= -10 and c["dynamic"].max() <= 10
    ds = ClipWindowDataset(**c, delta_t=8)
    dyn, sta, t, me, ml, man, ts = ds[2]
    assert dyn.shape == (3, 1, 8, 10, 14) and dyn.dtype == np.float32
    assert sta.shape == (2, 10, 14) and t.shape == (8,) and ts.shape == (1,)
    assert me.shape == (10, 14) and ml.shape == (10, 14) and ml.dtype == np.uint8 and man.shape == (3, 8, 10, 14)
    assert np.array_equal(dyn[:, 0, 0], c["dynamic"][:, 9])          # newest time step first
    assert int(ts[0]) == int(c["timestep"][9])
    assert np.array_equal(ml, (c["extreme"][2:10].sum(0) > 0).astype(np.uint8))
    with pytest.raises(ValueError):
        ClipWindowDataset(**c, delta_t=13)


def test_prefetch_loader_cpu_batches_and_sharding():
    c = synthetic_cube(V=2, T_total=19, H=6, W=5, seed=9)
    ds = ClipWindowDataset(**c, delta_t=8)
    assert len(ds) == 12
    ld = PrefetchLoader(ds, batch_size=4, device="cpu", shuffle=False)
    got = list(ld)
    assert len(got) == len(ld) == 3
    x, me, ml, ts = got[1]
    assert x.shape == (4, 2, 1, 8, 6, 5) and me.shape == (4, 6, 5) and ml.dtype == torch.float32 and ts.dtype == torch.int64
    for k in range(4):
        item = ds[4 + k]
        assert torch.equal(x[k], torch.from_numpy(item[0])) and int(ts[k]) == int(item[6][0])
        assert torch.equal(me[k], torch.from_numpy(item[3].astype(np.float32)))
    # shuffled + sharded: the two ranks see disjoint halves of the same permutation, and epochs differ
    parts = []
    for rank in range(2):
        l2 = PrefetchLoader(ds, batch_size=3, device="cpu", shuffle=True, seed=3, rank=rank, world=2)
        parts.append(torch.cat([b[3] for b in l2]))
    assert len(parts[0]) == len(parts[1]) == 6 and not set(parts[0].tolist()) & set(parts[1].tolist())
    l3 = PrefetchLoader(ds, batch_size=3, device="cpu", shuffle=True, seed=3)
    e0 = torch.cat([b[3] for b in l3])
    l3.set_epoch(1)
    e1 = torch.cat([b[3] for b in l3])
    assert sorted(e0.tolist()) == sorted(e1.tolist()) and e0.tolist() != e1.tolist()
    # ragged tail
    l4 = PrefetchLoader(ds, batch_size=5, device="cpu", shuffle=False, drop_last=False)
    assert [b[0].shape[0] for b in l4] == [5, 5, 2]


def test_prefetch_loader_constructor_loads_no_item():
    """Building the loader reads nothing from the dataset (no IO, no draw from an augmenting dataset's RNG): the batches of a
    seeded augmenting dataset equal the items drawn in loader order from a twin dataset with the same seed."""
    class Counting(ClipWindowDataset):
        loads = 0

        def __getitem__(self, i):
            Counting.loads += 1
            return super().__getitem__(i)
    c = synthetic_cube(V=2, T_total=14, H=6, W=6, seed=4)
    ds = Counting(**c, delta_t=8, is_aug=True, rng=np.random.RandomState(7))
    ld = PrefetchLoader(ds, batch_size=2, device="cpu", shuffle=False)
    assert Counting.loads == 0 and ld.layout == "synthetic"
    twin = ClipWindowDataset(**c, delta_t=8, is_aug=True, rng=np.random.RandomState(7))
    k = 0
    for x, me, ml, ts in ld:
        for j in range(x.shape[0]):
            item = twin[k]
            assert torch.equal(x[j], torch.from_numpy(item[0])) and int(ts[j]) == int(item[6][0])
            k += 1
    assert k == Counting.loads == 6
    # an explicit layout overrides; an unknown one is refused
    assert PrefetchLoader(ds, batch_size=2, device="cpu", layout="era5").layout == "era5"
    with pytest.raises(ValueError):
        PrefetchLoader(ds, batch_size=2, device="cpu", layout="netcdf")


def test_prefetch_loader_worker_error_surfaces():
    class Bad(ClipWindowDataset):
        def __getitem__(self, i):
            if i == 5:
                raise RuntimeError("boom")
            return super().__getitem__(i)
    c = synthetic_cube(V=1, T_total=16, H=4, W=4, seed=1)
    with pytest.raises(RuntimeError, match="boom"):
        list(PrefetchLoader(Bad(**c, delta_t=8), batch_size=2, device="cpu", shuffle=False))


@pytest.mark.gpu
def test_prefetch_loader_feeds_train_epoch_on_gpu():
    from idee_amd.config import default_config
    from idee_amd.harness import train_epoch
    from idee_amd.train import SyntheticTrainStep
    c = synthetic_cube(V=2, T_total=14, H=16, W=24, seed=11, p_extreme=0.2)
    ds = ClipWindowDataset(**c, delta_t=8, is_aug=True, rng=np.random.RandomState(0))
    ld = PrefetchLoader(ds, batch_size=2, device="cuda:0", shuffle=True, seed=1)
    ref = PrefetchLoader(ClipWindowDataset(**c, delta_t=8, is_aug=True, rng=np.random.RandomState(0)), batch_size=2,
                         device="cpu", shuffle=True, seed=1)
    dev_batches, cpu_batches = list(ld), list(ref)
    assert len(dev_batches) == len(cpu_batches) == 3
    for d, h in zip(dev_batches, cpu_batches):
        assert all(t.is_cuda for t in d)
        for td, th in zip(d, h):
            assert torch.equal(td.cpu(), th)
    cfg = default_config(encoder="Swin_3D", in_channels=1, in_channels_dynamic=2)
    step = SyntheticTrainStep(cfg, device="cuda:0")
    ld.set_epoch(1)
    mean_loss, _ = train_epoch(step, ld)
    assert np.isfinite(mean_loss)


def test_era5_items_match_reference_getitem():
    """idee_amd.data.ERA5ClipDataset against items produced by the reference's own ERA5_Land_Dataset.__getitem__ /
    __generate_mask on the same in-memory arrays (tests/golden/era5_items.npz; incl. NaN / inf fill, multi-file weeks and
    the seeded rot180 / flip augmentation): bit-equal."""
    import numpy as np
    from conftest import GOLDEN
    from idee_amd.data import ERA5ClipDataset
    z = np.load(GOLDEN + "/era5_items.npz")
    Tt = z["dynamic"].shape[2]
    dt = int(z["delta_t"])
    vci, tci, cold = ([z[f"{k}/{w}"] for w in range(Tt)] for k in ("vci", "tci", "cold"))
    names = ("dynamic", "mask_drought", "mask_drought_loss", "mask_cold_surface", "mask_cold_surface_loss", "mask_sea", "mask_no_vegetation")
    n_checked = 0
    for is_aug in (False, True):
        ds = ERA5ClipDataset(z["dynamic"], vci, tci, cold, z["mask_water"], z["mask_no_vegetation"], z["mean_var"], z["std_var"],
                             delta_t=dt, threshold=26.0, alpha=0.5, nan_fill=0.0, is_aug=is_aug)
        assert len(ds) == Tt - dt + 1
        for index in (0, 3, Tt - dt):
            for rs in ((0,) if not is_aug else (0, 1, 2, 3, 4, 5)):
                np.random.seed(2234 + rs)
                item = ds[index]
                for n, v in zip(names, item[:7]):
                    ref = z[f"item/aug{int(is_aug)}/i{index}/s{rs}/{n}"]
                    assert v.shape == ref.shape and v.dtype == ref.dtype, (n, v.shape, ref.shape, v.dtype, ref.dtype)
                    assert np.array_equal(v, ref, equal_nan=True), (is_aug, index, rs, n)
                    n_checked += 1
    assert n_checked == 7 * (3 + 18)


def _era5_dataset(is_aug=False):
    from conftest import GOLDEN
    from idee_amd.data import ERA5ClipDataset
    z = np.load(GOLDEN + "/era5_items.npz")
    Tt = z["dynamic"].shape[2]
    vci, tci, cold = ([z[f"{k}/{w}"] for w in range(Tt)] for k in ("vci", "tci", "cold"))
    return ERA5ClipDataset(z["dynamic"], vci, tci, cold, z["mask_water"], z["mask_no_vegetation"], z["mean_var"], z["std_var"],
                           delta_t=int(z["delta_t"]), threshold=26.0, alpha=0.5, nan_fill=0.0, is_aug=is_aug)


def test_prefetch_loader_era5_layout_cpu():
    """ERA5ClipDataset 8-tuples through the loader: batches (x_d, drought, drought_loss, cold, cold_loss, None) — exactly the
    tensors train_ERA5_Land.py:168-176 moves to the device — equal to the stacked items."""
    ds = _era5_dataset()
    ld = PrefetchLoader(ds, batch_size=2, device="cpu", shuffle=False, drop_last=True)
    assert ld.layout == "era5"
    batches = list(ld)
    assert len(batches) == len(ds) // 2
    for b, batch in enumerate(batches):
        assert len(batch) == 6 and batch[5] is None
        items = [ds[2 * b], ds[2 * b + 1]]
        for pos, src in enumerate((0, 1, 2, 3, 4)):
            want = torch.from_numpy(np.stack([np.asarray(it[src], dtype=np.float32) for it in items]))
            assert batch[pos].dtype == torch.float32 and torch.equal(batch[pos], want), (b, src)
    V, C, T, H, W = ds[0][0].shape
    assert batches[0][0].shape == (2, V, C, T, H, W) and batches[0][4].shape == (2, T, H, W)


@pytest.mark.gpu
def test_era5_dataset_loader_train_epoch_on_gpu():
    """The real-data path end to end: ERA5ClipDataset -> PrefetchLoader -> harness.train_epoch(ERA5TrainStep) with the
    threshold-0.35 evaluator on the valid pixels (train_ERA5_Land.py:160-229), one small epoch."""
    from idee_amd.config import default_config
    from idee_amd.harness import ConfusionEvaluator, train_epoch
    from idee_amd.train import ERA5TrainStep
    # (the golden cube has delta_t = 4, below the classifier's three stride-2 temporal kernels: a seeded cube with delta_t = 8)
    from idee_amd.data import ERA5ClipDataset
    rs = np.random.RandomState(5)
    V, Tt, H, W, dt = 3, 11, 12, 16, 8
    dyn = rs.randn(V, 2, Tt, H, W).astype(np.float32)
    dyn[rs.rand(*dyn.shape) < 0.01] = np.nan
    vci = [rs.rand(2, H, W).astype(np.float32) * 100 for _ in range(Tt)]
    tci = [rs.rand(2, H, W).astype(np.float32) * 100 for _ in range(Tt)]
    cold = [(rs.rand(2, H, W) < 0.05).astype(np.float32) for _ in range(Tt)]
    ds = ERA5ClipDataset(dyn, vci, tci, cold, (rs.rand(H, W) < 0.05).astype(np.float32), (rs.rand(H, W) < 0.05).astype(np.float32),
                         np.zeros(V, np.float32), np.ones(V, np.float32), delta_t=dt, threshold=40.0, is_aug=True,
                         rng=np.random.RandomState(1))
    V, C, T, H, W = ds[0][0].shape
    ld = PrefetchLoader(ds, batch_size=2, device="cuda:0", shuffle=True, seed=3)
    cfg = default_config(encoder="Swin_3D", in_channels=C, in_channels_dynamic=V, delta_t=T)
    step = ERA5TrainStep(cfg, device="cuda:0")
    ev = ConfusionEvaluator(threshold=0.35, device="cuda:0")
    mean_loss, res = train_epoch(step, ld, evaluator=ev)
    assert np.isfinite(mean_loss)
    assert res is not None and res["seen_all"] > 0
