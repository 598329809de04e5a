"""Input pipeline for the training hot path.

`ClipWindowDataset` restates the item contract of the reference's `Synthetic_Dataset.__getitem__`
(dataset/Synthetic_dataset.py:310-376: sliding delta_t window, time axis flipped to "newest first", the
extreme / extreme-loss / anomaly masks, rot180 + flip augmentation) on an in-memory data cube; the cube is
either handed in by the caller (e.g. read from the reference's NetCDF files with xarray where that is
installed — not needed, and not importable, here) or generated (`synthetic_cube`) with the value range the
reference's normalisation produces (clip to [-10, 10], Synthetic_dataset.py:215).  Golden parity of the
item logic: tests/test_data_pipeline.py against items produced by the reference's own `__getitem__`
(tests/golden/dataset_items.npz).

`PrefetchLoader` replaces torch's DataLoader + per-iteration `.to(device)` (train_synthetic.py:79-90,175-182):
a worker thread collates batches into a ring of PINNED host buffers and copies them to the GPU on a side HIP
stream, so the training stream never waits for PCIe (30.7 MB per 4-clip batch at the headline config).
It yields exactly what `harness.train_epoch` consumes.  On a CPU-only host it degrades to plain CPU tensors
(that is a host-side data path, not a compute fallback: the model still refuses CPU tensors)."""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch


def synthetic_cube(V=6, T_total=60, H=200, W=200, V_static=0, seed=1234, p_extreme=0.02, p_anomaly=0.05):
    """A random cube with the shapes / dtypes / ranges the reference's dataset holds after normalisation."""
    rng = np.random.default_rng(seed)
    return dict(
        dynamic=np.clip(rng.standard_normal((V, T_total, H, W), dtype=np.float32), -10.0, 10.0),
        static=rng.standard_normal((V_static, H, W), dtype=np.float32),
        time=(np.arange(T_total) % 52).astype(np.float32),
        timestep=np.arange(1, T_total + 1).astype(np.int64),
        extreme=(rng.random((T_total, H, W)) < p_extreme).astype(np.uint8),
        anomaly=(rng.random((V, T_total, H, W)) < p_anomaly).astype(np.uint8),
    )


class ClipWindowDataset(torch.utils.data.Dataset):
    """Items as Synthetic_Dataset.__getitem__ returns them (Synthetic_dataset.py:310-376):
        dynamic (V, 1, delta_t, H, W) f32   newest time step first
        static (Vs, H, W), t (delta_t) week numbers + 1, newest first
        mask_extreme (H, W)        last time step's extreme mask, labels > 1 zeroed
        mask_extreme_loss (H, W)   uint8: any extreme inside the window
        mask_anomaly (V, delta_t, H, W), tstep (1,)"""

    item_layout = "synthetic"      # what PrefetchLoader batches (it does not load an item to find out)

    def __init__(self, dynamic, static, time, timestep, extreme, anomaly, delta_t=8, is_aug=False, rng=None):
        self.dynamic, self.static, self.time, self.timestep = dynamic, static, time, timestep
        self.extreme, self.anomaly = extreme, anomaly
        self.delta_t, self.is_aug = int(delta_t), bool(is_aug)
        self.rng = rng if rng is not None else np.random          # the reference draws from the global numpy RNG
        if dynamic.shape[1] < self.delta_t:
            raise ValueError(f"cube has {dynamic.shape[1]} time steps, fewer than delta_t={self.delta_t}")

    def __len__(self):
        return self.dynamic.shape[1] - self.delta_t + 1                                     # :378-385

    def __getitem__(self, index):
        dt = self.delta_t
        dyn = np.flip(self.dynamic[:, index:index + dt], 1)[:, None]                        # :331-332
        sta = self.static
        t = np.flip(self.time[index:index + dt] + 1)                                        # :336
        tstep = np.array([self.timestep[index + dt - 1]])                                   # :338
        m_ext = self.extreme[index + dt - 1].copy()                                         # :342-343
        m_ext[m_ext > 1] = 0
        m_loss = self.extreme[index:index + dt].sum(axis=0)                                 # :346-349 (flip is sum-invariant)
        m_loss[m_loss > 1] = 1
        m_loss = m_loss.astype(np.uint8)
        m_an = np.flip(self.anomaly[:, index:index + dt], 1)                                # :352
        if self.is_aug:                                                                     # :355-372
            if self.rng.choice(2):
                dyn, sta, m_ext, m_loss, m_an = (np.rot90(a, k=2, axes=(-1, -2)) for a in (dyn, sta, m_ext, m_loss, m_an))
            if self.rng.choice(2):
                ax = self.rng.randint(1, 3)
                dyn, sta, m_ext, m_loss, m_an = (np.flip(a, axis=-ax) for a in (dyn, sta, m_ext, m_loss, m_an))
        return tuple(np.ascontiguousarray(a) for a in (dyn, sta, t, m_ext, m_loss, m_an, tstep))


class ERA5ClipDataset(torch.utils.data.Dataset):
    """Items as ERA5_Land_Dataset.__getitem__ returns them (dataset/ERA5_Land_dataset.py:521-625), on in-memory arrays
    (NetCDF reading is the caller's; window_size = 1, i.e. no spatial reduction):
        dynamic (V, 2, delta_t, H, W) f32   weekly mean / std statistics, target week first (:213-292 orders the files so),
                                            normalised (:571-583), clipped to [-10, 10], NaN / inf -> nan_fill (:586)
        mask_drought (H, W)                 VHI < threshold at the target week (:562), cold / no-vegetation / water pixels zeroed
        mask_drought_loss (H, W)            any VHI < 35 inside the window (:552-560)
        mask_cold_surface (H, W)            target week
        mask_cold_surface_loss (delta_t, H, W)
        mask_sea (H, W), mask_no_vegetation (H, W), name
    Inputs: dynamic (V, 2, T_total, H, W) already in the orientation __load_datacube returns (latitude flipped, :517-519);
    vci / tci / cold: one array (n_files, H, W) PER WEEK in file orientation (the masks are flipped here as
    __generate_mask does, :417-484: VHI = nanmean_files(alpha VCI + (1 - alpha) TCI)); mean_var / std_var (V,) (:579-581)
    or, with `clima=(mean, std)` arrays (V, 2, T_total, H, W) in the cube's week order, the per-pixel climatology scaling
    (:573-577)."""

    item_layout = "era5"

    def __init__(self, dynamic, vci, tci, cold, mask_water, mask_no_vegetation, mean_var=None, std_var=None, clima=None,
                 delta_t=8, threshold=26.0, alpha=0.5, nan_fill=0.0, is_norm=True, is_aug=False, names=None, rng=None):
        self.dynamic, self.vci, self.tci, self.cold = dynamic, vci, tci, cold
        self.mask_water, self.mask_no_vegetation = mask_water, mask_no_vegetation
        self.mean_var, self.std_var, self.clima = mean_var, std_var, clima
        self.delta_t, self.threshold, self.alpha, self.nan_fill = int(delta_t), float(threshold), float(alpha), float(nan_fill)
        self.is_norm, self.is_aug, self.names = bool(is_norm), bool(is_aug), names
        self.rng = rng if rng is not None else np.random
        if dynamic.shape[2] < self.delta_t:
            raise ValueError(f"cube has {dynamic.shape[2]} weeks, fewer than delta_t={self.delta_t}")

    def __len__(self):
        return self.dynamic.shape[2] - self.delta_t + 1

    def _mask(self, week, thr):
        """__generate_mask (:417-484) for one week."""
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            vhi = np.nanmean(self.alpha * self.vci[week] + (1 - self.alpha) * self.tci[week], axis=0)
        cold = np.flip(self.cold[week].sum(axis=0), 0).astype(np.float32)
        cold[cold > 1] = 1
        drought = (np.flip(vhi, 0) < thr).astype(np.float32)
        drought[np.isnan(drought)] = 0
        drought[cold == 1] = 0
        drought[self.mask_no_vegetation == 1] = 0
        drought[self.mask_water == 1] = 0
        return drought, cold

    def __getitem__(self, index):
        dt = self.delta_t
        weeks = [index + dt - 1 - k for k in range(dt)]                  # target week first, then back in time
        dyn = self.dynamic[:, :, weeks].astype(np.float32, copy=True)
        m_loss = np.zeros((dt,) + dyn.shape[-2:], dtype=np.float32)
        c_loss = np.zeros_like(m_loss)
        for k, w in enumerate(weeks):
            m_loss[k], c_loss[k] = self._mask(w, 35)
        m_loss = m_loss.sum(axis=0)
        m_loss[m_loss > 1] = 1
        m_drought, m_cold = self._mask(weeks[0], self.threshold)
        sea, noveg = self.mask_water.copy(), self.mask_no_vegetation.copy()
        if self.is_norm:
            import warnings
            if self.clima is not None:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", category=RuntimeWarning)
                    dyn = (dyn - self.clima[0][:, :, weeks]) / self.clima[1][:, :, weeks]
            else:
                for v in range(dyn.shape[0]):
                    dyn[v, 0] = (dyn[v, 0] - self.mean_var[v]) / self.std_var[v]
                    dyn[v, 1] = dyn[v, 1] / self.std_var[v]
            dyn = np.clip(dyn, a_min=-10., a_max=10.)
        dyn[np.logical_or(np.isnan(dyn), np.isinf(dyn))] = self.nan_fill
        if self.is_aug:
            if self.rng.choice(2):
                dyn, m_drought, m_loss, m_cold, c_loss, sea, noveg = (np.rot90(a, k=2, axes=(-1, -2)) for a in
                                                                      (dyn, m_drought, m_loss, m_cold, c_loss, sea, noveg))
            if self.rng.choice(2):
                ax = self.rng.randint(1, 3)
                dyn, m_drought, m_loss, m_cold, c_loss, sea, noveg = (np.flip(a, axis=-ax) for a in
                                                                      (dyn, m_drought, m_loss, m_cold, c_loss, sea, noveg))
        name = self.names[weeks[0]] if self.names is not None else str(weeks[0])
        return tuple(np.ascontiguousarray(a) for a in (dyn, m_drought, m_loss, m_cold, c_loss, sea, noveg)) + (name,)


class PrefetchLoader:
    """Device batches produced `depth` batches ahead by a worker thread through pinned buffers and a side stream.
    rank / world shard the (shuffled) index list like a DistributedSampler with drop_last over ranks.  Two item layouts
    (`layout=` argument, else the dataset's `item_layout` attribute, else told from the first item the worker loads — the
    constructor never loads an item: no IO and no draw from an augmenting dataset's RNG before the first batch):
      synthetic  (SyntheticClipDataset items) -> (x_d (B,V,1,T,H,W) f32, mask_extreme (B,H,W), mask_extreme_loss (B,H,W),
                 timestep (B,) int64)                      — what train_synthetic.py:171-177 moves to the device;
      era5       (ERA5ClipDataset 8-tuples)   -> (x_d (B,V,2,T,H,W) f32, mask_drought (B,H,W), mask_drought_loss (B,H,W),
                 mask_cold_surface (B,H,W), mask_cold_surface_loss (B,T,H,W), None)   — train_ERA5_Land.py:168-176
    each a tuple harness.train_epoch hands to SyntheticTrainStep / ERA5TrainStep (the last element is the time step index
    for the anomaly collector, None for ERA5 whose items carry a file name instead)."""

    def __init__(self, dataset, batch_size, device, shuffle=True, drop_last=True, seed=0, rank=0, world=1, depth=2, layout=None):
        self.ds, self.bs, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.drop_last, self.seed, self.rank, self.world, self.depth = shuffle, drop_last, seed, rank, world, depth
        self.epoch = 0
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.layout = layout or getattr(dataset, "item_layout", None)
        if self.layout not in (None, "synthetic", "era5"):
            raise ValueError(f"PrefetchLoader: layout must be 'synthetic' or 'era5', got {self.layout!r}")
        self.fields = None          # set from the first item of the first batch (_describe)

    def _describe(self, first):
        """(index in the item, leading-dims-free shape, dtype) of every tensor of a batch, from an item already loaded."""
        if self.layout is None:
            self.layout = "era5" if (len(first) == 8 and isinstance(first[-1], str)) else "synthetic"
        if self.layout == "era5":
            self.fields = [(0, first[0].shape, torch.float32)] + [(i, first[i].shape, torch.float32) for i in (1, 2, 3, 4)]
        else:
            self.fields = [(0, first[0].shape, torch.float32), (3, first[3].shape, torch.float32), (4, first[4].shape, torch.float32),
                           (6, (), torch.int64)]

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _indices(self):
        idx = np.arange(len(self.ds))
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        per = len(idx) // self.world
        return idx[self.rank * per:(self.rank + 1) * per] if self.world > 1 else idx

    def __len__(self):
        n = len(self._indices())
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def _collate(self, items, bufs):
        for k, it in enumerate(items):
            for buf, (src, shape, dtype) in zip(bufs, self.fields):
                if shape == ():
                    buf[k] = int(np.asarray(it[src]).reshape(-1)[0])
                else:
                    buf[k].copy_(torch.from_numpy(np.ascontiguousarray(it[src], dtype=np.float32)))
        return len(items)

    def __iter__(self):
        idx = self._indices()
        nb = len(self)
        ring = []                   # pinned staging buffers, allocated by the worker once the first items are loaded
        free, ready = queue.Queue(), queue.Queue(maxsize=self.depth)
        for r in range(self.depth + 1):
            free.put(r)
        stop = threading.Event()

        def work():
            try:
                for b in range(nb):
                    r = free.get()
                    if stop.is_set():
                        return
                    items = [self.ds[int(i)] for i in idx[b * self.bs:(b + 1) * self.bs]]
                    if not ring:
                        if self.fields is None:
                            self._describe(items[0])
                        ring.extend(tuple(torch.empty((self.bs,) + tuple(shape), dtype=dtype, pin_memory=self.cuda)
                                          for _, shape, dtype in self.fields) for _ in range(self.depth + 1))
                    n = self._collate(items, ring[r])
                    if self.cuda:
                        with torch.cuda.stream(self.stream):
                            dev = tuple(t[:n].to(self.device, non_blocking=True) for t in ring[r])
                            ev = torch.cuda.Event()
                            ev.record(self.stream)
                    else:
                        dev, ev = tuple(t[:n].clone() for t in ring[r]), None
                    ready.put((r, dev, ev))
                ready.put(None)
            except BaseException as e:      # surface worker errors in the consumer
                ready.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                got = ready.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                r, dev, ev = got
                if self.layout == "era5":
                    dev = dev + (None,)
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)     # the training stream waits for THIS batch only
                    for t in dev:
                        if t is not None:
                            t.record_stream(torch.cuda.current_stream(self.device))
                    ev.synchronize()                                          # the pinned buffer may be refilled now
                free.put(r)
                yield dev
        finally:
            stop.set()
            free.put(0)
            th.join(timeout=5)
