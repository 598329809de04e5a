import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, numpy as np
from idee_amd.config import default_config
from idee_amd.train import SyntheticTrainStep
from idee_amd.data import ClipWindowDataset, PrefetchLoader, synthetic_cube
from idee_amd.harness import train_epoch, ConfusionEvaluator
cfg = default_config(encoder="Swin_3D", in_channels=1)
step = SyntheticTrainStep(cfg, device="cuda:0")
c = synthetic_cube(V=6, T_total=40, H=96, W=128, seed=3, p_extreme=0.1)
ds = ClipWindowDataset(**c, delta_t=8, is_aug=True, rng=np.random.RandomState(0))
ld = PrefetchLoader(ds, batch_size=4, device="cuda:0", shuffle=True, seed=0)
torch.cuda.reset_peak_memory_stats()
for ep in range(6):
    ld.set_epoch(ep)
    t0 = time.time()
    loss, _ = train_epoch(step, ld)
    torch.cuda.synchronize()
    print(f"epoch {ep}: mean loss {loss:.5f}  {len(ld)} steps in {time.time()-t0:.2f}s  peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
    assert np.isfinite(loss)
