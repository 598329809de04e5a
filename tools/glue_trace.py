import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idee_amd import _lib as _L; _L.set_option("IDEE_SWIN_MATH", "f22"); _L.set_option("IDEE_CONV_MATH", "f22")
import bench
from idee_amd.config import default_config
from idee_amd.train import SyntheticTrainStep
enc, H, W, T, Cin, batch, over, kind, label = bench.CONFIGS["2"]
cfg = default_config(encoder=enc, in_channels=Cin, **over)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ts = SyntheticTrainStep(cfg, device=dev)
x, masks = bench.synthetic_batch(cfg, kind, batch, T, H, W, 0, dev)
for _ in range(3):
    ts.step(x, *masks)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts.step(x, *masks)
    torch.cuda.synchronize()
c = collections.Counter()
dur = collections.Counter()
for e in prof.events():
    if e.device_type.name == "CPU" and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::cat", "aten::clone", "aten::mul", "aten::sum", "aten::_foreach_zero_", "aten::zeros_like", "aten::div", "aten::sub", "aten::neg", "aten::ones_like", "aten::mean"):
        key = (e.name, str(e.input_shapes)[:90])
        c[key] += 1
        dur[key] += e.device_time_total if hasattr(e, "device_time_total") else 0
for k, n in sorted(c.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(n, k[0], k[1], f"{dur[k]:.0f}us")
