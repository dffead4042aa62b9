"""Developer probe: who issues device-to-device copies in a c5 (sampling) step -- torch profiler, aten::copy_ / clone with Python stacks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
import bench
from ipoke_amd import configs
from torch.profiler import profile, ProfilerActivity

cfg = dict(configs.BENCH_CONFIGS["c5"])
model = bench.build_model(cfg, "bf16", "cuda")
batch = bench.synthetic_batch(cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], 1, "cuda")
with torch.no_grad():
    model.forward_density(batch)
bench.randomise_couplings(model)
for _ in range(2):
    model.forward_sample(batch, n_samples=1, n_logged_vids=1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.forward_sample(batch, n_samples=1, n_logged_vids=1)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "ipoke_amd" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in cnt.most_common(14):
    print(f"{n:5d} x {name:16s} {where}")
kern = collections.Counter(ev.name for ev in prof.events() if "Memcpy" in ev.name or "copyBuffer" in ev.name)
print(kern.most_common(5))
