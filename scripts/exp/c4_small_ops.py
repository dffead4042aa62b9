"""Developer probe: which Python lines issue the small torch kernels of a c4 step (torch.profiler with stacks)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from ipoke_amd import configs
from ipoke_amd.first_stage import SpadeCondMotionModel
from ipoke_amd.first_stage_train import FirstStageTrainer
cfg = dict(configs.BENCH_CONFIGS["c4"]); B, T, size, z = cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], cfg["z_dim"]
torch.manual_seed(0)
model = SpadeCondMotionModel(configs.first_stage_config(size, z, T), dirs={}, dtype="bf16").to("cuda")
tr = FirstStageTrainer(model)
batch = bench.synthetic_batch(B, T, size, seed=1, device="cuda")
eps = torch.randn(B, z, 8, 8).cuda()
for i in range(3):
    tr.step(batch["images"], eps)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(batch["images"], eps)
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::div", "aten::sum") and ev.device_type == torch.autograd.DeviceType.CPU:
        st = [s for s in (ev.stack or []) if "ipoke_amd" in s or "bench" in s]
        shp = tuple(tuple(s) for s in (ev.input_shapes or [])) if hasattr(ev, "input_shapes") else ()
        cnt[(ev.name, st[0] if st else "(autograd / no python frame)")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:5d}  {name:12s} {where}")
