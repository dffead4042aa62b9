"""Developer probe: time of the frozen encoders (make_flow_input) at the bench config, with a per-kernel breakdown."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ipoke_amd import configs
cfg = dict(configs.BENCH_CONFIGS["c2"])
model = bench.build_model(cfg, "bf16", torch.device("cuda", 0))
batch = bench.synthetic_batch(cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], 1, "cuda")
for _ in range(3):
    model.make_flow_input(batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
t0 = time.time()
for _ in range(10):
    model.make_flow_input(batch)
host = (time.time() - t0) / 10 * 1e3
e1.record(); torch.cuda.synchronize()
print(f"make_flow_input: gpu {e0.elapsed_time(e1)/10:.2f} ms, host enqueue {host:.2f} ms")
