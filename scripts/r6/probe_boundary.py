"""Developer probe (round 6): what bounds the step boundary of the c2 train step?  Same loop as bench.py's timed region.
  mode base   -- the benchmarked step
  mode reuse  -- the frozen encoders' outputs of the first batch are reused (NO encoder kernels): step time without the encoders
Combine with IPOKE_PROBE_NO_READY_JOIN=1 (the next forward does not wait for the optimizer queue: wrong numbers, right clock)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from ipoke_amd import configs
from ipoke_amd.trainer import SecondStageTrainer

mode = sys.argv[1] if len(sys.argv) > 1 else "base"
cfg = dict(configs.BENCH_CONFIGS["c2"])
dev = torch.device("cuda", 0)
model = bench.build_model(cfg, "bf16", dev)
tr = SecondStageTrainer(model)
batch = bench.synthetic_batch(cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], seed=1, device=dev)
tr.sync_initial_state(batch)
bench.randomise_couplings(model)
if mode == "reuse":
    with torch.no_grad():
        fi, cd = model.make_flow_input(batch)
    model.make_flow_input = lambda b, *a, **k: (fi, cd)
    tr.prefetch_stream = None
for i in range(8):
    tr.train_step(batch, i, next_batch=batch)
import gc; gc.collect(); gc.freeze()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for i in range(N):
    tr.train_step(batch, 8 + i, next_batch=batch)
torch.cuda.synchronize()
print(f"{mode} NO_JOIN={os.environ.get('IPOKE_PROBE_NO_READY_JOIN', '0')} SKIP_ADAM={os.environ.get('IPOKE_PROBE_SKIP_ADAM', '0')}: {(time.perf_counter() - t0) / N * 1e3:.2f} ms per step")
