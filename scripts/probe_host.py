"""Host-side cost of a second-stage train step: wall time the host spends inside train_step (queueing only, no synchronisation)
against the GPU time per step -- tells whether the loop is launch-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ipoke_amd import configs  # noqa: E402
from ipoke_amd.trainer import SecondStageTrainer  # noqa: E402

cfg = dict(configs.BENCH_CONFIGS["c2"])
dev = torch.device("cuda", 0)
model = bench.build_model(cfg, "bf16", dev)
tr = SecondStageTrainer(model)
batch = bench.synthetic_batch(cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], seed=1, device=dev)
tr.sync_initial_state(batch)
bench.randomise_couplings(model)
for i in range(5):
    tr.train_step(batch, i, next_batch=batch)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for i in range(20):
    t0 = time.perf_counter()
    tr.train_step(batch, 5 + i, next_batch=batch)
    host.append(time.perf_counter() - t0)
t_queued = time.perf_counter() - t_all
torch.cuda.synchronize()
t_total = time.perf_counter() - t_all
host.sort()
print(f"host per step: median {host[10] * 1e3:.1f} ms (min {host[0] * 1e3:.1f}, max {host[-1] * 1e3:.1f}); all 20 steps queued after {t_queued * 1e3:.0f} ms, "
      f"GPU done after {t_total * 1e3:.0f} ms ({t_total / 20 * 1e3:.1f} ms per step)")
