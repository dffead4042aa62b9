"""Developer probe: where the HOST spends its time in a c4 step (first-stage L1 + KL train step, B = 20) -- cProfile over 6 steps with
the GPU running asynchronously, and the pure host time of a step (no synchronisation inside).  Usage: python scripts/probe_c4_host.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import configs
from ipoke_amd.first_stage import SpadeCondMotionModel
from ipoke_amd.first_stage_train import FirstStageTrainer

B, T, size, z = 20, 16, 128, 32
torch.manual_seed(0)
model = SpadeCondMotionModel(configs.first_stage_config(size, z, T), dirs={}, dtype="bf16").cuda()
trainer = FirstStageTrainer(model)
X = torch.rand(B, T, 3, size, size, device="cuda") * 2 - 1
eps = torch.randn(B, z, 8, 8).cuda()
for i in range(4):
    trainer.step(X, eps)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6):
    trainer.step(X, eps)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"6 steps: host returned after {t_host * 1e3 / 6:.1f} ms per step, GPU done after {t_all * 1e3 / 6:.1f} ms per step")
pr = cProfile.Profile()
pr.enable()
for i in range(6):
    trainer.step(X, eps)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
