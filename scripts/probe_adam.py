"""Isolated timing of the optimizer step of the full z = 64 flow: linear Adam + relayout of everything vs the fused path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import configs, optim as O
from ipoke_amd.flow import SupervisedMacowTransformer
m = SupervisedMacowTransformer(configs.flow_arch(64), dtype="bf16", device="cuda", max_batch=2)
eng = m.engine
eng.prepare_weights()
m.bind_grads().normal_(0, 1e-3)
opt = O.FusedAdamAmsgrad(m, lr=1e-3, weight_decay=1e-5)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
for fused in (False, True, False, True):
    O._FUSE_SHADOWS = fused
    print(f"fused={fused}: whole-buffer step {timed(opt.step):.2f} ms")
# piecewise with 128-block grids (as inside the train step)
n = eng.params.numel()
offs = sorted({off for name, off, shape, kind in eng.tensors if kind == 0})
cuts = [0] + [offs[len(offs) * i // 12] for i in range(1, 12)] + [n]
def pieces():
    opt.begin_step()
    for i in range(12):
        opt.step_range(cuts[i], cuts[i + 1])
    opt.finish_step()
for fused in (False, True, False, True):
    O._FUSE_SHADOWS = fused
    print(f"fused={fused}: 12 pieces, 128-block grids {timed(pieces):.2f} ms")
