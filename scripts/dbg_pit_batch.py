"""GPU debug: train-mode (power iteration) first-stage forward for batches of copies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import configs, first_stage_train as FT
from ipoke_amd.utils.detfill import deterministic_fill_
from ipoke_amd.first_stage import SpadeCondMotionModel
T = 16
X1 = (torch.rand(1, T, 3, 128, 128, generator=torch.Generator().manual_seed(131)) * 2 - 1).cuda()
eps1 = torch.randn(1, 32, 8, 8, generator=torch.Generator().manual_seed(5)).cuda()
def fresh():
    m = SpadeCondMotionModel(configs.first_stage_config(128, 32, T), dirs={}, train=False, dtype="f32")
    deterministic_fill_(m, prefix="first_stage.")
    return m.cuda().train()
ref = None
for ahead in (True, False):
    for hoist in (True, False):
        FT._SN_AHEAD, FT._HOIST_SPADE = ahead, hoist
        for B in (1, 2, 4):
            m = fresh()
            l, xh, mu, _ = m.training_loss(X1.repeat(B, 1, 1, 1, 1), eps1.repeat(B, 1, 1, 1))
            if ref is None:
                ref = xh[0].clone()
            print(f"ahead={ahead} hoist={hoist} B={B}: loss {l.item():.6f} X_hat per-slot err vs first run {[round((xh[b] - ref).abs().max().item(), 5) for b in range(B)]}",
                  "per-frame slot0", [round((xh[0, t] - ref[t]).abs().max().item(), 4) for t in range(T - 1)] if B > 1 else "")
            del m, l, xh
