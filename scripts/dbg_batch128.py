"""GPU debug: does the 128x128 first stage give per-slot identical results for a batch of copies?  (bisects encoder / GRU / decoder
and the halo kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import configs
from ipoke_amd.utils.detfill import deterministic_fill_
from ipoke_amd.first_stage import SpadeCondMotionModel

T = int(os.environ.get("DBG_T", "16"))
dtype = os.environ.get("DBG_DTYPE", "f32")
m = SpadeCondMotionModel(configs.first_stage_config(128, 32, T), dirs={}, train=False, dtype=dtype)
deterministic_fill_(m, prefix="first_stage.")
m = m.cuda().eval()
X1 = (torch.rand(1, T, 3, 128, 128, generator=torch.Generator().manual_seed(131)) * 2 - 1).cuda()
eps1 = torch.randn(1, 32, 8, 8, generator=torch.Generator().manual_seed(5)).cuda()
with torch.no_grad():
    z1, mu1, lv1 = m.enc_motion(X1.transpose(1, 2), eps=eps1)
    f1 = m.decode(z1, X1[:, 0], T - 1)
for B in (2, 4):
    X = X1.repeat(B, 1, 1, 1, 1); eps = eps1.repeat(B, 1, 1, 1)
    with torch.no_grad():
        z, mu, lv = m.enc_motion(X.transpose(1, 2), eps=eps)
        fr = m.decode(z1.repeat(B, 1, 1, 1), X[:, 0], T - 1)
    print(f"B={B} eval: encoder mu per-slot err {[(mu[b] - mu1[0]).abs().max().item() for b in range(B)]}")
    print(f"B={B} eval: decoder per-slot err {[(fr[b] - f1[0]).abs().max().item() for b in range(B)]}")
# training path, eval-mode spectral norm and train mode
m.train()
l1, xh1, _, _ = m.training_loss(X1, eps1, power_iteration=False)
for B in (2, 4):
    X = X1.repeat(B, 1, 1, 1, 1); eps = eps1.repeat(B, 1, 1, 1)
    l, xh, mu, _ = m.training_loss(X, eps, power_iteration=False)
    print(f"B={B} training_loss(pit=False): loss {l.item():.6f} vs {l1.item():.6f}; X_hat per-slot err {[(xh[b] - xh1[0]).abs().max().item() for b in range(B)]}")
    print(f"      per-frame err slot 0: {[round((xh[0, t] - xh1[0, t]).abs().max().item(), 4) for t in range(T - 1)]}")
