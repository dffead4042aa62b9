import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from ipoke_amd import configs
from ipoke_amd.flow import SupervisedMacowTransformer
from ipoke_amd.utils.detfill import deterministic_fill_
g7 = np.load("/root/repo/tests/golden/g7_sample_64.npz"); g6 = np.load("/root/repo/tests/golden/g6_glue_64.npz")
for dtype in ("f32",):
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    m = SupervisedMacowTransformer(arch, dtype=dtype, device="cuda", init="none", max_batch=2)
    deterministic_fill_(m, prefix="flow."); m.sync_buffers()
    x = torch.from_numpy(g6["flow_input"]).cuda(); cond = torch.from_numpy(g6["cond"]).cuda()
    with torch.no_grad():
        out, ld = m(x, cond)
        rev = m(out, cond, reverse=True)
        print(dtype, "roundtrip err", (rev - x).abs().max().item(), "out nan", torch.isnan(out).any().item())
        z = torch.from_numpy(g7["z"]).cuda()
        mot = m(z, cond, reverse=True)
        print("reverse(z) nan:", torch.isnan(mot).any().item(), "max", mot.abs().max().item(), "ref max", np.abs(g7["motion"]).max())
        z1 = z[:1].contiguous(); c1 = cond[:1].contiguous()
        mot1 = m(z1, c1, reverse=True)
        print("B=1 reverse nan:", torch.isnan(mot1).any().item(), (mot1.cpu() - torch.from_numpy(g7["motion"][:1])).abs().max().item())
