"""Debug: the gradient tensor handed to the strided-conv backward is the head of a NaN-filled slab: any read past its end shows up."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch.nn.functional as F
from ipoke_amd import first_stage_train as FT, nn as K
from ipoke_amd.first_stage import _Conv
from tests.test_vae_bwd_units_gpu import _cl5, _nchw
for dtype in ("bf16", "f32"):
    for st, dhw, cin, cout in [((2, 2, 2), (4, 16, 16), 16, 24), ((2, 1, 1), (6, 8, 8), 24, 16), ((2, 2, 2), (5, 9, 10), 8, 12), ((1, 2, 2), (3, 8, 12), 16, 16)]:
        gen = torch.Generator().manual_seed(13)
        mod = _Conv(cin, cout, (3, 3, 3), st, (1, 1, 1), bias=False, dims=3).to("cuda")
        x = torch.randn(2, cin, *dhw, generator=gen)
        y = F.conv3d(x, mod.weight.detach().cpu(), None, stride=st, padding=1)
        dy = torch.randn(y.shape, generator=gen)
        for phases in (True, False):
            FT._DG_PHASES = phases
            mod.weight.grad = None
            xc = _cl5(x, dtype); xc.t.requires_grad_(True)
            out = FT.conv(mod, xc, dtype)
            g = _cl5(dy, dtype).t
            slab = torch.full((g.shape[0] + 4096, g.shape[1]), float("nan"), dtype=g.dtype, device=g.device)
            slab[:g.shape[0]] = g
            out.t.backward(slab[:g.shape[0]])
            gx = xc.t.grad.float()
            bad = torch.isnan(gx).any(1).nonzero().flatten().tolist()
            print(dtype, st, dhw, cin, cout, "phases" if phases else "27-tap", "NaN rows of dx:", len(bad), bad[:4], bad[-4:], "| weight grad NaN", int(torch.isnan(mod.weight.grad).sum()))
