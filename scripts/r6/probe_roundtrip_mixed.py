"""Developer probe (round 6): would a wider operand in the INVERSE of the masked-conv flows buy round-trip margin?
Upper bound of that idea: the whole reverse pass in exact-f32 arithmetic applied to the output of the bf16 forward pass of the same
parameters (z = 64 flow, golden pair repeated to B).  Prints max |reverse(forward(x)) - x| for (forward, reverse) in (bf16, bf16),
(bf16, f32), (f32, f32)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.test_bench_configs_gpu import full_flow
from tests.conftest import GOLDEN, t

g = {k: v for k, v in np.load(os.path.join(GOLDEN, "g3_full_flow_z64.npz"), allow_pickle=False).items()}
mb = full_flow(g, 64, "bf16", 40)
mf = full_flow(g, 64, "f32", 40)
for B in (20, 40):
    n = B // 2
    x = t(g["x"], "cuda").repeat(n, 1, 1, 1)
    cond = t(g["cond"], "cuda").repeat(n, 1, 1, 1)
    with torch.no_grad():
        ob, _ = mb(x, cond)
        of, _ = mf(x, cond)
        bb = (mb(ob, cond, reverse=True) - x).abs().max().item()
        bf = (mf(ob, cond, reverse=True) - x).abs().max().item()
        ff = (mf(of, cond, reverse=True) - x).abs().max().item()
    print(f"B = {B}: round trip max error -- bf16 forward / bf16 reverse {bb:.3e}; bf16 forward / f32 reverse {bf:.3e}; f32 / f32 {ff:.3e} "
          f"(|x| max {x.abs().max().item():.2f}; forward outputs differ by {(ob - of).abs().max().item():.3e})")
