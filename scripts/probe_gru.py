"""Developer probe: the ConvGRU forward unroll (T steps x L cells on the 8 x 8 latent) -- fused kernel vs launch-per-phase form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib, nn as K
from ipoke_amd import first_stage_train as FT
from ipoke_amd.first_stage import ConvGRU
from ipoke_amd._lib import check

for B, T, L, Z in ((20, 15, 4, 32), (32, 15, 4, 64)):
    rnn = ConvGRU(Z, Z, 3, L, dtype="bf16").cuda()
    x = K.from_nchw(torch.randn(B, Z, 8, 8, device="cuda"), "bf16"); h = K.from_nchw(torch.randn(B, Z, 8, 8, device="cuda"), "bf16")
    for mode in (0, 1):
        check(_lib.lib().ipoke_gru_set_fused(mode))
        with torch.no_grad():
            for _ in range(3):
                FT.gru_unroll(rnn, x, h, T, "bf16")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                FT.gru_unroll(rnn, x, h, T, "bf16")
            e1.record(); torch.cuda.synchronize()
        print(f"B={B} T={T} L={L} Z={Z} {'fused' if mode else 'per-phase'}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per unroll ({e0.elapsed_time(e1) / 10 * 1e3 / (T * L):.1f} us per cell-step)")
check(_lib.lib().ipoke_gru_set_fused(-1))
