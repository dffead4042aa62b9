"""Developer probe: GroupNorm forward passes (stats, apply) at the decoder's SPADE shapes, isolated; GB/s per kernel.
python scripts/r6/probe_norm.py [frames clips]"""
import sys, time
from ctypes import byref
import torch
from ipoke_amd import _lib
from ipoke_amd._lib import NormDesc

frames, clips = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (15, 32)
N = frames * clips
L = _lib.lib()
dev = "cuda"
s = _lib.current_stream()
for (H, C) in ((32, 256), (64, 128), (128, 64)):
    S, G = H * H, 16
    x = torch.randn(N * S, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    mg = torch.randn(clips * S, C, device=dev).bfloat16() * 0.1
    mb = torch.randn(clips * S, C, device=dev).bfloat16() * 0.1
    ws = torch.empty(int(L.ipoke_groupnorm_workspace_floats(N, S, G)), device=dev)
    d = NormDesc()
    d.x = x.data_ptr(); d.ldx = C; d.y = y.data_ptr(); d.ldy = C; d.N, d.S, d.C, d.G, d.eps = N, S, C, G, 1e-5
    d.mod_gamma = mg.data_ptr(); d.mod_beta = mb.data_ptr(); d.ld_mod = C; d.mod_samples = clips; d.act = _lib.ACT_NONE
    d.workspace = ws.data_ptr()
    def t(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_all = t(lambda: _lib.check(L.ipoke_groupnorm(byref(d), _lib.BF16, s)))
    t_st = t(lambda: _lib.check(L.ipoke_groupnorm_stats(x.data_ptr(), C, N, S, C, G, 1e-5, ws.data_ptr(), _lib.BF16, s)))
    d2 = NormDesc()
    d2.x = x.data_ptr(); d2.ldx = C; d2.y = y.data_ptr(); d2.ldy = C; d2.N, d2.S, d2.C, d2.G, d2.eps = N, S, C, C, 1e-5
    d2.act = _lib.ACT_RELU; d2.workspace = ws.data_ptr()
    ws2 = torch.empty(int(L.ipoke_groupnorm_workspace_floats(N, S, C)), device=dev)
    d2.workspace = ws2.data_ptr()
    t_in = t(lambda: _lib.check(L.ipoke_groupnorm(byref(d2), _lib.BF16, s)))
    t_in_st = t(lambda: _lib.check(L.ipoke_groupnorm_stats(x.data_ptr(), C, N, S, C, C, 1e-5, ws2.data_ptr(), _lib.BF16, s)))
    nb = x.numel() * 2
    print(f"   InstanceNorm + ReLU (no modulation): stats {t_in_st:.0f} us = {nb / t_in_st / 1e6:.2f} TB/s; apply {t_in - t_in_st:.0f} us = {2 * nb / (t_in - t_in_st) / 1e6:.2f} TB/s")
    t_cp = t(lambda: _lib.check(L.ipoke_add_act(x.data_ptr(), C, None, 0, y.data_ptr(), C, N * S, C, _lib.ACT_RELU, _lib.BF16, s)))
    t_ad = t(lambda: _lib.check(L.ipoke_add_act(x.data_ptr(), C, y.data_ptr(), C, y.data_ptr(), C, N * S, C, _lib.ACT_NONE, _lib.BF16, s)))
    print(f"   flat element-wise: relu copy {t_cp:.0f} us = {2 * nb / t_cp / 1e6:.2f} TB/s (1 R + 1 W); add in place {t_ad:.0f} us = {3 * nb / t_ad / 1e6:.2f} TB/s (2 R + 1 W)")
    print(f"{H}x{H} C={C} N={N}: tensor {nb / 1e6:.0f} MB; stats+finalize {t_st:.0f} us = {nb / t_st / 1e6:.2f} TB/s; "
          f"apply {t_all - t_st:.0f} us = {2 * nb / (t_all - t_st) / 1e6:.2f} TB/s (read + write)")
