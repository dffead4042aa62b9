"""Isolated timing of the 2-D 3x3 convolutions of the SPADE decoder / VGG stack: run once with IPOKE_HALO=1 and once with
IPOKE_HALO=0 (halo-staged kernel vs the implicit-GEMM kernel); native back-to-back launches, HIP events."""
import os
import sys
from ctypes import byref

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipoke_amd import _lib, nn as K, ops  # noqa: E402

SHAPES = [(32, 128, 128, 64, 64), (32, 64, 64, 128, 128), (32, 32, 32, 256, 256), (32, 16, 16, 256, 256), (32, 128, 128, 64, 3),
          (300, 128, 128, 64, 64), (300, 64, 64, 128, 128), (300, 32, 32, 256, 256), (300, 16, 16, 512, 512)]
for N, H, W, cin, cout in SHAPES:
    x = torch.randn(N * H * W, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, cin, 1, 3, 3, device="cuda") / (3 * cin ** 0.5))
    wop, kc = K.weight_operand(w, "bf16")
    ldc = K.round_up(cout, 8)
    y = torch.empty(N * H * W, ldc, device="cuda", dtype=torch.bfloat16)
    d = ops.conv_desc(N, (1, H, W), (1, H, W), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = x.data_ptr(); d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = H * W * cin, H * W * cin, W * cin, cin, 1
    d.Kc_real = d.Kc = kc; d.W = wop.data_ptr(); d.ldw = wop.shape[1]; d.Nout = cout; d.C = y.data_ptr(); d.ldc = ldc; d.act = _lib.ACT_RELU
    s = _lib.current_stream()
    _lib.check(_lib.lib().ipoke_conv_forward_repeat(byref(d), 1, 3, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().ipoke_conv_forward_repeat(byref(d), 1, 20, s))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gf = 2.0 * N * H * W * cout * 9 * cin / 1e9
    mb = (N * H * W * (cin + ldc) * 2) / 1e6
    print(f"HALO={os.environ.get('IPOKE_HALO', '1')} N={N} {H}x{W} {cin}->{cout}: {us:8.1f} us  {gf / us * 1e-3:7.1f} TFLOP/s  ({mb / us * 1e-3:5.2f} TB/s of in+out bytes)")
