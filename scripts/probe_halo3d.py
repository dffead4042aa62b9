"""Isolated timing of the 3 x 3 x 3 convolutions of the motion encoder / temporal discriminator: IPOKE_HALO3D=1 (depth-halo kernel)
against IPOKE_HALO3D=0 (27-tap implicit GEMM); native back-to-back launches, HIP events."""
import os
import sys
from ctypes import byref

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipoke_amd import _lib, nn as K, ops  # noqa: E402

for N, D, H, W, cin, cout in [(20, 16, 64, 64, 64, 64), (20, 8, 32, 32, 128, 128), (20, 4, 16, 16, 256, 256), (20, 12, 32, 32, 64, 64),
                              (20, 6, 32, 32, 128, 128), (20, 3, 16, 16, 256, 256), (20, 2, 16, 16, 512, 512)]:
    x = torch.randn(N * D * H * W, cin, device="cuda").to(torch.bfloat16)
    wop, kc = K.weight_operand(torch.randn(cout, cin, 3, 3, 3, device="cuda") / (5.2 * cin ** 0.5), "bf16")
    y = torch.empty(N * D * H * W, cout, device="cuda", dtype=torch.bfloat16)
    d = ops.conv_desc(N, (D, H, W), (D, H, W), (3, 3, 3), (1, 1, 1), (1, 1, 1))
    d.A = x.data_ptr(); d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = D * H * W * cin, H * W * cin, W * cin, cin, 1
    d.Kc_real = d.Kc = kc; d.W = wop.data_ptr(); d.ldw = wop.shape[1]; d.Nout = cout; d.C = y.data_ptr(); d.ldc = cout; d.act = _lib.ACT_RELU
    s = _lib.current_stream()
    _lib.check(_lib.lib().ipoke_conv_forward_repeat(byref(d), 1, 3, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().ipoke_conv_forward_repeat(byref(d), 1, 10, s))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    gf = 2.0 * N * D * H * W * cout * 27 * cin / 1e9
    print(f"HALO3D={os.environ.get('IPOKE_HALO3D', '1')} N={N} {D}x{H}x{W} {cin}->{cout}: {us:8.1f} us  {gf / us * 1e-3:6.0f} TFLOP/s")
