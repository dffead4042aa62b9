"""Developer probe: the filter-resident stride-1 convolution on huge maps (conv3x3_c64_kernel) at the decoder's 128 x 128 layers --
64 -> 64 and the image head 64 -> 3 (fp32 output), 300 and 480 images -- per patch time; with IPOKE_LIB_PATH pointing at a
-DIPOKE_C64_ABL=n build: 1 no MFMA, 2 no epilogue, 3 no image stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib, nn as K

DEV = "cuda"
for N, cout, f32 in ((300, 64, False), (300, 3, True), (480, 64, False), (480, 3, True)):
    x = torch.randn(N * 128 * 128, 64, device=DEV).to(torch.bfloat16)
    w = torch.randn(cout, 64, 1, 3, 3, device=DEV) / (64 * 9) ** 0.5
    wop, kc = K.weight_operand(w, "bf16")
    f = lambda: K.conv(K.CL(x, N, (1, 128, 128), 64), wop, kc, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16", out_f32=f32)
    for _ in range(3):
        y = f()
    assert _lib.lib().ipoke_last_conv_kernel() == _lib.KERNEL_C64, _lib.lib().ipoke_last_conv_kernel()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    patches = N * 64
    print(f"N={N} 64->{cout}{' fp32 out' if f32 else ''}: {us:7.1f} us, {us / (patches / 256):5.2f} us per patch and CU, input {N * 128 * 128 * 128 / us * 1e-6:5.2f} TB/s")
