"""Developer probe: does the row pitch of the GEMM operands matter (L2-channel camping)?  conv2 shape, padded pitches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref
from ipoke_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hid, M, dev = 2048, B * 64, "cuda"
td = torch.bfloat16
for pa, pw, pc in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (128, 128, 128), (32, 32, 32), (192, 192, 192)]:
    lda, ldw, ldc = hid + pa, hid + pw, hid + pc
    NS = 6                                        # rotate operand sets: 6 x (8.6 + 5.4) MB
    sets = []
    for i in range(NS):
        a = torch.randn(M, lda, device=dev).to(td)
        w = (torch.randn(hid, ldw, device=dev) / hid ** 0.5).to(td)
        c = torch.empty(M, ldc, device=dev, dtype=td)
        d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0))
        d.A = a.data_ptr(); d.a_sn = 64 * lda; d.a_sh = 8 * lda; d.a_sw = lda; d.a_sc = 1; d.Kc_real = hid; d.Kc = hid
        d.W = w.data_ptr(); d.ldw = ldw; d.Nout = hid; d.act = _lib.ACT_ELU; d.C = c.data_ptr(); d.ldc = ldc
        sets.append((d, a, w, c))
    dt, stream = ops._dt("bf16"), _lib.current_stream()
    L = _lib.lib()
    for d, *_ in sets:
        _lib.check(L.ipoke_conv_forward(byref(d), dt, stream))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 240
    e0.record()
    for i in range(n):
        _lib.check(L.ipoke_conv_forward(byref(sets[i % NS][0]), dt, stream))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"pad A {pa:3d} W {pw:3d} C {pc:3d} elems: {us:6.2f} us  {2.0 * M * hid * hid / us * 1e-6:7.1f} TF/s")
