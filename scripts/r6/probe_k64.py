"""Developer probe: conv1 of a coupling net (3x3, cin -> 2048 on the 8x8 latent, ELU, bf16) isolated, back-to-back launches."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ctypes import byref
from ipoke_amd import _lib, ops
from tests.helpers import shadow_nt
L = _lib.lib()
for B, Cin in ((20, 32), (20, 16), (20, 64), (32, 32), (40, 32)):
    M, Cout, kc = B * 64, 2048, -(-Cin // 8) * 8
    xa = torch.randn(M, kc, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 1, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    ws = shadow_nt(w, kc, dtype="bf16")
    out = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = xa.data_ptr(); d.a_sn = 64 * kc; d.a_sd = 0; d.a_sh = 8 * kc; d.a_sw = kc; d.a_sc = 1; d.Kc_real = kc; d.Kc = kc
    d.W = ws.data_ptr(); d.ldw = ws.shape[1]; d.Nout = Cout; d.act = _lib.ACT_ELU; d.C = out.data_ptr(); d.c_f32 = 0; d.ldc = Cout
    s = _lib.current_stream()
    _lib.check(L.ipoke_conv_forward_repeat(byref(d), _lib.BF16, 10, s)); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(L.ipoke_conv_forward_repeat(byref(d), _lib.BF16, 200, s)); e1.record(); torch.cuda.synchronize()
    print(f"B={B} cin={Cin}: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch (kernel family {L.ipoke_last_conv_kernel()}, IPOKE_K64={os.environ.get('IPOKE_K64', '1')})")
