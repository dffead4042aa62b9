"""Developer probe: isolated timing of the weight-gradient (TN) GEMM at the NICE conv shapes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib
from ipoke_amd._lib import check, WgradDesc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
PAD = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # extra elements of row pitch on both operands
hid = 2048
dev = "cuda"; M = B * 64
lib = _lib.lib(); s = _lib.current_stream()
NW = 8
def mk(k, pad, Kc, Nout, ldy, w_sn, w_sc, w_st):
    ds = []
    for i in range(NW):
        A = torch.randn(M, Kc + PAD, device=dev).to(torch.bfloat16)
        dY = torch.randn(M, ldy + PAD, device=dev).to(torch.bfloat16)
        dW = torch.empty(Nout * w_sn + 16, device=dev)
        w = WgradDesc()
        w.NB = B; w.Di = 1; w.Hi = 8; w.Wi = 8; w.Do = 1; w.Ho = 8; w.Wo = 8; w.kd = 1; w.kh = w.kw = k
        w.sd = w.sh = w.sw = 1; w.ph = w.pw = pad
        w.A = A.data_ptr(); w.a_sn = 64 * (Kc + PAD); w.a_sh = 8 * (Kc + PAD); w.a_sw = Kc + PAD; w.a_sc = 1; w.Kc_real = Kc; w.Kc = Kc
        w.dY = dY.data_ptr(); w.ldy = ldy + PAD; w.Nout = Nout
        w.dW = dW.data_ptr(); w.w_sn = w_sn; w.w_sc = w_sc; w.w_st = w_st
        ds.append((w, A, dY, dW))
    return ds
def run(ds, n=200):
    for w, *_ in ds: check(lib.ipoke_conv_wgrad(ctypes.byref(w), _lib.BF16, s))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): check(lib.ipoke_conv_wgrad(ctypes.byref(ds[i % NW][0]), _lib.BF16, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
c2 = mk(1, 0, hid, hid, hid, hid, 1, 0)
c3 = mk(3, 1, hid, 64, 64, hid * 9, 9, 1)
c1 = mk(3, 1, 32, hid, hid, 32 * 9, 9, 1)
print(f"B={B} pad={PAD}: conv2 {run(c2):.1f} us  conv3 {run(c3):.1f} us  conv1 {run(c1):.1f} us")
