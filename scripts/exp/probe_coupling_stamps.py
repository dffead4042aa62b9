"""Developer probe: wall-clock (100 MHz) time line of the fused conv3 + coupling launch (ipoke_conv3x3_coupling) through the stamped
probe build scripts/exp/ab/libgemm_probe.so (hipcc -DIPOKE_GEMM_STAMPS -shared gemm.hip common.cpp): per workgroup
0 entry, 1 first K-block, 2 K loop done, 4 partial rows sent, 5 partners' rows arrived (owners), 6 sums staged, 3 exit.
usage: python scripts/exp/probe_coupling_stamps.py [B] [K] [mode]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ctypes import byref
from ipoke_amd import _lib, ops
from ipoke_amd._lib import AffineDesc, CouplingEpi, ConvDesc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
Cp, ld, M, dev = 32, 64, B * 64, "cuda"
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ.get("PROBE_LIB", "libgemm_probe.so")))
P.ipoke_conv3x3_coupling.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
P.ipoke_conv_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
P.ipoke_gemm_set_stamps.argtypes = [ctypes.c_void_p]
P.ipoke_conv3x3_coupling_xchg_bytes.restype = ctypes.c_int64
P.ipoke_conv3x3_coupling_xchg_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
P.ipoke_conv3x3_coupling_splitk.argtypes = [ctypes.c_int] * 3
ns = P.ipoke_conv3x3_coupling_splitk(M, K, _lib.BF16)
stream = _lib.current_stream()
xchg = torch.empty(P.ipoke_conv3x3_coupling_xchg_bytes(), dtype=torch.uint8, device=dev)
assert P.ipoke_conv3x3_coupling_xchg_init(xchg.data_ptr(), stream) == 0
NS, NL = 6, 24
sets = []
for i in range(NS):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(2 * Cp, 9 * K, device=dev) / (9 * K) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(2 * Cp, device=dev) * 0.1
    x = torch.randn(M, ld, device=dev); y = torch.empty(M, ld, device=dev); y2 = torch.empty(M, ld, device=dev)
    sc = torch.empty(M, Cp, device=dev); slots = torch.empty(B, 4, device=dev)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = a.data_ptr(); d.a_sn = 64 * K; d.a_sh = 8 * K; d.a_sw = K; d.a_sc = 1; d.Kc_real = K; d.Kc = K
    d.W = w.data_ptr(); d.ldw = 9 * K; d.Nout = 2 * Cp
    af = AffineDesc(); af.bias = bias.data_ptr(); af.Cp = Cp; af.t_off = 0; af.t_stride = 2; af.P = 64; af.ld = ld
    e = CouplingEpi(); e.mode = mode; e.inp = x.data_ptr(); e.out = y.data_ptr(); e.xchg = xchg.data_ptr()
    if mode != 2:
        e.scale_out = sc.data_ptr(); e.logdet_slot = slots.data_ptr(); e.slot_stride = 4
    if mode == 1:
        e.out2 = y2.data_ptr(); e.an_c0 = 0; e.an_C = ld
    sets.append((d, af, e, a, w, bias, x, y, y2, sc, slots))
st = torch.zeros(NL, 4096, 16, dtype=torch.int64, device=dev)
for s_ in sets:
    assert P.ipoke_conv3x3_coupling(byref(s_[0]), byref(s_[1]), byref(s_[2]), B, _lib.BF16, stream) == 0
torch.cuda.synchronize()
P.ipoke_gemm_set_stamps(ctypes.c_void_p(st.data_ptr()))
for i in range(NL):
    s_ = sets[i % NS]
    assert P.ipoke_conv3x3_coupling(byref(s_[0]), byref(s_[1]), byref(s_[2]), B, _lib.BF16, stream) == 0
torch.cuda.synchronize()
P.ipoke_gemm_set_stamps(None)
assert int(xchg[:4].view(torch.int32).item()) == 0, "hand-off time-outs"
t = st.cpu()
nwg = int((t[0, :, 0] > 0).sum())
t = t[:, :nwg].double() * 10.0          # ns
own = (torch.arange(nwg) % ns) < min(ns, 8)
print(f"B={B} K={K} mode={mode}: {nwg} workgroups, {ns} splits, {int(own.sum())} owners; ns relative to the first workgroup's entry (min/med/max)")
print("median over workgroups (owners from 'sent' on), ns from the first workgroup's entry:")
print("launch  first-blk loop-end   sent  polled-1st arrived  tree  staged  copied transformed logdet  exit | max exit, period")
for i in range(4, NL - 1):
    z = t[i, :, 0].min()
    med = lambda k, m=None: float((t[i, :, k] if m is None else t[i, m, k]).median() - z)
    per = t[i + 1, :, 0].min() - z
    print(f"{i:4d}   {med(1):7.0f} {med(2):7.0f} {med(4, own):7.0f} {med(7, own):7.0f} {med(5, own):7.0f} {med(8, own):7.0f} {med(6, own):7.0f} {med(9, own):7.0f} "
          f"{med(10, own):7.0f} {med(11, own):7.0f} {med(3, own):7.0f} | {float(t[i, own, 3].max() - z):7.0f} {per:7.0f}")
