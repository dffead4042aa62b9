"""Developer probe: wall-clock (100 MHz) time line of back-to-back NT GEMM launches through the stamped probe build
scripts/exp/libgemm_probe.so (hipcc -DIPOKE_GEMM_STAMPS gemm.hip common.cpp): per launch, when the first/last workgroup
enters, has its first K-block, leaves the main loop, exits -- and the gap to the next launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref
from ipoke_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
pad = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kind = sys.argv[4] if len(sys.argv) > 4 else "conv2"      # conv2: 1x1 hid->hid;  conv3: 3x3 K -> 64 columns, split-K partials
N, M, dev = (2048 if kind == "conv2" else 64), B * 64, "cuda"
td = torch.bfloat16
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", os.environ.get("PROBE_LIB", "libgemm_probe.so")))
P.ipoke_conv_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
P.ipoke_gemm_set_stamps.argtypes = [ctypes.c_void_p]
lda, ldw, ldc = K + pad, K + pad, N + pad
NS, NL = 6, 24
sets = []
for i in range(NS):
    a = torch.randn(M, lda, device=dev).to(td)
    w = (torch.randn(N, ldw, device=dev) / K ** 0.5).to(td)
    if kind == "conv2":
        c = torch.empty(M, ldc, device=dev, dtype=td)
        d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0))
        d.A = a.data_ptr(); d.a_sn = 64 * lda; d.a_sh = 8 * lda; d.a_sw = lda; d.a_sc = 1; d.Kc_real = K; d.Kc = K
        d.W = w.data_ptr(); d.ldw = ldw; d.Nout = N; d.act = _lib.ACT_ELU; d.C = c.data_ptr(); d.ldc = ldc
    else:
        SK = int(os.environ.get("SPLITK", "20"))
        w = (torch.randn(N, 9 * K, device=dev) / K ** 0.5).to(td)
        c = torch.empty(SK, M, 64, device=dev, dtype=torch.float32)
        d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
        d.A = a.data_ptr(); d.a_sn = 64 * lda; d.a_sh = 8 * lda; d.a_sw = lda; d.a_sc = 1; d.Kc_real = K; d.Kc = K
        d.W = w.data_ptr(); d.ldw = 9 * K; d.Nout = N; d.act = 0; d.C = c.data_ptr(); d.ldc = 64; d.c_f32 = 1; d.splitk = SK
    sets.append((d, a, w, c))
dt, stream = ops._dt("bf16"), _lib.current_stream()
st = torch.zeros(NL, 4096, 4, dtype=torch.int64, device=dev)
for d, *_ in sets:
    assert P.ipoke_conv_forward(byref(d), dt, stream) == 0
torch.cuda.synchronize()
P.ipoke_gemm_set_stamps(ctypes.c_void_p(st.data_ptr()))
for i in range(NL):
    assert P.ipoke_conv_forward(byref(sets[i % NS][0]), dt, stream) == 0
torch.cuda.synchronize()
P.ipoke_gemm_set_stamps(None)
t = st.cpu()
nwg = int((t[0, :, 0] > 0).sum())
t = t[:, :nwg].double() * 10.0          # ns
print(f"M={M} N={N} K={K} pad={pad}: {nwg} workgroups; per launch (ns, relative to the first workgroup's entry)")
print("launch  entry(last)  first-block(min/med/max)   loop-end(min/med/max)   exit(min/med/max)   gap-to-next-entry  period")
for i in range(4, NL - 1):
    z = t[i, :, 0].min()
    r = lambda k: (t[i, :, k].min() - z, t[i, :, k].median() - z, t[i, :, k].max() - z)
    e, f, l, x = r(0), r(1), r(2), r(3)
    gap = t[i + 1, :, 0].min() - t[i, :, 3].max()
    per = t[i + 1, :, 0].min() - z
    print(f"{i:4d}   {e[2]:7.0f}     {f[0]:6.0f}/{f[1]:6.0f}/{f[2]:6.0f}       {l[0]:6.0f}/{l[1]:6.0f}/{l[2]:6.0f}    {x[0]:6.0f}/{x[1]:6.0f}/{x[2]:6.0f}   {gap:7.0f}   {per:7.0f}")
