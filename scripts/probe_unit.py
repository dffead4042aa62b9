"""Developer probe: isolated timing of the fused MaCowUnit kernels with cold (rotating) weight sets, and -- through the
stamped probe build scripts/exp/libunit_probe.so (hipcc -DIPOKE_UNIT_STAMPS mcf_unit.hip common.cpp) -- the shader-clock
time line of one workgroup."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib, ops
from ipoke_amd._lib import check

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NU = 24                                     # distinct units (4 layers each): 24 x 1.2 MB of weights > one XCD's L2
dt, tdt, dev = "bf16", torch.bfloat16, "cuda"
dm = ops.mcf_dims(C, 128, dt)
g = torch.Generator(device=dev).manual_seed(0)
M, ld = B * 64, 64
x = torch.randn(M, ld, device=dev, generator=g)
cond = torch.randn(M, 128, device=dev, generator=g).to(tdt)
dy = torch.randn(M, ld, device=dev, generator=g); dx = torch.empty_like(x)
dld = torch.randn(B, device=dev, generator=g)


def rnd(*s):
    return (torch.randn(*s, device=dev, generator=g) * 0.05).to(tdt)


W = [[dict(W1=rnd(dm["Hr"], dm["K1p"]), W2=rnd(dm["N2r"], dm["K2p"]), W1T=rnd(dm["Cr"], 6 * dm["Hq"]), W2T=rnd(dm["Hr"], dm["K3p"]))
      for _ in range(4)] for _ in range(NU)]
bias2 = torch.zeros(2 * C, device=dev); pls = torch.zeros(C, device=dev); pb = torch.zeros(C, device=dev)
ys = [torch.empty(M, ld, device=dev) for _ in range(4)]
a2 = [torch.empty(M, dm["K2p"], device=dev, dtype=tdt) for _ in range(4)]
sc = [torch.empty(M, C, device=dev) for _ in range(4)]
dps = [torch.empty(M, dm["K3p"], device=dev, dtype=tdt) for _ in range(4)]
dcs = [torch.empty(M, dm["Hq"], device=dev, dtype=tdt) for _ in range(4)]
dbp = [torch.empty(B, 2 * C, device=dev) for _ in range(4)]; pp = [torch.empty(B, 2 * C, device=dev) for _ in range(4)]
slot = torch.zeros(4, B, 4, device=dev)


def descs(u):
    d4 = (_lib.McfDesc * 4)()
    ins = [x, ys[0], ys[1], ys[2]]
    for k in range(4):
        d = d4[k]
        d.ld, d.C, d.B, d.cond, d.Cc, d.order, d.rows_per_block = ld, C, B, cond.data_ptr(), 128, k, 16
        w = W[u][k]
        d.W1, d.W2, d.W1T, d.W2T, d.bias2 = w["W1"].data_ptr(), w["W2"].data_ptr(), w["W1T"].data_ptr(), w["W2T"].data_ptr(), bias2.data_ptr()
        d.x = ins[k].data_ptr(); d.y = ys[k].data_ptr(); d.a2_save = a2[k].data_ptr(); d.scale_save = sc[k].data_ptr()
        d.logdet_slot = slot[k].data_ptr()
        d.dparams_save = dps[k].data_ptr(); d.dc_save = dcs[k].data_ptr(); d.dbias_part = dbp[k].data_ptr()
        if k in (1, 3):
            d.post_log_scale, d.post_bias, d.y_post, d.post_part = pls.data_ptr(), pb.data_ptr(), ys[k].data_ptr(), pp[k].data_ptr()
    d4[3].dy = dy.data_ptr(); d4[0].dx = dx.data_ptr(); d4[0].dld = dld.data_ptr()
    return d4


D = [descs(u) for u in range(NU)]
lib = _lib.lib(); s = _lib.current_stream()


def run(fn, n=240):
    for u in range(NU):
        check(fn(D[u], _lib.BF16, s))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        check(fn(D[i % NU], _lib.BF16, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"C={C} B={B}: fused unit fwd {run(lib.ipoke_macow_unit_fwd):.1f} us  bwd {run(lib.ipoke_macow_unit_bwd):.1f} us  (per unit = 4 layers)")

probe_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libunit_probe.so")
if os.path.exists(probe_path):
    P = ctypes.CDLL(probe_path)
    P.ipoke_macow_unit_fwd.argtypes = [ctypes.POINTER(_lib.McfDesc), ctypes.c_int, ctypes.c_void_p]
    P.ipoke_macow_unit_set_stamps.argtypes = [ctypes.c_void_p]
    st = torch.zeros(64, dtype=torch.int64, device=dev)
    P.ipoke_macow_unit_set_stamps(ctypes.c_void_p(st.data_ptr()))
    for u in range(NU):                                  # rotate so that the stamped launch sees cold weights
        assert P.ipoke_macow_unit_fwd(D[u], _lib.BF16, s) == 0
    torch.cuda.synchronize()
    t = st.cpu().tolist()
    names = ["entry(loads issued)", "staged"]
    for k in range(4):
        names += [f"L{k} gemm1", f"L{k} sync", f"L{k} a2save", f"L{k} gemm2", f"L{k} sync", f"L{k} epilogue+sum"]
    prev = t[0]
    for i, nme in enumerate(names):
        print(f"  {nme:24s} +{t[i] - prev:7d} cyc   (t = {t[i] - t[0]:7d})")
        prev = t[i]
    P.ipoke_macow_unit_bwd.argtypes = [ctypes.POINTER(_lib.McfDesc), ctypes.c_int, ctypes.c_void_p]
    st.zero_()
    for u in range(NU):
        assert P.ipoke_macow_unit_bwd(D[u], _lib.BF16, s) == 0
    torch.cuda.synchronize()
    t = st.cpu().tolist()
    names = ["entry(loads issued)"]
    for k in (3, 2, 1, 0):
        names += [f"L{k} top", f"L{k} (a)", f"L{k} sync", f"L{k} colsum+(b)", f"L{k} sync", f"L{k} (c)+2 syncs"]
    prev = t[0]
    print("backward:")
    for i, nme in enumerate(names):
        print(f"  {nme:24s} +{t[i] - prev:7d} cyc   (t = {t[i] - t[0]:7d})")
        prev = t[i]
    for q, k in enumerate((3, 2, 1, 0)):
        b = 32 + 4 * q
        print(f"  L{k} (c) detail: sync->pass lo {t[b] - t[5 + 6 * q]}, pass hi {t[b + 1] - t[b]}, w1t issue {t[b + 2] - t[b + 1]}, "
              f"exchange+2 syncs {t[b + 3] - t[b + 2]}")


# ---- inverse (sampling): x = unit^-1(y), two samples per workgroup, 4 layers x 8 strips
yinv = torch.randn(M, ld, device=dev, generator=g)
xinv = [torch.empty(M, ld, device=dev) for _ in range(4)]


def inv_descs(u):
    d4 = (_lib.McfDesc * 4)()
    for k in range(4):
        d = d4[k]
        d.ld, d.C, d.B, d.cond, d.Cc, d.order, d.rows_per_block = ld, C, B, cond.data_ptr(), 128, k, 16
        w = W[u][k]
        d.W1, d.W2, d.bias2 = w["W1"].data_ptr(), w["W2"].data_ptr(), bias2.data_ptr()
        if k in (1, 3):
            d.post_log_scale, d.post_bias = pls.data_ptr(), pb.data_ptr()
    d4[3].x = yinv.data_ptr(); d4[0].y = xinv[0].data_ptr()
    return d4


DI = [inv_descs(u) for u in range(NU)]
D, D_fb = DI, D
print(f"C={C} B={B}: fused unit inverse {run(lib.ipoke_macow_unit_inv):.1f} us")
if os.path.exists(probe_path):
    P.ipoke_macow_unit_inv.argtypes = [ctypes.POINTER(_lib.McfDesc), ctypes.c_int, ctypes.c_void_p]
    st.zero_()
    for u in range(NU):
        assert P.ipoke_macow_unit_inv(DI[u], _lib.BF16, s) == 0
    torch.cuda.synchronize()
    t = st.cpu().tolist()
    print(f"inverse: prologue done at t = 0; layer ends at {[t[49 + q] - t[0] for q in range(4)]}")
    names = ["top", "cond rows", "gemm1+ELU", "sync", "gemm2", "sync"]
    for step in range(8):
        row = [t[1 + 6 * step + q] for q in range(6)]
        nxt = t[1 + 6 * (step + 1)] if step < 7 else t[49]
        d = [row[q + 1] - row[q] for q in range(5)] + [nxt - row[5]]
        print(f"  layer D strip {step}: cond rows {d[0]:5d}  gemm1+ELU {d[1]:5d}  sync {d[2]:5d}  gemm2 {d[3]:5d}  sync {d[4]:5d}  coupling+sync {d[5]:5d}   total {nxt - row[0]:6d}")
D = D_fb


def contended(kind, fn, n=120):
    """Unit kernel timed while a second stream keeps the chip busy: 'hbm' = large device copies, 'mfma' = large bf16 GEMMs,
    'tn' = the weight-gradient GEMM of the coupling nets' conv2 (what the side stream runs during the backward pass)."""
    side = torch.cuda.Stream()
    big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
    ma = torch.randn(4096, 4096, device=dev, dtype=tdt); mb = torch.randn(4096, 4096, device=dev, dtype=tdt)
    ta = torch.randn(1280, 2048, device=dev, dtype=tdt); tb = torch.randn(1280, 2048, device=dev, dtype=tdt)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(60 if kind != "tn" else 400):
            if kind == "hbm":
                big_b.copy_(big_a)
            elif kind == "mfma":
                torch.mm(ma, mb)
            else:
                torch.mm(ta.t(), tb)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        check(fn(D[i % NU], _lib.BF16, s))
    e1.record(); e1.synchronize()
    busy = not side.query()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, busy


for kind in ("hbm", "mfma", "tn"):
    f, fb = contended(kind, lib.ipoke_macow_unit_fwd); b_, bb = contended(kind, lib.ipoke_macow_unit_bwd)
    print(f"next to a '{kind}' stream: fwd {f:.1f} us (side still busy at the end: {fb})  bwd {b_:.1f} us ({bb})")
