import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from ipoke_amd import _lib, ops
C, B, NU = 64, 32, 24
dt, tdt, dev = "bf16", torch.bfloat16, "cuda"
dm = ops.mcf_dims(C, 128, dt)
g = torch.Generator(device=dev).manual_seed(0)
M, ld = B * 64, 64
cond = torch.randn(M, 128, device=dev, generator=g).to(tdt)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(tdt)
W = [[dict(W1=rnd(dm["Hr"], dm["K1p"]), W2=rnd(dm["N2r"], dm["K2p"])) for _ in range(4)] for _ in range(NU)]
bias2 = torch.zeros(2 * C, device=dev); pls = torch.zeros(C, device=dev); pb = torch.zeros(C, device=dev)
yinv = torch.randn(M, ld, device=dev, generator=g); xinv = torch.empty(M, ld, device=dev)
def descs(u):
    d4 = (_lib.McfDesc * 4)()
    for k in range(4):
        d = d4[k]
        d.ld, d.C, d.B, d.cond, d.Cc, d.order, d.rows_per_block = ld, C, B, cond.data_ptr(), 128, k, 16
        w = W[u][k]
        d.W1, d.W2, d.bias2 = w["W1"].data_ptr(), w["W2"].data_ptr(), bias2.data_ptr()
        if k in (1, 3):
            d.post_log_scale, d.post_bias = pls.data_ptr(), pb.data_ptr()
    d4[3].x = yinv.data_ptr(); d4[0].y = xinv.data_ptr()
    return d4
D = [descs(u) for u in range(NU)]
s = _lib.current_stream()
for path in sys.argv[1:]:
    P = ctypes.CDLL(path)
    P.ipoke_macow_unit_inv.argtypes = [ctypes.POINTER(_lib.McfDesc), ctypes.c_int, ctypes.c_void_p]
    for u in range(NU): assert P.ipoke_macow_unit_inv(D[u], _lib.BF16, s) == 0
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(240): P.ipoke_macow_unit_inv(D[i % NU], _lib.BF16, s)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 240 * 1e3)
    print(os.path.basename(path), " ".join(f"{t:.1f}" for t in res), "us per unit inverse")
