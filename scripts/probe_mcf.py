"""Developer probe: isolated timing of the MCF forward / backward kernels with cold (rotating) weight sets."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib, ops
from ipoke_amd._lib import check, ptr

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NW = 64
dt = "bf16"; tdt = torch.bfloat16
dm = ops.mcf_dims(C, 128, dt)
H = 4 * C
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = B * 64
x = torch.randn(M, 64, device=dev, generator=g); y = torch.empty_like(x)
dy = torch.randn(M, 64, device=dev, generator=g); dx = torch.empty_like(x)
cond = torch.randn(M, 128, device=dev, generator=g).to(tdt)
dld = torch.randn(B, device=dev, generator=g)
W1 = [(torch.randn(dm["Hr"], dm["K1p"], device=dev, generator=g) * 0.05).to(tdt) for _ in range(NW)]
W2 = [(torch.randn(dm["N2r"], dm["K2p"], device=dev, generator=g) * 0.05).to(tdt) for _ in range(NW)]
W1T = [(torch.randn(dm["Cr"], 6 * dm["Hq"], device=dev, generator=g) * 0.05).to(tdt) for _ in range(NW)]
W2T = [(torch.randn(dm["Hr"], dm["K3p"], device=dev, generator=g) * 0.05).to(tdt) for _ in range(NW)]
bias2 = torch.zeros(2 * C, device=dev)
a2 = torch.empty(M, dm["K2p"], device=dev, dtype=tdt); scale = torch.empty(M, C, device=dev)
dps = torch.empty(M, dm["K3p"], device=dev, dtype=tdt); dcs = torch.empty(M, dm["Hq"], device=dev, dtype=tdt)
dbp = torch.empty(B, 2 * C, device=dev); slot = torch.zeros(B, 4, device=dev)
lib = _lib.lib(); s = _lib.current_stream()

def desc(i):
    d = ops.mcf_desc(x, C, B, cond, W1[i], W2[i], bias2, i % 4)
    d.y = y.data_ptr(); d.a2_save = a2.data_ptr(); d.scale_save = scale.data_ptr(); d.logdet_slot = slot.data_ptr()
    d.rows_per_block = 16
    d.W1T = W1T[i].data_ptr(); d.W2T = W2T[i].data_ptr(); d.dy = dy.data_ptr(); d.dx = dx.data_ptr(); d.dld = dld.data_ptr()
    d.dparams_save = dps.data_ptr(); d.dc_save = dcs.data_ptr(); d.dbias_part = dbp.data_ptr()
    return d
descs = [desc(i) for i in range(NW)]
def run(fn, n=256):
    for i in range(NW): check(fn(ctypes.byref(descs[i]), _lib.BF16, s))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): check(fn(ctypes.byref(descs[i % NW]), _lib.BF16, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"C={C} B={B}: fwd {run(lib.ipoke_mcf_fwd):.1f} us  bwd {run(lib.ipoke_mcf_bwd):.1f} us")
