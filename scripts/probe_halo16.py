"""Developer probe: the wide halo-staged 3x3(x3) convolution (conv3x3_halo16_kernel) against the kernels the dispatcher used before
(IPOKE_HALO16=0), at the shapes of the 3-D encoder (B = 20) and of the decoder / discriminator stacks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib, nn as K

DEV = "cuda"
SHAPES = [  # N, D, H, W, cin, cout, stride_d
    (20, 4, 64, 64, 128, 128, 1), (20, 8, 64, 64, 64, 128, 2), (20, 2, 32, 32, 256, 256, 1), (20, 4, 32, 32, 128, 256, 2),
    (20, 1, 16, 16, 256, 256, 1), (20, 1, 64, 64, 128, 128, 1), (20, 1, 32, 32, 256, 256, 1), (20, 1, 128, 128, 64, 64, 1),
    (32, 1, 64, 64, 128, 128, 1), (20, 1, 16, 16, 512, 512, 1),
]
def run(shape, n=20):
    N, D, H, W, cin, cout, sd = shape
    deep = D > 1 or sd > 1
    x = torch.randn(N * D * H * W, cin, device=DEV).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3 if deep else 1, 3, 3, device=DEV) / (cin * 27) ** 0.5
    wop, kc = K.weight_operand(w, "bf16")
    k, st, pd = ((3, 3, 3), (sd, 1, 1), (1, 1, 1)) if deep else ((1, 3, 3), (1, 1, 1), (0, 1, 1))
    f = lambda: K.conv(K.CL(x, N, (D, H, W), cin), wop, kc, cout, k, st, pd, "bf16")
    for _ in range(3): y = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    gf = 2e-9 * y.M * cout * cin * (27 if deep else 9)
    return us, gf / us * 1e-3, y
for shape in SHAPES:
    res = {}
    for mode in ("0", "2"):
        os.environ["IPOKE_HALO16"] = mode
        res[mode] = run(shape)
    d = (res["0"][2].t.float() - res["2"][2].t.float()).abs().max().item()
    print(f"{shape}: before {res['0'][0]:7.1f} us ({res['0'][1]:5.0f} TF/s)   halo16 {res['2'][0]:7.1f} us ({res['2'][1]:5.0f} TF/s)   max diff {d:.3g}")
