"""End-to-end time of one epoch's FVD (utils/metrics.py:774-781 as validation_epoch_end calls it): 1 000 generated and 1 000 original
16-frame 128x128 clips resident on the GPU -> resize + I3D (fp32, batch 8) + float64 moments on the device + sqrtm on the host."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipoke_amd import fvd  # noqa: E402
from ipoke_amd.utils.detfill import deterministic_fill_  # noqa: E402

net = fvd.I3D(400, "rgb", dtype=sys.argv[1] if len(sys.argv) > 1 else "f32", device="cpu")
deterministic_fill_(net, prefix="i3d.")
net.to("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
orig = torch.rand(1000, 16, 3, 128, 128, device="cuda", generator=g) * 2 - 1
gen = (orig + 0.3 * torch.randn(1000, 16, 3, 128, 128, device="cuda", generator=g)).clamp(-1, 1)
fvd.calculate_FVD(net, gen[:16], orig[:16], batch_size=8)            # warm-up (operands, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
acts = fvd._activations(net, gen, 8, fvd._resized_min(gen), resize=(224, 224))
torch.cuda.synchronize()
t1 = time.perf_counter()
val = fvd.calculate_FVD(net, gen, orig, batch_size=8)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"I3D logits of 1 000 clips: {t1 - t0:.2f} s ({16000 / (t1 - t0):.0f} frames/s); calculate_FVD(1 000 vs 1 000) end to end: {t2 - t1:.2f} s; FVD = {val:.3f}")
