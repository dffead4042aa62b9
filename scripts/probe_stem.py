"""Developer probe: conv1 of the 3-D motion encoder (B clips of 16 x 128 x 128) -- folded forward (inference and training entry points),
its weight gradient, and the in-place form.  Usage: python scripts/probe_stem.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import configs, first_stage as FS, first_stage_train as FT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = FS.SpadeCondMotionModel(configs.first_stage_config(128, 32, 16), dirs={}, train=False, dtype="bf16").cuda()
enc = m.enc_motion
x = torch.rand(B, 3, 16, 128, 128, device="cuda") * 2 - 1


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    print(f"inference _stem (clip_to_cl4 + folded conv): {timed(lambda: enc._stem(x)):.1f} us")
w = enc.conv1.weight
y = FT._StemFn.apply(x, w, "bf16")
dy = torch.randn_like(y)


def fwd_bwd():
    w.grad = None
    yy = FT._StemFn.apply(x, w, "bf16")
    yy.backward(dy)


print(f"training _StemFn forward + weight gradient: {timed(fwd_bwd):.1f} us")
with torch.no_grad():
    print(f"training _StemFn forward only: {timed(lambda: FT._StemFn.apply(x, w, 'bf16')):.1f} us")
