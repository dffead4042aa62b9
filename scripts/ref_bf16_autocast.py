"""How far does the REFERENCE itself move when its convolutions run in bf16 (torch.autocast on CPU)?  Same clip, same metrics as
tests/test_train_mode_gpu.py::grad_report.

    python scripts/ref_bf16_autocast.py [size] [T] [fixture.npz]

Needs /root/reference (build container only).  With a third argument the measured deviations are merged into that fixture under the
key prefix ``s<size>_T<T>_``: tests/golden/g15_ref_bf16_autocast.npz is the yardstick the bf16 bounds of tests/test_train_mode_gpu.py cite
(1.5 x the reference's own deviation), asserted by tests/test_bf16_yardstick_cpu.py."""
import sys, copy, zlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import ref_import, make_goldens as mg
from ipoke_amd import configs
from ipoke_amd.utils.detfill import deterministic_fill_
torch.set_num_threads(8)
SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
OUT = sys.argv[3] if len(sys.argv) > 3 else None
g = np.load('/root/repo/tests/golden/g13_first_stage_train_mode_128.npz')
fsm = ref_import.ref("models.first_stage_motion_model")
losses = ref_import.ref("utils.losses")
def run(autocast):
    cfg = configs.first_stage_config(SIZE, 32, T)
    m = fsm.SpadeCondMotionModel(copy.deepcopy(cfg), dirs={}, train=False)
    deterministic_fill_(m, prefix="first_stage.")
    m.train()
    X = torch.rand(1, T, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(int(131))) * 2 - 1
    torch.manual_seed(79)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            Xh, mu, lv = m(X)
        Xh, mu, lv = Xh.float(), mu.float(), lv.float()
    else:
        Xh, mu, lv = m(X)
    loss = 10 * (X[:, 1:] - Xh).abs().mean() + 1e-7 * losses.KL(mu, lv)
    loss.backward()
    return Xh.detach(), loss.item(), {k: p.grad.detach().double().flatten() for k, p in m.named_parameters() if p.grad is not None}
X0, l0, g0 = run(False)
X1, l1, g1 = run(True)
print("X_hat err max %.3e mean %.3e, loss %.6f vs %.6f" % ((X0 - X1).abs().max().item(), (X0 - X1).abs().mean().item(), l1, l0))
names = list(g0)
rows = []
for k in names:
    a, b = g0[k], g1[k]
    ref_abs = max(a.abs().sum().item(), 1e-12)
    wkey = k.replace(".bias", ".weight_orig")
    if k.endswith(".bias") and wkey in g0 and ref_abs <= 1e-4 * g0[wkey].abs().sum().item():
        continue
    idx = torch.randint(0, a.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(k.encode())))
    e_sum = max(abs(b.sum().item() - a.sum().item()), abs(b.abs().sum().item() - a.abs().sum().item())) / ref_abs
    e_smp = max(abs(b[i].item() - a[i].item()) for i in idx.tolist()) / max(b.abs().max().item(), 1e-12)
    e_max = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12)
    rows.append((k, e_sum, e_smp, e_max, a.numel()))
rows.sort(key=lambda r: -max(r[1], r[2]))
print("reference bf16-autocast vs reference fp32, %d tensors: worst sum %.3e, sampled max %.3e mean %.3e, worst max-element %.3e" % (
    len(rows), max(r[1] for r in rows), max(r[2] for r in rows), float(np.mean([r[2] for r in rows])), max(r[3] for r in rows)))
for r in rows[:25]:
    print("   %-50s sum %.3f smp %.3f maxel %.3f n=%d" % r)

if OUT:
    import os
    old = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    pfx = "s%d_T%d_" % (SIZE, T)
    old[pfx + "x_max"] = np.float64((X0 - X1).abs().max().item()); old[pfx + "x_mean"] = np.float64((X0 - X1).abs().mean().item())
    old[pfx + "loss_rel"] = np.float64(abs(l1 - l0) / max(1.0, abs(l0)))
    old[pfx + "sum"] = np.float64(max(r[1] for r in rows)); old[pfx + "smp_max"] = np.float64(max(r[2] for r in rows))
    old[pfx + "smp_mean"] = np.float64(np.mean([r[2] for r in rows])); old[pfx + "n_tensors"] = np.int64(len(rows))
    np.savez(OUT, **old)
    print("merged into", OUT)
