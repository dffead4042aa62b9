"""Developer probe: shader-clock stamps of the LAST fused-unit backward launch of a c2 train step, taken INSIDE the step (side streams
running).  Needs a full library built with -DIPOKE_UNIT_STAMPS for mcf_unit.hip / mcf_unit_split.hip, loaded through IPOKE_LIB_PATH."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from ipoke_amd import _lib, configs
from ipoke_amd.trainer import SecondStageTrainer
lib = ctypes.CDLL(os.environ["IPOKE_LIB_PATH"])
lib.ipoke_macow_unit_set_stamps.argtypes = [ctypes.c_void_p]
cfg = dict(configs.BENCH_CONFIGS["c2"])
B, T, size = cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"]
model = bench.build_model(cfg, "bf16", torch.device("cuda", 0))
trainer = SecondStageTrainer(model)
batch = bench.synthetic_batch(B, T, size, seed=1, device="cuda")
trainer.sync_initial_state(batch)
bench.randomise_couplings(model)
for i in range(4):
    trainer.train_step(batch, i, next_batch=batch)
torch.cuda.synchronize()
st = torch.zeros(256, dtype=torch.int64, device="cuda")
lib.ipoke_macow_unit_set_stamps(ctypes.c_void_p(st.data_ptr()))
names = ["entry", "staged"]
for k in (3, 2, 1, 0):
    names += [f"L{k} (a)", f"L{k} colsum+(b)+publish", f"L{k} sync", f"L{k} indep tile/halo req", f"L{k} halo+sync", f"L{k} pass c",
              f"L{k} w1t issue+exchange+2 syncs", "(unused slot)"]
acc = None
N = 6
for rep in range(N):
    trainer.train_step(batch, 4 + rep, next_batch=batch)
    torch.cuda.synchronize()
    t = st.cpu().view(4, 64)[:, :len(names)].double()
    for q in range(4):                       # slot 9 + 8 q is not stamped: carry the previous stamp so that the deltas stay aligned
        t[:, 9 + 8 * q] = t[:, 8 + 8 * q]
    d = t[:, 1:] - t[:, :-1]
    acc = d if acc is None else acc + d
    tot = (t[:, 32] - t[:, 0])
    print(f"step {rep}: last unit backward, cycles entry -> end per workgroup 0..3: {[int(x) for x in tot.tolist()]}")
acc /= N
print("mean cycles per phase (workgroups 0..3 of the last unit-backward launch of the step):")
for i, nme in enumerate(names[1:]):
    print(f"  {nme:28s} " + "  ".join(f"{int(acc[w, i]):7d}" for w in range(4)))
