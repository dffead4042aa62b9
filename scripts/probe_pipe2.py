import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from ipoke_amd import configs
from ipoke_amd.utils import streams as S
dev = torch.device("cuda:0")
cfg = dict(configs.BENCH_CONFIGS["c5"])
B, T, size = cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"]
batch = bench.synthetic_batch(B, T, size, seed=1, device=dev)
m = bench.build_model(cfg, "bf16", dev)
with torch.no_grad():
    m.forward_density(batch)
bench.randomise_couplings(m)
for _ in range(3):
    m.forward_sample(batch)
main = torch.cuda.current_stream()
single = min(S._spin_pair_ms(main, None, 100000) for _ in range(3))
cyc = int(100000 * 4.0 / single)        # ~4 ms
print("spin cycles for 4 ms:", cyc, flush=True)
for prio in (0, -1):
    for k in range(8):
        st = torch.cuda.Stream(priority=prio)
        S._spin_pair_ms(main, st, 100000)
        sl = [round(S._waiter_slowdown(main, st, cyc), 2) for _ in range(2)]
        m._decode_stream = st
        ts = []
        for r_ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = sum(1 for _ in m.sample_stream([batch] * 8))
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
        print(f"torch stream prio {prio} #{k}: chain slowdown with a waiter {sl}; sample_stream {ts[1]:.1f} ms per batch", flush=True)
