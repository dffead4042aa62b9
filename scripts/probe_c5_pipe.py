"""Developer probe: host issue time vs device time of the two halves of a sampling step (c5), and what two streams overlap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ipoke_amd import configs

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
X, poke = batch["images"], m._poke_of(batch)
z = torch.randn(B, 64, 8, 8).to(dev)
sync = torch.cuda.synchronize
N = 10

def timed(fn, n=N):
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    th = time.perf_counter() - t0
    sync(); tt = time.perf_counter() - t0
    return th / n * 1e3, tt / n * 1e3

with torch.no_grad():
    motion = m._sample_motion(X, poke, z)
    print("stage 1 (encoders + reverse flow): host %.2f ms, device-complete %.2f ms per call" % timed(lambda: m._sample_motion(X, poke, z)))
    print("stage 2 (ConvGRU + decode):        host %.2f ms, device-complete %.2f ms per call" % timed(lambda: m.decode_first_stage(motion, X)))
    print("both, one stream:                  host %.2f ms, device-complete %.2f ms per call" % timed(lambda: m.decode_first_stage(m._sample_motion(X, poke, z), X)))
    side = torch.cuda.Stream()
    def both_two_streams():
        mo = m._sample_motion(X, poke, z)
        side.wait_stream(torch.cuda.current_stream())
        mo.record_stream(side)
        with torch.cuda.stream(side):
            return m.decode_first_stage(mo, X)
    both_two_streams(); sync()
    print("both, decode on a second stream:   host %.2f ms, device-complete %.2f ms per call" % timed(both_two_streams))
    # independent halves on two streams, no dependency at all: the overlap the device offers
    def indep():
        with torch.cuda.stream(side):
            m.decode_first_stage(motion, X)
        m._sample_motion(X, poke, z)
    indep(); sync()
    print("independent halves on two streams: host %.2f ms, device-complete %.2f ms per call" % timed(indep))
    try:
        print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
    except Exception as e:
        print("priority_range:", e)
    for pr_side, pr_main in ((0, -1), (1, 0), (1, -1), (0, 0)):
        try:
            side2 = torch.cuda.Stream(priority=pr_side)
            main2 = torch.cuda.Stream(priority=pr_main)
        except Exception as e:
            print("streams", pr_side, pr_main, e); continue
        def indep2():
            with torch.cuda.stream(side2):
                m.decode_first_stage(motion, X)
            with torch.cuda.stream(main2):
                m._sample_motion(X, poke, z)
        indep2(); sync()
        print("independent halves, decode prio %d, flow prio %d (own streams): host %.2f ms, device-complete %.2f ms" % ((pr_side, pr_main) + timed(indep2)))
from ipoke_amd.utils.streams import overlapping_stream
rep_ = []; overlapping_stream(report=rep_); print("candidates (index, pair/single):", rep_)
for rep in range(4):
    if rep == 2:
        m._decode_stream = torch.cuda.Stream(); print("decode stream := a new torch.cuda.Stream()")
    sync(); t0 = time.perf_counter()
    n = sum(1 for _ in m.sample_stream([batch] * 20))
    sync(); print("sample_stream: %.2f ms per batch" % ((time.perf_counter() - t0) / n * 1e3))
for rep in range(2):
    sync(); t0 = time.perf_counter()
    for _ in range(20):
        m.forward_sample(batch)
    sync(); print("forward_sample: %.2f ms per batch" % ((time.perf_counter() - t0) / 20 * 1e3))
