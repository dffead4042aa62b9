"""Picking a second stream that really runs beside the caller's.

HIP streams are multiplexed onto a few hardware queues (four by default); which queue a new stream lands on depends on how many
streams the process has created and used before (PyTorch's pools, the flow engine's side streams, RCCL).  Two streams that share a
queue serialise -- and with cross-stream waits between them the host loses its lead as well: measured on `sample_stream`, 80 ms per
batch on an unlucky stream against 41.5 ms on a lucky one (44.6 ms without any second stream).  `overlapping_stream` therefore tries
a few candidates with a spin kernel on each side and keeps the first that overlaps."""
import time

import torch


def _spin_pair_ms(main, other, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(main):
        torch.cuda._sleep(cycles)
    if other is not None:
        with torch.cuda.stream(other):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def overlapping_stream(candidates=8, spin_ms=0.5, report=None):
    """A torch.cuda.Stream whose work overlaps with the current stream's (checked with two spin kernels), or the best of
    ``candidates`` if none does.  ``report`` (a list) receives (candidate index, pair time / single time) per candidate tried."""
    main = torch.cuda.current_stream()
    cycles = 100_000
    _spin_pair_ms(main, None, cycles)                                   # warm: module load
    single = min(_spin_pair_ms(main, None, cycles) for _ in range(3))
    cycles = min(max(int(cycles * spin_ms / max(single, 1e-3)), 1000), 5_000_000)   # a spin of about spin_ms, whatever the counter's unit is (bounded)
    single = min(_spin_pair_ms(main, None, cycles) for _ in range(3))
    best, best_ratio = None, None
    for i in range(candidates):
        cand = torch.cuda.Stream()
        _spin_pair_ms(main, cand, cycles)                               # first use of the stream binds its hardware queue
        ratio = min(_spin_pair_ms(main, cand, cycles) for _ in range(3)) / single
        if report is not None:
            report.append((i, round(ratio, 3)))
        if best_ratio is None or ratio < best_ratio:
            best, best_ratio = cand, ratio
        if ratio < 1.3:
            break
    return best
