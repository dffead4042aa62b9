"""Picking a second stream that really runs beside the caller's.

HIP streams are multiplexed onto a few hardware queues (four by default); which queue a new stream lands on depends on how many
streams the process has created before (PyTorch's pools hand them out in order, the flow engine creates five, RCCL its own).  Measured
on an MI355X in a sampling process (`scripts/probe_pipe2.py`), for consecutive streams of PyTorch's pool, period four:

* two of four run beside the caller's stream (two spin kernels take 1.1-1.2x the time of one);
* one shares the caller's hardware queue (1.9-2.0x: no overlap at all, harmless otherwise);
* one overlaps on an idle chip but is poison as soon as it WAITS for an event of the caller's stream: while its wait is pending, a
  chain of small launches on the caller's stream runs 3-15x slower (1.9x beside a waiting stream of the first kind -- a pending
  cross-stream wait is never free).  `sample_stream` with its decode stream parked in such a wait for the whole reverse flow ran at
  80 ms per batch instead of 41.5; it now waits on the host instead, and this module keeps such streams out of the way.

`overlapping_stream` tries a few candidates on an idle device and keeps the first that passes both checks."""
import time

import os

import torch

_SMALL = {}


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


def _chain_ms_with_waiter(main, other, spin_cycles):
    """Device time of a chain of 300 small launches on ``main`` while ``other`` (if given) sits in a wait for an event recorded behind
    that chain."""
    dev = torch.cuda.current_device()
    if dev not in _SMALL:
        _SMALL[dev] = torch.zeros(1024, dtype=torch.float32, device="cuda")
    small = _SMALL[dev]
    torch.cuda.synchronize()
    c0, c1, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    with torch.cuda.stream(main):
        torch.cuda._sleep(spin_cycles)              # the host queues everything below while this runs
        c0.record()
        for _ in range(300):
            small.add_(1.0)
        c1.record()
        done.record()
    if other is not None:
        other.wait_event(done)
        with torch.cuda.stream(other):
            small.add_(0.0)
    torch.cuda.synchronize()
    return c0.elapsed_time(c1)


def _waiter_slowdown(main, other, spin_cycles):
    _chain_ms_with_waiter(main, other, spin_cycles)
    alone = min(_chain_ms_with_waiter(main, None, spin_cycles) for _ in range(2))
    return min(_chain_ms_with_waiter(main, other, spin_cycles) for _ in range(2)) / alone


def overlapping_stream(candidates=8, spin_ms=0.5, report=None):
    """A torch.cuda.Stream whose work overlaps with the current stream's and whose pending waits do not stall it, or the best of
    ``candidates`` if none passes.  ``report`` (a list) receives (candidate index, spin pair / single, chain slowdown beside the
    waiting candidate) per candidate tried."""
    # IPOKE_SIDE_STREAM=plain: no probing at all -- a fresh stream of PyTorch's pool (multi-rank jobs and shared GPUs: the probe below
    # times spin kernels on what it assumes to be an idle device, and its device-wide synchronisations are not free there)
    if os.environ.get("IPOKE_SIDE_STREAM", "") == "plain":
        return torch.cuda.Stream()
    main = torch.cuda.current_stream()
    cycles = 100_000
    _spin_pair_ms(main, None, cycles)                                   # warm: module load
    single = min(_spin_pair_ms(main, None, cycles) for _ in range(3))
    per_ms = cycles / max(single, 1e-3)                                 # spin cycles per millisecond, whatever the counter's unit is
    cycles = min(max(int(per_ms * spin_ms), 1000), 5_000_000)
    hold = min(max(int(per_ms * 4.0), 1000), 40_000_000)                # ~4 ms: long enough to queue the chain behind it
    single = min(_spin_pair_ms(main, None, cycles) for _ in range(3))
    best, best_score = None, None
    for i in range(candidates):
        cand = torch.cuda.Stream()
        _spin_pair_ms(main, cand, cycles)                               # first use of the stream binds its hardware queue
        ratio = min(_spin_pair_ms(main, cand, cycles) for _ in range(3)) / single
        slow = _waiter_slowdown(main, cand, hold) if ratio < 1.3 else float("inf")
        if report is not None:
            report.append((i, round(ratio, 3), round(slow, 2) if slow != float("inf") else None))
        score = ratio + (slow if slow != float("inf") else 100.0)
        if best_score is None or score < best_score:
            best, best_score = cand, score
        if ratio < 1.3 and slow < 2.5:
            return cand
    # no candidate passed both checks (a busy device makes the timings noise): say so instead of silently returning the least bad one
    import warnings
    warnings.warn(f"overlapping_stream: none of {candidates} streams passed the overlap / stalled-launch checks on this device "
                  f"(best score {best_score:.2f}); using the best candidate -- set IPOKE_SIDE_STREAM=plain to skip the probe", RuntimeWarning)
    return best


def _pair_ratio(a, b, cycles, single):
    """(time of one spin kernel on ``a`` and one on ``b``) / (time of one): ~1.0-1.2 when the two streams sit on different hardware
    queues, ~2.0 when they share one."""
    _spin_pair_ms(a, b, cycles)                                         # first use binds the queue
    return min(_spin_pair_ms(a, b, cycles) for _ in range(3)) / single


def distinct_streams(n, against=(), candidates=32, spin_ms=0.4, report=None):
    """``n`` streams of PyTorch's pool that run BESIDE each other and beside every stream in ``against`` (the caller's stream, the
    flow engine's weight-gradient stream, ...): each accepted stream passes a two-kernel overlap check against all of them.  The train
    step keeps four streams busy at once; two of them on one hardware queue serialise (measured: a busy stream on the chain's queue
    doubles the step, 111.8 vs 49.5 ms), and which queue a stream of the pool lands on depends on every stream the process -- PyTorch,
    this library, RCCL -- has created before.  Falls back to the best candidates with a warning when fewer than ``n`` pass (a busy or
    shared device makes the timings noise); IPOKE_SIDE_STREAM=plain skips the probe."""
    # (several ranks sharing one GPU -- the single-GPU test mode of the data-parallel path -- would time each other's kernels)
    if os.environ.get("IPOKE_SIDE_STREAM", "") == "plain" or os.environ.get("IPOKE_DIST_SINGLE_GPU") == "1" or n <= 0:
        return [torch.cuda.Stream() for _ in range(n)]
    main = torch.cuda.current_stream()
    cycles = 100_000
    _spin_pair_ms(main, None, cycles)
    single = min(_spin_pair_ms(main, None, cycles) for _ in range(3))
    per_ms = cycles / max(single, 1e-3)
    cycles = min(max(int(per_ms * spin_ms), 1000), 5_000_000)
    hold = min(max(int(per_ms * 4.0), 1000), 40_000_000)
    single = min(_spin_pair_ms(main, None, cycles) for _ in range(3))
    chosen, rejected = [], []
    fixed = list(against)
    for i in range(candidates):
        cand = torch.cuda.Stream()
        if any(cand.cuda_stream == s_.cuda_stream for s_ in fixed + chosen):
            continue
        worst = max(_pair_ratio(o, cand, cycles, single) for o in fixed + chosen) if fixed or chosen else 1.0
        # ... and a pending wait of the candidate for an event of the caller's stream must not stall that stream's launches (see above)
        slow = _waiter_slowdown(main, cand, hold) if worst < 1.35 else float("inf")
        if report is not None:
            report.append((i, round(worst, 3), None if slow == float("inf") else round(slow, 2)))
        if worst < 1.35 and slow < 2.5:
            chosen.append(cand)
            if len(chosen) == n:
                return chosen
        else:
            rejected.append((worst + (0.0 if slow == float("inf") else slow), cand))
    import warnings
    warnings.warn(f"distinct_streams: only {len(chosen)} of {n} streams passed the overlap check against {len(fixed)} busy streams "
                  "(is the device shared or busy?); filling up with the least overlapping candidates -- set IPOKE_SIDE_STREAM=plain to "
                  "skip the probe", RuntimeWarning)
    rejected.sort(key=lambda t_: t_[0])
    while len(chosen) < n:
        chosen.append(rejected.pop(0)[1] if rejected else torch.cuda.Stream())
    return chosen
