"""CPU, world_size 2 over gloo: the data-parallel plumbing (flat-gradient all-reduce in large buckets, parameter
broadcast after the data-dependent init, max-over-ranks timing) used by bench.py / SecondStageTrainer."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ipoke_amd import dist as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world
    n = 1003                                                   # not a multiple of the bucket count / of 4
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    D.allreduce_flat_(flat, n_buckets=8)
    expect = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok_sum = torch.equal(flat, expect)
    p = torch.full((17,), float(rank)).requires_grad_(True)      # parameter buffers require grad
    D.broadcast_(p, src=0)
    ok_bcast = bool((p == 0).all())
    mx = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    # fused-step semantics: the mean is applied as grad_scale = 1/world on the summed gradients
    g = torch.full((8,), 2.0 * (rank + 1)); D.allreduce_flat_(g, 2)
    ok_mean = torch.allclose(g / world, torch.full((8,), 3.0))
    # overlapped exchange: the flat buffer is reduced piece by piece (last levels first) with async handles, as
    # SecondStageTrainer does from the engine's gradient-ready callback
    flat2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    pieces = [(800, 1003), (400, 800), (96, 400), (0, 96)]
    handles = [D.allreduce_async(flat2[b:e]) for b, e in pieces]
    for h in handles:
        h.wait()
    ok_pieces = torch.equal(flat2, expect)
    # ZeRO-1 exchange of a slice (reduce-scatter -> update of the own shard -> all-gather, trailing elements all-reduced):
    # same parameters on every rank as all-reduce + replicated update, for slice lengths that do not divide
    ok_zero1 = True
    for nn in (1003, 64, 5, 4 * world, 4 * world + 3):
        g_loc = torch.arange(nn, dtype=torch.float32) * (rank + 1) + rank
        p0 = torch.linspace(-1, 1, nn)
        ref_g = g_loc.clone(); D.allreduce_flat_(ref_g, 1)
        ref_p = p0 - 0.1 * ref_g / world
        sh, main = D.shard_layout(nn, world)
        assert sh % 4 == 0 and main <= nn and nn - main < 4 * world
        p1 = p0.clone()
        if sh > 0:
            red = torch.empty(sh)
            D.reduce_scatter_async(red, g_loc[:main].clone()).wait()
            lo = rank * sh
            own = p1[lo:lo + sh] - 0.1 * red / world
            D.all_gather_async(p1[:main], own.clone()).wait()
        if main < nn:
            tail = g_loc[main:].clone()
            D.allreduce_async(tail).wait()
            p1[main:] -= 0.1 * tail / world
        ok_zero1 = ok_zero1 and torch.allclose(p1, ref_p, rtol=0, atol=1e-6)
    D.barrier()
    out[rank] = (ok_sum, ok_bcast, mx, ok_mean and ok_pieces and ok_zero1)
    dist.destroy_process_group()


def test_gloo_world2_allreduce_broadcast():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for rank in range(world):
        ok_sum, ok_bcast, mx, ok_mean = out[rank]
        assert ok_sum and ok_bcast and ok_mean and mx == 2.0


def test_single_process_is_a_noop():
    from ipoke_amd import dist as D
    t = torch.ones(5)
    assert D.allreduce_flat_(t) is t and D.world_size() == 1 and D.max_over_ranks(3.5, torch.device("cpu")) == 3.5
