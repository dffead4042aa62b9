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
    ok_mean = torch.allclose(g / world, torch.full((8,), float(world + 1)))      # mean of 2 * (rank + 1) over the ranks
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


def test_gloo_world8_allreduce_broadcast():
    """the same worker at the world size of the MI355X node (8 ranks; arithmetic only, gloo on the CPU): bucketed
    all-reduce of a length that divides into neither 8 buckets nor 8 shards, broadcast, max-over-ranks, the async pieces and the
    ZeRO-1 exchange for slice lengths around the 4 * world granule (5, 32, 35, 64, 1003 floats)."""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for rank in range(world):
        ok_sum, ok_bcast, mx, ok_rest = out[rank]
        assert ok_sum and ok_bcast and ok_rest and mx == 8.0, rank


def _piece_ranges(z, npieces):
    from ctypes import byref, c_int64, c_void_p
    from ipoke_amd import _lib, configs
    lib = _lib.lib()
    arch = configs.flow_arch(z)
    cfg = _lib.FlowConfig()
    cfg.z_channels, cfg.hidden, cfg.cond_channels, cfg.factor = z, arch["flow_mid_channels"], 128, 16
    cfg.n_levels = len(arch["num_steps"])
    for i, st in enumerate(arch["num_steps"]):
        cfg.num_steps[i] = st
    cfg.kernel_h, cfg.kernel_w, cfg.dtype, cfg.max_batch = 2, 3, _lib.BF16, 20
    h = c_void_p()
    _lib.check(lib.ipoke_flow_create(byref(cfg), byref(h)))
    n = lib.ipoke_flow_piece_ranges(h, npieces, None, 0)
    assert n >= npieces
    buf = (c_int64 * (3 * n))()
    assert lib.ipoke_flow_piece_ranges(h, npieces, buf, n) == n
    total = lib.ipoke_flow_param_count(h)
    lib.ipoke_flow_destroy(h)
    return [(int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2])) for i in range(n)], int(total)


def test_shard_layout_of_the_real_slices_at_world8():
    """The slices the engine's piecewise backward announces for BOTH shipped flows (z = 64: 1 237 326 840 parameters, z = 32: 1 054 426 620;
    24 pieces = the trainer's default, and 16), cut for 8 ranks as FusedAdamAmsgrad.step_range_sharded cuts them: every slice begins 16-byte
    aligned, shards are multiples of 4 floats, the replicated tail is shorter than 4 * world floats, the ranks' shards and the tail tile
    the slice exactly, and the slices tile the flat parameter buffer."""
    from ipoke_amd import dist as D
    world = 8
    for z, expected, npieces in ((64, 1237326840, 24), (32, 1054426620, 24), (64, 1237326840, 16)):      # 24: the trainer's default
        ranges, total = _piece_ranges(z, npieces)
        assert expected <= total < expected + 4 * 6995          # the flat buffer pads every tensor to 16 bytes
        assert len({p for p, _, _ in ranges}) == npieces            # one or two regions (layers.*, priors.*) per piece
        assert [p for p, _, _ in ranges] == sorted(p for p, _, _ in ranges)
        covered = 0
        spans = sorted((b, e) for _, b, e in ranges)
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
            assert e0 == b1, "slices must tile the flat buffer without gaps or overlap"
        tails = []
        for _, b, e in ranges:
            n = e - b
            assert b % 4 == 0 and n > 0
            sh, main = D.shard_layout(n, world)
            assert sh % 4 == 0 and main == sh * world and 0 <= n - main < 4 * world
            own = [(b + r * sh, b + (r + 1) * sh) for r in range(world)]
            assert own[0][0] == b and own[-1][1] == b + main and all(lo % 4 == 0 for lo, _ in own)
            covered += main + (n - main)
            tails.append(n - main)
        assert covered == total
        # the exchange is dominated by the shards: the replicated tails are a few floats per slice
        assert sum(tails) < 4 * world * len(ranges)
        per_rank_state = sum(D.shard_layout(e - b, world)[0] for _, b, e in ranges) + sum(tails)
        assert abs(per_rank_state - total / world) <= 4 * world * len(ranges)


def test_bench_self_launch_command_for_8_gpus():
    """`python bench.py --gpus 8` outside a torchrun environment re-executes itself as the launch line of the driver's contract."""
    import bench
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "5", "--warmup", "2"], 8, {}, script="/root/repo/bench.py", port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29511",
                       "/root/repo/bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert bench.self_launch_command(["--gpus", "8"], 8, {"WORLD_SIZE": "8"}) is None          # already a rank
    assert bench.self_launch_command([], 1, {}) is None


def test_single_process_is_a_noop():
    from ipoke_amd import dist as D
    t = torch.ones(5)
    assert D.allreduce_flat_(t) is t and D.world_size() == 1 and D.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def _fake_opt(numel, world=None, rank=0):
    """FusedAdamAmsgrad over a stand-in flow (host tensors; only the state bookkeeping is exercised, no kernel runs)."""
    import types
    from ipoke_amd.optim import FusedAdamAmsgrad
    flow = types.SimpleNamespace(flat_params=torch.zeros(numel, requires_grad=True))
    opt = FusedAdamAmsgrad(flow, lr=1e-3, weight_decay=1e-5)
    if world is not None:
        opt.enable_sharding(world, rank)
    return opt


def test_sharded_optimizer_state_roundtrip_and_conversion():
    """ZeRO-1 optimizer state: per-rank save / load round trip, layout validation (a changed slice layout raises instead of
    zeroing the moments), replicated -> sharded (shards cut out of the full tensors) and sharded -> replicated (merge of the
    ranks' files)."""
    import pytest
    from ipoke_amd import dist as D
    from ipoke_amd.optim import FusedAdamAmsgrad
    numel, world = 1003, 2
    pieces = [(600, 1003), (0, 600)]
    full = [torch.arange(numel, dtype=torch.float32) + 1000 * k for k in range(3)]
    states = []
    for rank in range(world):
        opt = _fake_opt(numel, world, rank)
        for b, e in pieces:                                     # what step_range_sharded touches
            sh, main = D.shard_layout(e - b, world)
            lo = b + rank * sh
            for st, f in zip(opt._shard_state(b, lo, sh), full):
                st.copy_(f[lo:lo + sh])
            if main < e - b:
                for st, f in zip(opt._shard_state(-(b + 1), b + main, e - b - main), full):
                    st.copy_(f[b + main:e])
        opt.steps = 7
        sd = opt.state_dict()
        assert sd["sharded"] and sd["layout"][0][0] == rank * D.shard_layout(600, world)[0]
        states.append(sd)
        # round trip into a fresh optimizer of the same (world, rank)
        opt2 = _fake_opt(numel, world, rank)
        opt2.load_state_dict(sd)
        assert opt2.steps == 7
        b, e = pieces[0]
        sh, _ = D.shard_layout(e - b, world)
        got = opt2._shard_state(b, b + rank * sh, sh)
        assert torch.equal(got[1], full[1][b + rank * sh:b + rank * sh + sh])
        with pytest.raises(RuntimeError):                       # other slice layout: must not silently reset the moments
            opt2._shard_state(b, b + rank * sh, sh - 4)
        with pytest.raises(ValueError):                         # other rank
            _fake_opt(numel, world, 1 - rank).load_state_dict(sd)
    # sharded -> replicated
    merged = FusedAdamAmsgrad.merge_sharded(states, numel)
    for k, f in zip(("exp_avg", "exp_avg_sq", "max_exp_avg_sq"), full):
        assert torch.equal(merged[k], f)
    rep = _fake_opt(numel)
    rep.load_state_dict(merged)
    assert rep.steps == 7 and torch.equal(rep.max_exp_avg_sq, full[2])
    with pytest.raises(ValueError):
        rep.load_state_dict(states[0])
    with pytest.raises(ValueError):
        FusedAdamAmsgrad.merge_sharded(states[:1], numel)        # a rank's file is missing
    # replicated -> sharded: the shards are cut out of the full tensors on first use
    for rank in range(world):
        opt = _fake_opt(numel, world, rank)
        opt.load_state_dict(rep.state_dict())
        b, e = pieces[1]
        sh, main = D.shard_layout(e - b, world)
        lo = b + rank * sh
        m, v, vmax = opt._shard_state(b, lo, sh)
        assert torch.equal(m, full[0][lo:lo + sh]) and torch.equal(vmax, full[2][lo:lo + sh])
    # single-process full_state_dict of a sharded optimizer (world 1) equals the replicated layout
    one = _fake_opt(numel, 1, 0)
    one.load_state_dict(merged)
    for b, e in pieces:
        sh, main = D.shard_layout(e - b, 1)
        one._shard_state(b, b, sh)
        if main < e - b:
            one._shard_state(-(b + 1), b + main, e - b - main)
    fsd = one.full_state_dict()
    assert torch.equal(fsd["exp_avg_sq"], full[1])


def test_bench_self_launch_decision():
    """`python bench.py --gpus N` as the driver runs it (no torchrun environment) must turn itself into the N-rank job of the contract;
    a process that already is a rank, N = 1 and the CPU-baseline child must not relaunch."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.self_launch_command(argv, 8, {}, script="/x/bench.py", port=29123)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29123"
    assert cmd[-len(argv) - 1:] == ["/x/bench.py"] + argv                       # the script and ITS arguments, unchanged, last
    auto = bench.self_launch_command(argv, 2, {})
    assert 1024 <= int(auto[auto.index("--master-port") + 1]) < 65536 and auto[-len(argv) - 1].endswith("bench.py")
    assert bench.self_launch_command(argv, 8, {"WORLD_SIZE": "8", "RANK": "3"}) is None      # already a rank of a launched job
    assert bench.self_launch_command(["--gpus", "1"], 1, {}) is None
    assert bench.self_launch_command(["--cpu-baseline-only", "--gpus", "2"], 2, {}) is None
