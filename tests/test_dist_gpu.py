"""Two data-parallel ranks on ONE MI355X over gloo (RCCL needs one GPU per rank; the test box has a single GPU): the
overlapped gradient exchange of SecondStageTrainer (backward in pieces; from the engine's gradient-ready callback either
reduce-scatter + sharded update + all-gather (ZeRO-1, default) or all-reduce + replicated update) must give the same
parameters as the plain exchange after the backward pass.  When two GPUs are visible the same runs use RCCL ("nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, overlap, out, env=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.update(env or {})
    from ipoke_amd import configs, dist as D
    from ipoke_amd.second_stage import PokeMotionModel
    from ipoke_amd.trainer import SecondStageTrainer
    from ipoke_amd.utils.detfill import deterministic_fill_
    two_gpus = torch.cuda.device_count() >= 2            # RCCL needs one GPU per rank
    if two_gpus:
        os.environ["LOCAL_RANK"] = str(rank)
    D.init_from_env(backend="nccl" if two_gpus else "gloo")
    torch.cuda.set_device(rank if two_gpus else 0)
    torch.manual_seed(1234 + rank)        # the encoder's reparameterisation noise comes from the CPU generator
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    model = PokeMotionModel(conf, dirs={}, dtype="f32", device=f"cuda:{torch.cuda.current_device()}", max_batch=2)
    for part, pfx in ((model.first_stage_model, "first_stage."), (model.poke_embedder, "poke_embedder."),
                      (model.conditioner, "conditioner."), (model.flow, "flow.")):
        deterministic_fill_(part, prefix=pfx)
    model.flow.sync_buffers()
    g = torch.Generator().manual_seed(10 + rank)                  # every rank its own micro-batch
    batch = {"images": (torch.rand(2, 16, 3, 64, 64, generator=g) * 2 - 1).cuda(),
             "flow": torch.randn(2, 2, 64, 64, generator=g).cuda(),
             "poke": [torch.zeros(2, 2, 64, 64).cuda(), torch.zeros(2, 5, 2, dtype=torch.int64).cuda()]}
    trainer = SecondStageTrainer(model, n_grad_buckets=3, overlap=overlap)
    assert trainer.overlap == overlap
    assert trainer.zero1 == (overlap and (env or {}).get("IPOKE_NO_ZERO1") != "1")
    trainer.sync_initial_state(batch)             # as bench.py does: init pass + broadcast of rank 0's parameters
    losses = [trainer.train_step(batch, i).item() for i in range(3)]
    torch.cuda.synchronize()
    p = model.flow.flat_params.detach().cpu()
    out[rank] = (losses, p.double().sum().item(), p.abs().double().sum().item(), p[::997].clone())
    D.barrier()
    dist.destroy_process_group()


_RUNS = {}


def _run(overlap, env=None):
    """Memoised per (overlap, env): the plain / default-overlapped runs are the baselines of several tests below."""
    key = (overlap, tuple(sorted((env or {}).items())))
    if key not in _RUNS:
        _RUNS[key] = _run_once(overlap, env)
    return _RUNS[key]


def _run_once(overlap, env=None):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), overlap, out, env), nprocs=2, join=True)
    return out[0], out[1]


@pytest.mark.parametrize("mode", ["zero1", "allreduce"])
def test_overlapped_gradient_exchange_matches_plain(mode):
    """plain = all-reduce of the flat buffer after the backward pass + replicated fused Adam.  zero1 (the default): per
    slice reduce-scatter -> sharded Adam-amsgrad (optimizer state sharded over the ranks) -> all-gather of the updated
    parameters, overlapped with the backward pass; allreduce: overlapped all-reduce + replicated update."""
    a0, a1 = _run(False)
    b0, b1 = _run(True, {"IPOKE_NO_ZERO1": "1"} if mode == "allreduce" else {})
    # both ranks hold identical parameters after every exchange ...
    assert torch.equal(a0[3], a1[3]) and torch.equal(b0[3], b1[3])
    # ... and overlapping the exchange with the backward pass does not change them (fp32 atomics in the split-K data
    # gradients leave run-to-run noise at the 1e-6 level, which Adam's m/sqrt(v) normalisation passes on to the update)
    scale = a0[3].abs().max().item()
    assert (a0[3] - b0[3]).abs().max().item() <= 1e-4 * scale
    print("losses plain", a0[0], "overlapped", b0[0])
    assert abs(a0[1] - b0[1]) <= 1e-6 * a0[2]
    # step 0 runs on identical parameters; later steps see Adam's amplification of the gradient noise (the update of a
    # parameter whose gradient is rounding noise is +-lr regardless of its size)
    assert abs(a0[0][0] - b0[0][0]) <= 1e-5 * max(1.0, abs(a0[0][0]))
    assert all(abs(x - y) <= 2e-5 * max(1.0, abs(x)) for x, y in zip(a0[0], b0[0]))


def test_bf16_gradient_exchange_stays_close():
    """IPOKE_GRAD_EXCHANGE=bf16: the reduce-scatter moves bf16 gradients (half the xGMI bytes); Adam normalises the
    update by sqrt(v), so three steps stay within a few 1e-3 of lr-sized updates of the fp32 exchange."""
    a0, a1 = _run(True)
    b0, b1 = _run(True, {"IPOKE_GRAD_EXCHANGE": "bf16"})
    assert torch.equal(b0[3], b1[3])
    assert all(abs(x - y) <= 1e-3 * max(1.0, abs(x)) for x, y in zip(a0[0], b0[0])), (a0[0], b0[0])
    assert (a0[3] - b0[3]).abs().max().item() <= 3 * 3 * 1e-3      # at most lr per step and parameter


def test_bench_gpus2_self_launches_over_gloo():
    """`python bench.py --gpus 2` with no launcher around it: the script re-executes itself under torch.distributed.run, two ranks share
    this box's one GPU over gloo (IPOKE_DIST_BACKEND / IPOKE_DIST_SINGLE_GPU: the only difference to the RCCL run is the backend string
    and the device index), rank 0 prints ONE JSON line whose value is the whole job's.  c1 = BASELINE configs[0] (plants_64, z = 32)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IPOKE_DIST_BACKEND="gloo", IPOKE_DIST_SINGLE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c1",
                          "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 * 16 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
