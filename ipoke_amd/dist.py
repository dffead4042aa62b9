"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) / gloo on CPU.

The flow's gradients live in ONE flat fp32 buffer, so the gradient exchange of a step (N1 in SURVEY.md: 4.2-4.95 GB)
is a handful of large collectives over contiguous slices instead of DDP's ~200 25-MB buckets; xGMI is
point-to-point, large messages keep every link busy.  Default exchange (ipoke_amd.trainer, ipoke_amd.optim): per slice a
reduce-scatter, the fused Adam-amsgrad update of this rank's 1/world shard (optimizer state sharded, ZeRO-1), an all-gather
of the updated parameters; the mean is folded into the update (grad_scale).  The plain all-reduce + replicated update is
kept as the reference path (IPOKE_NO_ZERO1=1) and for the non-overlapped mode.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # IPOKE_DIST_BACKEND=gloo: test hook -- several ranks sharing one GPU (RCCL needs one GPU per rank)
            backend = os.environ.get("IPOKE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("IPOKE_DIST_SINGLE_GPU") == "1":
            local = 0
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def allreduce_flat_(flat, n_buckets=8):
    """Sum ``flat`` (1-D tensor) over all ranks in ``n_buckets`` large, 16-byte aligned slices issued back to back."""
    if world_size() == 1:
        return flat
    n = flat.numel()
    step = -(-n // n_buckets)
    step = -(-step // 4) * 4
    works = []
    for lo in range(0, n, step):
        works.append(dist.all_reduce(flat[lo:min(n, lo + step)], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return flat


def allreduce_(t):
    """In-place sum of ``t`` over all ranks (blocking on the stream's timeline; identity in single-process runs)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


# Stream budget: the train step keeps FOUR streams busy -- chain, weight gradients, ready / optimizer, encoder prefetch --
# and a further busy stream costs +18 ms per step, or 2x when its hardware queue is the chain's (HIP multiplexes streams onto
# GPU_MAX_HW_QUEUES queues; tests/test_dist_gpu.py).  With ``async_op=True`` ProcessGroupNCCL runs every collective on an INTERNAL
# stream of PyTorch's pool -- a fifth busy stream on a queue nobody chose.  The collectives of the overlapped exchange are therefore
# issued as synchronous ops (``async_op=False``): since PyTorch 2.7 (AllreduceOptions.asyncOp) a synchronous op is enqueued on the
# CURRENT stream, i.e. the RCCL kernel runs on the trainer's ready stream in front of the sharded update that consumes it, where it
# is ordered by the stream itself.  The host does not block (NCCL / RCCL; gloo blocks the host in either form).
class _Done:
    def wait(self):
        return True


def _on_current_stream():
    """True when this PyTorch runs synchronous collectives on the caller's stream (2.7+: the ``asyncOp`` option exists)."""
    return hasattr(dist, "AllreduceOptions") and hasattr(dist.AllreduceOptions(), "asyncOp")


def allreduce_async(t):
    """Sum of ``t`` over all ranks, ordered on the current stream's timeline (see the note above); returns a handle whose ``wait()``
    orders the then-current stream after the collective (a no-op handle in single-process runs and for the on-stream form)."""
    if world_size() == 1:
        return _Done()
    if _on_current_stream():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=False)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_layout(n, world):
    """(shard, main): a slice of n floats is cut into `world` equal shards of `shard` floats (multiples of 4: 16-byte
    aligned shard boundaries for the fused optimizer kernel); the trailing n - main < 4 * world floats do not divide."""
    shard = (n // (4 * world)) * 4
    return shard, shard * world


def reduce_scatter_async(out, inp):
    """out <- this rank's 1/world slice of sum_ranks(inp) (inp.numel() == world * out.numel()); handle as allreduce_async."""
    if _on_current_stream():
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, async_op=False)
        return _Done()
    return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, async_op=True)


def all_gather_async(out, inp):
    """out <- concatenation over ranks of inp (out.numel() == world * inp.numel())."""
    if _on_current_stream():
        dist.all_gather_into_tensor(out, inp, async_op=False)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, async_op=True)


def broadcast_(t, src=0):
    """In-place broadcast (also of parameter buffers that require grad: the collective works on the detached storage)."""
    if world_size() > 1:
        with torch.no_grad():
            dist.broadcast(t.detach(), src=src)
    return t


def barrier():
    if world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
