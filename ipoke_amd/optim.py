"""Fused Adam(amsgrad) over the flow's flat parameter buffer.

Same update rule as ``torch.optim.Adam(lr, betas=(0.9, 0.999), weight_decay, amsgrad=True)`` which the
reference constructs in ``PokeMotionModel.configure_optimizers`` (models/second_stage_video.py:632-662),
executed as ONE kernel over the 1.05-1.24 B contiguous fp32 parameters (plus m, v, v_max of the same
layout) instead of ~3000 per-tensor updates; afterwards the matrix-core weight shadows are refreshed.
It is a ``torch.optim.Optimizer`` so LR schedules that write ``param_groups[i]['lr']`` (the reference's
``on_train_batch_start``) work unchanged.
"""
import os

import torch

from . import _lib
from ._lib import check, ptr

# IPOKE_ADAM_FUSION=1: Adam-amsgrad fused with the shadow refresh of the plain 1x1 weights (ipoke_flow_adam_range).  Built, bit-identical
# to the plain path (tests/test_flow_gpu.py) and measured on MI355X (round 3, c2): it saves 3.6 GB of HBM reads per step and 73 % of the
# relayout work, but its 64 x 64-tile access pattern streams HBM worse than the linear kernel -- optimizer + refresh of the whole buffer
# 14.4 ms against 12.6 ms alone, 17.5 against 15.2 ms in twelve pieces on 128-workgroup grids (with the loads of the next tile issued ahead
# of the arithmetic; 20.0 ms without), and the train step 62.1-62.2 ms against 61.8 ms (scripts/probe_adam.py).  Off by default.
_FUSE_DEFAULT = os.environ.get("IPOKE_ADAM_FUSION", "0") == "1"
_FUSE_SHADOWS = _FUSE_DEFAULT
# persistent grid of the engine-issued update underneath backward (c2, round 4, with the linear conv2 kernel: 64 / 128 / 160 / 192 / 224 / 256
# workgroups -> 65.0 / 57.8 / 57.4 / 57.35 / 57.4 / 59.2 ms)
# round 5 (unit kernels on 80 CUs): 160 workgroups with 24 backward pieces 54.6 / 54.8 ms against 55.6 / 55.5 at 192 / 16 (one call, two repeats)
_NATIVE_BLOCKS = int(os.environ.get("IPOKE_NATIVE_ADAM_BLOCKS", "160"))
_TILE_BLOCKS = int(os.environ.get("IPOKE_ADAM_TILE_BLOCKS", "128"))     # developer A/B: persistent grid of the fused tile kernel underneath backward


class FusedAdamAmsgrad(torch.optim.Optimizer):
    def __init__(self, flow, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=True):
        if not amsgrad:
            raise NotImplementedError("the reference trains with amsgrad=True; the fused kernel implements that variant")
        self.flow = flow
        flat = flow.flat_params
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=True)
        super().__init__([{"params": [flat], "name": "flow"}], defaults)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.max_exp_avg_sq = torch.zeros_like(flat)
        self.steps = 0
        self.world, self.rank, self.grad_dtype, self._shards = 1, 0, torch.float32, None

    def zero_grad(self, set_to_none=False):
        # the engine overwrites the flat gradient buffer on every backward; nothing to clear
        return None

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.steps += 1
        flat, grads = self.flow.flat_params, self.flow.flat_grads
        if _FUSE_SHADOWS and not self.flow.engine.shadow_stale:
            # update + shadow refresh in one pass (the conv2 operands leave the optimizer kernel's registers)
            self.flow.engine.adam_range(0, flat.numel(), self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq, g["lr"], g["betas"], g["eps"],
                                        g["weight_decay"], self.steps, grad_scale)
            return loss
        check(_lib.lib().ipoke_adam_amsgrad_step(
            ptr(flat), ptr(grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.max_exp_avg_sq), flat.numel(),
            float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
            self.steps, float(grad_scale), _lib.current_stream()))
        self.flow.engine.prepare_weights()
        return loss

    # ---- piecewise step: the update of a slice can start as soon as its gradients are final (overlap with backward) ----
    def begin_step(self):
        self.steps += 1
        self._covered = 0

    @torch.no_grad()
    def step_range(self, begin, end, grad_scale=1.0):
        """Adam-amsgrad update of flat[begin:end] on the current stream (same kernel, same step count for every slice)."""
        g = self.param_groups[0]
        flat, grads = self.flow.flat_params, self.flow.flat_grads
        if _FUSE_SHADOWS and not self.flow.engine.shadow_stale:
            self.flow.engine.adam_range(begin, end, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq, g["lr"], g["betas"], g["eps"],
                                        g["weight_decay"], self.steps, grad_scale, max_blocks=_TILE_BLOCKS)
            self._covered += end - begin
            return
        sl = slice(begin, end)
        check(_lib.lib().ipoke_adam_amsgrad_step_grid(
            ptr(flat[sl]), ptr(grads[sl]), ptr(self.exp_avg[sl]), ptr(self.exp_avg_sq[sl]), ptr(self.max_exp_avg_sq[sl]), end - begin,
            float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
            self.steps, float(grad_scale), 128, _lib.current_stream()))      # persistent grid on half the CUs: runs underneath backward
        # (a fused MaCowUnit workgroup cannot share a CU with one of these: 2 x 240 + 52 registers > 512; with a workgroup on
        # every CU the chain stalled until the slice update had finished -- 74.3 vs 72.6 ms per step)
        self.flow.engine.prepare_weights_range(begin, end)                   # ... and so does the refresh of its shadows
        self._covered += end - begin

    # ---- ZeRO-1 over the flat buffer: reduce-scatter -> update of this rank's shard -> all-gather of the parameters ----
    def _shard_state(self, key, lo, n):
        """Adam state (m, v, v_max) of flat[lo:lo+n] held by THIS rank under ``key`` (a slice's begin for its 1/world shard,
        -(begin+1) for its replicated trailing elements).  Allocated on first use -- 3 P / world in total -- as zeros, or cut
        out of a replicated checkpoint that ``load_state_dict`` parked in ``_full_state``.  A stored shard of another size
        means the slice layout (IPOKE_PIECES, bucket count, world size) changed since the state was saved: that is an error,
        never a silent reset of the moments."""
        st = self._shards.get(key)
        if st is not None:
            if st[0].numel() != n or self._layout.get(key) != (lo, n):
                raise RuntimeError(f"optimizer shard {key}: stored layout {self._layout.get(key)} / {st[0].numel()} elements, requested "
                                   f"({lo}, {n}) -- the gradient-slice layout or world size differs from the one the state was saved with")
            return st
        dev = self.flow.flat_params.device
        full = getattr(self, "_full_state", None)
        if full is not None:
            st = tuple(f[lo:lo + n].to(dev, torch.float32).clone() for f in full)
        else:
            st = tuple(torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3))
        self._shards[key] = st
        self._layout[key] = (lo, n)
        return st

    def enable_sharding(self, world, rank, grad_dtype=torch.float32):
        """Switch to the sharded update (call before the first step).  ``grad_dtype=torch.bfloat16`` exchanges the gradients
        in bf16 (half the xGMI bytes of the reduce-scatter; parameters are always gathered in fp32)."""
        if self.steps != 0:
            raise RuntimeError("enable_sharding must be called before the first optimizer step")
        self.world, self.rank, self.grad_dtype = int(world), int(rank), grad_dtype
        self._shards, self._layout, self._full_state = {}, {}, None
        self.exp_avg = self.exp_avg_sq = self.max_exp_avg_sq = None      # replicated state is not kept

    @torch.no_grad()
    def step_range_sharded(self, begin, end, grad_scale=1.0):
        """Sum grads[begin:end] over the ranks and update flat[begin:end] on every rank, on the current stream:
        the slice is cut into ``world`` equal shards (multiples of 4 floats); reduce-scatter hands this rank the summed
        gradients of its shard, the fused Adam-amsgrad kernel updates that shard only (1/world of the optimizer's HBM
        traffic and state), an all-gather distributes the updated parameters.  The < 4*world trailing elements that do not
        divide are all-reduced and updated on every rank."""
        from . import dist as D
        g = self.param_groups[0]
        flat, grads = self.flow.flat_params, self.flow.flat_grads
        W, n = self.world, end - begin
        sh, main = D.shard_layout(n, W)
        L = _lib.lib()
        hyper = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                 self.steps, float(grad_scale))
        if sh > 0:
            src = grads[begin:begin + main]
            if self.grad_dtype != torch.float32:
                src = src.to(self.grad_dtype)
            red = torch.empty(sh, dtype=src.dtype, device=src.device)
            D.reduce_scatter_async(red, src).wait()
            if red.dtype != torch.float32:
                red = red.float()
            lo = begin + self.rank * sh
            m, v, vmax = self._shard_state(begin, lo, sh)
            check(L.ipoke_adam_amsgrad_step_grid(ptr(flat[lo:lo + sh]), ptr(red), ptr(m), ptr(v), ptr(vmax), sh, *hyper, 128,
                                                 _lib.current_stream()))
            own = flat[lo:lo + sh].clone()                 # out-of-place input: valid for every backend
            D.all_gather_async(flat[begin:begin + main], own).wait()
        if main < n:
            tail = grads[begin + main:end]
            D.allreduce_async(tail).wait()
            m, v, vmax = self._shard_state(-(begin + 1), begin + main, n - main)       # replicated state of the few trailing elements
            # the trailing slice starts 16-byte aligned (begin and main are multiples of 4 floats)
            check(L.ipoke_adam_amsgrad_step_grid(ptr(flat[begin + main:end]), ptr(tail), ptr(m), ptr(v), ptr(vmax), n - main, *hyper, 1,
                                                 _lib.current_stream()))
        self.flow.engine.prepare_weights_range(begin, end)
        self._covered += n

    # ---- native piecewise step: the engine applies the update of every piece itself (single process) ------------------
    def arm_native(self, grad_scale=1.0, max_blocks=_NATIVE_BLOCKS):
        """Hand this step's update to the engine: ``ipoke_flow_backward_pieces`` then queues the Adam-amsgrad update and the shadow
        refresh of every piece on the ready stream as soon as the piece is final (ipoke_flow_set_native_adam) -- the launches of
        ``step_range`` without the host callback between the chain's kernels.  Call after ``begin_step``; ``finish_native`` after
        the backward pass."""
        g = self.param_groups[0]
        check(_lib.lib().ipoke_flow_set_native_adam(self.flow.engine.handle, ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.max_exp_avg_sq),
                                                    float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                    float(g["weight_decay"]), int(self.steps), float(grad_scale), int(max_blocks)))

    def disarm_native(self):
        check(_lib.lib().ipoke_flow_set_native_adam(self.flow.engine.handle, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 1.0, 0))

    def finish_native(self):
        self.disarm_native()
        self._covered = self.flow.flat_params.numel()          # every piece was updated and refreshed by the engine
        self.finish_step()

    def finish_step(self):
        if self._covered != self.flow.flat_params.numel():
            raise RuntimeError(f"piecewise optimizer step covered {self._covered} of {self.flow.flat_params.numel()} parameters")
        self.flow.engine.shadow_stale = False       # every slice refreshed its shadows right after its update
        # a replicated checkpoint loaded into the sharded optimizer is only a source for the shards (_shard_state): after one full step
        # every shard has been cut out of it -- release the 3 x P floats (they would undo ZeRO-1's 3P / world saving)
        if getattr(self, "_full_state", None) is not None and getattr(self, "_shards", None) is not None:
            self._full_state = None

    def _groups(self):
        return [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]

    def state_dict(self):
        """Replicated mode: the full m / v / v_max.  Sharded (ZeRO-1) mode: THIS RANK's shards only, with their (offset, length)
        layout -- every rank has to save its own (``torch.save(opt.state_dict(), f"opt_rank{rank}.pt")``), or call the collective
        ``full_state_dict()`` and save its result on rank 0."""
        if getattr(self, "_shards", None) is not None:
            return {"steps": self.steps, "sharded": True, "world": self.world, "rank": self.rank,
                    "shards": {k: tuple(t.clone() for t in v) for k, v in self._shards.items()}, "layout": dict(self._layout),
                    "param_groups": self._groups()}
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "max_exp_avg_sq": self.max_exp_avg_sq, "param_groups": self._groups()}

    @staticmethod
    def merge_sharded(states, numel):
        """Per-rank sharded state dicts (all ranks of one run) -> a replicated state dict."""
        full = [torch.zeros(numel, dtype=torch.float32) for _ in range(3)]
        covered = torch.zeros(numel, dtype=torch.bool)
        for sd in states:
            for key, (lo, n) in sd["layout"].items():
                for f, t_ in zip(full, sd["shards"][key]):
                    f[lo:lo + n] = t_.detach().to("cpu", torch.float32)
                covered[lo:lo + n] = True
        if not bool(covered.all()):
            raise ValueError(f"sharded optimizer states cover {int(covered.sum())} of {numel} elements (a rank's file is missing)")
        return {"steps": states[0]["steps"], "exp_avg": full[0], "exp_avg_sq": full[1], "max_exp_avg_sq": full[2],
                "param_groups": states[0]["param_groups"]}

    def full_state_dict(self):
        """Collective in sharded mode: gathers every rank's shards into the replicated layout (on the parameters' device)."""
        if getattr(self, "_shards", None) is None:
            return self.state_dict()
        import torch.distributed as td
        numel, dev = self.flow.flat_params.numel(), self.flow.flat_params.device
        full = [torch.zeros(numel, dtype=torch.float32, device=dev) for _ in range(3)]
        for key, (lo, n) in self._layout.items():
            if key < 0 and self.rank != 0:
                continue                                    # trailing elements are replicated: rank 0 contributes them
            for f, t_ in zip(full, self._shards[key]):
                f[lo:lo + n] = t_
        if self.world > 1:
            for f in full:
                td.all_reduce(f)                            # every element is owned by exactly one contributing rank
        return {"steps": self.steps, "exp_avg": full[0], "exp_avg_sq": full[1], "max_exp_avg_sq": full[2], "param_groups": self._groups()}

    def load_state_dict(self, sd):
        """Accepts both layouts in both modes: a replicated checkpoint loaded into a sharded optimizer is cut into this rank's
        shards when the slices are first touched; a sharded per-rank checkpoint loads into the same (world, rank) only (merge the
        ranks' files with ``merge_sharded`` for anything else).  State lands on the parameters' device whatever ``map_location``
        the checkpoint was read with."""
        dev = self.flow.flat_params.device
        sharded_opt = getattr(self, "_shards", None) is not None
        if sd.get("sharded"):
            if not sharded_opt:
                raise ValueError("per-rank sharded optimizer state cannot be loaded into a replicated optimizer: merge the ranks' "
                                 "files with FusedAdamAmsgrad.merge_sharded(states, numel) (or save full_state_dict()) first")
            if (sd["world"], sd["rank"]) != (self.world, self.rank):
                raise ValueError(f"sharded optimizer state belongs to (world, rank) = ({sd['world']}, {sd['rank']}), this is "
                                 f"({self.world}, {self.rank})")
            for key, (lo, n) in sd["layout"].items():
                if any(t_.numel() != n for t_ in sd["shards"][key]):
                    raise ValueError(f"optimizer shard {key}: tensors do not match the stored layout ({lo}, {n})")
            self._shards = {k: tuple(t_.detach().to(dev, torch.float32).clone() for t_ in v) for k, v in sd["shards"].items()}
            self._layout = {k: tuple(v) for k, v in sd["layout"].items()}
            self._full_state = None
        elif sharded_opt:
            n = self.flow.flat_params.numel()
            full = (sd["exp_avg"], sd["exp_avg_sq"], sd["max_exp_avg_sq"])
            if any(f.numel() != n for f in full):
                raise ValueError("replicated optimizer state does not match the flat parameter buffer")
            self._shards, self._layout = {}, {}
            self._full_state = tuple(f.detach() for f in full)          # shards are cut out on first use (_shard_state)
        else:
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.max_exp_avg_sq.copy_(sd["max_exp_avg_sq"])
        self.steps = int(sd["steps"])
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update(s_)
