"""Fused Adam(amsgrad) over the flow's flat parameter buffer.

Same update rule as ``torch.optim.Adam(lr, betas=(0.9, 0.999), weight_decay, amsgrad=True)`` which the
reference constructs in ``PokeMotionModel.configure_optimizers`` (models/second_stage_video.py:632-662),
executed as ONE kernel over the 1.05-1.24 B contiguous fp32 parameters (plus m, v, v_max of the same
layout) instead of ~3000 per-tensor updates; afterwards the matrix-core weight shadows are refreshed.
It is a ``torch.optim.Optimizer`` so LR schedules that write ``param_groups[i]['lr']`` (the reference's
``on_train_batch_start``) work unchanged.
"""
import torch

from . import _lib
from ._lib import check, ptr


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
    def _shard_state(self, begin, n):
        """Adam state of this rank's shard of the slice starting at ``begin`` (allocated on first use: 3 P / world in total;
        the replicated ``exp_avg*`` buffers of the constructor are released by ``enable_sharding``)."""
        st = self._shards.get(begin)
        if st is None or st[0].numel() != n:
            dev = self.flow.flat_params.device
            st = tuple(torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3))
            self._shards[begin] = st
        return st

    def enable_sharding(self, world, rank, grad_dtype=torch.float32):
        """Switch to the sharded update (call before the first step).  ``grad_dtype=torch.bfloat16`` exchanges the gradients
        in bf16 (half the xGMI bytes of the reduce-scatter; parameters are always gathered in fp32)."""
        if self.steps != 0:
            raise RuntimeError("enable_sharding must be called before the first optimizer step")
        self.world, self.rank, self.grad_dtype = int(world), int(rank), grad_dtype
        self._shards = {}
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
            m, v, vmax = self._shard_state(begin, sh)
            check(L.ipoke_adam_amsgrad_step_grid(ptr(flat[lo:lo + sh]), ptr(red), ptr(m), ptr(v), ptr(vmax), sh, *hyper, 128,
                                                 _lib.current_stream()))
            own = flat[lo:lo + sh].clone()                 # out-of-place input: valid for every backend
            D.all_gather_async(flat[begin:begin + main], own).wait()
        if main < n:
            tail = grads[begin + main:end]
            D.allreduce_async(tail).wait()
            m, v, vmax = self._shard_state(-(begin + 1), n - main)       # replicated state of the few trailing elements
            # the trailing slice starts 16-byte aligned (begin and main are multiples of 4 floats)
            check(L.ipoke_adam_amsgrad_step_grid(ptr(flat[begin + main:end]), ptr(tail), ptr(m), ptr(v), ptr(vmax), n - main, *hyper, 1,
                                                 _lib.current_stream()))
        self.flow.engine.prepare_weights_range(begin, end)
        self._covered += n

    def finish_step(self):
        if self._covered != self.flow.flat_params.numel():
            raise RuntimeError(f"piecewise optimizer step covered {self._covered} of {self.flow.flat_params.numel()} parameters")
        self.flow.engine.shadow_stale = False       # every slice refreshed its shadows right after its update

    def state_dict(self):
        if getattr(self, "_shards", None) is not None:
            return {"steps": self.steps, "sharded": True, "world": self.world, "rank": self.rank,
                    "shards": {k: tuple(t.clone() for t in v) for k, v in self._shards.items()},
                    "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "max_exp_avg_sq": self.max_exp_avg_sq, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                         for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        if sd.get("sharded"):
            if (sd["world"], sd["rank"]) != (self.world, self.rank):
                raise ValueError("sharded optimizer state belongs to another (world, rank)")
            self._shards = {k: tuple(t.clone() for t in v) for k, v in sd["shards"].items()}
            for g, s_ in zip(self.param_groups, sd["param_groups"]):
                g.update(s_)
            return
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.max_exp_avg_sq.copy_(sd["max_exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
