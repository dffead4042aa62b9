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
            self.steps, float(grad_scale), 256, _lib.current_stream()))      # one workgroup per CU: runs underneath backward
        self.flow.engine.prepare_weights_range(begin, end)                   # ... and so does the refresh of its shadows
        self._covered += end - begin

    def finish_step(self):
        if self._covered != self.flow.flat_params.numel():
            raise RuntimeError(f"piecewise optimizer step covered {self._covered} of {self.flow.flat_params.numel()} parameters")
        self.flow.engine.shadow_stale = False       # every slice refreshed its shadows right after its update

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "max_exp_avg_sq": self.max_exp_avg_sq, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                         for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.max_exp_avg_sq.copy_(sd["max_exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
