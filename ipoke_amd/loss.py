"""FlowLoss of the reference (models/modules/INN/loss.py:6-31, 75-79) on the HIP NLL kernel."""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr


class _NllFunction(torch.autograd.Function):
    """(loss, nll, nlogdet) = FlowLoss(out, logdet); one kernel also produces d loss/d out and d loss/d logdet."""

    @staticmethod
    def forward(ctx, sample, logdet, weight):
        _lib.require_gpu()
        B = sample.shape[0]
        n = sample[0].numel()
        s = sample.detach().float().contiguous()
        ld = logdet.detach().float().contiguous()
        scal = torch.empty(3, dtype=torch.float32, device=s.device)
        d_out = torch.empty_like(s)
        dld = torch.empty_like(ld)
        # the sum of squares is layout agnostic: treat each sample as one row of n channels
        check(_lib.lib().ipoke_flow_nll(ptr(s), ptr(ld), B, 1, n, n, float(weight), ptr(scal), ptr(d_out), ptr(dld),
                                        _lib.current_stream()))
        ctx.save_for_backward(d_out, dld)
        return scal[0], scal[1], scal[2]

    @staticmethod
    def backward(ctx, g_loss, g_nll, g_nld):
        d_out, dld = ctx.saved_tensors
        # only `loss` is used for optimisation; nll / nlogdet are logged values
        return d_out * g_loss, dld * g_loss, None


def nll(sample, spatial_mean=False):
    """mean over the batch of loss.py:75-79 ``nll``; with spatial_mean the sum over positions becomes their mean (a 1 / (h w) factor)."""
    z = torch.zeros(sample.shape[0], device=sample.device)
    v = _NllFunction.apply(sample, z, 0.0)[1]
    return v / (sample.shape[-2] * sample.shape[-1]) if spatial_mean else v


class FlowLoss(nn.Module):
    def __init__(self, spatial_mean=False, logdet_weight=1.0):
        super().__init__()
        self.spatial_mean = spatial_mean
        self.logdet_weight = logdet_weight

    def forward(self, sample, logdet):
        assert len(logdet.shape) == 1
        loss, nll_loss, nlogdet_loss = _NllFunction.apply(sample, logdet, self.logdet_weight)
        if self.spatial_mean:       # loss.py:14-20: both terms carry 1 / (h w); the gradient scales with the loss
            hw = float(sample.shape[-2] * sample.shape[-1])
            loss, nll_loss, nlogdet_loss = loss / hw, nll_loss / hw, nlogdet_loss / hw
        with torch.no_grad():       # logged only; consumes the device RNG like the reference's randn_like
            reference_nll_loss = nll(torch.randn_like(sample), self.spatial_mean)
        log = {"flow_loss": loss, "reference_nll_loss": reference_nll_loss, "nlogdet_loss": nlogdet_loss,
               "nll_loss": nll_loss, "logdet_weight": self.logdet_weight}
        return loss, log
