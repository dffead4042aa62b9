"""Image metrics of the validation logging (reference utils/metrics.py:450-481, second_stage_video.py:511-512): ``SSIM_custom`` and
``PSNR_custom`` -- running means over validation batches of pytorch_lightning.metrics.functional.ssim / psnr (library defaults) -- on
the device: one pass for the ranges and the squared error, one separable-Gaussian pass for the SSIM map (csrc/eval.hip)."""
import torch

from . import _lib
from ._lib import check, ptr

_ws = {}


def psnr_ssim(preds, target):
    """(psnr, ssim) of two [N, C, H, W] tensors as a device tensor of two floats (no host synchronisation)."""
    _lib.require_gpu()
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError(f"expected two [N, C, H, W] tensors of one shape, got {tuple(preds.shape)} and {tuple(target.shape)}")
    p, t = preds.float().contiguous(), target.float().contiguous()
    N, C, H, W = p.shape
    need = _lib.lib().ipoke_image_metrics_workspace_bytes(N * C, H, W)
    key = (p.device, torch.cuda.current_stream().cuda_stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _ws[key] = torch.empty(need, dtype=torch.uint8, device=p.device)
    out = torch.empty(2, dtype=torch.float32, device=p.device)
    check(_lib.lib().ipoke_psnr_ssim(ptr(p), ptr(t), N * C, H, W, ptr(ws), ptr(out), _lib.current_stream()))
    return out


def psnr(preds, target):
    return psnr_ssim(preds, target)[0]


def ssim(preds, target):
    return psnr_ssim(preds, target)[1]


class _RunningMean:
    """metrics.py:450-481: ``update`` adds one batch value, ``compute`` returns the mean over the batches seen."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.acc, self.total = None, 0

    def update_value(self, v):
        self.acc = v if self.acc is None else self.acc + v
        self.total += 1

    def compute(self):
        return self.acc / self.total

    def __call__(self, preds, target):
        v = self.update(preds, target)
        return v


class SSIM_custom(_RunningMean):
    def update(self, preds, targets):
        v = ssim(preds, targets)
        self.update_value(v)
        return v


class PSNR_custom(_RunningMean):
    def update(self, preds, targets):
        v = psnr(preds, targets)
        self.update_value(v)
        return v
