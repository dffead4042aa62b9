"""The oracle's restatement of pytorch_lightning.metrics.functional.ssim / psnr (oracle/metrics_ref.py; the library is absent here:
parity unpinned) against an independent formulation: scipy's separable correlate1d on the un-padded images, interior positions only
(the reflect padding of the library's implementation is cropped away again), numpy's log10 for the PSNR."""
import numpy as np
import pytest
import torch
from scipy.ndimage import correlate1d

from oracle import metrics_ref


def _ssim_scipy(a, b):
    a, b = a.double().numpy(), b.double().numpy()
    d = np.arange(11) - 5
    g = np.exp(-(d / 1.5) ** 2 / 2); g /= g.sum()
    f = lambda x: correlate1d(correlate1d(x, g, axis=-1, mode="constant"), g, axis=-2, mode="constant")[..., 5:-5, 5:-5]
    R = max(a.max() - a.min(), b.max() - b.min())
    c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    ma, mb = f(a), f(b)
    saa, sbb, sab = f(a * a) - ma * ma, f(b * b) - mb * mb, f(a * b) - ma * mb
    return (((2 * ma * mb + c1) * (2 * sab + c2)) / ((ma * ma + mb * mb + c1) * (saa + sbb + c2))).mean()


@pytest.mark.parametrize("shape,seed", [((2, 3, 32, 40), 0), ((1, 1, 11, 11), 1), ((3, 3, 64, 64), 2)])
def test_ssim_psnr_restatement_against_scipy(shape, seed):
    g = torch.Generator().manual_seed(seed)
    target = torch.rand(shape, generator=g) * 2 - 1
    preds = (target + 0.3 * torch.randn(shape, generator=g)).clamp(-1, 1)
    got = metrics_ref.ssim(preds, target).item()
    assert abs(got - _ssim_scipy(preds, target)) <= 2e-5
    t, p = target.double().numpy(), preds.double().numpy()
    want = 10 * np.log10((t.max() - t.min()) ** 2 / np.mean((p - t) ** 2))
    assert abs(metrics_ref.psnr(preds, target).item() - want) <= 1e-4


def test_ssim_of_identical_images_is_one():
    x = torch.rand(2, 3, 24, 24)
    assert abs(metrics_ref.ssim(x, x.clone()).item() - 1.0) <= 1e-6
