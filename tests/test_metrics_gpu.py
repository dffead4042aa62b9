"""GPU parity of ipoke_psnr_ssim (csrc/eval.hip) against the oracle's restatement of pytorch_lightning.metrics.functional.ssim / psnr
(oracle/metrics_ref.py), at small sizes and at the shape the validation loop logs (frames of 128 x 128), and of the running means."""
import pytest
import torch

from ipoke_amd import metrics
from oracle import metrics_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("shape,noise", [((2, 3, 32, 40), 0.3), ((1, 1, 11, 11), 0.2), ((5, 3, 43, 75), 0.05), ((30, 3, 128, 128), 0.4),
                                         ((2, 3, 64, 64), 0.0)])
def test_psnr_ssim_against_oracle(shape, noise):
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    target = torch.rand(shape, generator=g) * 2 - 1
    preds = (target + noise * torch.randn(shape, generator=g)).clamp(-1, 1) if noise else torch.rand(shape, generator=g) * 0.5
    both = metrics.psnr_ssim(preds.to(DEV), target.to(DEV)).cpu()
    want_p, want_s = metrics_ref.psnr(preds, target).item(), metrics_ref.ssim(preds, target).item()
    print(f"{shape}: psnr {both[0].item():.5f} vs {want_p:.5f}, ssim {both[1].item():.6f} vs {want_s:.6f}")
    assert abs(both[0].item() - want_p) <= 1e-4 * max(1.0, abs(want_p))
    assert abs(both[1].item() - want_s) <= 2e-5


def test_identical_images_and_running_means():
    x = torch.rand(4, 3, 32, 32, device=DEV)
    assert abs(metrics.ssim(x, x.clone()).item() - 1.0) <= 1e-6
    s, p = metrics.SSIM_custom(), metrics.PSNR_custom()
    vals = []
    for k in range(3):
        y = (x + 0.1 * (k + 1) * torch.randn_like(x)).clamp(0, 1)
        vals.append((p(y, x).item(), s(y, x).item()))
    assert abs(p.compute().item() - sum(v[0] for v in vals) / 3) <= 1e-4
    assert abs(s.compute().item() - sum(v[1] for v in vals) / 3) <= 1e-6
    with pytest.raises(ValueError):
        metrics.psnr_ssim(x, x[:, :2])


@pytest.mark.parametrize("zeros_everywhere", [True, False])
def test_exact_zero_minimum(zeros_everywhere):
    """Images in [0, 1] whose darkest pixels are exactly 0: the negated minimum is -0.0f, whose bit pattern is INT_MIN (ADVICE r3:
    the integer-atomic float maximum must pick its flavour by the sign bit).  ``zeros_everywhere``: every wavefront sees a zero (the range
    table stayed at its NaN initial value before the fix); otherwise only a few do (the minimum was silently over-estimated)."""
    g = torch.Generator().manual_seed(5)
    shape = (3, 3, 48, 64)
    target = torch.rand(shape, generator=g) * 0.9 + 0.1
    preds = (target + 0.1 * torch.randn(shape, generator=g)).clamp(0.05, 1)
    if zeros_everywhere:
        target[..., ::2] = 0.0
        preds[..., 1::3] = 0.0
    else:
        target[1, 2, 17, 5] = 0.0
        preds[2, 0, 40, 63] = 0.0
    both = metrics.psnr_ssim(preds.to(DEV), target.to(DEV)).cpu()
    want_p, want_s = metrics_ref.psnr(preds, target).item(), metrics_ref.ssim(preds, target).item()
    assert torch.isfinite(both).all()
    assert abs(both[0].item() - want_p) <= 1e-4 * max(1.0, abs(want_p))
    assert abs(both[1].item() - want_s) <= 2e-5
