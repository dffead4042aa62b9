"""TEST INFRASTRUCTURE (oracle): CPU restatement of the image metrics of the reference's validation logging.

`SSIM_custom` / `PSNR_custom` (reference utils/metrics.py:450-481) call `pytorch_lightning.metrics.functional.ssim / psnr` with their
defaults (second_stage_video.py:511-512).  pytorch_lightning (pinned 1.1.7 in the reference's environment file) is NOT present in this
image and not vendored in /root/reference, so these functions restate its published algorithm (functional/ssim.py `_ssim_compute`,
functional/psnr.py `_psnr_compute`):

* psnr: 10 log10(data_range^2 / mse), data_range = target.max() - target.min(), mse over all elements;
* ssim: 11 x 11 Gaussian window (sigma 1.5, normalised outer product of two 1-D windows), inputs reflect-padded by 5, the five moments
  filtered with a depth-wise valid convolution, c1 = (0.01 R)^2, c2 = (0.03 R)^2 with R = max(range(preds), range(target)), the map cropped
  by the padding again and averaged over everything.

PARITY UNPINNED: no golden vector of the library itself can be generated here; tests/test_metrics_cpu.py checks this restatement against
an independent scipy formulation (separable correlate1d on the un-padded image, interior positions only)."""
import torch
import torch.nn.functional as F


def psnr(preds, target):
    data_range = target.max() - target.min()
    mse = torch.mean((preds.double() - target.double()) ** 2)
    return (10.0 * (2.0 * torch.log(data_range.double()) - torch.log(mse)) / torch.log(torch.tensor(10.0, dtype=torch.float64))).float()


def _gaussian(kernel_size, sigma, dtype):
    dist = torch.arange(start=(1 - kernel_size) / 2, end=(1 + kernel_size) / 2, step=1, dtype=dtype)
    gauss = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return (gauss / gauss.sum()).unsqueeze(dim=0)


def ssim(preds, target, kernel_size=(11, 11), sigma=(1.5, 1.5), k1=0.01, k2=0.03):
    """preds, target: [N, C, H, W]."""
    data_range = max(preds.max() - preds.min(), target.max() - target.min())
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    channel, dtype = preds.size(1), preds.dtype
    kernel = torch.matmul(_gaussian(kernel_size[0], sigma[0], dtype).t(), _gaussian(kernel_size[1], sigma[1], dtype))
    kernel = kernel.expand(channel, 1, kernel_size[0], kernel_size[1])
    pad_w, pad_h = (kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2
    preds = F.pad(preds, (pad_w, pad_w, pad_h, pad_h), mode="reflect")
    target = F.pad(target, (pad_w, pad_w, pad_h, pad_h), mode="reflect")
    inputs = torch.cat((preds, target, preds * preds, target * target, preds * target))
    outputs = F.conv2d(inputs, kernel, groups=channel)
    n = preds.size(0)
    o = [outputs[x * n:(x + 1) * n] for x in range(5)]
    mu_pred_sq, mu_target_sq, mu_pred_target = o[0].pow(2), o[1].pow(2), o[0] * o[1]
    sigma_pred_sq, sigma_target_sq, sigma_pred_target = o[2] - mu_pred_sq, o[3] - mu_target_sq, o[4] - mu_pred_target
    upper = 2 * sigma_pred_target + c2
    lower = sigma_pred_sq + sigma_target_sq + c2
    ssim_idx = ((2 * mu_pred_target + c1) * upper) / ((mu_pred_sq + mu_target_sq + c1) * lower)
    ssim_idx = ssim_idx[..., pad_h:-pad_h, pad_w:-pad_w]
    return ssim_idx.mean()
