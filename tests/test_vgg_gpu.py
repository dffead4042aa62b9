"""VGG perceptual loss on the HIP kernels (ipoke_amd/vgg.py, reference utils/losses.py:6-82) against golden G12 -- the reference's
VGGLoss run on the restated vgg19 feature stack: loss value, the five feature maps, and the gradient w.r.t. the generated
frames (backward through 13 adjoint convolutions, 4 max-pool selections and 5 L1 terms)."""
import zlib

import numpy as np
import pytest
import torch

from ipoke_amd.utils.detfill import deterministic_fill_
from ipoke_amd.vgg import VGGLoss
from oracle import vgg_ref
from tests.conftest import t

pytestmark = pytest.mark.gpu
DEV = "cuda"
# f32: fp32 matrix-core GEMMs vs the reference's CPU convolutions; bf16: bf16 maps through 13 conv layers, the L1 gradient is
# sign(f2 - f1) of bf16-rounded maps, so single gradient elements move while the sum of |dy| agrees (as for the discriminators)
# f32, measured: loss exact to 1e-7, sum |dy| to 2e-5, while the largest single-element deviation is 1e-2 of max |dy|: the L1
# gradient is sign(f2 - f1), and a handful of the 2.6 M map elements have |f2 - f1| below the fp32 rounding of two different
# summation orders (mean deviation 1.3e-3 of mean |dy|; bf16: 0.2 / 0.17), so their sign -- and a few pixels of dy under that element's receptive field -- flips.  Hence an element-wise
# bound (dy), a mean bound (dy_mean) and the abs-sum (dy_sum).
TOL = {"f32": dict(loss=2e-6, fmap=2e-5, dy=5e-2, dy_mean=5e-3, dy_sum=1e-4), "bf16": dict(loss=5e-3, fmap=5e-2, dy=0.8, dy_mean=0.4, dy_sum=3e-2)}


def _checksum(x, key):
    x = x.detach().double().flatten().cpu()
    idx = torch.randint(0, x.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(key.encode())))
    return np.array([x.sum().item(), x.abs().sum().item(), *x[idx].tolist()])


def _loss_module(dtype):
    m = VGGLoss(dtype=dtype)
    feats = vgg_ref.vgg19_features()
    deterministic_fill_(feats, prefix="vgg19.features.")
    m.vgg.load_torchvision_features({"features." + k: v for k, v in feats.state_dict().items()})     # torchvision key layout
    return m.to(DEV)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_vgg_loss_value_maps_gradient(golden, dtype):
    g = golden("g12_vgg_loss")
    tol = TOL[dtype]
    m = _loss_module(dtype)
    assert list(m.vgg.state_dict()) == [str(k) for k in g["keys"]]
    x, y = t(g["x_true"], DEV), t(g["x_hat"], DEV).requires_grad_(True)
    loss = m(x, y)
    loss.backward()
    e_loss = abs(loss.item() - float(g["loss"]))
    dy, want = y.grad.cpu(), t(g["dy"])
    e_dy = (dy - want).abs().max().item() / want.abs().max().item()
    e_sum = abs(dy.abs().sum().item() - want.abs().sum().item()) / want.abs().sum().item()
    e_mean = (dy - want).abs().mean().item() / want.abs().mean().item()
    print(f"[{dtype}] loss {loss.item():.6f} vs {float(g['loss']):.6f}; dy max-rel err {e_dy:.2e}, mean-rel err {e_mean:.2e}, |dy| sum rel err {e_sum:.2e}")
    assert e_loss <= tol["loss"] * max(1.0, float(g["loss"])) and e_dy <= tol["dy"] and e_mean <= tol["dy_mean"] and e_sum <= tol["dy_sum"]
    with torch.no_grad():
        fm = m.vgg(y.detach())
    for i, f in enumerate(fm):
        full = f.t[:, :f.C].float().reshape(f.N, f.dhw[1], f.dhw[2], f.C).permute(0, 3, 1, 2)
        want = t(g[f"fmap{i}_slice"])
        err = (full[:2, :4, :4, :4].cpu() - want).abs().max().item()
        cs, ws = _checksum(full, f"vgg{i}"), g[f"fmap{i}_checksum"]
        print(f"[{dtype}] relu{i + 1}_1 {tuple(full.shape)}: slice err {err:.2e} (max {want.abs().max():.2f}); abs-sum {cs[1]:.5e} vs {ws[1]:.5e}")
        assert err <= tol["fmap"] * max(1.0, want.abs().max().item())
        assert abs(cs[1] - ws[1]) <= (1e-4 if dtype == "f32" else 1e-2) * ws[1]
    assert all(p.grad is None for p in m.parameters())                # frozen: no weight gradient was formed


def test_vgg_term_in_the_first_stage_step():
    """FirstStageTrainer with w_vgg: the loss of one step equals L1 + KL (same model, w_vgg = 0) plus w_vgg x the VGG loss of
    its own reconstruction, and the step runs backward through both paths."""
    from ipoke_amd import configs
    from ipoke_amd.first_stage import SpadeCondMotionModel
    from ipoke_amd.first_stage_train import FirstStageTrainer, first_stage_forward_loss
    torch.manual_seed(0)
    model = SpadeCondMotionModel(configs.first_stage_config(64, 32, 4), dirs={}, dtype="f32").to(DEV)
    deterministic_fill_(model, prefix="first_stage.")
    vgg = _loss_module("f32")
    X = (torch.rand(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(DEV)
    eps = torch.randn(2, 32, 8, 8, generator=torch.Generator().manual_seed(5)).to(DEV)
    import copy
    twin = copy.deepcopy(model).train()                               # same weights, same power iteration as the step below
    with torch.no_grad():
        base, X_hat, _, _ = first_stage_forward_loss(twin, X, eps)
        lv = vgg(X[:, 1:].reshape(-1, 3, 64, 64), X_hat.reshape(-1, 3, 64, 64))
    before = [p.detach().clone() for p in model.parameters()]
    tr = FirstStageTrainer(model, vgg_loss=vgg, w_vgg=10.0)
    loss, _ = tr.step(X, eps)
    print(f"L1+KL {base.item():.5f} + 10 x VGG {lv.item():.5f} = {base.item() + 10 * lv.item():.5f}; step loss {loss.item():.5f}")
    assert abs(loss.item() - (base.item() + 10.0 * lv.item())) <= 1e-4 * abs(loss.item())
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
