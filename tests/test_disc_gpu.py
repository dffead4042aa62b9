"""First-stage temporal discriminator (ipoke_amd/discriminator.py, reference patchgan_3d.py:171-304) against golden G8, which
holds the reference module's own outputs: predictions, feature maps, hinge loss with every parameter gradient, the
generator-side loss with the gradient w.r.t. the fake clip, and a train-mode forward (power iterations)."""
import zlib

import numpy as np
import pytest
import torch

from ipoke_amd.discriminator import TemporalDiscriminator
from ipoke_amd.utils.detfill import deterministic_fill_
from tests.conftest import t

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = {"bce_loss": False, "gp_weight": 1.0, "num_classes": 1, "patch_temp_disc": False}
# f32: HIP fp32 GEMMs vs the reference's fp32 convs; bf16: bf16 activations/weights through 17 conv + GroupNorm layers
# pred / loss bounds are relative to the magnitude of the predictions (|pred| ~ 4e2 with the name-keyed fill: fc.weight ~ 1)
# dx (gradient w.r.t. the clip): element-wise bound relative to the largest element, plus the abs-sum of the whole tensor.  In
# bf16 the L1 feature-matching gradient is sign(f1 - f2) of bf16-rounded maps pushed back through ReLU masks and max-pool
# selections, so single elements move by up to ~25 % of the maximum (measured) while the abs-sum agrees to 0.3 %.
TOL = {"f32": dict(pred=2e-6, fmap=2e-4, loss=2e-6, grad=5e-3, dx=5e-3, dx_sum=1e-3),
       "bf16": dict(pred=4e-3, fmap=8e-2, loss=4e-3, grad=1.5e-1, dx=4e-1, dx_sum=1e-2)}


def _checksum(x, key):
    x = x.detach().double().flatten().cpu()
    idx = torch.randint(0, x.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(key.encode())))
    return np.array([x.sum().item(), x.abs().sum().item(), *x[idx].tolist()])


def _ncdhw(f):
    D, H, W = f.dhw
    return f.t[:, :f.C].float().reshape(f.N, D, H, W, f.C).permute(0, 4, 1, 2, 3)


def _model(dtype):
    m = TemporalDiscriminator(64, CFG, dtype=dtype)
    deterministic_fill_(m, prefix="disc_t.")
    return m.to(DEV)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_discriminator_side(golden, dtype):
    g = golden("g8_temporal_disc_64")
    tol = TOL[dtype]
    m = _model(dtype).eval()
    Xt, Xf = t(g["X_true"], DEV), t(g["X_fake"], DEV)
    pf, _ = m(Xf)
    pt, fm = m(Xt)
    e_pf, e_pt = (pf.cpu() - t(g["pred_fake"])).abs().max().item(), (pt.cpu() - t(g["pred_true"])).abs().max().item()
    print(f"[{dtype}] pred err fake {e_pf:.2e} true {e_pt:.2e} (|pred| <= {np.abs(g['pred_true']).max():.2f})")
    scale = max(1.0, float(np.abs(g["pred_true"]).max()))
    assert e_pf <= tol["pred"] * scale and e_pt <= tol["pred"] * scale
    for i, f in enumerate(fm):
        full = _ncdhw(f)
        want = t(g[f"fmap{i}_slice"])
        err = (full[:, :4, :, :3, :3].cpu() - want).abs().max().item()
        cs, ws = _checksum(full, f"fmap{i}"), g[f"fmap{i}_checksum"]
        print(f"[{dtype}] fmap{i} {tuple(full.shape)} slice err {err:.2e}; abs-sum {cs[1]:.4e} vs {ws[1]:.4e}")
        assert err <= tol["fmap"] * max(1.0, want.abs().max().item())
        assert abs(cs[1] - ws[1]) <= (2e-4 if dtype == "f32" else 1e-2) * ws[1]
    loss = (m.loss(pf, real=False) + m.loss(pt, real=True)) / 2.0
    assert abs(loss.item() - float(g["loss_d"])) <= tol["loss"] * max(1.0, abs(float(g["loss_d"])))
    loss.backward()
    grads = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, want in zip(g["grad_names"], g["grad_checksums"]):
        got = _checksum(grads[str(k)].grad, str(k))
        rel = abs(got[1] - want[1]) / max(want[1], 1e-12)                      # abs-sum of the whole gradient tensor
        if rel > worst[1]:
            worst = (str(k), rel)
        assert rel <= tol["grad"], (k, got, want)
        assert np.allclose(got[2:], want[2:], rtol=tol["grad"] * 4, atol=tol["grad"] * want[1] / grads[str(k)].numel() * 20), (k, got, want)
    print(f"[{dtype}] worst parameter-gradient abs-sum deviation {worst[1]:.2e} ({worst[0]}) over {len(g['grad_names'])} tensors")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gradient_penalty_forward_over_reverse(golden, dtype):
    """gp2 (patchgan_3d.py:285-294): value and every parameter gradient of the penalty against the reference's double
    backward (golden G8), computed here by a tangent pass + one ordinary backward pass on the HIP kernels."""
    g = golden("g8_temporal_disc_64")
    m = _model(dtype).eval()
    Xt = t(g["X_true"], DEV)
    gp = m.gp2(Xt)
    want = float(g["gp"])
    print(f"[{dtype}] gp {gp.item():.6f} vs {want:.6f}")
    assert abs(gp.item() - want) <= (2e-4 if dtype == "f32" else 2e-2) * want
    gp.backward()
    grads = dict(m.named_parameters())
    worst = ("", 0.0)
    tol = 5e-3 if dtype == "f32" else 1e-1
    for k, wsum in zip(g["gp_grad_names"], g["gp_grad_checksums"]):
        p = grads[str(k)]
        assert p.grad is not None, k
        got = _checksum(p.grad, "gp." + str(k))
        if wsum[1] < 1e-6:
            assert got[1] <= 1e-2, (k, got, wsum)
            continue
        rel = abs(got[1] - wsum[1]) / wsum[1]
        if rel > worst[1]:
            worst = (str(k), rel)
        assert rel <= tol, (k, got, wsum)
        if dtype == "f32":
            assert np.allclose(got[2:], wsum[2:], rtol=2e-2, atol=2e-2 * wsum[1] / p.numel() * 20), (k, got, wsum)
    print(f"[{dtype}] gp: worst parameter-gradient abs-sum deviation {worst[1]:.2e} ({worst[0]}) over {len(g['gp_grad_names'])} tensors")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_generator_side_and_train_mode(golden, dtype):
    g = golden("g8_temporal_disc_64")
    tol = TOL[dtype]
    m = _model(dtype).eval()
    Xt = t(g["X_true"], DEV)
    xf = t(g["X_fake"], DEV).requires_grad_(True)
    pg, ff = m(xf)
    with torch.no_grad():
        _, ft = m(Xt)
    lg = -pg.mean() + m.fmap_loss(ff, ft)
    assert abs(lg.item() - float(g["loss_g"])) <= tol["loss"] * max(1.0, abs(float(g["loss_g"])))
    lg.backward()
    want = t(g["dx_fake_slice"])
    err = (xf.grad[:, :, :2, :6, :6].cpu() - want).abs().max().item()
    cs, ws = _checksum(xf.grad, "dx_fake"), g["dx_fake_checksum"]
    print(f"[{dtype}] d loss_g / d X_fake slice err {err:.2e} (max {want.abs().max():.2e}); abs-sum {cs[1]:.4e} vs {ws[1]:.4e}")
    assert err <= tol["dx"] * want.abs().max().item()
    assert abs(cs[1] - ws[1]) <= tol["dx_sum"] * ws[1]
    # train mode: every spectral-normalised conv runs one power iteration per forward call
    m.train()
    with torch.no_grad():
        ptr_, _ = m(Xt)
    assert (ptr_.cpu() - t(g["pred_true_train"])).abs().max().item() <= tol["pred"] * max(1.0, float(np.abs(g["pred_true_train"]).max()))
    for k in ("conv1", "layer2.0.downsample.0", "layer4.1.conv2"):
        mod = m.get_submodule(k)
        assert (mod.weight_u.cpu() - t(g[f"u1.{k}"])).abs().max().item() <= 1e-5
        assert np.allclose(_checksum(mod.weight_v, "v1." + k), g[f"v1_checksum.{k}"], rtol=1e-4, atol=1e-5)


PCFG = {"bce_loss": False, "gp_weight": 0.0}


def _nchw2(f):
    return f.t[:, :f.C].float().reshape(f.N, f.dhw[1], f.dhw[2], f.C).permute(0, 3, 1, 2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_patch_discriminator(golden, dtype):
    """2-D PatchGAN (patchgan.py:368-470) against golden G9: prediction map, feature maps (16 / 8 / 7 pixels), hinge loss with
    every parameter gradient, generator-side gradient w.r.t. the fake frames, train-mode power iterations."""
    from ipoke_amd.discriminator import PatchDiscriminator
    g = golden("g9_patch_disc_64")
    tol = TOL[dtype]
    m = PatchDiscriminator(PCFG, dtype=dtype)
    deterministic_fill_(m, prefix="disc_s.")
    m = m.to(DEV).eval()
    xt, xf = t(g["x_true"], DEV), t(g["x_fake"], DEV)
    pf, _ = m(xf)
    pt, fm = m(xt)
    scale = max(1.0, float(np.abs(g["pred_true"]).max()))
    e = max((pf.cpu() - t(g["pred_fake"])).abs().max().item(), (pt.cpu() - t(g["pred_true"])).abs().max().item())
    print(f"[{dtype}] patch pred {tuple(pt.shape)} err {e:.2e} (|pred| <= {scale:.2f})")
    assert tuple(pt.shape) == tuple(g["pred_true"].shape) and e <= max(tol["pred"], 2e-5 if dtype == "f32" else 2e-2) * scale
    for i, f in enumerate(fm):
        full = _nchw2(f)
        want = t(g[f"fmap{i}_slice"])
        err = (full[:, :4, :3, :3].cpu() - want).abs().max().item()
        cs, ws = _checksum(full, f"fmap{i}"), g[f"fmap{i}_checksum"]
        assert err <= tol["fmap"] * max(1.0, want.abs().max().item()), (i, err)
        assert abs(cs[1] - ws[1]) <= (2e-4 if dtype == "f32" else 1e-2) * ws[1]
    loss = (m.loss(pf, real=False) + m.loss(pt, real=True)) / 2.0
    assert abs(loss.item() - float(g["loss_d"])) <= max(tol["loss"], 1e-5 if dtype == "f32" else 1e-2) * max(1.0, abs(float(g["loss_d"])))
    loss.backward()
    grads = dict(m.named_parameters())
    worst = 0.0
    for k, want in zip(g["grad_names"], g["grad_checksums"]):
        got = _checksum(grads[str(k)].grad, str(k))
        if want[1] < 1e-6:            # biases in front of an InstanceNorm: the exact gradient is zero, only round-off remains
            assert got[1] <= (2e-3 if dtype == "f32" else 5e-2), (k, got, want)
            continue
        rel = abs(got[1] - want[1]) / max(want[1], 1e-12)
        worst = max(worst, rel)
        assert rel <= tol["grad"], (k, got, want)
    print(f"[{dtype}] patch worst parameter-gradient abs-sum deviation {worst:.2e}")
    xg = xf.clone().requires_grad_(True)
    pg, ff = m(xg)
    with torch.no_grad():
        _, ft = m(xt)
    lg = -pg.mean() + m.fmap_loss(ff, ft)
    assert abs(lg.item() - float(g["loss_g"])) <= max(tol["loss"], 1e-5 if dtype == "f32" else 1e-2) * max(1.0, abs(float(g["loss_g"])))
    lg.backward()
    want = t(g["dx_fake_slice"])
    err = (xg.grad[:, :, :6, :6].cpu() - want).abs().max().item()
    cs, ws = _checksum(xg.grad, "dx_fake"), g["dx_fake_checksum"]
    print(f"[{dtype}] patch d loss_g / d x_fake slice err {err:.2e} (max {want.abs().max():.2e}); abs-sum {cs[1]:.4e} vs {ws[1]:.4e}")
    assert err <= tol["dx"] * want.abs().max().item() and abs(cs[1] - ws[1]) <= tol["dx_sum"] * ws[1]
    m.train()
    with torch.no_grad():
        ptr_, _ = m(xt)
    assert (ptr_.cpu() - t(g["pred_true_train"])).abs().max().item() <= max(tol["pred"], 2e-5 if dtype == "f32" else 2e-2) * scale
    assert (m.in_conv.weight_u.cpu() - t(g["u1.in_conv"])).abs().max().item() <= 1e-5
    assert (m.out_conv.weight_u.cpu() - t(g["u1.out_conv"])).abs().max().item() <= 1e-5
