"""FVD evaluation on the GPU (ipoke_amd/fvd.py + csrc/eval.hip, reference utils/metrics.py:622-1099) against golden G10, which
holds the reference I3D's logits, intermediate maps, activation moments and calculate_FVD values for 64x64 clips of 16 and 15
frames; plus unit tests of the element-wise kernels against the oracle / torch."""
import zlib

import numpy as np
import pytest
import torch

from ipoke_amd import _lib, fvd
from ipoke_amd.nn import CL
from ipoke_amd.utils.detfill import deterministic_fill_
from oracle import fvd_ref
from tests.conftest import t

pytestmark = pytest.mark.gpu
DEV = "cuda"
# f32: fp32 matrix-core GEMMs (different summation order than the reference's CPU convolutions) through 58 conv layers, logits
# of magnitude ~15; bf16: bf16 weights and activations through the same stack.
# measured on MI355X: f32 logits 2.3e-5, maps 1.5e-6 of their maximum; bf16 logits 7.7e-2, maps 7e-3 of their maximum
TOL = {"f32": dict(logits=1e-4, tap=1e-5, fvd=2e-3), "bf16": dict(logits=0.2, tap=2e-2, fvd=None)}


def _checksum(x, key):
    x = x.detach().double().flatten().cpu()
    idx = torch.randint(0, x.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(key.encode())))
    return np.array([x.sum().item(), x.abs().sum().item(), *x[idx].tolist()])


_models = {}


def _model(dtype):
    if dtype not in _models:
        m = fvd.I3D(400, "rgb", dtype=dtype, device="cpu")
        deterministic_fill_(m, prefix="i3d.")
        _models[dtype] = m.to(DEV)
    return _models[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("T", [16, 15])
def test_i3d_logits(golden, dtype, T):
    g = golden("g10_fvd")
    m = _model(dtype)
    for which in ("orig", "gen"):
        vids = t(g[f"videos_{which}"]).float()[:, 16 - T:].to(DEV)
        minval = fvd._resized_min(vids)
        assert minval.item() < 0
        got = fvd._activations(m, vids, 3, minval, resize=(224, 224)).cpu().double()
        want = t(g[f"logits_{which}_T{T}"])
        err = (got - want).abs().max().item()
        print(f"[{dtype}] T={T} {which}: logits err {err:.2e} (|logits| <= {want.abs().max():.2f})")
        assert err <= TOL[dtype]["logits"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_i3d_intermediate_maps(golden, dtype):
    g = golden("g10_fvd")
    m = _model(dtype)
    vids = t(g["videos_orig"]).float().to(DEV)
    x = fvd_ref.preprocess(vids.cpu())[:1].to(DEV)                  # [1, T, 3, 224, 224]
    taps = {}
    m(x.permute(0, 2, 1, 3, 4), taps)                                # the reference-signature entry on a permuted view, as get_activations passes it
    for k, v in taps.items():
        want = t(g[f"tap_{k}_slice"])
        err = (v[0, :6, :2, :5, :5].cpu() - want).abs().max().item()
        cs, ws = _checksum(v, k), g[f"tap_{k}_checksum"]
        print(f"[{dtype}] {k} {tuple(v.shape)}: slice err {err:.2e} (max {want.abs().max():.2f}); abs-sum {cs[1]:.5e} vs {ws[1]:.5e}")
        assert err <= TOL[dtype]["tap"] * max(1.0, want.abs().max().item())
        assert abs(cs[1] - ws[1]) <= (1e-4 if dtype == "f32" else 1e-2) * ws[1]


def test_fvd_value(golden):
    """calculate_FVD end to end (streaming resize + I3D + float64 moments on the device, sqrtm on the host) vs the reference."""
    g = golden("g10_fvd")
    m = _model("f32")
    for T in (16, 15):
        vg, vo = t(g["videos_gen"]).float()[:, 16 - T:], t(g["videos_orig"]).float()[:, 16 - T:]
        val = fvd.calculate_FVD(m, vg, vo, batch_size=3)
        want = float(g[f"fvd_T{T}"])
        print(f"T={T}: FVD {val:.6f} vs reference {want:.6f}")
        assert abs(val - want) <= TOL["f32"]["fvd"] * abs(want)
        mu, sigma = fvd._moments_device(fvd._activations(m, vg.to(DEV), 3, fvd._resized_min(vg), resize=(224, 224)))
        assert np.abs(mu - g[f"mu_gen_T{T}"]).max() <= 5e-4
        assert abs(_checksum(torch.from_numpy(sigma), "sigma")[1] - g[f"sigma_gen_checksum_T{T}"][1]) <= 2e-3 * g[f"sigma_gen_checksum_T{T}"][1]
    # the metric object of the validation loop
    metric = fvd.FVD(n_samples=6, i3d=m)
    vg, vo = t(g["videos_gen"]).float(), t(g["videos_orig"]).float()
    metric.update(vg[:3], vo[:3]); metric.update(vg[3:], vo[3:])
    # per-update de-normalisation decisions coincide here (every half contains negative pixels)
    assert abs(metric.compute() - float(g["fvd_T16"])) <= TOL["f32"]["fvd"] * float(g["fvd_T16"])


def test_preprocess_matches_interpolate():
    gen = torch.Generator().manual_seed(5)
    for lo in (-1.0, 0.0):                                  # with and without negative values -> with and without (x + 1) / 2
        v = torch.rand(2, 3, 3, 40, 56, generator=gen) * (1.0 - lo) + lo
        want = fvd_ref.preprocess(v)
        a, b = fvd.preprocess(v.to(DEV), v.to(DEV))
        assert a.shape == want.shape
        assert (a.cpu() - want).abs().max().item() <= 1e-5 and torch.equal(a, b)
    # identity size: exact copy into padded channels-last rows, zero border
    x = torch.randn(2, 3, 4, 6, 10, generator=gen).to(DEV)
    s = x.stride()
    dst = torch.full((2 * 4 * 6 * 15, 3), 7.0, device=DEV)
    mn = torch.empty(1, device=DEV)
    _lib.check(_lib.lib().ipoke_min_reset(_lib.ptr(mn), _lib.current_stream()))
    _lib.check(_lib.lib().ipoke_video_to_cl(_lib.ptr(x), s[0], s[2], s[1], s[3], s[4], 2, 4, 3, 6, 10, _lib.ptr(dst), 6, 10, 2, 3, _lib.ptr(mn),
                                            _lib.current_stream()))
    d = dst.view(2, 4, 6, 15, 3)
    assert torch.equal(d[:, :, :, 2:12], x.permute(0, 2, 3, 4, 1)) and d[:, :, :, :2].abs().max() == 0 and d[:, :, :, 12:].abs().max() == 0
    assert mn.item() == x.min().item()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("k,s,dhw", [((1, 3, 3), (1, 2, 2), (4, 14, 14)), ((3, 3, 3), (2, 2, 2), (8, 12, 12)), ((3, 3, 3), (2, 2, 2), (7, 9, 9)),
                                     ((2, 2, 2), (2, 2, 2), (4, 14, 14)), ((2, 2, 2), (2, 2, 2), (3, 7, 7)), ((3, 3, 3), (1, 1, 1), (4, 7, 7))])
def test_pool_same(dtype, k, s, dhw):
    """MaxPool3dTFPadding semantics incl. NEGATIVE inputs (the zero border then wins) and ceil_mode overhang."""
    m = fvd.I3D(400, "rgb", dtype=dtype, device="cpu")
    gen = torch.Generator().manual_seed(11)
    C = 24
    x = torch.randn(2, C, *dhw, generator=gen) - 0.5
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    x = x.to(td).float()
    want = fvd_ref.pool_same(x, k, s)
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, C).to(td).to(DEV).contiguous()
    y = m._pool(CL(rows, 2, dhw, C), k, s)
    got = y.t.float().reshape(2, *y.dhw, C).permute(0, 4, 1, 2, 3).cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.equal(got, want)


def test_activation_moments():
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(37, 400, generator=gen) * 3 + torch.randn(400, generator=gen)
    a[5] = float("nan")                                     # a row without a single finite entry is dropped
    mu, sigma = fvd.calculate_moments(a.numpy())
    wmu, wsig = fvd_ref.moments(a.double().numpy())
    assert np.abs(mu - wmu).max() <= 1e-12 and np.abs(sigma - wsig).max() <= 1e-11


def test_validation_loop_fvd():
    """validation_step / validation_epoch_end (second_stage_video.py:490-584) on a reduced second-stage model: the FVD the loop
    logs equals the oracle's FVD of the very clips the loop collected (generated on the device and kept there)."""
    from ipoke_amd import configs
    from ipoke_amd.second_stage import PokeMotionModel
    from tests.helpers import synthetic_batch
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=3, arch=arch)
    conf["logging"]["n_fvd_samples"] = 6
    conf["first_stage"]["logging"]["bs_i3d"] = 3
    model = PokeMotionModel(conf, dirs={}, dtype="f32", device=DEV, max_batch=3)
    for name in ("first_stage_model", "poke_embedder", "conditioner", "flow"):
        deterministic_fill_(getattr(model, name), prefix=name + ".")
    model.flow.sync_buffers()
    model.attach_fvd(i3d=_model("f32"))
    kept = {}
    for i in range(2):
        batch = synthetic_batch(3, 16, 64, seed=20 + i, device=DEV)
        out = model.validation_step(batch, i)
        assert torch.isfinite(out["loss"]).all() and "val/nll_loss" in model.logged
        # ssim-val / psnr-val of this batch (second_stage_video.py:511-512) against the oracle on the clips the step kept
        from oracle import metrics_ref
        fake, true = model._fvd_fake[-1].cpu(), model._fvd_true[-1].cpu()
        fake, true = fake.reshape(-1, *fake.shape[2:]), true.reshape(-1, *true.shape[2:])
        assert abs(float(model.logged["ssim-val"]) - metrics_ref.ssim(fake, true).item()) <= 2e-5
        assert abs(float(model.logged["psnr-val"]) - metrics_ref.psnr(fake, true).item()) <= 1e-3
    kept = [torch.cat(x).cpu() for x in (model._fvd_fake, model._fvd_true, model._fvd_fake_x0, model._fvd_true_x0)]
    assert kept[0].shape == (6, 15, 3, 64, 64) and kept[3].shape == (6, 16, 3, 64, 64)
    fvd_val, fvd_x0 = model.validation_epoch_end()
    o = fvd_ref.I3D(400)
    deterministic_fill_(o, prefix="i3d.")
    o.eval()
    want, want_x0 = fvd_ref.fvd(o, kept[0], kept[1], 3), fvd_ref.fvd(o, kept[2], kept[3], 3)
    print(f"FVD-val {fvd_val:.5f} (oracle {want:.5f}); FVD-val-x0 {fvd_x0:.5f} (oracle {want_x0:.5f})")
    assert abs(fvd_val - want) <= 2e-3 * abs(want) and abs(fvd_x0 - want_x0) <= 2e-3 * abs(want_x0)
    assert model.logged["FVD-val"] == fvd_val and not model._fvd_fake


def test_first_stage_validation_loop():
    """SpadeCondMotionModel.validation_step / validation_epoch_end (first_stage_motion_model.py:303-367): rec_loss and the two FVDs
    of the reconstructions against the oracle's values for the same clips."""
    from ipoke_amd import configs
    from ipoke_amd.first_stage import SpadeCondMotionModel
    conf = configs.first_stage_config(64, 32, 16)
    conf["logging"].update(bs_i3d=3, n_samples_fvd=6)
    model = SpadeCondMotionModel(conf, dirs={}, dtype="f32").to(DEV).eval()
    deterministic_fill_(model, prefix="first_stage.")
    model.attach_fvd(i3d=_model("f32"))
    kept_hat, kept_x, ssims, psnrs = [], [], [], []
    for i in range(2):
        X = (torch.rand(3, 16, 3, 64, 64, generator=torch.Generator().manual_seed(40 + i)) * 2 - 1).to(DEV)
        X_hat = model.validation_step({"images": X}, i)
        want = (X[:, 1:] - X_hat).abs().mean().item()
        assert abs(model.logged["val/rec_loss"].item() - want) <= 1e-5 * max(1.0, want)
        from oracle import metrics_ref
        fake, true = X_hat.cpu().reshape(-1, *X_hat.shape[2:]), X[:, 1:].cpu().reshape(-1, *X_hat.shape[2:])
        ssims.append(metrics_ref.ssim(fake, true).item()); psnrs.append(metrics_ref.psnr(fake, true).item())
        assert abs(float(model.logged["ssim-val_step"]) - ssims[-1]) <= 2e-5
        assert abs(float(model.logged["psnr-val_step"]) - psnrs[-1]) <= 1e-3
        # logged with on_epoch=True (first_stage_motion_model.py:323-324): the epoch value is the mean over the batches so far
        assert abs(float(model.logged["ssim-val"]) - sum(ssims) / len(ssims)) <= 2e-5
        assert abs(float(model.logged["psnr-val"]) - sum(psnrs) / len(psnrs)) <= 1e-3
        kept_hat.append(X_hat.cpu()); kept_x.append(X.cpu())
    fvd_val, fvd_x0 = model.validation_epoch_end()
    o = fvd_ref.I3D(400)
    deterministic_fill_(o, prefix="i3d.")
    o.eval()
    Xh, Xt = torch.cat(kept_hat), torch.cat(kept_x)
    want, want_x0 = fvd_ref.fvd(o, Xh, Xt[:, 1:], 3), fvd_ref.fvd(o, torch.cat([Xt[:, :1], Xh], 1), Xt, 3)
    print(f"first stage: FVD-val {fvd_val:.5f} (oracle {want:.5f}); FVD-val-x0 {fvd_x0:.5f} (oracle {want_x0:.5f})")
    assert abs(fvd_val - want) <= 2e-3 * abs(want) and abs(fvd_x0 - want_x0) <= 2e-3 * abs(want_x0)


@pytest.mark.parametrize("T", [9, 32])
def test_i3d_other_clip_lengths(T):
    """Clip lengths the goldens do not hold: 9 frames (odd remainders on the time axis at every strided layer) and 32 frames (three
    windows of the (2, 7, 7) average pool, i.e. the head's weighted time mean with unequal frame weights) against the oracle's I3D."""
    m = _model("f32")
    o = fvd_ref.I3D(400)
    deterministic_fill_(o, prefix="i3d.")
    o.eval()
    vids = torch.rand(2, T, 3, 48, 48, generator=torch.Generator().manual_seed(T)) * 2 - 1
    with torch.no_grad():
        want = o(fvd_ref.preprocess(vids).permute(0, 2, 1, 3, 4))
    got = fvd._activations(m, vids.to(DEV), 2, fvd._resized_min(vids.to(DEV)), resize=(224, 224)).cpu()
    err = (got - want).abs().max().item()
    print(f"T={T}: logits err {err:.2e} (|logits| <= {want.abs().max():.2f})")
    assert err <= 1e-4
