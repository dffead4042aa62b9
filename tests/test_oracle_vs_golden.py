"""CPU: the oracle (oracle/flow_ref.py, oracle/vae_ref.py) against the golden vectors generated from the REFERENCE's own
modules (oracle/make_goldens.py).  This is what pins the oracle; tolerances are those of SURVEY.md §8c."""
import copy

import numpy as np
import pytest
import torch

from ipoke_amd import configs
from ipoke_amd.utils.detfill import deterministic_fill_, fill_value
from oracle import flow_ref, vae_ref
from tests.conftest import t


@pytest.mark.parametrize("C", [8, 32])
def test_unit_layers(golden, C):
    g = golden("g1_flow_units")
    x, h = t(g[f"x_{C}"]), t(g[f"h_{C}"])
    # ActNorm: data-dependent init from the reference's pre-init draw, forward, inverse
    an = flow_ref.ActNorm2dFlow(C)
    with torch.no_grad():
        an.log_scale.copy_(t(g[f"actnorm_{C}_pre_log_scale"]))
    y, ld = an(t(g[f"actnorm_{C}_init_x"]))
    assert (an.log_scale - t(g[f"actnorm_{C}_post_log_scale"])).abs().max() < 1e-6
    assert (y - t(g[f"actnorm_{C}_y"])).abs().max() < 1e-6 and (ld - t(g[f"actnorm_{C}_logdet"])).abs().max() < 1e-4
    assert (an(y, reverse=True) - t(g[f"actnorm_{C}_inv"])).abs().max() < 1e-6
    # Shuffle: bit exact, indices are int64 buffers, argsort relation
    sh = flow_ref.Shuffle(C)
    deterministic_fill_(sh, prefix=f"shuffle{C}.")
    assert sh.forward_shuffle_idx.dtype == torch.int64
    assert torch.equal(sh.forward_shuffle_idx, t(g[f"shuffle_{C}_fwd_idx"]))
    assert torch.equal(sh.backward_shuffle_idx, torch.argsort(sh.forward_shuffle_idx))
    ys, zero = sh(x)
    assert zero == 0 and torch.equal(ys, t(g[f"shuffle_{C}_y"])) and torch.equal(sh(ys, reverse=True), x)
    # affine transform
    mu, sc = flow_ref.affine_params(t(g[f"affine_{C}_raw"]))
    ya, la = flow_ref.affine_fwd(x, mu, sc)
    assert (ya - t(g[f"affine_{C}_y"])).abs().max() < 1e-6 and (la - t(g[f"affine_{C}_logdet"])).abs().max() < 1e-4
    assert (flow_ref.affine_inv(ya, mu, sc) - t(g[f"affine_{C}_inv"])).abs().max() < 1e-6
    for order, ks in (("A", (2, 3)), ("B", (2, 3)), ("C", (3, 2)), ("D", (3, 2))):
        sconv = flow_ref.ShiftedConv2d(C, 4 * C, ks, order)
        deterministic_fill_(sconv, prefix=f"sc{C}{order}.")
        assert (sconv(x) - t(g[f"shiftconv_{C}_{order}"])).abs().max() < 1e-5
        m = flow_ref.MaskedConvFlow(C, ks, order, 128)
        deterministic_fill_(m, prefix=f"mcf{C}{order}.")
        xg = x.clone().requires_grad_(True)
        ym, lm = m(xg, h=h)
        assert (ym - t(g[f"mcf_{C}_{order}_y"])).abs().max() < 1e-5
        assert (lm - t(g[f"mcf_{C}_{order}_logdet"])).abs().max() < 1e-4
        assert (m(ym.detach(), h=h, reverse=True) - t(g[f"mcf_{C}_{order}_inv"])).abs().max() < 1e-5
        (0.5 * (ym ** 2).sum() - lm.sum()).backward()
        for name, grad in (("dx", xg.grad), ("dshift", m.net.shift_conv.weight.grad), ("dv", m.net.conv1x1.conv.weight_v.grad),
                           ("dg", m.net.conv1x1.conv.weight_g.grad), ("db", m.net.conv1x1.conv.bias.grad)):
            ref = t(g[f"mcf_{C}_{order}_{name}"])
            assert (grad - ref).abs().max() <= 1e-4 * (1 + ref.abs().max()), (order, name)
        # autoregressive property: output row i of order A does not depend on input rows >= i
        if order == "A":
            x2 = x.clone(); x2[:, :, 4:] += 1.0
            y2, _ = m(x2, h=h)
            # the affine transform multiplies x itself, so compare the *parameters* through rows above the change
            assert torch.allclose((y2 - ym)[:, :, :4], torch.zeros_like(ym[:, :, :4]), atol=1e-6)
    for split in ("continuous", "skip"):
        for order in ("up", "down"):
            tag = f"nice_{C}_{split}_{order}"
            n = flow_ref.NICE2d(C, 64, split, order)
            deterministic_fill_(n, prefix=tag + ".")
            yn, ln = n(x)
            assert (yn - t(g[tag + "_y"])).abs().max() < 1e-5 and (ln - t(g[tag + "_logdet"])).abs().max() < 1e-4
            assert (n(yn, reverse=True) - t(g[tag + "_inv"])).abs().max() < 1e-5
    pr = flow_ref.MultiScalePrior(C, 64, 4)
    deterministic_fill_(pr, prefix=f"prior{C}.")
    yp, lp = pr(x, h=h)
    assert (yp - t(g[f"prior_{C}_y"])).abs().max() < 1e-5 and (lp - t(g[f"prior_{C}_logdet"])).abs().max() < 1e-4
    assert (pr(yp, h=h, reverse=True) - t(g[f"prior_{C}_inv"])).abs().max() < 1e-5


def test_reduced_flow_full_topology(golden):
    g = golden("g2_reduced_flow")
    o = flow_ref.SupervisedMacowTransformer(configs.reduced_flow_arch())
    deterministic_fill_(o, prefix="flow.")
    x, cond = t(g["x"]), t(g["cond"])
    out, logdet = o(x, cond)
    assert (out - t(g["out"])).abs().max() <= 2e-5 and (logdet - t(g["logdet"])).abs().max() <= 1e-3
    assert (o(out.detach(), cond, reverse=True) - t(g["reverse"])).abs().max() <= 5e-5
    torch.manual_seed(1234)
    loss, log = flow_ref.FlowLoss()(out, logdet)
    assert abs(loss.item() - float(g["loss"])) <= 1e-3
    assert abs(log["reference_nll_loss"].item() - float(g["reference_nll_loss"])) <= 1e-4      # same RNG consumption
    loss.backward()
    for k, p in o.named_parameters():
        ref = t(g["grad." + k])
        assert (p.grad - ref).abs().max() <= 1e-4 * (ref.abs().max() + 1e-6), k


def _condition_nice_arch():
    arch = configs.reduced_flow_arch()
    arch["condition_nice"] = True
    arch["h_channels"] = 32
    return arch


def test_condition_nice_flow(golden):
    """condition_nice (macow2.py:1024-1060, 553; macow_utils.py:275-283, 328-332): every NICE net sees h behind conv2 -- the
    reference's activations, reverse pass, loss and every parameter gradient (G16)."""
    g = golden("g16_condition_nice")
    o = flow_ref.SupervisedMacowTransformer(_condition_nice_arch())
    assert o.state_dict()["flow.layers.0.0.coupling1_up.net.conv3.conv.weight_v"].shape[1] == 64 + 32
    assert o.state_dict()["flow.priors.0.coupling.net.conv3.conv.weight_v"].shape[1] == 64 + 32
    deterministic_fill_(o, prefix="flow.")
    x, cond = t(g["x"]), t(g["cond"])
    out, logdet = o(x, cond)
    assert (out - t(g["out"])).abs().max() <= 2e-5 and (logdet - t(g["logdet"])).abs().max() <= 1e-3
    assert (o(out.detach(), cond, reverse=True) - t(g["reverse"])).abs().max() <= 5e-5
    loss = (0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()
    assert abs(loss.item() - float(g["loss"])) <= 1e-3
    loss.backward()
    for k, p in o.named_parameters():
        ref = t(g["grad." + k])
        assert (p.grad - ref).abs().max() <= 1e-4 * (ref.abs().max() + 1e-6), k
    with torch.no_grad():                     # the conditioning columns matter
        assert (o(x, torch.zeros_like(cond))[0] - out).abs().max() > 1e-3


def test_lu_conv_unit_and_flow(golden):
    """InvertibleConvLU1d (macow2.py:596-649): the reference's unit golden, and the reduced flow with use1x1 (G2-LU)."""
    g = golden("g1_flow_units")
    lu = flow_ref.InvertibleConvLU1d(8)
    with torch.no_grad():
        for k in ("permutated", "sign_s", "l", "u", "log_s"):
            getattr(lu, k).copy_(t(g["lu_8_" + k]))
    x = t(g["x_8"])
    y, ld = lu(x)
    assert (y - t(g["lu_8_y"])).abs().max() <= 1e-6 and (ld - t(g["lu_8_logdet"])).abs().max() <= 1e-5
    assert (lu(y, reverse=True) - t(g["lu_8_inv"])).abs().max() <= 1e-5
    g = golden("g2_reduced_flow_lu")
    arch = configs.reduced_flow_arch(); arch["use1x1"] = True
    o = flow_ref.SupervisedMacowTransformer(arch)
    deterministic_fill_(o, prefix="flow.")
    o.load_state_dict({k[3:]: t(v) for k, v in g.items() if k.startswith("lu.")}, strict=False)
    x, cond = t(g["x"]), t(g["cond"])
    out, logdet = o(x, cond)
    assert (out - t(g["out"])).abs().max() <= 2e-5 and (logdet - t(g["logdet"])).abs().max() <= 1e-3
    assert (o(out.detach(), cond, reverse=True) - t(g["reverse"])).abs().max() <= 5e-5
    (0.5 * (out ** 2).sum(dim=[1, 2, 3]).mean() - logdet.mean()).backward()
    for k, p in o.named_parameters():
        ref = t(g["grad." + k])
        assert (p.grad - ref).abs().max() <= 1e-4 * (ref.abs().max() + 1e-6), k


def test_identity_at_init_and_data_init(golden):
    """Known answers: after the initialising forward every coupling is the identity; logdet = 64 * sum log_scale."""
    g = golden("g2_reduced_flow_init")
    o = flow_ref.SupervisedMacowTransformer(configs.reduced_flow_arch())
    sd = o.state_dict()
    big = set(g["big_keys"].tolist())
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(fill_value("flow." + k, v) if k in big else t(g["pre." + k]))
    with torch.no_grad():
        out, logdet = o(t(g["x"]), t(g["cond"]))
    assert (out - t(g["out_filled"])).abs().max() <= 1e-5 and (logdet - t(g["logdet_filled"])).abs().max() <= 1e-3
    assert (logdet - logdet[0]).abs().max() < 1e-4                       # batch independent
    ls_sum = sum(v.sum() for k, v in o.state_dict().items() if k.endswith("log_scale"))
    assert abs(float(ls_sum) * 64 - float(logdet[0])) <= 1e-2
    for k, v in o.state_dict().items():
        if k.endswith("weight_g"):
            assert float(v.abs().max()) == 0.0
        if k.endswith("initialized"):
            assert int(v) == 1


def test_flow_loss_known_answers():
    fl = flow_ref.FlowLoss()
    z = torch.zeros(3, 4, 8, 8)
    assert fl(z, torch.zeros(3))[0].item() == 0.0
    assert abs(fl(torch.ones(3, 4, 8, 8), torch.zeros(3))[0].item() - 0.5 * 4 * 64) < 1e-4
    with pytest.raises(AssertionError):
        fl(z, torch.zeros(3, 1))                       # reference asserts len(logdet.shape) == 1 (loss.py:15)
    # spatial_mean (loss.py:14-20, 75-77): both terms divided by h * w; known answers from the reference's FlowLoss
    sm = flow_ref.FlowLoss(spatial_mean=True, logdet_weight=0.5)
    assert abs(sm(torch.ones(3, 4, 8, 8), torch.full((3,), 6.4))[0].item() - (0.5 * 4 - 0.5 * 6.4 / 64)) < 1e-5


def test_lr_schedule(golden):
    g = golden("g6_glue_64")
    for it, lr in zip(g["lr_its"], g["lr_vals"]):
        assert abs(flow_ref.lr_at(int(it)) - float(lr)) < 1e-12


def test_motion_encoder_and_decoder(golden):
    g4, g5 = golden("g4_encoder_64"), golden("g5_decoder_64")
    m = vae_ref.SpadeCondMotionModel(configs.first_stage_config(64, 32, 16)).eval()
    deterministic_fill_(m, prefix="first_stage.")
    with torch.no_grad():
        z, mu, lv = m.enc_motion(t(g4["X"]).transpose(1, 2), eps=t(g4["eps"]))
        frames = m.decode(t(g5["z"]), t(g5["x0"]), 3)
    assert (mu - t(g4["mu"])).abs().max() <= 2e-5 and (lv - t(g4["logvar"])).abs().max() <= 2e-5
    assert (z - t(g4["z"])).abs().max() <= 2e-5
    assert (frames - t(g5["frames"])).abs().max() <= 5e-5


def test_spectral_norm_train_mode_power_iteration(golden):
    g = golden("g5_spectral_train")
    m = vae_ref.SpadeCondMotionModel(configs.first_stage_config(64, 32, 16)).eval()
    deterministic_fill_(m, prefix="first_stage.")
    blk = m.gen.blocks[0].conv1
    assert torch.allclose(blk.conv.weight_u, t(g["u0"]))
    blk.conv.spectral_power_iter()
    assert (blk.conv.weight_u - t(g["u1"])).abs().max() < 1e-5 and (blk.conv.weight_v - t(g["v1"])).abs().max() < 1e-5
    with torch.no_grad():
        y = blk(t(g["x"]))
    assert (y - t(g["y"])).abs().max() <= 2e-5 * max(1.0, float(np.abs(g["y"]).max()))


def test_first_stage_l1_kl_training_slice(golden):
    g = golden("g5_first_stage_train_64")
    m = vae_ref.SpadeCondMotionModel(configs.first_stage_config(64, 32, 4)).eval()
    deterministic_fill_(m, prefix="first_stage.")
    X = t(g["X"])
    Xh, mu, lv = m(X, eps=t(g["eps"]))
    loss = vae_ref.first_stage_loss(X, Xh, mu, lv)
    assert (Xh - t(g["X_hat"])).abs().max() <= 5e-5 and abs(loss.item() - float(g["loss"])) <= 1e-4


def test_first_stage_train_mode_step_128(golden):
    """The oracle in TRAIN mode (one power iteration per decoder call: 15 sigma per weight per step) against the reference's own
    train-mode step at the c4 size (golden g13: 128x128, z = 32, T = 16): reconstruction, loss, u buffers after the step, and the
    checksums of a few gradient tensors."""
    g = golden("g13_first_stage_train_mode_128")
    m = vae_ref.SpadeCondMotionModel(configs.first_stage_config(128, 32, 16)).train()
    deterministic_fill_(m, prefix="first_stage.")
    X = torch.rand(1, 16, 3, 128, 128, generator=torch.Generator().manual_seed(int(g["X_seed"]))) * 2 - 1
    Xh, mu, lv = m(X, eps=t(g["eps"]))
    loss = vae_ref.first_stage_loss(X, Xh, mu, lv)
    assert (Xh[:, [0, 7, 14]] - t(g["X_hat_frames"])).abs().max() <= 1e-4 and abs(loss.item() - float(g["loss"])) <= 1e-4
    sd = m.state_dict()
    for k in g["u_names"].tolist():
        assert (sd[k] - t(g["u." + k])).abs().max() <= 1e-5, k
    loss.backward()
    names = g["grad_names"].tolist()
    grads = dict(m.named_parameters())
    for k in ("gen.out_conv.conv.weight", "gen.in_block.conv1.conv.weight_orig", "rnn.cells.0.out_gate.weight", "enc_motion.conv1.weight"):
        cs, ref = _checksum(grads[k].grad, k), g["grad_checksums"][names.index(k)]
        assert abs(cs[0] - ref[0]) <= 1e-2 * ref[1] and abs(cs[1] - ref[1]) <= 1e-2 * ref[1], (k, cs, ref)


def test_adapt_cond_oracle(golden):
    """The conditioner with a 4x4 latent + the transposed adapter block (second_stage_video.py:120-129) from the oracle's blocks
    against the reference's make_flow_input (golden g14)."""
    g = golden("g14_adapt_cond_64")
    ccfg = configs.encoder2d_config(64, 3)
    ccfg["architecture"]["min_spatial_size"] = 4
    oc = vae_ref.FirstStageWrapper(ccfg).eval()
    ob = vae_ref.Conv2dTransposeBlock(64, 64, 3, 2, 1, norm="none", activation="elu", snorm=False)
    deterministic_fill_(oc, prefix="conditioner."); deterministic_fill_(ob, prefix="conv_adapt_cond.")
    X0 = torch.rand(2, 16, 3, 64, 64, generator=torch.Generator().manual_seed(int(g["batch_seed"])))[:, 0] * 2 - 1
    with torch.no_grad():
        lat = oc.encoder(X0)[0]
        cond = ob(lat)
    assert (lat - t(g["cond_latent_4x4"])).abs().max() <= 2e-5 and (cond - t(g["cond"])[:, :64]).abs().max() <= 2e-5


def _checksum(x, key):
    import zlib
    x = x.detach().double().flatten().cpu()
    idx = torch.randint(0, x.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(key.encode())))
    return np.array([x.sum().item(), x.abs().sum().item(), *x[idx].tolist()])


def test_temporal_discriminator(golden):
    """G8: oracle/disc_ref.py against the reference's patchgan_3d.resnet outputs (predictions, feature maps, hinge loss and
    every parameter gradient, gradient penalty, generator-side loss and input gradient)."""
    from oracle import disc_ref
    g = golden("g8_temporal_disc_64")
    cfg = {"bce_loss": False, "gp_weight": 1.0, "num_classes": 1, "patch_temp_disc": False}
    o = disc_ref.TemporalDiscriminator(64, cfg)
    deterministic_fill_(o, prefix="disc_t.")
    o.eval()
    Xt, Xf = t(g["X_true"]), t(g["X_fake"])
    xt = Xt.clone().requires_grad_(True)
    pf, _ = o(Xf)
    pt, fm = o(xt)
    assert (pf - t(g["pred_fake"])).abs().max() <= 2e-5 and (pt - t(g["pred_true"])).abs().max() <= 2e-5
    for i, f in enumerate(fm):
        assert (f[:, :4, :, :3, :3] - t(g[f"fmap{i}_slice"])).abs().max() <= 5e-5
        assert np.allclose(_checksum(f, f"fmap{i}"), g[f"fmap{i}_checksum"], rtol=1e-4, atol=1e-3)
    loss = (o.loss(pf, real=False) + o.loss(pt, real=True)) / 2.0
    gp = o.gp2(pt, xt)
    assert abs(loss.item() - float(g["loss_d"])) <= 1e-5 and abs(gp.item() - float(g["gp"])) <= 1e-4 * float(g["gp"])
    loss.backward()
    grads = dict(o.named_parameters())
    for k, want in zip(g["grad_names"], g["grad_checksums"]):
        got = _checksum(grads[str(k)].grad, str(k))
        assert np.allclose(got, want, rtol=2e-3, atol=1e-5 * max(1.0, abs(want[1]))), k


def test_patch_discriminator(golden):
    """G9: oracle/disc_ref.PatchDiscriminator against the reference's 2-D PatchGAN outputs."""
    from oracle import disc_ref
    g = golden("g9_patch_disc_64")
    o = disc_ref.PatchDiscriminator({"bce_loss": False, "gp_weight": 0.0})
    deterministic_fill_(o, prefix="disc_s.")
    o.eval()
    pf, _ = o(t(g["x_fake"]))
    pt, fm = o(t(g["x_true"]))
    assert (pf - t(g["pred_fake"])).abs().max() <= 2e-5 and (pt - t(g["pred_true"])).abs().max() <= 2e-5
    for i, f in enumerate(fm):
        assert (f[:, :4, :3, :3] - t(g[f"fmap{i}_slice"])).abs().max() <= 5e-5
    loss = (o.loss(pf, real=False) + o.loss(pt, real=True)) / 2.0
    assert abs(loss.item() - float(g["loss_d"])) <= 1e-5
    loss.backward()
    grads = dict(o.named_parameters())
    for k, want in zip(g["grad_names"], g["grad_checksums"]):
        got = _checksum(grads[str(k)].grad, str(k))
        assert np.allclose(got, want, rtol=2e-3, atol=1e-5 * max(1.0, abs(want[1]))), k


def test_fvd_oracle(golden):
    """G10: oracle/fvd_ref (I3D, preprocess, moments, Frechet distance) against the reference's logits, moments and FVD value."""
    from oracle import fvd_ref
    g = golden("g10_fvd")
    o = fvd_ref.I3D(400)
    deterministic_fill_(o, prefix="i3d.")
    o.eval()
    vo, vg = t(g["videos_orig"]).float()[:, 1:], t(g["videos_gen"]).float()[:, 1:]
    a = fvd_ref.activations(o, fvd_ref.preprocess(vg), 3)
    assert np.abs(a - g["logits_gen_T15"]).max() <= 5e-5
    mu, sigma = fvd_ref.moments(a)
    assert np.abs(mu - g["mu_gen_T15"]).max() <= 5e-5
    assert np.allclose(_checksum(torch.from_numpy(sigma), "sigma"), g["sigma_gen_checksum_T15"], rtol=1e-4, atol=1e-4)
    mo, so = fvd_ref.moments(g["logits_orig_T15"])
    assert abs(fvd_ref.frechet_distance(mu, sigma, mo, so) - float(g["fvd_T15"])) <= 1e-3 * float(g["fvd_T15"])
    del vo


def test_fvd_state_dict_keys():
    """The product I3D (parameter container only on CPU) and the oracle share the reference's state-dict keys and shapes."""
    from ipoke_amd import fvd
    from oracle import fvd_ref
    a = fvd.I3D(400, "rgb", device="cpu").state_dict()
    b = fvd_ref.I3D(400).state_dict()
    assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a)
    assert "mixed_3b.branch_3.1.conv3d.weight" in a and "conv3d_0c_1x1.conv3d.bias" in a and len(a) == 344


def test_data_path_oracle(golden):
    """G11: oracle/data_ref (flow resize, poke simulation with injected draws) against the reference's _get_flow / _get_poke."""
    from oracle import data_ref
    g = golden("g11_data_path")
    for ci in range(int(g["n_cases"])):
        size, poke_size, zero, equal, fix = (int(v) for v in g[f"meta{ci}"])
        flow = data_ref.get_flow(g[f"raw{ci}"], (size, size), True)
        assert torch.equal(flow, t(g[f"flow{ci}"]))
        poke, centers = data_ref.get_poke(flow, poke_size, 5, data_ref.UniformDraws(g[f"u{ci}"], 5, bool(fix), bool(zero)), zero=bool(zero),
                                          fix_n_pokes=bool(fix), equal_poke_val=bool(equal))
        assert np.array_equal(centers.numpy(), g[f"centers{ci}"])
        assert np.array_equal(poke.nonzero().numpy().astype(np.int16), g[f"poke_nz{ci}"])
        assert np.array_equal(poke[poke != 0].numpy(), g[f"poke_val{ci}"])
