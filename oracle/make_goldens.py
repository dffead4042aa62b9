"""Generate the golden vectors under tests/golden/ from the REFERENCE's own modules.

Run inside the build container only (needs the read-only checkout at
/root/reference):

    python -m oracle.make_goldens            # writes tests/golden/*.npz

Each fixture holds inputs and the outputs the reference produced for them.
Weights are NOT stored: every parameter/buffer is overwritten with the
name-keyed deterministic fill of ``ipoke_amd.utils.detfill`` (SURVEY.md §7
step 0), which any implementation sharing the state-dict keys regenerates.
Exceptions are stored explicitly (pre-init ActNorm draws, data-initialised
ActNorm parameters of the full-size flow).

While generating, the CPU oracle (oracle/flow_ref.py, oracle/vae_ref.py) is run
on the same inputs and asserted against the reference -- this is what pins it.
"""
import copy
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ipoke_amd import configs                                    # noqa: E402
from ipoke_amd.utils.detfill import deterministic_fill_          # noqa: E402
from oracle import data_ref, disc_ref, flow_ref, fvd_ref, ref_import, vae_ref, vgg_ref      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen(seed):
    return torch.Generator().manual_seed(seed)


def rn(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=gen(seed)) * scale


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def close(a, b, atol, what):
    err = (a.double() - b.double()).abs().max().item()
    assert err <= atol, f"oracle != reference for {what}: {err} > {atol}"
    return err


def checksum(t, key):
    """sum, abs-sum and three name-keyed sampled elements of a tensor."""
    t = t.detach().double().flatten()
    g = gen(zlib.crc32(key.encode()))
    idx = torch.randint(0, t.numel(), (3,), generator=g)
    return np.array([t.sum().item(), t.abs().sum().item(), *t[idx].tolist()])


# ---------------------------------------------------------------------------
def g1_units(Cs=(8, 32), name="g1_flow_units", with_lu=True):
    """G1: per-layer goldens of the flow's building blocks."""
    m2 = ref_import.ref("models.modules.INN.macow2")
    mu_ = ref_import.ref("models.modules.INN.macow_utils")
    fb = ref_import.ref("models.modules.INN.flow_blocks")
    out = {}
    for C in Cs:
        x = rn((2, C, 8, 8), 100 + C)
        h = rn((2, 128, 8, 8), 200 + C)
        out[f"x_{C}"], out[f"h_{C}"] = x, h

        # ActNorm: init path then fwd / inverse
        torch.manual_seed(7 + C)
        an = m2.ActNorm2dFlow(C)
        pre = an.log_scale.detach().clone()
        xi = rn((4, C, 8, 8), 300 + C, 2.0) + 0.5
        y, ld = an(xi)
        out[f"actnorm_{C}_pre_log_scale"], out[f"actnorm_{C}_init_x"] = pre, xi
        out[f"actnorm_{C}_post_log_scale"], out[f"actnorm_{C}_post_bias"] = an.log_scale, an.bias
        out[f"actnorm_{C}_y"], out[f"actnorm_{C}_logdet"] = y, ld
        out[f"actnorm_{C}_inv"] = an(y, reverse=True)
        o = flow_ref.ActNorm2dFlow(C)
        with torch.no_grad():
            o.log_scale.copy_(pre)
        yo, ldo = o(xi)
        close(yo, y, 1e-6, "actnorm init fwd"); close(ldo, ld, 1e-4, "actnorm logdet")
        close(o(yo, reverse=True), an(y, reverse=True), 1e-6, "actnorm inv")

        # Shuffle (bit exact)
        sh = fb.Shuffle(C)
        deterministic_fill_(sh, prefix=f"shuffle{C}.")
        ys, zero = sh(x)
        assert zero == 0
        out[f"shuffle_{C}_fwd_idx"], out[f"shuffle_{C}_bwd_idx"] = sh.forward_shuffle_idx, sh.backward_shuffle_idx
        out[f"shuffle_{C}_y"], out[f"shuffle_{C}_inv"] = ys, sh(ys, reverse=True)
        assert torch.equal(sh(ys, reverse=True), x)
        o = flow_ref.Shuffle(C); deterministic_fill_(o, prefix=f"shuffle{C}.")
        assert torch.equal(o(x)[0], ys) and torch.equal(o(ys, reverse=True), x)

        # Affine transform
        raw = rn((2, 2 * C, 8, 8), 400 + C)
        aff = mu_.Affine(dim=1, alpha=1.0)
        p = aff.calc_params(raw)
        ya, lda = aff.fwd(x, p)
        xa, ldb = aff.bwd(ya, p)
        out[f"affine_{C}_raw"], out[f"affine_{C}_y"], out[f"affine_{C}_logdet"] = raw, ya, lda
        out[f"affine_{C}_inv"] = xa
        mo, so = flow_ref.affine_params(raw)
        close(flow_ref.affine_fwd(x, mo, so)[0], ya, 1e-6, "affine fwd")
        close(flow_ref.affine_inv(ya, mo, so), xa, 1e-6, "affine inv")

        # shifted convs + MCF for the four orders
        for order, ks in (("A", (2, 3)), ("B", (2, 3)), ("C", (3, 2)), ("D", (3, 2))):
            sc = mu_.ShiftedConv2d(C, 4 * C, ks, order=order, bias=False)
            deterministic_fill_(sc, prefix=f"sc{C}{order}.")
            out[f"shiftconv_{C}_{order}"] = sc(x)
            o = flow_ref.ShiftedConv2d(C, 4 * C, ks, order); deterministic_fill_(o, prefix=f"sc{C}{order}.")
            close(o(x), sc(x), 1e-5, "shiftconv")

            mcf = m2.MaskedConvFlow(C, ks, h_channels=128, order=order, activation="elu", transform="affine", alpha=1.0)
            deterministic_fill_(mcf, prefix=f"mcf{C}{order}.")
            ym, ldm = mcf(x, h=h)
            xm = mcf(ym, h=h, reverse=True)
            out[f"mcf_{C}_{order}_y"], out[f"mcf_{C}_{order}_logdet"], out[f"mcf_{C}_{order}_inv"] = ym, ldm, xm
            # gradients of a scalar objective wrt input and the two weights
            xg = x.clone().requires_grad_(True)
            yg, lg = mcf(xg, h=h)
            (0.5 * (yg ** 2).sum() - lg.sum()).backward()
            out[f"mcf_{C}_{order}_dx"] = xg.grad
            out[f"mcf_{C}_{order}_dshift"] = mcf.net.shift_conv.weight.grad
            out[f"mcf_{C}_{order}_dv"] = mcf.net.conv1x1.conv.weight_v.grad
            out[f"mcf_{C}_{order}_dg"] = mcf.net.conv1x1.conv.weight_g.grad
            out[f"mcf_{C}_{order}_db"] = mcf.net.conv1x1.conv.bias.grad
            o = flow_ref.MaskedConvFlow(C, ks, order, 128); deterministic_fill_(o, prefix=f"mcf{C}{order}.")
            yo, lo = o(x, h=h)
            close(yo, ym, 1e-5, "mcf fwd"); close(lo, ldm, 1e-4, "mcf logdet")
            close(o(yo, h=h, reverse=True), xm, 1e-5, "mcf inverse")

        # NICE couplings
        for split in ("continuous", "skip"):
            for order in ("up", "down"):
                nice = m2.NICE2d(C, hidden_channels=64, h_channels=0, split_type=split, order=order, factor=2,
                                 transform="affine", alpha=1.0, type="conv", activation="elu")
                tag = f"nice_{C}_{split}_{order}"
                deterministic_fill_(nice, prefix=tag + ".")
                yn, ldn = nice(x)
                out[tag + "_y"], out[tag + "_logdet"], out[tag + "_inv"] = yn, ldn, nice(yn, reverse=True)
                o = flow_ref.NICE2d(C, 64, split, order); deterministic_fill_(o, prefix=tag + ".")
                yo, lo = o(x)
                close(yo, yn, 1e-5, tag); close(lo, ldn, 1e-4, tag + " logdet")
                close(o(yo, reverse=True), nice(yn, reverse=True), 1e-5, tag + " inv")

        # prior
        f = 4
        pr = m2.MultiScalePrior(C, hidden_channels=64, h_channels=128, factor=f, transform="affine", alpha=1.0,
                                coupling_type="conv", h_type=None, activation="elu", normalize=None, num_groups=None)
        deterministic_fill_(pr, prefix=f"prior{C}.")
        yp, ldp = pr(x, h=h)
        out[f"prior_{C}_y"], out[f"prior_{C}_logdet"], out[f"prior_{C}_inv"] = yp, ldp, pr(yp, h=h, reverse=True)
        o = flow_ref.MultiScalePrior(C, 64, f); deterministic_fill_(o, prefix=f"prior{C}.")
        yo, lo = o(x, h=h)
        close(yo, yp, 1e-5, "prior"); close(lo, ldp, 1e-4, "prior logdet")

    if not with_lu:
        npz(name, **out)
        return
    # LU-parametrised invertible 1x1 conv (optional path: use1x1)
    np.random.seed(3)
    lu = m2.InvertibleConvLU1d(8)
    x = out["x_8"]
    ylu, ldlu = lu(x)
    out["lu_8_permutated"], out["lu_8_sign_s"] = lu.permutated, lu.sign_s
    out["lu_8_l"], out["lu_8_u"], out["lu_8_log_s"] = lu.l, lu.u, lu.log_s
    out["lu_8_y"], out["lu_8_logdet"], out["lu_8_inv"] = ylu, ldlu, lu(ylu, reverse=True)
    npz(name, **out)


# ---------------------------------------------------------------------------
def _loss_and_grads(model, loss_mod, x, cond, seed):
    model.zero_grad()
    out, logdet = model(x, cond)
    torch.manual_seed(seed)
    loss, log = loss_mod(out, logdet)
    loss.backward()
    return out, logdet, loss, log


def g2_reduced_flow():
    """G2: reduced full-topology flow -- activations, reverse, loss dict, every parameter gradient,
    plus the data-dependent init path (G3-iii of SURVEY.md §7)."""
    INN = ref_import.ref("models.modules.INN.INN")
    loss_m = ref_import.ref("models.modules.INN.loss")
    arch = configs.reduced_flow_arch()
    R = INN.SupervisedMacowTransformer(copy.deepcopy(arch))
    O = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch))
    assert list(R.state_dict()) == list(O.state_dict()), "state-dict keys/order differ from the reference"
    deterministic_fill_(R, prefix="flow."); deterministic_fill_(O, prefix="flow.")
    x, cond = rn((3, 16, 8, 8), 11), rn((3, 128, 8, 8), 12)
    out, logdet, loss, log = _loss_and_grads(R, loss_m.FlowLoss(), x, cond, 1234)
    oo, ol, oloss, olog = _loss_and_grads(O, flow_ref.FlowLoss(), x, cond, 1234)
    close(oo, out, 2e-5, "G2 out"); close(ol, logdet, 1e-3, "G2 logdet"); close(oloss, loss, 1e-3, "G2 loss")
    close(olog["reference_nll_loss"], log["reference_nll_loss"], 1e-4, "G2 reference nll (RNG)")
    rev = R(out.detach(), cond, reverse=True)
    close(O(oo.detach(), cond, reverse=True), rev, 5e-5, "G2 reverse")
    arrs = dict(x=x, cond=cond, out=out, logdet=logdet, loss=loss, reverse=rev,
                reference_nll_loss=log["reference_nll_loss"], nll_loss=log["nll_loss"],
                nlogdet_loss=log["nlogdet_loss"], roundtrip_err=(rev - x).abs().max())
    worst = 0.0
    for (k, p), (k2, q) in zip(R.named_parameters(), O.named_parameters()):
        assert k == k2
        arrs["grad." + k] = p.grad
        worst = max(worst, (p.grad - q.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12))
    assert worst < 1e-4, worst
    print(f"  G2 oracle-vs-reference worst relative grad error {worst:.2e}")
    npz("g2_reduced_flow", **arrs)

    # init path: construction-order RNG of the reference is NOT reproduced; instead the
    # freshly constructed (uninitialised) parameters are captured and stored.
    torch.manual_seed(5)
    R2 = INN.SupervisedMacowTransformer(copy.deepcopy(arch))
    pre = {k: v.clone() for k, v in R2.state_dict().items()}
    O2 = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch)); O2.load_state_dict(pre)
    xi, ci = rn((4, 16, 8, 8), 21, 2.0) + 0.5, rn((4, 128, 8, 8), 22)
    with torch.no_grad():
        yi, li = R2(xi, ci)
        yo, lo = O2(xi, ci)
    close(yo, yi, 1e-5, "init out"); close(lo, li, 1e-3, "init logdet")
    arrs = dict(x=xi, cond=ci, out=yi, logdet=li)
    for k, v in pre.items():
        if v.dtype.is_floating_point and v.numel() <= 4096 or not v.dtype.is_floating_point:
            arrs["pre." + k] = v          # small tensors verbatim (ActNorm draws, flags, shuffle idx, g, biases)
    for k, v in R2.state_dict().items():
        if k.endswith(("log_scale", "bias", "weight_g", "initialized")):
            arrs["post." + k] = v
            close(O2.state_dict()[k].float(), v.float(), 1e-5, "post-init " + k)
    # big tensors of the pre-init state are regenerated by name-keyed fill instead of being stored
    big = [k for k, v in pre.items() if v.dtype.is_floating_point and v.numel() > 4096]
    R3 = INN.SupervisedMacowTransformer(copy.deepcopy(arch)); R3.load_state_dict(pre)
    sd3 = R3.state_dict()
    from ipoke_amd.utils.detfill import fill_value
    with torch.no_grad():
        for k in big:
            sd3[k].copy_(fill_value("flow." + k, sd3[k]))
        y3, l3 = R3(xi, ci)
    arrs["big_keys"] = np.array(big)
    arrs["out_filled"], arrs["logdet_filled"] = y3, l3
    for k, v in R3.state_dict().items():
        if k.endswith(("log_scale", "bias", "weight_g")):
            arrs["postfilled." + k] = v
    npz("g2_reduced_flow_init", **arrs)


def g2_lu_flow():
    """G2-LU: the reduced full-topology flow with ``use1x1`` (LU-parametrised invertible 1x1 convolutions as the per-level
    shuffle layers, macow2.py:596-649, 862): out, log-det, reverse, every parameter gradient.  The LU buffers / parameters come
    from the reference constructor's numpy draws (seeded) and are stored; everything else is the name-keyed fill."""
    INN = ref_import.ref("models.modules.INN.INN")
    loss_m = ref_import.ref("models.modules.INN.loss")
    arch = configs.reduced_flow_arch(); arch["use1x1"] = True
    np.random.seed(11)
    R = INN.SupervisedMacowTransformer(copy.deepcopy(arch))
    lu_state = {k: v.clone() for k, v in R.state_dict().items() if ".shuffle_layers." in k}
    np.random.seed(12)
    O = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch))
    assert list(R.state_dict()) == list(O.state_dict()), "state-dict keys/order differ from the reference (use1x1)"
    deterministic_fill_(R, prefix="flow."); deterministic_fill_(O, prefix="flow.")
    R.load_state_dict(lu_state, strict=False); O.load_state_dict(lu_state, strict=False)
    x, cond = rn((3, 16, 8, 8), 13), rn((3, 128, 8, 8), 14)
    out, logdet, loss, log = _loss_and_grads(R, loss_m.FlowLoss(), x, cond, 1234)
    oo, ol, oloss, olog = _loss_and_grads(O, flow_ref.FlowLoss(), x, cond, 1234)
    close(oo, out, 2e-5, "G2-LU out"); close(ol, logdet, 1e-3, "G2-LU logdet"); close(oloss, loss, 1e-3, "G2-LU loss")
    rev = R(out.detach(), cond, reverse=True)
    close(O(oo.detach(), cond, reverse=True), rev, 5e-5, "G2-LU reverse")
    arrs = dict(x=x, cond=cond, out=out, logdet=logdet, loss=loss, reverse=rev, roundtrip_err=(rev - x).abs().max())
    for k, v in lu_state.items():
        arrs["lu." + k] = v
    worst = 0.0
    for (k, p_), (k2, q) in zip(R.named_parameters(), O.named_parameters()):
        assert k == k2
        arrs["grad." + k] = p_.grad
        worst = max(worst, (p_.grad - q.grad).abs().max().item() / (p_.grad.abs().max().item() + 1e-12))
    assert worst < 1e-4, worst
    print(f"  G2-LU oracle-vs-reference worst relative grad error {worst:.2e}")
    npz("g2_reduced_flow_lu", **arrs)


# ---------------------------------------------------------------------------
def g3_full_flow(z_dim=32):
    """G3: full-size flow (1.05 B parameters): activations, log-det, loss, per-parameter gradient checksums."""
    INN = ref_import.ref("models.modules.INN.INN")
    loss_m = ref_import.ref("models.modules.INN.loss")
    arch = configs.flow_arch(z_dim)
    t = time.time()
    R = INN.SupervisedMacowTransformer(copy.deepcopy(arch))
    deterministic_fill_(R, prefix="flow.")
    # The analytic inverse of 800 chained autoregressive flows amplifies round-off by the coupling
    # gain; scaling the weight-norm gains keeps the random-weight flow well conditioned so that the
    # reverse pass can be compared value by value (recorded in the fixture as g_scale).
    g_scale = 0.3
    with torch.no_grad():
        for k, v in R.state_dict().items():
            if k.endswith("weight_g"):
                v.mul_(g_scale)
    print(f"  built+filled reference flow z={z_dim} in {time.time() - t:.0f}s")
    # ActNorm parameters come from the reference's own data-dependent init on a seeded batch
    for k, v in R.state_dict().items():
        if "actnorm" in k and k.endswith("initialized"):
            v.fill_(0)
    xi, ci = rn((8, z_dim, 8, 8), 31), rn((8, 128, 8, 8), 32)
    with torch.no_grad():
        R(xi, ci)
    actn = {k: v.clone() for k, v in R.state_dict().items() if "actnorm" in k and not k.endswith("initialized")}
    x, cond = rn((2, z_dim, 8, 8), 33), rn((2, 128, 8, 8), 34)
    out, logdet, loss, log = _loss_and_grads(R, loss_m.FlowLoss(), x, cond, 1234)
    with torch.no_grad():
        rev = R(out.detach(), cond, reverse=True)
    arrs = dict(x=x, cond=cond, out=out, logdet=logdet, loss=loss, reverse=rev, init_x=xi, init_cond=ci, g_scale=g_scale)
    names, sums = [], []
    for k, p in R.named_parameters():
        names.append(k); sums.append(checksum(p.grad, k))
    arrs["grad_names"], arrs["grad_checksums"] = np.array(names), np.stack(sums)
    for k, v in actn.items():
        arrs["actnorm." + k] = v
    # pin the oracle at full size too
    O = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch))
    O.load_state_dict(R.state_dict())
    oo, ol, oloss, _ = _loss_and_grads(O, flow_ref.FlowLoss(), x, cond, 1234)
    e1 = close(oo, out, 5e-4, "G3 out"); e2 = close(ol, logdet, 5e-2, "G3 logdet")
    print(f"  G3 oracle-vs-reference: out {e1:.2e}, logdet {e2:.2e}, roundtrip {(rev - x).abs().max().item():.2e}")
    npz(f"g3_full_flow_z{z_dim}", **arrs)


# ---------------------------------------------------------------------------
def _ref_first_stage(size, z_dim, n_frames):
    fsm = ref_import.ref("models.first_stage_motion_model")
    cfg = configs.first_stage_config(size, z_dim, n_frames)
    m = fsm.SpadeCondMotionModel(copy.deepcopy(cfg), dirs={}, train=False)
    deterministic_fill_(m, prefix="first_stage.")
    return m.eval(), cfg


def g4_g5_first_stage():
    """G4 (3-D encoder) and G5 (ConvGRU + SPADE decoder), 64x64, T=16, z=32."""
    m, cfg = _ref_first_stage(64, 32, 16)
    o = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg)).eval()
    missing = set(m.state_dict()) ^ set(o.state_dict())
    assert not missing, missing
    o.load_state_dict(m.state_dict())
    X = torch.rand(2, 16, 3, 64, 64, generator=gen(41)) * 2 - 1
    eps = rn((2, 32, 8, 8), 42)
    with torch.no_grad():
        emb = m.enc_motion.conv1(X.transpose(1, 2))
        x = torch.relu(m.enc_motion.bn1(emb))
        shapes = [tuple(x.shape)]
        for name in ("layer1", "layer2", "layer3"):
            x = getattr(m.enc_motion, name)(x); shapes.append(tuple(x.shape))
        mu, logvar = m.enc_motion.conv_mu(x.squeeze(2)), m.enc_motion.conv_var(x.squeeze(2))
        z = eps * (0.5 * logvar).exp() + mu
        zo, muo, lvo = o.enc_motion(X.transpose(1, 2), eps=eps)
    close(muo, mu, 2e-5, "G4 mu"); close(lvo, logvar, 2e-5, "G4 logvar"); close(zo, z, 2e-5, "G4 z")
    npz("g4_encoder_64", X=X, eps=eps, mu=mu, logvar=logvar, z=z, stage_shapes=np.array(shapes))

    # G5: decode 3 frames from a fixed latent
    zin, x0 = rn((2, 32, 8, 8), 43), X[:, 0]
    with torch.no_grad():
        hidden = [zin] * m.n_layers
        in_rnn = torch.cat([m.motion_bias] * 2, dim=0)
        frames, hids = [], []
        for _ in range(3):
            hidden = m.rnn(in_rnn, hidden)
            hids.append(hidden[-1])
            frames.append(m.gen([hidden[-1]], x0, del_shape=True))
        frames = torch.stack(frames, 1)
        fo = o.decode(zin, x0, 3)
    close(fo, frames, 5e-5, "G5 frames")
    arrs = dict(z=zin, x0=x0, frames=frames, hidden_last=torch.stack(hids, 1))
    # unit goldens of the decoder's parts
    with torch.no_grad():
        t0 = m.gen.in_block(hids[0]); arrs["in_block"] = t0
        t1 = m.gen.blocks[0](t0); arrs["block0"] = t1
        arrs["block0_conv1"] = m.gen.blocks[0].conv1(t0)          # ConvTranspose + ("elu" -> ReLU)
        arrs["spade0"] = m.gen.spade_blocks[0](t1, x0)
        close(o.gen.in_block(hids[0]), t0, 2e-5, "in_block"); close(o.gen.blocks[0](t0), t1, 2e-5, "block0")
        close(o.gen.spade_blocks[0](t1, x0), arrs["spade0"], 2e-5, "spade0")
    npz("g5_decoder_64", **arrs)

    # train-mode spectral-norm power iteration (one forward of one transposed block)
    blk = m.gen.blocks[0].conv1
    u0, v0 = blk.conv.weight_u.clone(), blk.conv.weight_v.clone()
    blk.train()
    with torch.no_grad():
        yt = blk(t0)
    blk.eval()
    ob = o.gen.blocks[0].conv1
    ob.conv.spectral_power_iter()
    with torch.no_grad():
        close(ob(t0), yt, 2e-5, "train-mode spectral norm fwd")
    npz("g5_spectral_train", x=t0, u0=u0, v0=v0, u1=blk.conv.weight_u, v1=blk.conv.weight_v, y=yt)

    # full first-stage forward + L1/KL loss + two sampled grads (c4 path, small shape)
    Xs = X[:, :4]
    cfg4 = configs.first_stage_config(64, 32, 4)
    fsm = ref_import.ref("models.first_stage_motion_model")
    m4 = fsm.SpadeCondMotionModel(copy.deepcopy(cfg4), dirs={}, train=False)
    deterministic_fill_(m4, prefix="first_stage.")
    m4.eval()     # frozen spectral-norm u,v (train-mode iteration is pinned separately above)
    torch.manual_seed(77)
    eps4 = torch.FloatTensor(2, 32, 8, 8).normal_()
    torch.manual_seed(77)
    Xh, mu4, lv4 = m4(Xs)
    losses = ref_import.ref("utils.losses")
    loss = 10 * (Xs[:, 1:] - Xh).abs().mean() + 1e-7 * losses.KL(mu4, lv4)
    loss.backward()
    o4 = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg4)).eval(); o4.load_state_dict(m4.state_dict())
    Xo, muo, lvo = o4(Xs, eps=eps4)
    lo = vae_ref.first_stage_loss(Xs, Xo, muo, lvo)
    close(Xo, Xh, 5e-5, "first-stage X_hat"); close(lo, loss, 1e-4, "first-stage loss")
    lo.backward()
    arrs = dict(X=Xs, eps=eps4, X_hat=Xh, mu=mu4, logvar=lv4, loss=loss)
    names, sums = [], []
    for (k, p), (k2, q) in zip(m4.named_parameters(), o4.named_parameters()):
        if p.grad is None:
            continue
        names.append(k); sums.append(checksum(p.grad, k))
        assert (p.grad - q.grad).abs().max().item() <= 1e-4 * (p.grad.abs().max().item() + 1e-6) + 1e-7, k
    arrs["grad_names"], arrs["grad_checksums"] = np.array(names), np.stack(sums)
    npz("g5_first_stage_train_64", **arrs)


def g4_encoder_128():
    m, cfg = _ref_first_stage(128, 32, 16)
    o = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg)).eval(); o.load_state_dict(m.state_dict())
    X = torch.rand(1, 16, 3, 128, 128, generator=gen(45)) * 2 - 1
    with torch.no_grad():
        feats = m.enc_motion.conv1(X.transpose(1, 2))
        x = torch.relu(m.enc_motion.bn1(feats)); shapes = [tuple(x.shape)]
        for name in ("layer1", "layer2", "layer3", "layer4"):
            x = getattr(m.enc_motion, name)(x); shapes.append(tuple(x.shape))
        mu, logvar = m.enc_motion.conv_mu(x.squeeze(2)), m.enc_motion.conv_var(x.squeeze(2))
        _, muo, lvo = o.enc_motion(X.transpose(1, 2), eps=torch.zeros(1, 32, 8, 8))
    close(muo, mu, 5e-5, "G4-128 mu"); close(lvo, logvar, 5e-5, "G4-128 logvar")
    npz("g4_encoder_128", X_seed=45, mu=mu, logvar=logvar, stage_shapes=np.array(shapes))


# ---------------------------------------------------------------------------
def build_reference_poke_model(size, z_dim, n_frames, arch):
    """PokeMotionModel via __new__ (its __init__ needs checkpoints that are not in the container)."""
    ssv = ref_import.ref("models.second_stage_video")
    fsm = ref_import.ref("models.first_stage_motion_model")
    fcm = ref_import.ref("models.modules.autoencoders.fully_conv_models")
    INN = ref_import.ref("models.modules.INN.INN")
    loss_m = ref_import.ref("models.modules.INN.loss")
    cfg = configs.second_stage_config(size, z_dim, n_frames, arch=arch)
    M = ssv.PokeMotionModel.__new__(ssv.PokeMotionModel)
    torch.nn.Module.__init__(M)
    M.config = cfg
    M.first_stage_config = copy.deepcopy(cfg["first_stage"])
    M.first_stage_model = fsm.SpadeCondMotionModel(copy.deepcopy(cfg["first_stage"]), dirs={}, train=False)
    M.poke_embedder = fcm.FirstStageWrapper(copy.deepcopy(cfg["poke_embedder"]))
    M.conditioner = fcm.FirstStageWrapper(copy.deepcopy(cfg["conditioner_model"]))
    M.flow = INN.SupervisedMacowTransformer(copy.deepcopy(cfg["architecture"]))
    M.loss_func = loss_m.FlowLoss()
    M.use_cond, M.embed_poke_and_image, M.poke_key = True, False, "flow"
    M.adapt_poke_emb_ssize = M.adapt_cond_ssize = False
    M.augment_input, M.full_seq = False, True
    deterministic_fill_(M.first_stage_model, prefix="first_stage.")
    deterministic_fill_(M.poke_embedder, prefix="poke_embedder.")
    deterministic_fill_(M.conditioner, prefix="conditioner.")
    deterministic_fill_(M.flow, prefix="flow.")
    return M, cfg


def synthetic_batch(B, T, size, seed=1):
    g = gen(seed)
    images = torch.rand(B, T, 3, size, size, generator=g) * 2 - 1
    flow = torch.randn(B, 2, size, size, generator=g)
    mask = (torch.rand(B, 1, size, size, generator=g) < 0.05).float()
    poke = [torch.randn(B, 2, size, size, generator=g) * mask, torch.zeros(B, 5, 2, dtype=torch.int64)]
    return {"images": images, "flow": flow, "poke": poke, "sample_ids": torch.zeros(B, T, dtype=torch.int64)}


def g6_g7_glue():
    """G6 (make_flow_input, training step, Adam-amsgrad, LR rule) and G7 (forward_sample with injected z)."""
    arch = configs.reduced_flow_arch(); arch.update(flow_in_channels=32, factor=4)
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    M, cfg = build_reference_poke_model(64, 32, 16, arch)
    batch = synthetic_batch(2, 16, 64)
    torch.manual_seed(99)
    eps = torch.FloatTensor(2, 32, 8, 8).normal_()      # what reparameterize will draw
    torch.manual_seed(99)
    flow_input, cond = M.make_flow_input(batch)
    arrs = dict(images_seed=1, eps=eps, flow_input=flow_input, cond=cond)

    opt = torch.optim.Adam(M.flow.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-5, amsgrad=True)
    losses = []
    names = [k for k, _ in M.flow.named_parameters()]
    for step in range(2):
        opt.zero_grad()
        out, logdet = M.flow(flow_input.detach(), cond)
        torch.manual_seed(1234)
        loss, log = M.loss_func(out, logdet)
        loss.backward()
        if step == 0:
            arrs["out"], arrs["logdet"] = out, logdet
        opt.step()
        losses.append(loss.item())
        arrs[f"param_checksums_step{step + 1}"] = np.stack([checksum(p, k) for k, p in M.flow.named_parameters()])
    arrs["losses"], arrs["param_names"] = np.array(losses), np.array(names)
    gen_mod = ref_import.ref("utils.general")
    its = [0, 1, 250, 499, 500, 501, 100000, 199999, 200000]
    lrs = []
    for it in its:
        if it < 500:
            lrs.append(float(gen_mod.linear_var(it, start_it=0, end_it=500, start_val=0., end_val=1e-3, clip_min=0., clip_max=1e-3)))
        else:
            lrs.append(float(gen_mod.linear_var(it, start_it=500, end_it=200000, start_val=1e-3, end_val=0., clip_min=0., clip_max=1e-3)))
        assert abs(lrs[-1] - flow_ref.lr_at(it)) < 1e-12
    arrs["lr_its"], arrs["lr_vals"] = np.array(its), np.array(lrs)

    # oracle pin: the same two steps on the CPU restatement
    O = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch)); deterministic_fill_(O, prefix="flow.")
    oopt = torch.optim.Adam(O.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-5, amsgrad=True)
    for step in range(2):
        oopt.zero_grad()
        oo, ol = O(flow_input.detach(), cond)
        torch.manual_seed(1234)
        l, _ = flow_ref.FlowLoss()(oo, ol)
        l.backward(); oopt.step()
        assert abs(l.item() - losses[step]) < 1e-3 * max(1.0, abs(losses[step])), (l.item(), losses[step])
    npz("g6_glue_64", **arrs)

    # G7: sampling with an injected latent.  Fresh name-keyed weights; the weight-norm gains are scaled down so that the
    # analytic inverse of the randomly filled flow stays well conditioned (a reverse pass from z ~ N(0,1) through random,
    # unscaled couplings overflows in the reference itself).
    g_scale = 0.3
    deterministic_fill_(M.flow, prefix="flow.")
    with torch.no_grad():
        for k, v in M.flow.state_dict().items():
            if k.endswith("weight_g"):
                v.mul_(g_scale)
    zs = rn((2, 32, 8, 8), 55)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: zs.clone()
    try:
        vids = M.forward_sample(batch, n_samples=1, n_logged_vids=2)
    finally:
        torch.randn = real_randn
    with torch.no_grad():
        motion = M.flow(zs, cond, reverse=True)
    assert torch.isfinite(motion).all() and torch.isfinite(vids[0]).all()
    O7 = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch)); O7.load_state_dict(M.flow.state_dict())
    with torch.no_grad():
        close(O7(zs, cond, reverse=True), motion, 1e-4, "G7 reverse")
    print(f"  G7 motion max {motion.abs().max().item():.2f}")
    npz("g7_sample_64", z=zs, motion=motion, video=vids[0][:, :4], video_checksum=checksum(vids[0], "video"),
        video_shape=np.array(vids[0].shape), g_scale=g_scale)


def g_128():
    """128x128 goldens of the configurations the benchmark runs (c2 / c4 / c5): z = 64 encoder heads, the 5-stage SPADE
    decoder (z = 64), the 4-stage 2-D encoders through make_flow_input (z = 64), and the first-stage L1 + KL training slice
    (z = 32, first_stage.yaml as written).  Inputs are regenerated from the recorded seeds."""
    # --- G4-128, z = 64 heads (c2 / c5 encoder) and G5-128 decoder (c5)
    m, cfg = _ref_first_stage(128, 64, 16)
    o = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg)).eval(); o.load_state_dict(m.state_dict())
    X = torch.rand(1, 16, 3, 128, 128, generator=gen(46)) * 2 - 1
    eps = rn((1, 64, 8, 8), 47)
    with torch.no_grad():
        x = torch.relu(m.enc_motion.bn1(m.enc_motion.conv1(X.transpose(1, 2))))
        for name in ("layer1", "layer2", "layer3", "layer4"):
            x = getattr(m.enc_motion, name)(x)
        mu, logvar = m.enc_motion.conv_mu(x.squeeze(2)), m.enc_motion.conv_var(x.squeeze(2))
        z = eps * (0.5 * logvar).exp() + mu
        zo, muo, lvo = o.enc_motion(X.transpose(1, 2), eps=eps)
    close(muo, mu, 5e-5, "G4-128/z64 mu"); close(lvo, logvar, 5e-5, "G4-128/z64 logvar"); close(zo, z, 5e-5, "G4-128/z64 z")
    npz("g4_encoder_128_z64", X_seed=46, eps=eps, mu=mu, logvar=logvar, z=z)
    zin, x0 = rn((1, 64, 8, 8), 48), X[:, 0]
    with torch.no_grad():
        hidden = [zin] * m.n_layers
        frames, hids = [], []
        for _ in range(2):
            hidden = m.rnn(m.motion_bias, hidden)
            hids.append(hidden[-1])
            frames.append(m.gen([hidden[-1]], x0, del_shape=True))
        frames = torch.stack(frames, 1)
        fo = o.decode(zin, x0, 2)
    close(fo, frames, 1e-4, "G5-128 frames")
    npz("g5_decoder_128_z64", X_seed=46, z=zin, frames=frames, hidden_last=torch.stack(hids, 1))

    # --- first-stage training slice at 128x128 (c4: z = 32), B = 1, T = 3
    cfg4 = configs.first_stage_config(128, 32, 3)
    fsm = ref_import.ref("models.first_stage_motion_model")
    m4 = fsm.SpadeCondMotionModel(copy.deepcopy(cfg4), dirs={}, train=False)
    deterministic_fill_(m4, prefix="first_stage.")
    m4.eval()
    Xs = torch.rand(1, 3, 3, 128, 128, generator=gen(49)) * 2 - 1
    torch.manual_seed(78)
    eps4 = torch.FloatTensor(1, 32, 8, 8).normal_()
    torch.manual_seed(78)
    Xh, mu4, lv4 = m4(Xs)
    losses = ref_import.ref("utils.losses")
    loss = 10 * (Xs[:, 1:] - Xh).abs().mean() + 1e-7 * losses.KL(mu4, lv4)
    loss.backward()
    o4 = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg4)).eval(); o4.load_state_dict(m4.state_dict())
    Xo, muo, lvo = o4(Xs, eps=eps4)
    lo = vae_ref.first_stage_loss(Xs, Xo, muo, lvo)
    close(Xo, Xh, 1e-4, "first-stage-128 X_hat"); close(lo, loss, 1e-4, "first-stage-128 loss")
    lo.backward()
    arrs = dict(X_seed=49, eps=eps4, X_hat=Xh, mu=mu4, logvar=lv4, loss=loss)
    names, sums, worst = [], [], 0.0
    for (k, p), (k2, q) in zip(m4.named_parameters(), o4.named_parameters()):
        if p.grad is None:
            continue
        names.append(k); sums.append(checksum(p.grad, k))
        # the L1 sub-gradient sign(x_hat - x) flips for the few pixels whose residual is below the oracle-vs-reference
        # forward difference (1e-5), which perturbs every upstream gradient: bound 2e-3 of the tensor's range here
        e = (p.grad - q.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-6)
        if e > 1e-3:
            print(f"    {k}: rel {e:.2e} (|grad| max {p.grad.abs().max().item():.2e})")
        worst = max(worst, e)
    assert worst <= 1e-2, worst
    print(f"  first-stage-128 oracle-vs-reference worst relative grad error {worst:.2e}")
    arrs["grad_names"], arrs["grad_checksums"] = np.array(names), np.stack(sums)
    npz("g5_first_stage_train_128", **arrs)

    # --- make_flow_input at 128x128, z = 64 (4-stage 2-D encoders + 5-stage 3-D encoder as PokeMotionModel chains them)
    arch = configs.flow_arch(64, hidden=64, num_steps=[2, 1, 1], factor=4)
    M, cfg = build_reference_poke_model(128, 64, 16, arch)
    batch = synthetic_batch(1, 16, 128, seed=3)
    torch.manual_seed(98)
    epsg = torch.FloatTensor(1, 64, 8, 8).normal_()
    torch.manual_seed(98)
    flow_input, cond = M.make_flow_input(batch)
    npz("g6_glue_128", batch_seed=3, eps=epsg, flow_input=flow_input, cond=cond)


def g8_disc():
    """G8: the first-stage temporal discriminator (patchgan_3d.py:171-304, config d_t of config/first_stage.yaml:65-75) at
    64x64, 8 frames, B = 2: predictions, the four feature maps, hinge discriminator loss with every parameter gradient,
    the generator-side loss (-mean(pred) + feature matching) with its gradient w.r.t. the fake clip, the gradient penalty
    value, and one train-mode forward (power iteration of every spectral-normalised conv)."""
    d3 = ref_import.ref("models.modules.discriminators.patchgan_3d")
    cfg = {"bce_loss": False, "gp_weight": 1.0, "num_classes": 1, "patch_temp_disc": False}
    m = d3.resnet(config=dict(cfg), spatial_size=64, sequence_length=9)
    deterministic_fill_(m, prefix="disc_t.")
    o = disc_ref.TemporalDiscriminator(64, dict(cfg))
    assert set(m.state_dict()) == set(o.state_dict()), set(m.state_dict()) ^ set(o.state_dict())
    o.load_state_dict(m.state_dict())
    m.eval(); o.eval()
    X_true = (torch.rand(2, 3, 8, 64, 64, generator=gen(81)) * 2 - 1)
    X_fake = (torch.rand(2, 3, 8, 64, 64, generator=gen(82)) * 2 - 1)
    arrs = dict(X_true=X_true, X_fake=X_fake)

    def disc_side(net):
        net.zero_grad()
        xt = X_true.clone().requires_grad_(True)
        pf, _ = net(X_fake)
        pt, fm = net(xt)
        loss = (net.loss(pf, real=False) + net.loss(pt, real=True)) / 2.0
        gp = net.gp2(pt, xt)
        loss.backward(retain_graph=True)
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        net.zero_grad()
        gp.backward()
        gp_grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        return pf, pt, fm, loss, gp, grads, gp_grads

    pf, pt, fm, loss, gp, grads, gp_grads = disc_side(m)
    pfo, pto, fmo, losso, gpo, gradso, gp_gradso = disc_side(o)
    close(pfo, pf, 2e-5, "G8 pred fake"); close(pto, pt, 2e-5, "G8 pred true"); close(losso, loss, 1e-5, "G8 loss_d")
    assert abs(gpo.item() - gp.item()) <= 1e-4 * abs(gp.item()), (gpo.item(), gp.item())
    for a, b in zip(fmo, fm):
        close(a, b, 5e-5, "G8 fmap")
    worst = 0.0
    for k in grads:
        e = (grads[k] - gradso[k]).abs().max().item() / (grads[k].abs().max().item() + 1e-8)
        worst = max(worst, e)
        assert e <= 2e-3, (k, e)
    print(f"  oracle vs reference: worst relative parameter-gradient error {worst:.2e}; gp {gp.item():.6f}")
    arrs.update(pred_fake=pf, pred_true=pt, loss_d=loss, gp=gp)
    for i, f in enumerate(fm):
        arrs[f"fmap{i}_checksum"] = checksum(f, f"fmap{i}")
        arrs[f"fmap{i}_slice"] = f[:, :4, :, :3, :3]
    names = sorted(grads)
    arrs["grad_names"] = np.array(names)
    arrs["grad_checksums"] = np.stack([checksum(grads[k], k) for k in names])
    gnames = sorted(gp_grads)
    arrs["gp_grad_names"] = np.array(gnames)
    arrs["gp_grad_checksums"] = np.stack([checksum(gp_grads[k], "gp." + k) for k in gnames])

    def gen_side(net):
        xf = X_fake.clone().requires_grad_(True)
        pg, ff = net(xf)
        with torch.no_grad():
            _, ft = net(X_true)
        lg = -pg.mean() + net.fmap_loss(ff, ft)
        lg.backward()
        return lg, xf.grad

    lg, dxf = gen_side(m)
    lgo, dxfo = gen_side(o)
    close(lgo, lg, 1e-5, "G8 generator loss")
    assert (dxf - dxfo).abs().max().item() <= 2e-3 * dxf.abs().max().item()
    arrs.update(loss_g=lg, dx_fake_checksum=checksum(dxf, "dx_fake"), dx_fake_slice=dxf[:, :, :2, :6, :6])

    # train mode: one forward = one power iteration per spectral-normalised conv
    m.train(); o.train()
    with torch.no_grad():
        ptr_, _ = m(X_true)
        ptro, _ = o(X_true)
    close(ptro, ptr_, 2e-5, "G8 train-mode pred")
    arrs["pred_true_train"] = ptr_
    for k in ("conv1", "layer2.0.downsample.0", "layer4.1.conv2"):
        mod = m.get_submodule(k)
        arrs[f"u1.{k}"] = mod.weight_u.clone(); arrs[f"v1_checksum.{k}"] = checksum(mod.weight_v, "v1." + k)
        close(o.get_submodule(k).weight_u, mod.weight_u, 1e-5, "G8 u after power iteration")
    npz("g8_temporal_disc_64", **arrs)


def g9_patch_disc():
    """G9: the first-stage 2-D PatchGAN (patchgan.py:368-470, config d_s of config/first_stage.yaml:77-85) on 64x64 frames,
    B = 4: prediction map (6x6), the three feature maps (16, 8, 7 pixels: the stride-1 4x4 convolutions leave the powers of
    two), hinge discriminator loss with every parameter gradient, generator-side loss with the gradient w.r.t. the fake
    frames, and a train-mode forward."""
    pg = ref_import.ref("models.modules.discriminators.patchgan")
    cfg = {"bce_loss": False, "gp_weight": 0.0}
    m = pg.PatchDiscriminator(dict(cfg))
    deterministic_fill_(m, prefix="disc_s.")
    o = disc_ref.PatchDiscriminator(dict(cfg))
    assert set(m.state_dict()) == set(o.state_dict()), set(m.state_dict()) ^ set(o.state_dict())
    o.load_state_dict(m.state_dict())
    m.eval(); o.eval()
    x_true = torch.rand(4, 3, 64, 64, generator=gen(91)) * 2 - 1
    x_fake = torch.rand(4, 3, 64, 64, generator=gen(92)) * 2 - 1
    arrs = dict(x_true=x_true, x_fake=x_fake)

    def disc_side(net):
        net.zero_grad()
        pf, _ = net(x_fake)
        pt, fm = net(x_true)
        loss = (net.loss(pf, real=False) + net.loss(pt, real=True)) / 2.0
        loss.backward()
        return pf, pt, fm, loss, {k: p.grad.clone() for k, p in net.named_parameters()}

    pf, pt, fm, loss, grads = disc_side(m)
    pfo, pto, fmo, losso, gradso = disc_side(o)
    close(pfo, pf, 2e-5, "G9 pred fake"); close(pto, pt, 2e-5, "G9 pred true"); close(losso, loss, 1e-5, "G9 loss")
    for a, b in zip(fmo, fm):
        close(a, b, 5e-5, "G9 fmap")
    for k in grads:
        assert (grads[k] - gradso[k]).abs().max().item() <= 2e-3 * (grads[k].abs().max().item() + 1e-8), k
    arrs.update(pred_fake=pf, pred_true=pt, loss_d=loss)
    for i, f in enumerate(fm):
        arrs[f"fmap{i}_checksum"] = checksum(f, f"fmap{i}")
        arrs[f"fmap{i}_slice"] = f[:, :4, :3, :3]
    names = sorted(grads)
    arrs["grad_names"] = np.array(names)
    arrs["grad_checksums"] = np.stack([checksum(grads[k], k) for k in names])

    def gen_side(net):
        xf = x_fake.clone().requires_grad_(True)
        pg_, ff = net(xf)
        with torch.no_grad():
            _, ft = net(x_true)
        lg = -pg_.mean() + net.fmap_loss(ff, ft)
        lg.backward()
        return lg, xf.grad

    lg, dxf = gen_side(m)
    lgo, dxfo = gen_side(o)
    close(lgo, lg, 1e-5, "G9 generator loss")
    assert (dxf - dxfo).abs().max().item() <= 2e-3 * dxf.abs().max().item()
    arrs.update(loss_g=lg, dx_fake_checksum=checksum(dxf, "dx_fake"), dx_fake_slice=dxf[:, :, :6, :6])
    m.train(); o.train()
    with torch.no_grad():
        ptr_, _ = m(x_true)
        ptro, _ = o(x_true)
    close(ptro, ptr_, 2e-5, "G9 train-mode pred")
    arrs["pred_true_train"] = ptr_
    arrs["u1.in_conv"] = m.in_conv.weight_u.clone(); arrs["u1.out_conv"] = m.out_conv.weight_u.clone()
    npz("g9_patch_disc_64", **arrs)


def g10_fvd():
    """G10: the FVD evaluation (utils/metrics.py:679-800, 813-1099): I3D logits of 64x64 clips after the 224x224 bilinear
    preprocess, for 16-frame (FVD-val-x0) and 15-frame (FVD-val) clips, named intermediate maps of the first clip, the
    activation moments and the Frechet distance of N = 6 generated against 6 original clips (I3D batch 3)."""
    mt = ref_import.ref("utils.metrics")
    m = mt.I3D(400, "rgb")
    deterministic_fill_(m, prefix="i3d.")
    m.eval()
    o = fvd_ref.I3D(400)
    assert set(m.state_dict()) == set(o.state_dict()), set(m.state_dict()) ^ set(o.state_dict())
    o.load_state_dict(m.state_dict())
    o.eval()
    N = 6
    orig = torch.rand(N, 16, 3, 64, 64, generator=gen(101)) * 2 - 1
    gen_ = (orig + 0.35 * torch.randn(N, 16, 3, 64, 64, generator=gen(102))).clamp(-1, 1)
    arrs = dict(videos_orig=orig.half(), videos_gen=gen_.half())
    orig, gen_ = orig.half().float(), gen_.half().float()          # the fixture stores fp16: use exactly those values
    for T in (16, 15):
        vo, vg = orig[:, 16 - T:], gen_[:, 16 - T:]
        pg, po = mt.preprocess(vg, vo)
        close(fvd_ref.preprocess(vg), pg, 1e-6, "G10 preprocess"); close(fvd_ref.preprocess(vo), po, 1e-6, "G10 preprocess")
        a_ref = mt.get_activations(po, m, 3)
        a_or = fvd_ref.activations(o, po, 3)
        close(torch.from_numpy(a_or), torch.from_numpy(a_ref), 2e-5, f"G10 logits T={T}")
        f_ref = mt.calculate_FVD(m, vg, vo, batch_size=3, cuda=False)
        f_or = fvd_ref.fvd(o, vg, vo, 3)
        assert abs(f_ref - f_or) <= 1e-3 * abs(f_ref) + 1e-3, (f_ref, f_or)
        print(f"  T={T}: FVD reference {f_ref:.6f} oracle {f_or:.6f}; logits |max| {np.abs(a_ref).max():.3f}")
        mu, sg = mt.calculate_activation_statistics(pg, m, 3, cuda=False)
        arrs[f"logits_orig_T{T}"] = a_ref
        arrs[f"logits_gen_T{T}"] = mt.get_activations(pg, m, 3)
        arrs[f"mu_gen_T{T}"] = mu
        arrs[f"sigma_gen_checksum_T{T}"] = checksum(torch.from_numpy(sg), "sigma")
        arrs[f"fvd_T{T}"] = np.float64(f_ref)
        if T == 16:
            taps = {}
            with torch.no_grad():
                x = po[:1].permute(0, 2, 1, 3, 4)
                lo = o(x, taps)
                # the same maps from the reference's sub-modules
                r = m.maxPool3d_2a_3x3(m.conv3d_1a_7x7(x)); close(taps["pool2a"], r, 2e-5, "G10 pool2a")
                r = m.maxPool3d_3a_3x3(m.conv3d_2c_3x3(m.conv3d_2b_1x1(r))); close(taps["pool3a"], r, 2e-5, "G10 pool3a")
                r = m.mixed_3b(r); close(taps["mixed_3b"], r, 2e-5, "G10 mixed_3b")
                r = m.maxPool3d_4a_3x3(m.mixed_3c(r)); close(taps["pool4a"], r, 2e-5, "G10 pool4a")
            for k, t in taps.items():
                arrs[f"tap_{k}_checksum"] = checksum(t, k)
                arrs[f"tap_{k}_slice"] = t[0, :6, :2, :5, :5]
    npz("g10_fvd", **arrs)


def _synthetic_raw_flow(seed, hs, kind):
    """Smooth synthetic optical flow [2, hs, hs]: a few moving blobs on a still background ("blobs"), a nearly uniform field whose
    amplitude has no outliers beyond two standard deviations ("flat"), or a dense random field ("dense")."""
    g = gen(seed)
    if kind == "flat":
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, hs), torch.linspace(-1, 1, hs), indexing="ij")
        return torch.stack([1.0 + 0.5 * xx, 0.3 * yy]).numpy().astype(np.float32) * 6
    coarse = torch.randn(1, 2, 6, 6, generator=g)
    if kind == "blobs":
        coarse = coarse * (torch.rand(1, 1, 6, 6, generator=g) < 0.2)
    f = torch.nn.functional.interpolate(coarse, size=(hs, hs), mode="bicubic", align_corners=False)[0] * 12
    return (f + 0.05 * torch.randn(2, hs, hs, generator=g)).numpy().astype(np.float32)


def g11_data():
    """G11: the data path (data/base_dataset.py: _get_flow :651-693, _get_poke :507-648) run through the reference's own methods on
    synthetic raw flows: resized flows, pokes and poke centres for ordinary and zero-poke samples, equal_poke_val on and off, fixed
    and drawn poke counts, and a flat field that exercises the threshold fallbacks.  np.random.randint is replaced by
    oracle.data_ref.UniformDraws fed with the stored uniforms while the reference runs."""
    import tempfile
    import types
    np.int = int                                            # the reference predates numpy 1.24
    bd = ref_import.ref("data.base_dataset")
    tmp = tempfile.mkdtemp()
    cases = []
    for size, hs, poke_size in ((64, 96, 5), (128, 160, 5), (128, 160, 10)):
        for j, kind in enumerate(("blobs", "dense", "flat", "blobs")):
            for zero in (False, True):
                cases.append(dict(size=size, hs=hs, poke_size=poke_size, kind=kind, zero=zero, seed=1000 * size + 10 * j + int(zero) + poke_size,
                                  equal=(j != 3), fix=(j == 1 and poke_size == 10)))
    arrs = {"n_cases": np.int64(len(cases))}
    for ci, c in enumerate(cases):
        raw = _synthetic_raw_flow(c["seed"], c["hs"], c["kind"])
        path = os.path.join(tmp, f"flow{ci}.npy")
        np.save(path, raw)
        size = c["size"]
        cfg = {"spatial_size": (size, size), "n_pokes": 5, "poke_size": c["poke_size"]}
        fake = types.SimpleNamespace(config=cfg, datadict={"flow_paths": np.array([[path]])}, valid_lags=[0], scale_poke_to_res=True, geom_transfs=None,
                                     poke_size=c["poke_size"], valid_h=[c["poke_size"], size - c["poke_size"]], valid_w=[c["poke_size"], size - c["poke_size"]],
                                     filter_flow=False, fix_n_pokes=c["fix"], equal_poke_val=c["equal"])
        fake._get_flow = lambda ids, _f=fake, **kw: bd.BaseDataset._get_flow(_f, ids, **kw)
        ids = (0, -1 if c["zero"] else 3)
        flow = bd.BaseDataset._get_flow(fake, (0, 3))
        u = torch.rand(11, generator=gen(c["seed"] + 7)).numpy().astype(np.float32)
        keep = np.random.randint
        np.random.randint = data_ref.UniformDraws(u, 5, c["fix"], c["zero"])
        try:
            poke, centers = bd.BaseDataset._get_poke(fake, ids)
        finally:
            np.random.randint = keep
        flow_ret = bd.BaseDataset._get_flow(fake, ids)             # what the sample's "flow" entry holds (zeros for zero pokes)
        of = data_ref.get_flow(raw, (size, size), True)
        close(of, flow, 0.0, f"G11 flow {ci}")
        op, oc = data_ref.get_poke(of, c["poke_size"], 5, data_ref.UniformDraws(u, 5, c["fix"], c["zero"]), zero=c["zero"], fix_n_pokes=c["fix"],
                                   equal_poke_val=c["equal"])
        close(op, poke, 0.0, f"G11 poke {ci}")
        assert torch.equal(oc, centers), (ci, oc, centers)
        n = int((centers[:, 0] >= 0).sum())
        print(f"  case {ci}: {c['kind']:5s} {size}px poke_size {c['poke_size']} zero={int(c['zero'])} equal={int(c['equal'])} fix={int(c['fix'])}: "
              f"{n} pokes, |poke| {poke.abs().sum().item():.2f}")
        arrs.update({f"raw{ci}": raw, f"u{ci}": u, f"flow{ci}": flow, f"flow_ret_abs{ci}": np.float64(flow_ret.abs().sum().item()),
                     f"centers{ci}": centers.numpy(), f"poke_abs{ci}": np.float64(poke.abs().sum().item()),
                     f"poke_nz{ci}": poke.nonzero().numpy().astype(np.int16), f"poke_val{ci}": poke[poke != 0].numpy(),
                     f"meta{ci}": np.array([size, c["poke_size"], int(c["zero"]), int(c["equal"]), int(c["fix"])])})
    npz("g11_data_path", **arrs)


def g12_vgg():
    """G12: the VGG perceptual loss (utils/losses.py:6-82) as the first stage calls it (first_stage_motion_model.py:263) on 64x64
    frames: the reference's own VGGLoss / VGG classes run with ``torchvision.models.vgg19`` resolved to the restated feature stack
    (torchvision itself is absent); loss value, the five feature maps of the generated frames and the gradient w.r.t. them."""
    import types
    ls = ref_import.ref("utils.losses")
    feats = vgg_ref.vgg19_features()
    deterministic_fill_(feats, prefix="vgg19.features.")
    ls.torchvision.models.vgg19 = lambda pretrained=True: types.SimpleNamespace(features=feats)
    m = ls.VGGLoss()
    o = vgg_ref.VGGLoss()
    deterministic_fill_(o.vgg, prefix="unused.")
    o.vgg.load_state_dict(m.vgg.state_dict())
    assert list(m.vgg.state_dict()) == list(o.vgg.state_dict())
    x = torch.rand(6, 3, 64, 64, generator=gen(121)) * 2 - 1
    y = (x + 0.4 * torch.randn(6, 3, 64, 64, generator=gen(122))).clamp(-1, 1)
    arrs = dict(x_true=x, x_hat=y, keys=np.array(list(m.vgg.state_dict())))

    def run(net):
        yy = y.clone().requires_grad_(True)
        loss = net(x, yy).mean()
        loss.backward()
        with torch.no_grad():
            fm = net.vgg(y)
        return loss.detach(), yy.grad, fm

    loss, dy, fm = run(m)
    losso, dyo, fmo = run(o)
    close(losso, loss, 1e-6, "G12 loss"); close(dyo, dy, 1e-7, "G12 dy")
    for i, (a, b) in enumerate(zip(fmo, fm)):
        close(a, b, 1e-5, f"G12 fmap{i}")
        arrs[f"fmap{i}_checksum"] = checksum(b, f"vgg{i}")
        arrs[f"fmap{i}_slice"] = b[:2, :4, :4, :4]
    print(f"  loss {loss.item():.6f}; |dy| max {dy.abs().max().item():.3e}")
    arrs.update(loss=loss, dy=dy)
    npz("g12_vgg_loss", **arrs)


def g13_train_mode():
    """G13: the first-stage training step exactly as c4 runs it -- ``SpadeCondMotionModel`` at 128x128, z = 32, T = 16 in
    ``.train()`` mode, so that every spectral-normalised decoder convolution runs one power iteration per ``gen`` CALL
    (old-style torch.nn.utils.spectral_norm hook, models/modules/autoencoders/util.py:52, 252): 15 different sigma per weight per
    step (models/first_stage_motion_model.py:498-522).  X_hat (three frames verbatim + per-frame checksums), L1 + KL loss, the
    checksum of every parameter gradient, and ``weight_u`` / ``weight_v`` of every decoder convolution after the step."""
    T = 16
    cfg = configs.first_stage_config(128, 32, T)
    fsm = ref_import.ref("models.first_stage_motion_model")
    # ``train=False`` only keeps the constructor from building discriminators / metric nets (first_stage_motion_model.py:51-69);
    # the module's mode is what the spectral-norm hook looks at
    m = fsm.SpadeCondMotionModel(copy.deepcopy(cfg), dirs={}, train=False)
    deterministic_fill_(m, prefix="first_stage.")
    m.train()
    o = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg)); o.load_state_dict(m.state_dict()); o.train()
    X = torch.rand(1, T, 3, 128, 128, generator=gen(131)) * 2 - 1
    torch.manual_seed(79)
    eps = torch.FloatTensor(1, 32, 8, 8).normal_()
    torch.manual_seed(79)
    Xh, mu, lv = m(X)
    losses = ref_import.ref("utils.losses")
    loss = 10 * (X[:, 1:] - Xh).abs().mean() + 1e-7 * losses.KL(mu, lv)
    loss.backward()
    Xo, muo, lvo = o(X, eps=eps)
    lo = vae_ref.first_stage_loss(X, Xo, muo, lvo)
    e1 = close(Xo, Xh, 1e-4, "G13 X_hat"); close(lo, loss, 1e-4, "G13 loss")
    lo.backward()
    arrs = dict(X_seed=131, eps=eps, mu=mu, logvar=lv, loss=loss, X_hat_frames=Xh[:, [0, 7, 14]],
                X_hat_checksums=np.stack([checksum(Xh[:, i], f"frame{i}") for i in range(T - 1)]))
    names, sums, worst = [], [], 0.0
    for (k, p), (k2, q) in zip(m.named_parameters(), o.named_parameters()):
        assert k == k2
        if p.grad is None:
            continue
        names.append(k); sums.append(checksum(p.grad, k))
        e_ = (p.grad - q.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-6)
        if e_ > 2e-3:
            print(f"    {k}: rel {e_:.2e} (|grad| max {p.grad.abs().max().item():.2e})")
        worst = max(worst, e_)
    assert worst <= 3e-2, worst           # L1 sub-gradient sign flips below the oracle-vs-reference forward difference (see g_128)
    arrs["grad_names"], arrs["grad_checksums"] = np.array(names), np.stack(sums)
    un, worst_u = [], 0.0
    sdm, sdo = m.state_dict(), o.state_dict()
    for k in sdm:
        if k.startswith("gen.") and k.endswith("weight_u"):
            un.append(k)
            arrs["u." + k] = sdm[k]
            arrs["v_checksum." + k] = checksum(sdm[k[:-1] + "v"], k[:-1] + "v")
            worst_u = max(worst_u, (sdm[k] - sdo[k]).abs().max().item())
    assert worst_u <= 1e-5, worst_u
    arrs["u_names"] = np.array(un)
    print(f"  G13 oracle-vs-reference: X_hat {e1:.2e}, worst relative grad error {worst:.2e}, u after {T - 1} iterations {worst_u:.2e} "
          f"({len(un)} spectral-normalised decoder convolutions)")
    npz("g13_first_stage_train_mode_128", **arrs)


def g7_sample_128():
    """G7-128: ``forward_sample`` of the c5 model -- 128x128, the FULL z = 64 flow (1.237 B parameters, the weights and the
    data-initialised ActNorm parameters of golden ``g3_full_flow_z64``), 15 generated frames -- with an injected latent
    (models/second_stage_video.py:326-382).  Stored: the conditioning, the sampled motion latent, frames 0 / 7 / 14 of both videos
    and per-frame checksums."""
    g3 = np.load(os.path.join(OUT, "g3_full_flow_z64.npz"))
    arch = configs.flow_arch(64)
    t0 = time.time()
    M, cfg = build_reference_poke_model(128, 64, 16, arch)
    # a reverse pass from z ~ N(0, 1) through 800 randomly filled autoregressive inverses overflows in the reference itself
    # at g3's gain scale 0.3 (NaN; |motion| 20 at 0.2); at 0.15 it is well conditioned: |motion| <= 11.5, forward(reverse(z)) - z = 1.5e-5
    g_scale = 0.15
    with torch.no_grad():
        sd = M.flow.state_dict()
        for k, v in sd.items():
            if k.endswith("weight_g"):
                v.mul_(g_scale)
        for k in g3.files:
            if k.startswith("actnorm."):
                sd[k[len("actnorm."):]].copy_(torch.from_numpy(g3[k]))
    print(f"  built the reference c5 model in {time.time() - t0:.0f}s")
    batch = synthetic_batch(2, 16, 128, seed=5)
    zs = rn((2, 64, 8, 8), 56)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: zs.clone()
    try:
        vids = M.forward_sample(batch, n_samples=1, n_logged_vids=2)
    finally:
        torch.randn = real_randn
    with torch.no_grad():
        _, cond = M.make_flow_input(batch, reverse=True)
        motion = M.flow(zs, cond, reverse=True)
    v = vids[0]
    assert torch.isfinite(motion).all() and torch.isfinite(v).all() and tuple(v.shape) == (2, 15, 3, 128, 128)
    O7 = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch)); O7.load_state_dict(M.flow.state_dict())
    with torch.no_grad():
        e = close(O7(zs, cond, reverse=True), motion, 5e-4, "G7-128 reverse")
    ofs = vae_ref.SpadeCondMotionModel(copy.deepcopy(cfg["first_stage"])).eval(); ofs.load_state_dict(M.first_stage_model.state_dict())
    with torch.no_grad():
        e2 = close(ofs.decode(motion, batch["images"][:, 0], 15), v, 2e-4, "G7-128 decode")
    print(f"  G7-128 oracle-vs-reference: motion {e:.2e} (|motion| max {motion.abs().max().item():.2f}), video {e2:.2e}")
    npz("g7_sample_128", batch_seed=5, z=zs, cond=cond, motion=motion, video_frames=v[:, [0, 7, 14]],
        video_checksums=np.stack([checksum(v[:, i], f"frame{i}") for i in range(15)]), video_shape=np.array(v.shape),
        g_scale=g_scale)


def g14_adapt_cond():
    """G14: ``adapt_cond_ssize`` (models/second_stage_video.py:120-129, 286-287): a conditioner whose latent is 4x4 and the transposed
    adapter block that brings it to the first stage's 8x8 -- the one adapter variant of the reference that can run (the poke adapter
    resizes away from the target for either ratio, the conditioner adapter for larger latents is a stride-0 Conv2d).  64x64, z = 32."""
    util = ref_import.ref("models.modules.autoencoders.util")
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    M, cfg = build_reference_poke_model(64, 32, 16, arch)
    fcm = ref_import.ref("models.modules.autoencoders.fully_conv_models")
    ccfg = configs.encoder2d_config(64, 3)
    ccfg["architecture"]["min_spatial_size"] = 4
    M.conditioner = fcm.FirstStageWrapper(copy.deepcopy(ccfg))
    deterministic_fill_(M.conditioner, prefix="conditioner.")
    M.adapt_cond_ssize = True
    M.conv_adapt_cond = util.Conv2dTransposeBlock(64, 64, st=2, ks=3, padding=1)          # as __init__ builds it for factor = 8 / 4 (:127-129)
    deterministic_fill_(M.conv_adapt_cond, prefix="conv_adapt_cond.")
    batch = synthetic_batch(2, 16, 64, seed=6)
    torch.manual_seed(97)
    eps = torch.FloatTensor(2, 32, 8, 8).normal_()
    torch.manual_seed(97)
    flow_input, cond = M.make_flow_input(batch)
    assert tuple(cond.shape) == (2, 128, 8, 8)
    with torch.no_grad():
        raw, *_ = M.conditioner.encoder(batch["images"][:, 0])
    assert tuple(raw.shape) == (2, 64, 4, 4)
    # oracle: the same from the restated blocks
    oc = vae_ref.FirstStageWrapper(copy.deepcopy(ccfg)).eval(); oc.load_state_dict(M.conditioner.state_dict(), strict=False)
    ob = vae_ref.Conv2dTransposeBlock(64, 64, 3, 2, 1, norm="none", activation="elu", snorm=False); ob.load_state_dict(M.conv_adapt_cond.state_dict())
    with torch.no_grad():
        close(ob(oc.encoder(batch["images"][:, 0])[0]), cond[:, :64], 2e-5, "G14 adapted cond")
    npz("g14_adapt_cond_64", batch_seed=6, eps=eps, flow_input=flow_input, cond=cond, cond_latent_4x4=raw)


def g16_condition_nice():
    """G16: the reduced full-topology flow with ``condition_nice: True`` (macow2.py:1024-1060, 553; macow_utils.py:275-283, 328-332):
    every NICE coupling net (steps and priors) sees the conditioning map behind conv2 -- activations, reverse pass, loss dict and every
    parameter gradient of the reference, with a 32-channel conditioning map so that the widened conv3 (hidden + 32 inputs) keeps the
    fixture small.  Parameters filled by name as in G2."""
    INN = ref_import.ref("models.modules.INN.INN")
    loss_m = ref_import.ref("models.modules.INN.loss")
    arch = configs.reduced_flow_arch()
    arch["condition_nice"] = True
    arch["h_channels"] = 32
    R = INN.SupervisedMacowTransformer(copy.deepcopy(arch))
    O = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch))
    assert list(R.state_dict()) == list(O.state_dict()), "state-dict keys/order differ from the reference"
    assert [tuple(v.shape) for v in R.state_dict().values()] == [tuple(v.shape) for v in O.state_dict().values()]
    w3 = R.state_dict()["flow.layers.0.0.coupling1_up.net.conv3.conv.weight_v"]
    assert w3.shape[1] == 64 + 32, w3.shape
    deterministic_fill_(R, prefix="flow."); deterministic_fill_(O, prefix="flow.")
    x, cond = rn((3, 16, 8, 8), 61), rn((3, 32, 8, 8), 62)
    out, logdet, loss, log = _loss_and_grads(R, loss_m.FlowLoss(), x, cond, 1234)
    oo, ol, oloss, olog = _loss_and_grads(O, flow_ref.FlowLoss(), x, cond, 1234)
    close(oo, out, 2e-5, "G16 out"); close(ol, logdet, 1e-3, "G16 logdet"); close(oloss, loss, 1e-3, "G16 loss")
    rev = R(out.detach(), cond, reverse=True)
    close(O(oo.detach(), cond, reverse=True), rev, 5e-5, "G16 reverse")
    # the option must matter: the same parameters without the conditioning columns give another output
    with torch.no_grad():
        out0, _ = R(x, torch.zeros_like(cond))
    assert (out0 - out).abs().max().item() > 1e-3
    arrs = dict(x=x, cond=cond, out=out, logdet=logdet, loss=loss, reverse=rev, nll_loss=log["nll_loss"],
                nlogdet_loss=log["nlogdet_loss"], roundtrip_err=(rev - x).abs().max())
    worst = 0.0
    for (k, p_), (k2, q) in zip(R.named_parameters(), O.named_parameters()):
        assert k == k2
        arrs["grad." + k] = p_.grad
        worst = max(worst, (p_.grad - q.grad).abs().max().item() / (p_.grad.abs().max().item() + 1e-12))
    assert worst < 1e-4, worst
    print(f"  G16 oracle-vs-reference worst relative grad error {worst:.2e}")
    npz("g16_condition_nice", **arrs)


def main(which):
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    jobs = {"g1": g1_units, "g1_wide": lambda: g1_units((60, 64), "g1_flow_units_wide", with_lu=False),
            "g2": g2_reduced_flow, "g2_lu": g2_lu_flow, "g3": g3_full_flow, "g3_64": lambda: g3_full_flow(64), "g45": g4_g5_first_stage,
            "g4_128": g4_encoder_128, "g67": g6_g7_glue, "g128": g_128, "g8": g8_disc, "g9": g9_patch_disc, "g10": g10_fvd, "g11": g11_data, "g12": g12_vgg, "g13": g13_train_mode,
            "g7_128": g7_sample_128, "g14": g14_adapt_cond, "g16": g16_condition_nice}
    for name in (which or list(jobs)):
        print(f"[{name}]")
        t = time.time()
        jobs[name]()
        print(f"  done in {time.time() - t:.1f}s")


if __name__ == "__main__":
    main(sys.argv[1:])
