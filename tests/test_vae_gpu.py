"""GPU parity of the first-stage VAE (encoder, ConvGRU, SPADE decoder, 2-D encoders) against the reference goldens."""
import copy

import numpy as np
import pytest
import torch

from ipoke_amd import configs
from ipoke_amd.utils.detfill import deterministic_fill_
from tests.conftest import t

pytestmark = pytest.mark.gpu
# f32: exact-f32 matrix cores vs the reference's fp32 CPU run; bf16: 8-bit mantissa activations through ~20 conv+norm layers
TOL = {"f32": 2e-4, "bf16": 6e-2}


def first_stage(size, z, T, dtype):
    from ipoke_amd.first_stage import SpadeCondMotionModel
    m = SpadeCondMotionModel(configs.first_stage_config(size, z, T), dirs={}, train=False, dtype=dtype)
    deterministic_fill_(m, prefix="first_stage.")
    return m.to("cuda").eval()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_motion_encoder_64(golden, dtype):
    g = golden("g4_encoder_64")
    m = first_stage(64, 32, 16, dtype)
    X, eps = t(g["X"], "cuda"), t(g["eps"], "cuda")
    z, mu, lv = m.enc_motion(X.transpose(1, 2), eps=eps)
    for name, got in (("mu", mu), ("logvar", lv), ("z", z)):
        err = (got.cpu() - t(g[name])).abs().max().item()
        print(f"encoder64[{dtype}] {name} err {err:.3e} (max |ref| {np.abs(g[name]).max():.2f})")
        assert err <= TOL[dtype]


def test_motion_encoder_128_f32(golden):
    g = golden("g4_encoder_128")
    m = first_stage(128, 32, 16, "f32")
    X = torch.rand(1, 16, 3, 128, 128, generator=torch.Generator().manual_seed(int(g["X_seed"]))) * 2 - 1
    z, mu, lv = m.enc_motion(X.cuda().transpose(1, 2), eps=torch.zeros(1, 32, 8, 8, device="cuda"))
    assert (mu.cpu() - t(g["mu"])).abs().max().item() <= TOL["f32"]
    assert (lv.cpu() - t(g["logvar"])).abs().max().item() <= TOL["f32"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gru_and_spade_decoder_64(golden, dtype):
    g = golden("g5_decoder_64")
    m = first_stage(64, 32, 16, dtype)
    z, x0 = t(g["z"], "cuda"), t(g["x0"], "cuda")
    frames = m.decode(z, x0, 3)
    diff = (frames.cpu() - t(g["frames"])).abs()
    print(f"decoder64[{dtype}] frames max err {diff.max().item():.3e} mean err {diff.mean().item():.3e}")
    assert frames.shape == (2, 3, 3, 64, 64)
    # frames are tanh outputs in [-1, 1]; bf16: max error over 73k pixels after 12 GRU cells + 20 conv/norm layers
    assert diff.max().item() <= (TOL["f32"] if dtype == "f32" else 0.12) and diff.mean().item() <= (1e-5 if dtype == "f32" else 1.2e-2)
    # unit pieces
    from ipoke_amd import nn as K
    errs = {}
    h0 = t(g["hidden_last"], "cuda")[:, 0]
    ib = m.gen.in_block.run(K.from_nchw(h0, dtype), dtype)
    def rel(got, key):        # the un-normalised upsampling blocks reach |x| ~ 1e5 with random weights: compare relative to range
        ref = t(g[key])
        return (K.to_nchw(got, dtype).cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    errs["in_block"] = rel(ib, "in_block")
    tin = K.from_nchw(t(g["in_block"], "cuda"), dtype)
    b0 = m.gen.blocks[0].run(tin, dtype)
    errs["block0"] = rel(b0, "block0")
    c1 = m.gen.blocks[0].conv1.run(tin, dtype)                                   # ConvTranspose + ("elu" -> ReLU)
    errs["block0_conv1"] = rel(c1, "block0_conv1")
    mods = m.gen.modulations(x0)
    sp = m.gen.spade_blocks[0].run(K.from_nchw(t(g["block0"], "cuda"), dtype), mods[0], dtype)
    errs["spade0"] = rel(sp, "spade0")
    print(f"decoder64[{dtype}] unit errors {errs}")
    assert all(v <= TOL[dtype] for v in errs.values()), errs


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_glue_make_flow_input(golden, dtype):
    """poke / image encoders + motion encoder as PokeMotionModel.make_flow_input chains them (G6)."""
    from tests.helpers import synthetic_batch
    g = golden("g6_glue_64")
    from ipoke_amd.first_stage import FirstStageWrapper
    pe = FirstStageWrapper(configs.encoder2d_config(64, 2), dtype=dtype)
    ce = FirstStageWrapper(configs.encoder2d_config(64, 3), dtype=dtype)
    deterministic_fill_(pe, prefix="poke_embedder."); deterministic_fill_(ce, prefix="conditioner.")
    pe, ce = pe.cuda().eval(), ce.cuda().eval()
    m = first_stage(64, 32, 16, dtype)
    batch = synthetic_batch(2, 16, 64, device="cuda")
    poke_emb, *_ = pe.encoder(batch["flow"])
    cond, *_ = ce.encoder(batch["images"][:, 0])
    z, mu, lv = m.enc_motion(batch["images"].transpose(1, 2), eps=t(g["eps"], "cuda"))
    cond = torch.cat([cond, poke_emb], 1)
    e1 = (cond.cpu() - t(g["cond"])).abs().max().item()
    e2 = (z.cpu() - t(g["flow_input"])).abs().max().item()
    print(f"glue[{dtype}] cond err {e1:.3e} flow_input err {e2:.3e}")
    assert e1 <= TOL[dtype] * 2 and e2 <= TOL[dtype]


# ---------------------------------------------------------------------------------------------- first-stage training (a18 / c4)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_first_stage_train_slice(golden, dtype):
    """Forward + L1/KL loss + backward of the whole VAE on the HIP kernels against the reference's autograd
    (goldens: X_hat, loss and checksums of all 125 parameter gradients; eval-mode spectral norm as in the fixture)."""
    g = golden("g5_first_stage_train_64")
    m = first_stage(64, 32, 4, dtype)
    X, eps = t(g["X"], "cuda"), t(g["eps"], "cuda")
    loss, X_hat, mu, lv = m.training_loss(X, eps, power_iteration=False)
    loss.backward()
    err_x = (X_hat.detach().cpu() - t(g["X_hat"])).abs().max().item()
    err_l = abs(loss.item() - float(g["loss"]))
    print(f"first-stage train[{dtype}] X_hat err {err_x:.3e} loss {loss.item():.6f} (ref {float(g['loss']):.6f})")
    # bf16: max over 73k tanh pixels; the GroupNorm statistics use LDS float atomics, so the last bf16 bit of a few
    # activations (and with it this maximum) varies from run to run
    assert err_x <= (TOL["f32"] if dtype == "f32" else 0.2)
    assert err_l <= (2e-4 if dtype == "f32" else 5e-2) * max(1.0, abs(float(g["loss"])))
    assert (mu.detach().cpu() - t(g["mu"])).abs().max().item() <= TOL[dtype]
    params = dict(m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert set(names) == {k for k, p in params.items() if p.grad is not None}
    import zlib
    worst = 0.0
    bad = []
    for k, ck in zip(names, g["grad_checksums"]):
        gr = params[k].grad.detach().double().flatten().cpu()
        idx = torch.randint(0, gr.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(k.encode())))
        ref_sum, ref_abs = ck[0], ck[1]
        scale = max(ref_abs, 1e-12)
        e_sum = abs(gr.sum().item() - ref_sum) / scale
        e_abs = abs(gr.abs().sum().item() - ref_abs) / scale
        e_smp = max(abs(gr[i].item() - r) for i, r in zip(idx.tolist(), ck[2:])) / max(gr.abs().max().item(), 1e-12)
        # sums: 2e-3 (f32) -- the L1 sub-gradient sign(x_hat - x) flips for pixels whose residual is below the forward
        # error, which perturbs every upstream gradient at the 1e-3 level; single sampled elements see it undamped
        tol, tol_smp = (2e-3, 1.5e-2) if dtype == "f32" else (0.25, 0.6)
        # a bias in front of an Instance/GroupNorm has an analytically zero gradient: what both sides hold is rounding
        # noise of the cancellation, compared on the scale of the layer's weight gradient instead
        noise = ref_abs <= 1e-4 * g["grad_checksums"][names.index(k.replace(".bias", ".weight_orig"))][1] if (
            k.endswith(".bias") and k.replace(".bias", ".weight_orig") in names) else False
        if noise:
            wabs = g["grad_checksums"][names.index(k.replace(".bias", ".weight_orig"))][1]
            assert gr.abs().sum().item() <= 1e-3 * wabs, (k, gr.abs().sum().item(), wabs)
            continue
        worst = max(worst, e_sum, e_abs, e_smp)
        if not (e_sum <= tol and e_abs <= tol and e_smp <= tol_smp):
            bad.append((k, float(e_sum), float(e_abs), float(e_smp), float(ref_abs)))
    print(f"first-stage train[{dtype}] worst relative gradient checksum error {worst:.3e}")
    for b in bad:
        print("   BAD", b)
    assert not bad


def test_first_stage_train_power_iteration(golden):
    """Train-mode spectral norm: one power iteration per forward call (g5_spectral_train pins u1, v1 and the output)."""
    from ipoke_amd import first_stage_train as FT, nn as K
    g = golden("g5_spectral_train")
    m = first_stage(64, 32, 16, "f32")
    blk = m.gen.blocks[0].conv1
    with torch.no_grad():
        blk.conv.weight_u.copy_(t(g["u0"], "cuda")); blk.conv.weight_v.copy_(t(g["v0"], "cuda"))
    x = K.from_nchw(t(g["x"], "cuda"), "f32")
    y = FT.convT_block(blk, x, "f32", pit=True)
    assert (blk.conv.weight_u.cpu() - t(g["u1"])).abs().max().item() <= 1e-5
    assert (blk.conv.weight_v.cpu() - t(g["v1"])).abs().max().item() <= 1e-5
    got = K.to_nchw(y, "f32").cpu()
    assert (got - t(g["y"])).abs().max().item() <= 2e-4 * max(1.0, float(np.abs(g["y"]).max()))


def test_first_stage_trainer_step_decreases_loss():
    from ipoke_amd.first_stage_train import FirstStageTrainer
    m = first_stage(64, 32, 4, "bf16")
    tr = FirstStageTrainer(m, lr=2e-4)
    X = torch.rand(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(3)).cuda() * 2 - 1
    eps = torch.randn(2, 32, 8, 8, generator=torch.Generator().manual_seed(4)).cuda()
    losses = [tr.step(X, eps)[0].item() for _ in range(4)]
    print("first-stage trainer losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
