"""Optional branches of the second stage that the shipped configs leave off: augmented_input (second_stage_video.py:66-79, 304-308,
335-336) -- noise channels appended to the latent before the flow and stripped after sampling."""
import copy

import pytest
import torch

from ipoke_amd import configs
from ipoke_amd.second_stage import PokeMotionModel
from ipoke_amd.utils.detfill import deterministic_fill_
from oracle import flow_ref
from tests.helpers import synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_augmented_input():
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    arch.update(augmented_input=True, augment_channels=16, scale_augmentation=True)
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    model = PokeMotionModel(conf, dirs={}, dtype="f32", device=DEV, max_batch=2)
    assert model.config["architecture"]["flow_in_channels"] == 48
    assert "scale_augment" in dict(model.named_parameters()) and "shift_augment" in dict(model.named_buffers())
    for name in ("first_stage_model", "poke_embedder", "conditioner", "flow"):
        deterministic_fill_(getattr(model, name), prefix=name + ".")
    model.flow.sync_buffers()
    with torch.no_grad():
        model.scale_augment.fill_(2.0); model.shift_augment.fill_(0.5)
    batch = synthetic_batch(2, 16, 64, seed=4, device=DEV)
    torch.manual_seed(11)
    flow_input, cond = model.make_flow_input(batch)
    torch.manual_seed(11)                                             # same CPU draws in the same order: the encoder's reparameterisation
    latent, _ = model.encode_first_stage(batch["images"])             # noise first, then the augmentation noise
    noise = torch.randn(2, 16, 8, 8)
    assert flow_input.shape == (2, 48, 8, 8) and torch.equal(flow_input[:, :32], latent)
    assert torch.allclose(flow_input[:, 32:].cpu(), 2.0 * noise + 0.5, atol=1e-6)
    # the 48-channel flow against the oracle on that input
    with torch.no_grad():
        out, logdet = model.flow(flow_input, cond)
    o = flow_ref.SupervisedMacowTransformer(copy.deepcopy(model.config["architecture"]))
    deterministic_fill_(o, prefix="flow.")
    with torch.no_grad():
        oo, old = o(flow_input.cpu(), cond.cpu())
    assert (out.cpu() - oo).abs().max().item() < 2e-4 and (logdet.cpu() - old).abs().max().item() < 2e-2
    # training step and sampling run; sampled motion is cut back to z_dim channels before decoding
    from ipoke_amd.trainer import SecondStageTrainer
    tr = SecondStageTrainer(model)
    l0 = tr.train_step(batch).item()
    vids = model.forward_sample(batch, n_samples=1, n_logged_vids=2)
    assert vids[0].shape == (2, 15, 3, 64, 64) and torch.isfinite(vids[0]).all() and l0 == l0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_condition_nice_second_stage(dtype):
    """architecture.condition_nice (macow2.py:1024-1060): the whole second stage with conditioned NICE nets -- three optimizer steps (the
    first has lr = 0, second_stage_video.py:238-253), the conditioning columns of conv3 move, the trained state dict loads into the
    CPU oracle and gives the same flow output, sampling runs."""
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    arch["condition_nice"] = True
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    model = PokeMotionModel(conf, dirs={}, dtype=dtype, device=DEV, max_batch=2)
    for name in ("first_stage_model", "poke_embedder", "conditioner", "flow"):
        deterministic_fill_(getattr(model, name), prefix=name + ".")
    model.flow.sync_buffers()
    key = "flow.layers.0.0.coupling1_up.net.conv3.conv.weight_v"
    w0 = model.flow.state_dict()[key].clone()
    assert tuple(w0.shape) == (32, 64 + 128, 3, 3)
    batch = synthetic_batch(2, 16, 64, seed=4, device=DEV)
    from ipoke_amd.trainer import SecondStageTrainer
    tr = SecondStageTrainer(model)
    torch.manual_seed(17)                       # the encoder's reparameterisation noise is drawn on the CPU generator in every step
    losses = [tr.train_step(batch).item() for _ in range(3)]
    assert all(l == l for l in losses)
    w1 = model.flow.state_dict()[key]
    moved = (w1 - w0).abs()
    assert moved[:, :64].max().item() > 0 and moved[:, 64:].max().item() > 0, "both column groups of conv3 are trained"
    model.eval()
    torch.manual_seed(3)
    flow_input, cond = model.make_flow_input(batch)
    with torch.no_grad():
        out, logdet = model.flow(flow_input, cond)
    o = flow_ref.SupervisedMacowTransformer(copy.deepcopy(model.config["architecture"]))
    o.load_state_dict({k: v.cpu() for k, v in model.flow.state_dict().items()})
    with torch.no_grad():
        oo, old = o(flow_input.cpu(), cond.cpu())
    scale = oo.abs().max().item()
    # bf16: 13 coupling nets deep at |out| ~ 38; measured 0.5-1.0 % of the range over several noise draws (f32 mode is the tight check)
    tol_o, tol_l = (2e-4, 2e-2) if dtype == "f32" else (1.5e-2 * scale, 5e-3 * old.abs().max().item())
    assert (out.cpu() - oo).abs().max().item() <= tol_o and (logdet.cpu() - old).abs().max().item() <= tol_l
    vids = model.forward_sample(batch, n_samples=1, n_logged_vids=2)
    assert vids[0].shape == (2, 15, 3, 64, 64) and torch.isfinite(vids[0]).all()


@pytest.mark.parametrize("overlap", [True, False])
def test_gradient_accumulation_matches_the_large_batch_step(overlap):
    """training.min_acc_batch_size > data.batch_size (experiments/experiment.py:81-88 -> Lightning's accumulate_grad_batches = k):
    k = 2 micro-batches of 2 clips, each loss divided by k, one optimizer step -- against the CPU oracle + torch.optim.Adam(amsgrad)
    doing ONE step on the 4 clips at once (the mean loss of 4 = the mean of the two micro-batch means): parameters after the step,
    and a second optimizer step to check that the accumulation buffer is reset."""
    from ipoke_amd.trainer import SecondStageTrainer
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    conf["training"]["min_acc_batch_size"] = 4
    conf["training"]["lr_scaling"] = False                      # constant lr 1e-3 from the first step (the warm-up starts at 0)
    model = PokeMotionModel(conf, dirs={}, dtype="f32", device=DEV, max_batch=2)
    deterministic_fill_(model.flow, prefix="flow.")
    model.flow.sync_buffers()
    tr = SecondStageTrainer(model, n_grad_buckets=3, overlap=overlap)
    assert tr.accumulate_grad_batches == 2
    gen = torch.Generator().manual_seed(3)
    inputs = [(torch.randn(2, 32, 8, 8, generator=gen), torch.randn(2, 128, 8, 8, generator=gen)) for _ in range(4)]
    batches = [{"images": torch.zeros(2, 16, 3, 64, 64, device=DEV), "tag": i} for i in range(4)]
    model.make_flow_input = lambda batch, **kw: tuple(x.to(DEV) for x in inputs[batch["tag"]])      # frozen encoders bypassed
    # oracle: two optimizer steps on the concatenated micro-batches
    o = flow_ref.SupervisedMacowTransformer(copy.deepcopy(model.config["architecture"]))
    deterministic_fill_(o, prefix="flow.")
    opt = torch.optim.Adam(o.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-5, amsgrad=True)
    loss_fn = flow_ref.FlowLoss()
    ref_losses = []
    for s in range(2):
        x = torch.cat([inputs[2 * s][0], inputs[2 * s + 1][0]]); c = torch.cat([inputs[2 * s][1], inputs[2 * s + 1][1]])
        opt.zero_grad()
        out, ld = o(x, c)
        loss, _ = loss_fn(out, ld)
        loss.backward(); opt.step()
        ref_losses.append(loss.item())
    losses = [tr.train_step(batches[i], i).item() for i in range(4)]
    torch.cuda.synchronize()
    assert model.global_step == 2                                                      # optimizer steps, not batches
    for s in range(2):
        assert abs(0.5 * (losses[2 * s] + losses[2 * s + 1]) - ref_losses[s]) <= 2e-3 * max(1.0, abs(ref_losses[s]))
    worst = 0.0
    got = dict(model.flow.named_parameters())
    for k, p in o.named_parameters():
        d = (got[k].detach().cpu() - p.detach()).abs().max().item() / max(p.detach().abs().max().item(), 1e-3)
        worst = max(worst, d)
    print(f"gradient accumulation (overlap={overlap}): worst relative parameter deviation after 2 optimizer steps {worst:.2e}")
    assert worst <= 5e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_adapt_cond_ssize(golden, dtype):
    """adapt_cond_ssize (second_stage_video.py:120-129, 286-287): a conditioner with a 4x4 latent and the transposed adapter block
    (ConvTranspose2d k3 s2 + the "elu" -> ReLU quirk) that brings it to the first stage's 8x8, against the reference's own
    make_flow_input (golden g14).  The two adapter variants that cannot run in the reference raise with the reason."""
    from tests.conftest import t
    g = golden("g14_adapt_cond_64")
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    conf["conditioner_model"]["architecture"]["min_spatial_size"] = 4
    model = PokeMotionModel(conf, dirs={}, dtype=dtype, device=DEV, max_batch=2)
    assert model.adapt_cond_ssize and not model.adapt_poke_emb_ssize
    assert {"conv_adapt_cond.conv.weight", "conv_adapt_cond.conv.bias"} <= set(model.state_dict())
    for name in ("first_stage_model", "poke_embedder", "conditioner", "flow"):
        deterministic_fill_(getattr(model, name), prefix=("first_stage" if name == "first_stage_model" else name) + ".")
    deterministic_fill_(model.conv_adapt_cond, prefix="conv_adapt_cond.")
    model.flow.sync_buffers()
    batch = synthetic_batch(2, 16, 64, seed=int(g["batch_seed"]), device=DEV)
    torch.manual_seed(97)
    flow_input, cond = model.make_flow_input(batch)
    e_c = (cond.cpu() - t(g["cond"])).abs().max().item()
    e_z = (flow_input.cpu() - t(g["flow_input"])).abs().max().item()
    print(f"adapt_cond_ssize[{dtype}]: cond err {e_c:.3e} (|cond| max {abs(g['cond']).max():.2f}), flow_input err {e_z:.3e}")
    assert cond.shape == (2, 128, 8, 8)
    assert e_c <= (4e-4 if dtype == "f32" else 0.12) and e_z <= (2e-4 if dtype == "f32" else 6e-2)
    # sampling runs through the adapter too
    vids = model.forward_sample(batch, n_samples=1, n_logged_vids=1)
    assert vids[0].shape == (1, 15, 3, 64, 64)
    bad = configs.second_stage_config(64, 32, 16, batch_size=2, arch=configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4))
    bad["architecture"]["flow_mid_channels_factor"] = 2
    bad["conditioner_model"]["architecture"]["min_spatial_size"] = 16
    with pytest.raises(NotImplementedError, match="stride-0"):
        PokeMotionModel(bad, dirs={}, dtype=dtype, device=DEV, max_batch=2)
    bad["conditioner_model"]["architecture"]["min_spatial_size"] = 8
    bad["poke_embedder"]["architecture"]["min_spatial_size"] = 4
    with pytest.raises(NotImplementedError, match="wrong direction"):
        PokeMotionModel(bad, dirs={}, dtype=dtype, device=DEV, max_batch=2)
