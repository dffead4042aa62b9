"""GPU parity of the glue: Adam-amsgrad steps (G6), sampling (G7), encoder prefetch.  The full-size flows (G3, z = 32 and
z = 64) are in tests/test_bench_configs_gpu.py."""
import copy
import zlib

import numpy as np
import pytest
import torch

from ipoke_amd import configs
from ipoke_amd.utils.detfill import deterministic_fill_
from tests.conftest import t

pytestmark = pytest.mark.gpu


def checksum(x, key):
    x = x.detach().double().flatten().cpu()
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    idx = torch.randint(0, x.numel(), (3,), generator=g)
    return np.array([x.sum().item(), x.abs().sum().item(), *x[idx].tolist()])


def _glue_model(dtype):
    from ipoke_amd.second_stage import PokeMotionModel
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    m = PokeMotionModel(conf, dirs={}, dtype=dtype, device="cuda", max_batch=2)
    deterministic_fill_(m.first_stage_model, prefix="first_stage.")
    deterministic_fill_(m.poke_embedder, prefix="poke_embedder.")
    deterministic_fill_(m.conditioner, prefix="conditioner.")
    deterministic_fill_(m.flow, prefix="flow.")
    m.flow.sync_buffers()
    return m


def test_training_steps_adam_amsgrad(golden):
    """Two optimisation steps (flow fwd, FlowLoss, bwd, fused Adam-amsgrad with weight decay) vs torch.optim.Adam on the
    reference: losses and per-tensor parameter checksums after each step."""
    from ipoke_amd.optim import FusedAdamAmsgrad
    g = golden("g6_glue_64")
    m = _glue_model("f32")
    flow_input, cond = t(g["flow_input"], "cuda"), t(g["cond"], "cuda")
    opt = FusedAdamAmsgrad(m.flow, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-5, amsgrad=True)
    m.flow.train()
    names = g["param_names"].tolist()
    for step in range(2):
        out, logdet = m.flow(flow_input, cond)
        if step == 0:
            assert (out.detach().cpu() - t(g["out"])).abs().max() <= 1e-4
        loss, log = m.loss_func(out, logdet)
        assert abs(loss.item() - float(g["losses"][step])) <= 2e-3 * max(1.0, abs(float(g["losses"][step])))
        loss.backward()
        opt.step()
        ref = g[f"param_checksums_step{step + 1}"]
        params = dict(m.flow.named_parameters())
        for i, k in enumerate(names):
            cs = checksum(params[k], k)
            assert abs(cs[0] - ref[i][0]) <= 2e-4 * max(ref[i][1], 1e-3), (step, k, cs, ref[i])
            assert np.abs(cs[2:] - ref[i][2:]).max() <= 2e-4 * max(np.abs(ref[i][2:]).max(), 1e-2), (step, k)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_sample_with_injected_latent(golden, dtype):
    """forward_sample: reverse flow + GRU/SPADE decode of an injected z (the reference's torch.randn is patched)."""
    from tests.helpers import synthetic_batch
    g6, g7 = golden("g6_glue_64"), golden("g7_sample_64")
    m = _glue_model(dtype)
    with torch.no_grad():
        for k, p in m.flow.named_parameters():
            if k.endswith("weight_g"):
                p.mul_(float(g7["g_scale"]))
    m.flow.mark_weights_updated()
    batch = synthetic_batch(2, 16, 64, device="cuda")
    z = t(g7["z"])
    real = torch.randn
    torch.randn = lambda *a, **k: z.clone()
    try:
        vids = m.forward_sample(batch, n_samples=1, n_logged_vids=2)
    finally:
        torch.randn = real
    assert tuple(vids[0].shape) == tuple(g7["video_shape"]) and vids[0].device.type == "cpu"
    with torch.no_grad():
        motion = m.flow(z.cuda(), t(g6["cond"], "cuda"), reverse=True)
    e_m = (motion.cpu() - t(g7["motion"])).abs().max().item()
    e_v = (vids[0][:, :4] - t(g7["video"])).abs()
    print(f"[{dtype}] sample: motion err {e_m:.3e}, video max err {e_v.max().item():.3e} mean {e_v.mean().item():.3e}")
    assert e_m <= (1e-4 if dtype == "f32" else 0.1)
    assert e_v.max().item() <= (5e-4 if dtype == "f32" else 0.2) and e_v.mean().item() <= (2e-5 if dtype == "f32" else 2e-2)


@pytest.mark.parametrize("enc_graph", ["1", "0", "thread"])
def test_encoder_prefetch_does_not_change_training(enc_graph, monkeypatch):
    """train_step(batch, next_batch=...) runs the next batch's frozen encoders on a side stream during the current backward:
    same losses and parameters as the plain loop (the encoder's CPU-generator draws keep their order).  enc_graph = 1 (default): the
    prefetched encoders are replayed from one captured hipGraph (first prefetch eager, second captures, third and fourth replay)."""
    # enc_graph = "thread": eager encoders queued by a second host thread while the main thread is inside the backward call
    monkeypatch.setenv("IPOKE_ENC_GRAPH", "0" if enc_graph == "thread" else enc_graph)
    monkeypatch.setenv("IPOKE_PREFETCH_THREAD", "1" if enc_graph == "thread" else "0")
    from ipoke_amd import configs
    from ipoke_amd.second_stage import PokeMotionModel
    from ipoke_amd.trainer import SecondStageTrainer
    from ipoke_amd.utils.detfill import deterministic_fill_

    def run(prefetch):
        torch.manual_seed(99)
        arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
        arch["flow_mid_channels_factor"] = 2
        model = PokeMotionModel(configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch), dirs={}, dtype="f32",
                                device="cuda:0", max_batch=2)
        for part, pfx in ((model.first_stage_model, "first_stage."), (model.poke_embedder, "poke_embedder."),
                          (model.conditioner, "conditioner."), (model.flow, "flow.")):
            deterministic_fill_(part, prefix=pfx)
        model.flow.sync_buffers()
        batches = []
        for k in range(5):
            g = torch.Generator().manual_seed(20 + k)
            batches.append({"images": (torch.rand(2, 16, 3, 64, 64, generator=g) * 2 - 1).cuda(),
                            "flow": torch.randn(2, 2, 64, 64, generator=g).cuda(),
                            "poke": [torch.zeros(2, 2, 64, 64).cuda(), torch.zeros(2, 5, 2, dtype=torch.int64).cuda()]})
        tr = SecondStageTrainer(model)
        assert tr.prefetch_thread == (enc_graph == "thread")
        if not prefetch:
            tr.prefetch_stream = None
        losses = []
        for k in range(5):
            nxt = batches[k + 1] if k + 1 < 5 else None
            losses.append(tr.train_step(batches[k], k, next_batch=nxt).item())
        if prefetch and enc_graph == "1":
            assert any(e.get("state") == "ready" for e in model._enc_graphs.values())
        torch.cuda.synchronize()
        return losses, model.flow.flat_params.detach().cpu()[::499].clone()

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert all(abs(a - b) <= 2e-5 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
    assert (p0 - p1).abs().max().item() <= 2e-5 * p0.abs().max().item()


def test_lightning_checkpoint_loads_and_reproduces_the_oracle(tmp_path):
    """f4: a Lightning-layout ``.ckpt`` written from the oracle's modules (reference state-dict keys: ``flow.flow.*`` with
    weight_g / weight_v, int64 shuffle indices, uint8 ``initialized`` flags; ``first_stage_model.*`` with spectral-norm
    ``weight_orig / weight_u / weight_v``; ``poke_embedder.*``, ``conditioner.*``) loads through
    ``PokeMotionModel.load_checkpoint`` and reproduces the oracle's forward_density on the same batch."""
    from ipoke_amd.second_stage import PokeMotionModel
    from oracle import flow_ref, vae_ref
    from tests.helpers import synthetic_batch
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    ofs = vae_ref.SpadeCondMotionModel(copy.deepcopy(conf["first_stage"])).eval()
    ope = vae_ref.FirstStageWrapper(copy.deepcopy(conf["poke_embedder"])).eval()
    oce = vae_ref.FirstStageWrapper(copy.deepcopy(conf["conditioner_model"])).eval()
    oarch = copy.deepcopy(arch); oarch.update(flow_in_channels=32, h_channels=128, flow_mid_channels=64)
    oflow = flow_ref.SupervisedMacowTransformer(oarch)
    sd = {}
    for prefix, mod in (("first_stage_model.", ofs), ("poke_embedder.", ope), ("conditioner.", oce), ("flow.", oflow)):
        deterministic_fill_(mod, prefix="ckpt." + prefix)
        sd.update({prefix + k: v.clone() for k, v in mod.state_dict().items()})
    assert any(v.dtype == torch.int64 for v in sd.values()) and any(v.dtype == torch.uint8 for v in sd.values())
    assert any(k.endswith("weight_orig") for k in sd) and any(k.endswith("weight_g") for k in sd)
    path = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": sd, "epoch": 3, "global_step": 1234, "pytorch-lightning_version": "1.1.0",
                "optimizer_states": [], "lr_schedulers": []}, path)
    m = PokeMotionModel(conf, dirs={}, dtype="f32", device="cuda", max_batch=2)
    res = m.load_checkpoint(path)
    assert not res.missing_keys and not res.unexpected_keys
    batch = synthetic_batch(2, 16, 64, device="cuda")
    torch.manual_seed(5)
    out, logdet = m.forward_density(batch)
    torch.manual_seed(5)
    eps = torch.FloatTensor(2, 32, 8, 8).normal_()
    cpu = synthetic_batch(2, 16, 64)
    with torch.no_grad():
        poke_emb, *_ = ope.encoder(cpu["flow"])
        cond, *_ = oce.encoder(cpu["images"][:, 0])
        z, mu, _ = ofs.enc_motion(cpu["images"].transpose(1, 2), eps=eps)
        o_out, o_ld = oflow(z, torch.cat([cond, poke_emb], 1))
    e_out = (out.detach().cpu() - o_out).abs().max().item()
    e_ld = (logdet.detach().cpu() - o_ld).abs().max().item()
    print(f"ckpt round trip: out err {e_out:.3e} logdet err {e_ld:.3e}")
    assert e_out <= 2e-4 and e_ld <= 2e-2


def test_full_size_adam_amsgrad_matches_torch_optim():
    """the optimizer path the c2 / c3 benchmarks execute -- SecondStageTrainer.train_step with the engine-issued
    (native) Adam-amsgrad over 16 backward pieces: adam_cast over the conv2 tensors, adam_amsgrad_seg over the rest, the shadow refresh
    behind them -- on the FULL z = 32 flow (1.054 B parameters, 2048 hidden, f32 mode, B = 2) against torch.optim.Adam(amsgrad=True,
    weight_decay) itself, stepped on the GPU over the same flat buffer with the gradients the engine produced.  (Those gradients are
    pinned to the reference by the checksums of every tensor in test_bench_configs_gpu.py::test_full_size_flow.)  Two steps, so that
    the second one starts from carried optimizer state; per-tensor comparison of the parameter UPDATES for every tensor."""
    from ipoke_amd.second_stage import PokeMotionModel
    from ipoke_amd.trainer import SecondStageTrainer
    from tests.helpers import cached_fill_, synthetic_batch
    conf = configs.second_stage_config(64, 32, 16, batch_size=2)           # plants_64 (BASELINE configs[0]): the full-size z = 32 flow
    m = PokeMotionModel(conf, dirs={}, dtype="f32", device="cuda", max_batch=2)
    assert m.flow.engine.n_params >= 1_054_000_000
    deterministic_fill_(m.first_stage_model, prefix="first_stage.")
    deterministic_fill_(m.poke_embedder, prefix="poke_embedder.")
    deterministic_fill_(m.conditioner, prefix="conditioner.")
    cached_fill_(m.flow, "flow.")
    with torch.no_grad():
        for k, p in m.flow.named_parameters():
            if k.endswith("weight_g"):
                p.mul_(0.05)                                               # couplings away from the identity, outputs bounded
    m.flow.sync_buffers()
    m.flow.mark_weights_updated()
    tr = SecondStageTrainer(m, n_grad_buckets=16)
    assert tr.native_opt and tr.overlap, "the benchmarked single-GPU path: native Adam, updates overlapped with the backward pass"
    m.global_step = 2000                                                   # inside the LR warm-up ramp: a non-zero learning rate
    batch = synthetic_batch(2, 16, 64, seed=3, device="cuda")
    flat = m.flow.flat_params
    g0 = tr.opt.param_groups[0]
    ref_p = torch.nn.Parameter(flat.detach().clone())
    ref = torch.optim.Adam([ref_p], lr=g0["lr"], betas=g0["betas"], eps=g0["eps"], weight_decay=g0["weight_decay"], amsgrad=True)
    # tensor spans inside the flat buffer (views of it)
    spans = []
    for k, p in m.flow.named_parameters():
        off = (p.data_ptr() - flat.data_ptr()) // 4
        spans.append((k, off, p.numel()))
    offs = torch.tensor([o for _, o, _ in spans], device="cuda"); nums = torch.tensor([n for _, _, n in spans], device="cuda")
    zero = torch.zeros(1, dtype=torch.float64, device="cuda")
    worst = 0.0
    for step in range(2):
        before = flat.detach().clone()
        loss = tr.train_step(batch, step)
        torch.cuda.synchronize()
        assert torch.isfinite(loss)
        lr = tr.opt.param_groups[0]["lr"]
        assert lr > 0
        for pg in ref.param_groups:
            pg["lr"] = lr
        ref_p.grad = m.flow.flat_grads.detach().clone()                    # what the engine's update consumed (weight-norm backward included)
        ref.step()
        upd_e = (flat.detach() - before).double()
        upd_r = (ref_p.detach() - before).double()
        del before
        # per tensor: sum |update difference| / sum |reference update| through prefix sums over the flat buffer
        c = torch.cat([zero, (upd_e - upd_r).abs().cumsum(0)]); d_sum = c[offs + nums] - c[offs]; del c
        c = torch.cat([zero, upd_r.abs().cumsum(0)]); r_sum = c[offs + nums] - c[offs]; del c
        rel = (d_sum / r_sum.clamp_min(1e-30)).cpu().numpy()
        moved = (r_sum > 0).cpu().numpy()
        e_max = ((upd_e - upd_r).abs().max() / upd_r.abs().max()).item()
        k_worst = int(np.argmax(np.where(moved, rel, 0.0)))
        print(f"step {step + 1}: lr {lr:.3e}, loss {loss.item():.4f}, {int(moved.sum())} of {len(spans)} tensors moved; update error: "
              f"max-norm {e_max:.3e}, worst tensor {rel[k_worst]:.3e} ({spans[k_worst][0]})")
        assert moved.sum() >= 0.99 * len(spans)
        assert e_max <= 2e-4 and rel[moved].max() <= 2e-4, (e_max, rel[moved].max(), spans[k_worst][0])
        worst = max(worst, e_max)
        # the next step starts from the engine's parameters on both sides (the comparison is of the update rule and its carried state)
        with torch.no_grad():
            ref_p.copy_(flat)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_second_stage_train_steps_are_reproducible(dtype):
    """the c2 train step -- plants_128, the full z = 64 flow (1.237 B parameters), B = 20, native Adam-amsgrad over
    24 backward pieces, encoder prefetch on -- is BIT-reproducible: two ``SecondStageTrainer.train_step``s from the same parameters and
    optimizer state, three times; the flat gradient buffer, the parameters and the loss of every repetition are identical.  The
    reference trains with ``deterministic=True`` / ``cudnn.deterministic = True`` (experiments/experiment.py:33, 86).  Rounds 1-5 added
    the K slices of every conv1 data gradient (215 per backward pass) into the gradient state with fp32 atomics; they now meet in a
    scratch and are summed in a fixed order (ipoke_conv_desc.acc_scratch)."""
    from ipoke_amd.second_stage import PokeMotionModel
    from ipoke_amd.trainer import SecondStageTrainer
    from tests.helpers import cached_fill_, synthetic_batch
    cfg = configs.BENCH_CONFIGS["c2"]
    B = cfg["batch_size"]
    conf = configs.second_stage_config(cfg["spatial_size"], cfg["z_dim"], cfg["n_frames"], B)
    m = PokeMotionModel(conf, dirs={}, dtype=dtype, device="cuda", max_batch=B)
    assert m.flow.engine.n_params >= 1_237_000_000
    deterministic_fill_(m.first_stage_model, prefix="first_stage.")
    deterministic_fill_(m.poke_embedder, prefix="poke_embedder.")
    deterministic_fill_(m.conditioner, prefix="conditioner.")
    cached_fill_(m.flow, "flow.")
    with torch.no_grad():
        for k, p in m.flow.named_parameters():
            if k.endswith("weight_g"):
                p.mul_(0.05)                                               # couplings away from the identity, outputs bounded
    m.flow.sync_buffers()
    m.flow.mark_weights_updated()
    tr = SecondStageTrainer(m)
    assert tr.native_opt and tr.overlap and tr.prefetch_stream is not None, "the benchmarked single-GPU path"
    batch = synthetic_batch(B, cfg["n_frames"], cfg["spatial_size"], seed=3, device="cuda")
    flat = m.flow.flat_params
    snap = flat.detach().clone()
    state = (tr.opt.exp_avg, tr.opt.exp_avg_sq, tr.opt.max_exp_avg_sq)

    def two_steps():
        with torch.no_grad():
            flat.copy_(snap)
            for s_ in state:
                s_.zero_()
        tr.opt.steps = 0
        m.global_step = 20                                                 # early in the LR warm-up ramp: a small non-zero learning rate
        m.flow.mark_weights_updated()
        if isinstance(getattr(m, "_prefetched", None), dict):
            m._prefetched.clear()
        torch.manual_seed(11)                                              # the encoder's reparameterisation noise (CPU generator)
        losses = []
        for step in range(2):
            losses.append(tr.train_step(batch, step, next_batch=batch if step == 0 else None))
        torch.cuda.synchronize()
        assert tr.opt.param_groups[0]["lr"] > 0
        return m.flow.flat_grads.detach().clone(), flat.detach().clone(), torch.stack([l.detach() for l in losses]).cpu()

    g0, p0, l0 = two_steps()
    assert torch.isfinite(l0).all() and torch.isfinite(g0).all() and not torch.equal(p0, snap)
    for rep in range(2):
        g, p, l = two_steps()
        n_g, n_p = int((g != g0).sum()), int((p != p0).sum())
        print(f"{dtype} repetition {rep + 1}: losses {l.tolist()} vs {l0.tolist()}, differing gradient elements {n_g}, parameters {n_p}")
        assert torch.equal(l, l0) and n_g == 0 and n_p == 0
        del g, p
    assert m.flow.engine.handoff_timeouts() == (0, 0)


def test_adam_in_the_conv2_weight_gradient_epilogue_trains_bit_identically(monkeypatch):
    """IPOKE_WGRAD_ADAM=1 (opt-in, measured slower on c2: DESIGN.md §7): the single-GPU train step applies Adam-amsgrad to conv2 of every
    coupling net (plain 1 x 1, 73 % of the parameters at full size; macow_utils.py:270-281, second_stage_video.py:648-650) in the epilogue
    of its weight-gradient GEMM (ipoke_wgrad_desc.adam) instead of writing the gradient and running adam_cast over it (default): three
    train steps from the same state, bf16 -- parameters and optimizer state of the two forms bit-identical; mode 2 also leaves the same
    gradient in the flat buffer."""
    from ipoke_amd.second_stage import PokeMotionModel
    from ipoke_amd.trainer import SecondStageTrainer
    from tests.helpers import synthetic_batch
    arch = configs.flow_arch(32, hidden=128, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 4           # PokeMotionModel derives flow_mid_channels = factor * z_dim = 128 (whole 128 x 128 tiles)
    conf = configs.second_stage_config(64, 32, 16, batch_size=4, arch=arch)
    batch = synthetic_batch(4, 16, 64, seed=5, device="cuda")

    def run(mode):
        monkeypatch.setenv("IPOKE_WGRAD_ADAM", mode)
        m = PokeMotionModel(conf, dirs={}, dtype="bf16", device="cuda", max_batch=4)
        assert m.flow.engine.cfg.hidden == 128
        deterministic_fill_(m.first_stage_model, prefix="first_stage.")
        deterministic_fill_(m.poke_embedder, prefix="poke_embedder.")
        deterministic_fill_(m.conditioner, prefix="conditioner.")
        deterministic_fill_(m.flow, prefix="flow.")
        m.flow.sync_buffers()
        m.flow.mark_weights_updated()
        tr = SecondStageTrainer(m)
        assert tr.native_opt and tr.overlap
        m.global_step = 300
        torch.manual_seed(4)
        losses = [tr.train_step(batch, k).item() for k in range(3)]
        torch.cuda.synchronize()
        assert m.flow.engine.handoff_timeouts() == (0, 0)
        return (losses, m.flow.flat_params.detach().clone(), tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone(), tr.opt.max_exp_avg_sq.clone(),
                m.flow.flat_grads.detach().clone())

    plain, fused, keep = run("0"), run("1"), run("2")
    assert all(np.isfinite(plain[0])) and plain[0] == fused[0] == keep[0], (plain[0], fused[0], keep[0])
    for name, a, b, c in zip(("params", "m", "v", "vmax"), plain[1:5], fused[1:5], keep[1:5]):
        assert torch.equal(a, b) and torch.equal(a, c), (name, int((a != b).sum()), int((a != c).sum()))
    assert torch.equal(plain[5], keep[5]), "mode 2 writes the gradient the epilogue consumed"
    assert not torch.equal(plain[5], fused[5]), "the fused form must not have written conv2's gradient (is it active?)"
