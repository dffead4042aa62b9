"""GPU parity of the native flow engine against the reference's golden vectors and the CPU oracle."""
import copy

import numpy as np
import pytest
import torch

from ipoke_amd import configs
from ipoke_amd.utils.detfill import deterministic_fill_
from tests.conftest import t

pytestmark = pytest.mark.gpu

# tolerances: f32 MFMA mode vs reference fp32 CPU; bf16 mode looser
# f32: exact-f32 MFMA vs the reference's fp32 CPU run (4x the oracle's own 2e-5 / 1e-3 bounds).
# bf16: nets take bf16 inputs (8-bit mantissa, ~4e-3 relative per contraction), transforms stay fp32; the
# golden outputs reach |out| ~ 8, so 6e-2 absolute is < 1 % of range; logdet within 0.5 % of |logdet| ~ 65-75.
TOL = {"f32": dict(out=8e-5, logdet=4e-3, grad=2e-3, rev=2e-4),
       "bf16": dict(out=6e-2, logdet=0.35, grad=6e-2, rev=6e-2)}


def build(arch, dtype):
    from ipoke_amd.flow import SupervisedMacowTransformer
    m = SupervisedMacowTransformer(copy.deepcopy(arch), dtype=dtype, device="cuda", init="none")
    deterministic_fill_(m, prefix="flow.")
    m.sync_buffers()
    return m


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reduced_flow_forward_reverse(golden, dtype):
    g = golden("g2_reduced_flow")
    m = build(configs.reduced_flow_arch(), dtype).eval()
    x, cond = t(g["x"], "cuda"), t(g["cond"], "cuda")
    with torch.no_grad():
        out, logdet = m(x, cond)
    tol = TOL[dtype]
    e_out = (out.cpu() - t(g["out"])).abs().max().item()
    e_ld = (logdet.cpu() - t(g["logdet"])).abs().max().item()
    print(f"[{dtype}] fwd: out err {e_out:.3e}, logdet err {e_ld:.3e} (logdet {g['logdet']})")
    assert e_out <= tol["out"] and e_ld <= tol["logdet"]
    rev = m(t(g["out"], "cuda"), cond, reverse=True)
    e_rev = (rev.cpu() - t(g["reverse"])).abs().max().item()
    print(f"[{dtype}] reverse err {e_rev:.3e}")
    assert e_rev <= tol["rev"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reduced_flow_gradients(golden, dtype):
    g = golden("g2_reduced_flow")
    m = build(configs.reduced_flow_arch(), dtype).train()
    x, cond = t(g["x"], "cuda"), t(g["cond"], "cuda")
    out, logdet = m(x, cond)
    loss = (0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()
    assert abs(loss.item() - float(g["loss"])) <= 1e-3 * abs(float(g["loss"])) + TOL[dtype]["logdet"]
    loss.backward()
    worst, worst_key = 0.0, None
    for name, p in m.named_parameters():
        ref = t(g["grad." + name])
        got = p.grad.cpu()
        err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
        if err > worst:
            worst, worst_key = err, name
    print(f"[{dtype}] worst relative grad error {worst:.3e} at {worst_key}")
    assert worst <= TOL[dtype]["grad"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reduced_flow_with_lu_1x1_convs(golden, dtype):
    """use1x1: the per-level shuffle layers are LU-parametrised invertible 1x1 convolutions (macow2.py:596-649, 862) --
    forward, log-det, reverse and every parameter gradient (incl. l, u, log_s) against the reference (G2-LU)."""
    g = golden("g2_reduced_flow_lu")
    arch = configs.reduced_flow_arch(); arch["use1x1"] = True
    from ipoke_amd.flow import SupervisedMacowTransformer
    m = SupervisedMacowTransformer(copy.deepcopy(arch), dtype=dtype, device="cuda", init="none")
    deterministic_fill_(m, prefix="flow.")
    m.load_state_dict({k[3:]: t(v) for k, v in g.items() if k.startswith("lu.")}, strict=False)
    m.sync_buffers()
    m.train()
    x, cond = t(g["x"], "cuda"), t(g["cond"], "cuda")
    out, logdet = m(x, cond)
    tol = TOL[dtype]
    e_out = (out.detach().cpu() - t(g["out"])).abs().max().item()
    e_ld = (logdet.detach().cpu() - t(g["logdet"])).abs().max().item()
    print(f"[{dtype}] LU flow: out err {e_out:.3e}, logdet err {e_ld:.3e}")
    assert e_out <= tol["out"] and e_ld <= tol["logdet"]
    loss = (0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()
    loss.backward()
    worst, worst_key = 0.0, None
    for name, p in m.named_parameters():
        ref = t(g["grad." + name])
        err = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
        if err > worst:
            worst, worst_key = err, name
    print(f"[{dtype}] LU flow: worst relative grad error {worst:.3e} at {worst_key}")
    assert worst <= tol["grad"]
    with torch.no_grad():
        rev = m(t(g["out"], "cuda"), cond, reverse=True)
    e_rev = (rev.cpu() - t(g["reverse"])).abs().max().item()
    print(f"[{dtype}] LU flow: reverse err {e_rev:.3e}")
    assert e_rev <= tol["rev"]


def _condition_nice_arch(h_channels=32):
    arch = configs.reduced_flow_arch()
    arch["condition_nice"] = True
    arch["h_channels"] = h_channels
    return arch


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_condition_nice_flow(golden, dtype):
    """condition_nice (macow2.py:1024-1060, 553; macow_utils.py:275-283, 328-332): conv3 of every NICE net takes hidden + h_channels
    inputs, the last h_channels being ELU(h) -- forward, log-det, reverse and every parameter gradient against the reference (G16)."""
    g = golden("g16_condition_nice")
    m = build(_condition_nice_arch(), dtype).train()
    assert tuple(m.state_dict()["flow.layers.0.0.coupling1_up.net.conv3.conv.weight_v"].shape) == (16, 96, 3, 3)
    x, cond = t(g["x"], "cuda"), t(g["cond"], "cuda")
    out, logdet = m(x, cond)
    tol = dict(TOL[dtype])
    if dtype == "bf16":       # the same relative bounds as TOL["bf16"] (0.5 % of the range; G2 reaches |out| 12.8, |logdet| 70 -- G16 16.2 and 110)
        tol["out"] = tol["rev"] = 5e-3 * float(np.abs(g["out"]).max())
        tol["logdet"] = 5e-3 * float(np.abs(g["logdet"]).max())
    e_out = (out.detach().cpu() - t(g["out"])).abs().max().item()
    e_ld = (logdet.detach().cpu() - t(g["logdet"])).abs().max().item()
    print(f"[{dtype}] condition_nice: out err {e_out:.3e}, logdet err {e_ld:.3e}")
    assert e_out <= tol["out"] and e_ld <= tol["logdet"]
    loss = (0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()
    loss.backward()
    worst, worst_key = 0.0, None
    for name, p in m.named_parameters():
        ref = t(g["grad." + name])
        err = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
        if err > worst:
            worst, worst_key = err, name
    print(f"[{dtype}] condition_nice: worst relative grad error {worst:.3e} at {worst_key}")
    assert worst <= tol["grad"]
    with torch.no_grad():
        rev = m(t(g["out"], "cuda"), cond, reverse=True)
        e_rev = (rev.cpu() - t(g["reverse"])).abs().max().item()
        print(f"[{dtype}] condition_nice: reverse err {e_rev:.3e}")
        assert e_rev <= tol["rev"]
        out0, _ = m(x, torch.zeros_like(cond))
        assert (out0 - out).abs().max().item() > 1e-3            # the conditioning columns are live


def test_condition_nice_wide_vs_oracle():
    """The same option at the shipped conditioning width (128 channels), hidden = 256 (conv3 on the stationary 3x3 kernel, K = 384
    per tap), B = 5, f32 mode against the CPU oracle: forward, gradients of the conv3 weights (both column groups), reverse."""
    from oracle import flow_ref
    arch = configs.flow_arch(16, hidden=256, num_steps=[1, 1], factor=4)
    arch["condition_nice"] = True
    m = build(arch, "f32").train()
    o = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch))
    deterministic_fill_(o, prefix="flow.")
    gen = torch.Generator().manual_seed(5)
    x, cond = torch.randn(5, 16, 8, 8, generator=gen), torch.randn(5, 128, 8, 8, generator=gen)
    out, logdet = m(x.cuda(), cond.cuda())
    oo, ol = o(x, cond)
    assert (out.detach().cpu() - oo).abs().max().item() <= TOL["f32"]["out"] * 4
    assert (logdet.detach().cpu() - ol).abs().max().item() <= TOL["f32"]["logdet"] * 4
    ((0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()).backward()
    ((0.5 * (oo ** 2).sum(dim=[1, 2, 3])).mean() - ol.mean()).backward()
    og = dict(o.named_parameters())
    worst = 0.0
    for name, p in m.named_parameters():
        ref = og[name].grad
        worst = max(worst, (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6))
        if name.endswith("coupling1_up.net.conv3.conv.weight_v"):
            gc = p.grad.cpu()
            for lo, hi in ((0, 256), (256, 384)):           # hidden columns, conditioning columns
                e = (gc[:, lo:hi] - ref[:, lo:hi]).abs().max().item() / (ref[:, lo:hi].abs().max().item() + 1e-6)
                assert e <= TOL["f32"]["grad"], (name, lo, e)
    print(f"condition_nice wide: worst relative grad error {worst:.3e}")
    assert worst <= TOL["f32"]["grad"]
    with torch.no_grad():
        rev = m(oo.detach().cuda(), cond.cuda(), reverse=True)
        assert (rev.cpu() - x).abs().max().item() <= TOL["f32"]["rev"] * 4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_condition_nice_with_lu_convs_vs_oracle(dtype):
    """Both optional branches together (condition_nice + use1x1) at an odd batch size (B = 3: the reverse pass's 128-row tiles are ragged),
    against the CPU oracle: forward, log-det, every gradient, reverse."""
    from oracle import flow_ref
    arch = _condition_nice_arch(64)
    arch["use1x1"] = True
    np.random.seed(12)
    o = flow_ref.SupervisedMacowTransformer(copy.deepcopy(arch))
    lu_state = {k: v.clone() for k, v in o.state_dict().items() if ".shuffle_layers." in k}   # the constructor's P, L, U draws (a name-keyed fill is not a permutation)
    m = build(arch, dtype)
    m.load_state_dict(lu_state, strict=False)
    m.sync_buffers()
    m.train()
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    gen = torch.Generator().manual_seed(21)
    x, cond = torch.randn(3, 16, 8, 8, generator=gen), torch.randn(3, 64, 8, 8, generator=gen)
    out, logdet = m(x.cuda(), cond.cuda())
    oo, ol = o(x, cond)
    scale, lscale = oo.abs().max().item(), ol.abs().max().item()
    tol_o, tol_l, tol_g = (4e-4, 2e-2, 4e-3) if dtype == "f32" else (7.5e-3 * scale, 5e-3 * lscale, 6e-2)
    assert (out.detach().cpu() - oo).abs().max().item() <= tol_o and (logdet.detach().cpu() - ol).abs().max().item() <= tol_l
    ((0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()).backward()
    ((0.5 * (oo ** 2).sum(dim=[1, 2, 3])).mean() - ol.mean()).backward()
    og = dict(o.named_parameters())
    worst = max((p.grad.cpu() - og[n].grad).abs().max().item() / (og[n].grad.abs().max().item() + 1e-6) for n, p in m.named_parameters())
    print(f"[{dtype}] condition_nice + use1x1: worst relative grad error {worst:.3e}")
    assert worst <= tol_g
    with torch.no_grad():
        rev = m(oo.detach().cuda(), cond.cuda(), reverse=True)
    assert (rev.cpu() - x).abs().max().item() <= (1e-3 if dtype == "f32" else 7.5e-3 * scale)


def test_reduced_flow_data_init(golden):
    """First forward with initialized == 0 (data-dependent ActNorm init, zero-init couplings)."""
    g = golden("g2_reduced_flow_init")
    from ipoke_amd.flow import SupervisedMacowTransformer
    from ipoke_amd.utils.detfill import fill_value
    m = SupervisedMacowTransformer(configs.reduced_flow_arch(), dtype="f32", device="cuda", init="none")
    sd = m.state_dict()
    big = set(g["big_keys"].tolist())
    with torch.no_grad():
        for k, v in sd.items():
            if k in big:
                v.copy_(fill_value("flow." + k, v).to(v.device))
            else:
                v.copy_(t(g["pre." + k]).to(v.device))
    m.sync_buffers()
    assert not m._initialized
    with torch.no_grad():
        out, logdet = m(t(g["x"], "cuda"), t(g["cond"], "cuda"))
    assert (out.cpu() - t(g["out_filled"])).abs().max().item() <= 2e-5
    assert (logdet.cpu() - t(g["logdet_filled"])).abs().max().item() <= 2e-3
    for k, v in m.state_dict().items():
        if k.endswith(("log_scale", "bias", "weight_g")):
            assert (v.cpu() - t(g["postfilled." + k])).abs().max().item() <= 2e-5, k
        if k.endswith("initialized"):
            assert int(v) == 1


def test_backward_in_pieces_matches_and_covers_all_parameters():
    """ipoke_flow_backward_pieces (the overlap path of data-parallel training): same gradients as the one-shot backward,
    and the ready callback announces every parameter exactly once, last levels first."""
    from ipoke_amd import configs
    from ipoke_amd.flow import SupervisedMacowTransformer
    arch = configs.reduced_flow_arch()
    m = SupervisedMacowTransformer(arch, dtype="f32", max_batch=4, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(4, arch["flow_in_channels"], 8, 8, device="cuda", generator=gen)
    cond = torch.randn(4, arch["h_channels"], 8, 8, device="cuda", generator=gen)
    with torch.no_grad():
        m(x, cond)
        for name, p in m.named_parameters():
            if name.endswith("weight_g"):
                p.fill_(0.07)
    m.mark_weights_updated()
    m.train()

    def run():
        m.flat_grads.zero_()
        out, logdet = m(x, cond)
        ((out ** 2).sum() * 0.5 - logdet.sum()).backward()
        torch.cuda.synchronize()
        return m.flat_grads.clone()

    ref = run()
    again = run()
    noise = (again - ref).abs().max().item()        # the split-K data gradients accumulate with fp32 atomics
    ranges = []
    stream = torch.cuda.Stream()
    m.engine.grad_ready_hook = (5, stream, lambda b, e: ranges.append((b, e)))
    got = run()
    m.engine.grad_ready_hook = None
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= max(4 * noise, 1e-6 * scale), ((got - ref).abs().max().item(), noise, scale)
    assert len(ranges) >= 3
    covered = torch.zeros(m.engine.n_params, dtype=torch.int32)
    for b, e in ranges:
        covered[b:e] += 1
    assert int(covered.min()) == 1 and int(covered.max()) == 1
    # per region (layers.*, priors.*) the ranges arrive from the end of the flat buffer towards its start
    prior0 = min(off for name, off, shape, kind in m.engine.tensors if name.startswith("flow.priors.") and kind == 0)
    for region in (lambda b: b < prior0, lambda b: b >= prior0):
        starts = [b for b, _ in ranges if region(b)]
        assert starts and starts == sorted(starts, reverse=True)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_hipgraph_replay_is_bit_identical_to_eager(golden, dtype):
    """configs[4] (sampling, "hipGraph-captured"): reverse and forward replayed from captured graphs give the same bits as
    the eager launches (call 1 eager, call 2 captures, calls 3+ replay)."""
    g = golden("g2_reduced_flow")
    m = build(configs.reduced_flow_arch(), dtype).eval()
    x, cond = t(g["x"], "cuda"), t(g["cond"], "cuda")
    z = t(g["out"], "cuda")
    with torch.no_grad():
        ref_rev = m(z, cond, reverse=True).clone()
        ref_out, ref_ld = m(x, cond)
        ref_out, ref_ld = ref_out.clone(), ref_ld.clone()
        m.set_graph_mode(True)
        for it in range(4):
            rev = m(z, cond, reverse=True)
            out, ld = m(x, cond)
            assert torch.equal(rev, ref_rev) and torch.equal(out, ref_out) and torch.equal(ld, ref_ld), it
        m.set_graph_mode(False)
        assert torch.equal(m(z, cond, reverse=True), ref_rev)


@pytest.mark.parametrize("spatial_mean", [False, True])
def test_flow_loss_value_and_gradient(spatial_mean):
    """FlowLoss (loss.py:6-31, 75-79) incl. the spatial_mean form: value and both gradients against the oracle's autograd."""
    from ipoke_amd.loss import FlowLoss
    from oracle import flow_ref
    g = torch.Generator().manual_seed(2)
    x, ld = torch.randn(5, 32, 8, 8, generator=g), torch.randn(5, generator=g) * 30
    xo, ldo = x.clone().requires_grad_(True), ld.clone().requires_grad_(True)
    lo, _ = flow_ref.FlowLoss(spatial_mean=spatial_mean, logdet_weight=0.7)(xo, ldo)
    lo.backward()
    xg, ldg = x.cuda().requires_grad_(True), ld.cuda().requires_grad_(True)
    lg, log = FlowLoss(spatial_mean=spatial_mean, logdet_weight=0.7)(xg, ldg)
    lg.backward()
    assert abs(lg.item() - lo.item()) <= 2e-6 * abs(lo.item())
    assert (xg.grad.cpu() - xo.grad).abs().max().item() <= 1e-7 and (ldg.grad.cpu() - ldo.grad).abs().max().item() <= 1e-8
    ref_scale = 0.5 * 32 * (1 if spatial_mean else 64)             # E[reference_nll_loss] of a standard normal sample
    assert 0.7 * ref_scale < log["reference_nll_loss"].item() < 1.3 * ref_scale


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("hidden", [64, 192])
def test_fused_adam_shadow_refresh_is_bit_identical(dtype, hidden):
    """ipoke_flow_adam_range (Adam-amsgrad with the conv2 operands written by the optimizer kernel, the gaps updated by the
    segment kernel, the rest laid out by relayout) against the plain path (linear ipoke_adam_amsgrad_step over the flat buffer +
    ipoke_flow_prepare_weights): parameters, m, v, v_max and EVERY byte of the shadow buffer, over the whole buffer in one
    call and over the ranges a piecewise backward announces, three steps."""
    from ipoke_amd import optim as O
    arch = configs.flow_arch(16, hidden=hidden, num_steps=[2, 1, 1], factor=4)

    def run(fused, pieces):
        m = build(arch, dtype).train()
        eng = m.engine
        eng.prepare_weights()
        opt = O.FusedAdamAmsgrad(m, lr=1e-3, weight_decay=1e-5)
        n = eng.params.numel()
        grads = m.bind_grads()
        O._FUSE_SHADOWS = fused
        try:
            for step in range(3):
                # the same gradients for both paths (the engine's backward sums some of them with atomics, so two backward passes
                # do not agree to the last bit; the optimizer is what is compared here)
                grads.copy_(torch.randn(n, generator=torch.Generator().manual_seed(10 + step)).cuda() * 1e-2)
                if pieces is None:
                    opt.step()
                else:
                    opt.begin_step()
                    for b, e in pieces(eng, n):
                        opt.step_range(b, e)
                    opt.finish_step()
            torch.cuda.synchronize()
        finally:
            O._FUSE_SHADOWS = O._FUSE_DEFAULT
        return (eng.params.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.max_exp_avg_sq.clone(), eng.shadow.clone())

    def cut(eng, n):
        # tensor-aligned ranges: cut the flat buffer at three tensor starts (as the engine's level groups are)
        offs = sorted({off for name, off, shape, kind in eng.tensors if kind == 0})
        cuts = [0, offs[len(offs) // 4], offs[len(offs) // 2], offs[3 * len(offs) // 4], n]
        return [(cuts[i], cuts[i + 1]) for i in range(4)][::-1]

    # like against like: a cut may fall between a weight-norm gain and its direction tensor, whose shadow then depends on the order
    # of the ranges -- the plain and the fused path see the same order
    for pieces in (None, cut):
        ref = run(False, pieces)
        got = run(True, pieces)
        for name, a, b in zip(("params", "exp_avg", "exp_avg_sq", "max_exp_avg_sq", "shadow"), ref, got):
            assert torch.equal(a, b), (name, "whole" if pieces is None else "pieces", (a.float() - b.float()).abs().max().item())


@pytest.mark.parametrize("split", [2, 4])
def test_unit_row_split_engine_matches_one_workgroup_per_sample(golden, split, monkeypatch):
    """IPOKE_UNIT_SPLIT: the engine's fused MaCowUnit launches on 2 / 4 workgroups per sample (csrc/mcf_unit_split.hip).  States and
    data gradients are bit-identical by construction; log-dets and the bias / ActNorm gradients are sums over parts."""
    g = golden("g2_reduced_flow")
    arch = configs.reduced_flow_arch()
    x, cond = t(g["x"]).cuda(), t(g["cond"]).cuda()

    def run(s):
        monkeypatch.setenv("IPOKE_UNIT_SPLIT", str(s))
        m = build(arch, "bf16")
        m.train()
        m.flat_grads.zero_()
        out, logdet = m(x, cond)
        ((out ** 2).sum() * 0.5 - logdet.sum()).backward()
        torch.cuda.synchronize()
        return out.detach().clone(), logdet.detach().clone(), m.flat_grads.clone()

    o1, l1, g1 = run(1)
    o1b, l1b, g1b = run(1)
    noise = (g1b - g1).abs().max().item()           # the split-K data gradients of the coupling nets accumulate with fp32 atomics
    os_, ls_, gs_ = run(split)
    assert torch.equal(os_, o1)
    assert (ls_ - l1).abs().max().item() <= 1e-3
    scale = g1.abs().max().item()
    assert (gs_ - g1).abs().max().item() <= max(4 * noise, 2e-6 * scale), ((gs_ - g1).abs().max().item(), noise, scale)


@pytest.mark.parametrize("which", [0, 1])
def test_handoff_timeout_is_reported_and_the_scratch_recovers(which):
    """the row-split unit launches and the fused conv3 + coupling launches bound their spin-waits and count a
    time-out in word 0 of their exchange scratch -- which nothing read.  Now every eager pass ends with a poll of both words and the
    next entry point of the flow fails with IPOKE_ERR_STATE once it has seen a non-zero count (the pass that timed out finished on
    garbage), re-initialises the scratches, and the passes after that are clean again.  The time-out is injected through the test
    hook ipoke_flow_test_inject_timeout (a real one needs a wedged partner workgroup)."""
    import time
    from ipoke_amd import _lib, configs
    from ipoke_amd.flow import SupervisedMacowTransformer
    from ipoke_amd.utils.detfill import deterministic_fill_
    arch = configs.flow_arch(32, hidden=256, num_steps=[2, 1, 1], factor=4)
    m = SupervisedMacowTransformer(arch, dtype="bf16", device="cuda", init="none", max_batch=4)
    deterministic_fill_(m, prefix="flow.")
    m.sync_buffers()
    eng = m.engine
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 32, 8, 8, generator=g).cuda()
    cond = torch.randn(4, arch["h_channels"], 8, 8, generator=g).cuda()
    with torch.no_grad():
        ref, _ = m(x, cond)
        torch.cuda.synchronize()
        assert eng.handoff_timeouts() == (0, 0)
        _lib.check(eng.lib.ipoke_flow_test_inject_timeout(eng.handle, which, _lib.current_stream()))
        m(x, cond)                                   # this pass's poll carries the count ...
        torch.cuda.synchronize()
        counts = eng.handoff_timeouts()
        assert counts[which] == 1 and counts[1 - which] == 0
        with pytest.raises(RuntimeError, match="hand-off time-out"):
            m(x, cond)                               # ... and the next entry point refuses (and re-initialises the scratches)
        torch.cuda.synchronize()
        out, _ = m(x, cond)
        torch.cuda.synchronize()
        assert eng.handoff_timeouts() == (0, 0)
        assert torch.equal(out, ref), "the scratches must be in their initial state again"
