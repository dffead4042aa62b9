"""The first-stage training step as ``bench.py --config c4 / c4gan`` executes it: TRAIN mode (one spectral-norm power iteration
per decoder call, i.e. 15 different sigma per weight per step), the power iterations run AHEAD of the frame loop
(``ipoke_spectral_sigma_multi``), the SPADE maps hoisted out of the frame loop, 128x128, z = 32, T = 16.

Golden ``g13_first_stage_train_mode_128`` is the reference's own ``SpadeCondMotionModel`` in ``.train()`` on one clip
(oracle/make_goldens.py::g13_train_mode; models/first_stage_motion_model.py:498-522, models/modules/autoencoders/util.py:52, 252).
"""
import ctypes
import zlib

import numpy as np
import pytest
import torch

from ipoke_amd import _lib, configs
from ipoke_amd.utils.detfill import deterministic_fill_
from tests.conftest import t

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = 16

# f32 (exact-f32 matrix cores): the bounds of the eval-mode slices (tests/test_vae_gpu.py); measured on MI355X: X_hat 6.6e-5, u / v
# 1.2e-7, gradient sums 1.4e-3, sampled elements 3.0e-3 (max) / 6e-4 (mean) -- the L1 sub-gradient sign(x_hat - x) flips for pixels
# whose residual is below the forward difference, which perturbs every upstream gradient at the 1e-3 level.
# bf16: the max and the mean over the sampled elements are bounded separately.  The yardstick is the REFERENCE ITSELF with its
# convolutions in bf16 (scripts/ref_bf16_autocast.py: the reference's SpadeCondMotionModel, train mode, under
# torch.autocast(bfloat16) against its own fp32 run, 64x64, T = 4): X_hat moves by 0.17 (max) / 9.7e-3 (mean) and, in the metrics
# of grad_report, the gradients by 0.148 (worst sum / abs-sum), 0.33 (worst sampled element), 0.086 (mean sampled element) -- the
# same tensors that are worst here (bias and norm-affine vectors of the encoder, the GRU gate biases: sums of the loss gradient over
# all positions, where the sign flips of the L1 term do not average out), the same magnitudes (measured here, T = 16, 128x128:
# 0.17 / 0.32 / 0.075).  A bf16 run cannot be closer to the fp32 reference than bf16 arithmetic lets the reference be to itself;
# the bounds are 1.5x the reference's own deviation.  f32 mode is the tight check of the same code path.
TOL = {"f32": dict(x_max=2e-4, x_mean=1e-5, loss=2e-5, mu=2e-4, sum=5e-3, smp_max=3e-2, smp_mean=5e-3, u=2e-5),
       # bf16: every entry the yardstick fixture holds (tests/golden/g15_ref_bf16_autocast.npz, scripts/ref_bf16_autocast.py) is at most
       # 1.5x the reference's own bf16-autocast deviation -- asserted on the CPU by tests/test_bf16_yardstick_cpu.py.  Measured on the
       # MI355X (four variants, B = 1 / 4 / 10): x_max <= 0.204, x_mean 8.6e-3, loss 3.3e-5, sum <= 0.175, smp_max <= 0.339, smp_mean <= 0.081.
       # `mu` (the encoder's posterior mean, not in the fixture) is measured 5.0e-2.
       # `sum` and `smp_max` are MAXIMA over 152 tensors (extreme values of a noisy quantity: the run-to-run spread of the fp32 atomics alone
       # moves them between 0.10 and 0.21 resp. 0.27 and 0.48 over 24 runs, i.e. up to 1.40x / 1.46x the reference's single run): 2x there.
       # Round 5: the step is reproducible (no float atomics in the spectral norm: test_first_stage_train_steps_are_reproducible), so that
       # `sum` and `smp_max` are single values again and carry the same 1.5x as every other entry (0.148 -> 0.22, 0.33 -> 0.5).
       "bf16": dict(x_max=0.25, x_mean=1.4e-2, loss=1e-4, mu=6e-2, sum=0.22, smp_max=0.5, smp_mean=0.125, u=2e-5)}


def train_model(dtype):
    from ipoke_amd.first_stage import SpadeCondMotionModel
    m = SpadeCondMotionModel(configs.first_stage_config(128, 32, T), dirs={}, train=False, dtype=dtype)
    deterministic_fill_(m, prefix="first_stage.")
    return m.to(DEV).train()


def clip(g, copies=1):
    X = torch.rand(1, T, 3, 128, 128, generator=torch.Generator().manual_seed(int(g["X_seed"]))) * 2 - 1
    return X.repeat(copies, 1, 1, 1, 1).to(DEV), t(g["eps"], DEV).repeat(copies, 1, 1, 1)


def frame_checksum(x, key):
    x = x.detach().double().flatten().cpu()
    idx = torch.randint(0, x.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(key.encode())))
    return np.array([x.sum().item(), x.abs().sum().item(), *x[idx].tolist()])


def grad_report(m, g, dtype, tag):
    """Every parameter gradient against the reference's checksums.  Per tensor: |sum - ref| and |abs-sum - ref| relative to
    the reference abs-sum, three sampled elements relative to the tensor's largest gradient."""
    tol = TOL[dtype]
    params = dict(m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert set(names) == {k for k, p in params.items() if p.grad is not None}
    cks = g["grad_checksums"]
    worst_sum, worst_smp, smp_all, bad = ("", 0.0), ("", 0.0), [], []
    for k, ck in zip(names, cks):
        gr = params[k].grad.detach().double().flatten().cpu()
        idx = torch.randint(0, gr.numel(), (3,), generator=torch.Generator().manual_seed(zlib.crc32(k.encode())))
        ref_abs = max(ck[1], 1e-12)
        wkey = k.replace(".bias", ".weight_orig")
        if k.endswith(".bias") and wkey in names and ck[1] <= 1e-4 * cks[names.index(wkey)][1]:
            # a bias in front of an Instance / GroupNorm: analytically zero gradient, both sides hold cancellation noise
            assert gr.abs().sum().item() <= 1e-3 * cks[names.index(wkey)][1], k
            continue
        e_sum = max(abs(gr.sum().item() - ck[0]), abs(gr.abs().sum().item() - ck[1])) / ref_abs
        e_smp = max(abs(gr[i].item() - r) for i, r in zip(idx.tolist(), ck[2:])) / max(gr.abs().max().item(), 1e-12)
        smp_all.append(e_smp)
        if e_sum > worst_sum[1]:
            worst_sum = (k, e_sum)
        if e_smp > worst_smp[1]:
            worst_smp = (k, e_smp)
        if e_sum > tol["sum"] or e_smp > tol["smp_max"]:
            bad.append((k, float(e_sum), float(e_smp)))
    smp_mean = float(np.mean(smp_all))
    print(f"{tag}[{dtype}] gradients of {len(names)} tensors: worst sum/abs-sum error {worst_sum[1]:.3e} ({worst_sum[0]}), "
          f"sampled elements max {worst_smp[1]:.3e} ({worst_smp[0]}) mean {smp_mean:.3e}")
    for b in bad:
        print("   BAD", b)
    assert not bad and smp_mean <= tol["smp_mean"]


def check_step(m, g, dtype, tag, loss, X_hat, mu, slots=1):
    tol = TOL[dtype]
    ref_frames = t(g["X_hat_frames"])                       # frames 0, 7, 14 of the single golden clip
    for b in range(slots):
        d = (X_hat[b:b + 1, [0, 7, 14]].detach().cpu() - ref_frames).abs()
        assert d.max().item() <= tol["x_max"] and d.mean().item() <= tol["x_mean"], (tag, b, d.max().item(), d.mean().item())
    d0 = (X_hat[:1, [0, 7, 14]].detach().cpu() - ref_frames).abs()
    e_cs = 0.0
    for i in range(T - 1):                                   # every frame through its checksum (sum and abs-sum per pixel)
        cs = frame_checksum(X_hat[:1, i], f"frame{i}")
        ref = g["X_hat_checksums"][i]
        e_cs = max(e_cs, abs(cs[0] - ref[0]) / X_hat[0, i].numel(), abs(cs[1] - ref[1]) / X_hat[0, i].numel())
    e_l = abs(loss.item() - float(g["loss"])) / max(1.0, abs(float(g["loss"])))
    e_mu = (mu[:1].detach().cpu() - t(g["mu"])).abs().max().item()
    print(f"{tag}[{dtype}] X_hat err max {d0.max().item():.3e} mean {d0.mean().item():.3e}, per-frame mean-pixel checksum err {e_cs:.3e}, "
          f"loss {loss.item():.6f} (ref {float(g['loss']):.6f}), mu err {e_mu:.3e}")
    assert e_cs <= tol["x_mean"] and e_l <= tol["loss"] and e_mu <= tol["mu"]
    # u, v of every spectral-normalised decoder convolution after the step's 15 iterations (fp32 on both sides)
    sd = m.state_dict()
    e_u = 0.0
    for k in [str(n) for n in g["u_names"]]:
        e_u = max(e_u, (sd[k].cpu() - t(g["u." + k])).abs().max().item())
        vk = k[:-1] + "v"
        cs = frame_checksum(sd[vk], vk)
        e_u = max(e_u, float(np.abs(cs[2:] - g["v_checksum." + k][2:]).max()))
    print(f"{tag}[{dtype}] weight_u / weight_v after {T - 1} power iterations: err {e_u:.3e} ({len(g['u_names'])} convolutions)")
    assert e_u <= tol["u"]


@pytest.mark.parametrize("variant,dtype", [("default", "f32"), ("default", "bf16"), ("per_frame", "f32"), ("per_frame", "bf16"),
                                           ("no_sn_ahead", "f32"), ("no_spade_hoist", "f32")])
def test_first_stage_train_mode_step(golden, monkeypatch, variant, dtype):
    """One clip through the c4 code path (train mode): with the default switches (round 4: ALL frames decoded as one (frame, clip)-ordered
    batch, one operand of W_orig, 1 / sigma_t in the GEMM epilogues, one weight gradient per convolution + the rank-1 sigma terms);
    frame by frame (IPOKE_C4_PER_FRAME=1: the default of rounds 2-3); frame by frame with the power iterations at the call sites
    (IPOKE_NO_SN_AHEAD=1: the reference's literal order of work); and with per-frame SPADE maps (IPOKE_NO_SPADE_HOIST=1).  All must
    reproduce the reference's X_hat, loss, gradients and u / v buffers."""
    from ipoke_amd import first_stage_train as FT
    g = golden("g13_first_stage_train_mode_128")
    monkeypatch.setattr(FT, "_FRAME_BATCH", variant == "default")
    monkeypatch.setattr(FT, "_SN_AHEAD", variant != "no_sn_ahead")
    monkeypatch.setattr(FT, "_HOIST_SPADE", variant != "no_spade_hoist")
    m = train_model(dtype)
    X, eps = clip(g)
    loss, X_hat, mu, lv = m.training_loss(X, eps)           # power_iteration=None -> model.training -> True
    loss.backward()
    check_step(m, g, dtype, f"c4-train-mode/{variant}", loss, X_hat, mu)
    grad_report(m, g, dtype, f"c4-train-mode/{variant}")


@pytest.mark.parametrize("dtype,copies", [("f32", 4), ("bf16", 4), ("bf16", 10)])
def test_first_stage_train_mode_batch_of_copies(golden, dtype, copies):
    """B = 4 / 10, T = 16: clips are independent, so copies of the golden clip reproduce the golden reconstruction in every slot, and the
    mean-loss gradients equal the single clip's (checked against the reference's checksums of every tensor).  B = 10 is the size class
    of ``bench.py --config c4``: the decoder sees 150 images per launch -- 9 600 patches of 16 x 16 pixels at the
    128 x 128 layers, conv3x3_c64 by the default rule with ~38 patches per persistent workgroup (forward, data gradient, the four
    scattered sub-pixel phases), conv3x3_halo16 on the 64 x 64 / 128-channel layers, ~512-workgroup split-M weight gradients."""
    g = golden("g13_first_stage_train_mode_128")
    m = train_model(dtype)
    X, eps = clip(g, copies=copies)
    loss, X_hat, mu, lv = m.training_loss(X, eps)
    loss.backward()
    check_step(m, g, dtype, f"c4-train-mode/B={copies}", loss, X_hat, mu, slots=copies)
    grad_report(m, g, dtype, f"c4-train-mode/B={copies}")


def test_spectral_sigma_multi_matches_sequential_calls():
    """ipoke_spectral_sigma_multi (K iterations of several weights, 3 launches per iteration) against K sequential
    ipoke_spectral_sigma(iterate = 1) calls per weight -- sigma, 1/sigma, every u | v snapshot, the final buffers -- and against
    torch's own arithmetic (F.normalize(W^T u), F.normalize(W v), u.W v) in float64."""
    import torch.nn.functional as F
    L = _lib.lib()
    K_IT = 7
    gen = torch.Generator().manual_seed(5)
    shapes = [(256, 64, 9, 0), (64, 128, 9, 1), (3, 64, 9, 0), (256, 256, 9, 1), (128, 32, 1, 0)]      # rows, cols, taps, transposed
    ws, us, vs = [], [], []
    for r, c, tp, tr in shapes:
        k = int(tp ** 0.5)
        shape = (c, r, k, k) if tr else (r, c, k, k)
        ws.append((torch.randn(*shape, generator=gen) * 0.1).to(DEV))
        us.append(F.normalize(torch.randn(r, generator=gen), dim=0).to(DEV))
        vs.append(F.normalize(torch.randn(c * tp, generator=gen), dim=0).to(DEV))

    def wsz(r, c, tp):
        return int(L.ipoke_spectral_workspace_floats(r, c, tp))

    # ---- sequential single-weight calls
    seq = []
    for (r, c, tp, tr), w, u0, v0 in zip(shapes, ws, us, vs):
        u, v = u0.clone(), v0.clone()
        wsb = torch.zeros(wsz(r, c, tp), device=DEV)
        sig = torch.empty(K_IT, 2, device=DEV); snap = torch.empty(K_IT, r + c * tp, device=DEV)
        for k in range(K_IT):
            _lib.check(L.ipoke_spectral_sigma(_lib.ptr(w), r, c, tp, tr, _lib.ptr(u), _lib.ptr(v), 1, 1e-12, _lib.ptr(sig[k]), _lib.ptr(snap[k]),
                                              _lib.ptr(wsb), _lib.current_stream()))
        seq.append((sig, snap, u, v))
    # ---- all weights, all iterations
    jobs = (_lib.SnJob * len(shapes))()
    keep = []
    for i, ((r, c, tp, tr), w, u0, v0) in enumerate(zip(shapes, ws, us, vs)):
        u, v = u0.clone(), v0.clone()
        wsb = torch.zeros(wsz(r, c, tp), device=DEV)
        sig = torch.empty(K_IT, 2, device=DEV); snap = torch.empty(K_IT, r + c * tp, device=DEV)
        keep.append((sig, snap, u, v, wsb))
        j = jobs[i]
        j.w = w.data_ptr(); j.cout, j.cin, j.taps, j.transposed = r, c, tp, tr
        j.u = u.data_ptr(); j.v = v.data_ptr(); j.out = sig.data_ptr(); j.out_stride = 2
        j.snap = snap.data_ptr(); j.snap_stride = r + c * tp; j.workspace = wsb.data_ptr()
    jobs_dev = torch.empty(len(shapes) * int(L.ipoke_sn_job_size()), dtype=torch.uint8, device=DEV)
    _lib.check(L.ipoke_sn_jobs_upload(ctypes.byref(jobs), len(shapes), _lib.ptr(jobs_dev), _lib.current_stream()))
    _lib.check(L.ipoke_spectral_sigma_multi(_lib.ptr(jobs_dev), len(shapes), max(s[0] for s in shapes), max(s[1] * s[2] for s in shapes), K_IT,
                                            1e-12, _lib.current_stream()))
    torch.cuda.synchronize()
    for i, ((r, c, tp, tr), w, u0, v0) in enumerate(zip(shapes, ws, us, vs)):
        sig_s, snap_s, u_s, v_s = seq[i]
        sig_m, snap_m, u_m, v_m, _ = keep[i]
        e_seq = max((sig_s - sig_m).abs().max().item() / sig_s.abs().max().item(), (snap_s - snap_m).abs().max().item(),
                    (u_s - u_m).abs().max().item(), (v_s - v_m).abs().max().item())
        # torch arithmetic in float64
        W = (w.transpose(0, 1) if tr else w).reshape(r, -1).double().cpu()
        u, v = u0.double().cpu(), v0.double().cpu()
        e_ref = 0.0
        for k in range(K_IT):
            v = F.normalize(W.t() @ u, dim=0, eps=1e-12); u = F.normalize(W @ v, dim=0, eps=1e-12)
            sigma = torch.dot(u, W @ v).item()
            e_ref = max(e_ref, abs(sig_m[k, 0].item() - sigma) / abs(sigma), abs(sig_m[k, 1].item() * sigma - 1.0),
                        (snap_m[k, :r].double().cpu() - u).abs().max().item(), (snap_m[k, r:].double().cpu() - v).abs().max().item())
        print(f"sigma_multi weight {i} {shapes[i]}: vs sequential calls {e_seq:.2e}, vs float64 torch arithmetic {e_ref:.2e}")
        assert e_seq <= 1e-6 and e_ref <= 2e-5


@pytest.mark.parametrize("dtype,copies", [("f32", 1), ("bf16", 4)])
def test_weight_gradients_on_a_second_stream(golden, dtype, copies):
    """``wgrad_side_stream`` (FirstStageTrainer.step under IPOKE_C4_WGRAD_SIDE=1): the weight-gradient half of every convolution's backward
    queued on a second stream and parked on the parameter until the context ends.  Same kernels on the same data: the reference's
    reconstruction, loss and gradient checksums hold as for the plain backward (three repetitions on a fresh model each -- a missing
    ordering between the streams reads unfinished buffers), every parameter that has a gradient in the plain backward has one, and nothing
    is left parked.  (Two plain backward passes are not bit-identical either -- fp32 atomics in the norm statistics -- so the bar is the
    golden one.)"""
    from ipoke_amd import first_stage_train as FT
    g = golden("g13_first_stage_train_mode_128")
    X, eps = clip(g, copies=copies)
    side = torch.cuda.Stream()
    for rep in range(3):
        m = train_model(dtype)
        loss, X_hat, mu, lv = m.training_loss(X, eps)
        with FT.wgrad_side_stream(side):
            loss.backward()
            parked = sum(hasattr(p, "_side_grad") for p in m.parameters())
        assert parked > 20 and not any(hasattr(p, "_side_grad") for p in m.parameters())
        assert FT._WGRAD_SIDE["stream"] is None and not FT._WGRAD_SIDE["params"]
        check_step(m, g, dtype, f"c4-train-mode/wgrad-side/{rep}", loss, X_hat, mu, slots=copies)
        grad_report(m, g, dtype, f"c4-train-mode/wgrad-side/{rep}")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_first_stage_train_steps_are_reproducible(golden, dtype):
    """two c4 train steps (forward, L1 + KL, backward with the weight gradients on their second stream, Adam) from the
    same initial state are BIT-identical -- reconstruction, loss, every parameter and the spectral-norm u / v buffers after each of the
    two steps.  (Until round 4 the spectral-norm power iteration summed its row chunks with float atomics: sigma differed in its last
    bits from run to run, and bf16 roundings downstream turned that into gradient differences of several per cent of a tensor's
    maximum -- the run-to-run spread the bf16 bounds above had to absorb.)"""
    from ipoke_amd.first_stage_train import FirstStageTrainer
    g = golden("g13_first_stage_train_mode_128")
    X, eps = clip(g, copies=4)
    X2 = X.flip(0).contiguous() * 0.9

    def run():
        m = train_model(dtype)
        tr = FirstStageTrainer(m)
        l1, xh1 = tr.step(X, eps)
        l2, xh2 = tr.step(X2, eps)
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        return l1.clone(), xh1.clone(), l2.clone(), xh2.clone(), sd

    a = run()
    for rep in range(2):
        b = run()
        assert torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]), "reconstructions differ between identical runs"
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]), (a[0].item(), b[0].item(), a[2].item(), b[2].item())
        diff = [k for k in a[4] if not torch.equal(a[4][k], b[4][k])]
        assert not diff, f"{len(diff)} tensors differ after two identical steps, e.g. {diff[:5]}"
