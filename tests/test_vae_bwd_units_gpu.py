"""Per-kernel GPU parity of the first-stage VAE backward kernels (csrc/vae_bwd.hip) and of the weight-gradient launch
modes against torch autograd on the CPU -- the op-level counterpart of tests/test_vae_gpu.py::test_first_stage_train_slice."""
from ctypes import byref

import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, nn as K, ops
from ipoke_amd import first_stage_train as FT
from ipoke_amd._lib import check, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {"f32": 2e-4, "bf16": 4e-2}


def _cl(x_nchw, dtype):
    return K.from_nchw(x_nchw.to(DEV), dtype)


def _cl5(x, dtype):
    """[N, C, (D,) H, W] -> CL (a 5-D tensor goes through the 4-D converter with the depth folded into the rows)."""
    if x.dim() == 4:
        return _cl(x, dtype)
    N, C, D, H, W = x.shape
    c = K.from_nchw(x.reshape(N, C, D * H, W).to(DEV), dtype)
    return K.CL(c.t, N, (D, H, W), C)


def _nchw(cl, dtype):
    return K.to_nchw(cl, dtype).cpu()


def _rel(a, b):
    """max error relative to the tensor's range.  In bf16 an activation that rounds across a kink (ReLU at 0) flips one
    element's mask, so there the 99.9th percentile is compared instead of the maximum."""
    d = (a.float() - b.float()).abs().flatten()
    scale = max(b.abs().max().item(), 1e-6)
    if a.dtype == torch.bfloat16 or _rel.bf16:
        k = max(1, int(d.numel() * 0.999))
        return d.kthvalue(k).values.item() / scale
    return d.max().item() / scale


_rel.bf16 = False


@pytest.fixture(autouse=True)
def _metric(request):
    _rel.bf16 = "bf16" in request.node.name
    yield
    _rel.bf16 = False


CASES = [
    # name, N, C, H, groups (0 = InstanceNorm), affine, act, residual, spade
    ("gn_relu_res", 3, 32, 16, 16, True, "relu", True, False),
    ("gn_elu", 2, 64, 8, 16, True, "elu", False, False),
    ("instnorm", 2, 16, 16, 0, False, "none", False, False),
    ("spade", 2, 32, 16, 16, False, "none", False, True),
    # the residual joins BEHIND the activation (ipoke_norm_desc.res_post; backward: act' from the recomputed pre-activation value)
    ("post_inst_relu", 3, 64, 24, 0, False, "relu", True, False),
    ("post_group_elu", 2, 32, 16, 16, True, "elu", True, False),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_groupnorm_backward_vs_autograd(case, dtype):
    name, N, C, H, G, affine, act, use_res, spade = case
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(N, C, H, H, generator=gen, requires_grad=True)
    gamma = (1 + 0.3 * torch.randn(C, generator=gen)).requires_grad_(affine)
    beta = (0.2 * torch.randn(C, generator=gen)).requires_grad_(affine)
    res = torch.randn(N, C, H, H, generator=gen, requires_grad=use_res)
    mg = (0.5 * torch.randn(N, C, H, H, generator=gen)).requires_grad_(spade)
    mb = (0.5 * torch.randn(N, C, H, H, generator=gen)).requires_grad_(spade)
    groups = C if G == 0 else G
    y = F.group_norm(x, groups, gamma if affine else None, beta if affine else None, eps=1e-5)
    if spade:
        y = y * (1 + mg) + mb
    post = name.startswith("post_")
    if use_res and not post:
        y = y + res
    y = {"relu": torch.relu, "elu": F.elu, "none": lambda v: v}[act](y)
    if use_res and post:
        y = y + res
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)

    actc = {"relu": _lib.ACT_RELU, "elu": _lib.ACT_ELU, "none": _lib.ACT_NONE}[act]
    xc = _cl(x.detach(), dtype); xc.t.requires_grad_(True)
    g_d = gamma.detach().to(DEV).requires_grad_(affine) if affine else None
    b_d = beta.detach().to(DEV).requires_grad_(affine) if affine else None
    rc = _cl(res.detach(), dtype) if use_res else None
    if use_res:
        rc.t.requires_grad_(True)
    mod = None
    if spade:
        mod = (_cl(mg.detach(), dtype), _cl(mb.detach(), dtype))
        mod[0].t.requires_grad_(True); mod[1].t.requires_grad_(True)
    out = FT.group_norm(xc, groups, dtype, g_d, b_d, act=actc, res=rc, mod=mod, res_post=post)
    tol = TOL[dtype]
    assert _rel(_nchw(out, dtype).detach(), y.detach()) <= tol
    out.t.backward(_cl(dy, dtype).t)
    assert _rel(_nchw(K.CL(xc.t.grad, N, (1, H, H), C), dtype), x.grad) <= tol * 3
    if affine:
        assert _rel(g_d.grad.cpu(), gamma.grad) <= tol * 3 and _rel(b_d.grad.cpu(), beta.grad) <= tol * 3
    if use_res:
        assert _rel(_nchw(K.CL(rc.t.grad, N, (1, H, H), C), dtype), res.grad) <= tol * 3
    if spade:
        assert _rel(_nchw(K.CL(mod[0].t.grad, N, (1, H, H), C), dtype), mg.grad) <= tol * 3
        assert _rel(_nchw(K.CL(mod[1].t.grad, N, (1, H, H), C), dtype), mb.grad) <= tol * 3


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_block_backward_vs_autograd(dtype):
    """_ConvFn: fused activation backward, bias column sums, weight gradient (direct and transposed conv), data gradient."""
    from ipoke_amd.first_stage import _Conv
    gen = torch.Generator().manual_seed(7)
    for transposed in (False, True):
        cin, cout, H = 16, 24, 8
        mod = _Conv(cin, cout, 3, 2, 1, transposed=transposed).to(DEV)
        w = mod.weight.detach().cpu().clone().requires_grad_(True)
        b = (0.1 * torch.randn(cout, generator=gen)).requires_grad_(True)
        with torch.no_grad():
            mod.bias.copy_(b.detach().to(DEV))
        x = torch.randn(2, cin, H, H, generator=gen, requires_grad=True)
        if transposed:
            y = F.elu(F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1))
        else:
            y = F.elu(F.conv2d(x, w, b, stride=2, padding=1))
        dy = torch.randn(y.shape, generator=gen)
        y.backward(dy)
        xc = _cl(x.detach(), dtype); xc.t.requires_grad_(True)
        out = FT.conv(mod, xc, dtype, act=_lib.ACT_ELU)
        tol = TOL[dtype]
        assert _rel(_nchw(out, dtype).detach(), y.detach()) <= tol
        out.t.backward(_cl(dy, dtype).t)
        assert _rel(mod.weight.grad.cpu(), w.grad) <= tol * 3, transposed
        assert _rel(mod.bias.grad.cpu(), b.grad) <= tol * 3
        assert _rel(_nchw(K.CL(xc.t.grad, 2, (1, H, H), cin), dtype), x.grad) <= tol * 3


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(64, 3, 3, 1, 24), (32, 5, 3, 3, 8), (128, 12, 5, 1, 16), (64, 8, 1, 1, 16)])
def test_narrow_output_conv_weight_gradient_swapped(case, dtype, monkeypatch):
    """A convolution onto a handful of channels (the decoder's 64 -> 3 at the full resolution): its weight gradient runs with the roles
    swapped (the mirrored convolution from dY to X, first_stage_train._ConvFn.backward) -- against torch autograd and against the
    direct form, for 2-D and 3-D kernels, 3 / 5 / 8 / 12 output channels, kernel 1 / 3 / 5."""
    from ipoke_amd.first_stage import _Conv
    cin, cout, k, kd, H = case
    gen = torch.Generator().manual_seed(11)
    D = 4 if kd > 1 else 1
    mod = _Conv(cin, cout, (kd, k, k) if kd > 1 else k, 1, (kd // 2, k // 2, k // 2) if kd > 1 else k // 2, dims=3 if kd > 1 else 2).to(DEV)
    w = mod.weight.detach().cpu().clone().requires_grad_(True)
    x = torch.randn(2, cin, *((D, H, H) if kd > 1 else (H, H)), generator=gen)
    fwd = F.conv3d if kd > 1 else F.conv2d
    y = fwd(x, w, mod.bias.detach().cpu(), stride=1, padding=(kd // 2, k // 2, k // 2) if kd > 1 else k // 2)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    grads = {}
    for swap in (True, False):
        monkeypatch.setattr(FT, "_WGRAD_SWAP", swap)
        mod.weight.grad = None
        xc, dyc = _cl5(x, dtype), _cl5(dy, dtype)
        xc.t.requires_grad_(True)
        out = FT.conv(mod, xc, dtype)
        out.t.backward(dyc.t)
        grads[swap] = mod.weight.grad.detach().cpu().clone()
        assert _rel(grads[swap], w.grad) <= TOL[dtype] * 3, (case, swap)
    assert _rel(grads[True], grads[False]) <= (1e-5 if dtype == "f32" else 2e-3), case      # same products, another summation order


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [((2, 2, 2), (4, 16, 16), 16, 24), ((2, 1, 1), (6, 8, 8), 24, 16), ((2, 2, 2), (5, 9, 10), 8, 12),
                                  ((1, 2, 2), (3, 8, 12), 16, 16)])
def test_strided_conv3d_data_gradient_by_parity_phases(case, dtype, monkeypatch):
    """Data gradient of a 3 x 3 x 3 / padding 1 Conv3d with stride 2 along some dimensions (motion_encoder.py:80-91): one stride-1
    convolution per output parity class scattered into dX (ipoke_conv_desc.c_sd) -- against torch autograd and against the single
    27-tap launch, even and odd extents, strides (2,2,2) / (2,1,1) / (1,2,2)."""
    from ipoke_amd.first_stage import _Conv
    st, dhw, cin, cout = case
    gen = torch.Generator().manual_seed(13)
    mod = _Conv(cin, cout, (3, 3, 3), st, (1, 1, 1), bias=False, dims=3).to(DEV)
    w = mod.weight.detach().cpu().clone().requires_grad_(True)
    x = torch.randn(2, cin, *dhw, generator=gen, requires_grad=True)
    y = F.conv3d(x, w, None, stride=st, padding=1)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    got = {}
    for phases in (True, False):
        monkeypatch.setattr(FT, "_DG_PHASES", phases)
        mod.weight.grad = None
        xc = _cl5(x.detach(), dtype); xc.t.requires_grad_(True)
        out = FT.conv(mod, xc, dtype)
        assert tuple(out.dhw) == tuple(y.shape[2:])
        out.t.backward(_cl5(dy, dtype).t)
        dx = K.CL(xc.t.grad, 2, dhw, cin)
        got[phases] = _nchw(dx, dtype)
        assert _rel(got[phases], x.grad) <= TOL[dtype] * 3, (case, phases)
        assert _rel(mod.weight.grad.cpu(), w.grad) <= TOL[dtype] * 3
    assert _rel(got[True], got[False]) <= (1e-5 if dtype == "f32" else 2e-3), case


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gru_cell_backward_vs_autograd(dtype):
    from ipoke_amd.first_stage import ConvGRUCell
    from oracle import vae_ref
    gen = torch.Generator().manual_seed(11)
    Z = 32
    cell = ConvGRUCell(Z, Z).to(DEV)
    ref = vae_ref.ConvGRUCell(Z, Z)
    ref.load_state_dict({k: v.cpu() for k, v in cell.state_dict().items()})
    x = torch.randn(2, Z, 8, 8, generator=gen, requires_grad=True)
    h = torch.randn(2, Z, 8, 8, generator=gen, requires_grad=True)
    hn = ref(x, h)
    dh = torch.randn(hn.shape, generator=gen)
    hn.backward(dh)
    xc, hc = _cl(x.detach(), dtype), _cl(h.detach(), dtype)
    xc.t.requires_grad_(True); hc.t.requires_grad_(True)
    out = FT.gru_cell(cell, xc, hc, dtype)
    tol = TOL[dtype]
    assert _rel(_nchw(out, dtype).detach(), hn.detach()) <= tol
    out.t.backward(_cl(dh, dtype).t)
    assert _rel(_nchw(K.CL(xc.t.grad, 2, (1, 8, 8), Z), dtype), x.grad) <= tol * 3
    assert _rel(_nchw(K.CL(hc.t.grad, 2, (1, 8, 8), Z), dtype), h.grad) <= tol * 3
    for (k, p), (k2, q) in zip(cell.named_parameters(), ref.named_parameters()):
        assert k == k2 and _rel(p.grad.cpu(), q.grad) <= tol * 4, k


def test_l1_tanh_loss_and_reparam_backward():
    gen = torch.Generator().manual_seed(3)
    N, H = 2, 16
    pre = torch.randn(N * H * H, 3, generator=gen, requires_grad=True)
    x = torch.rand(N, 3, H, H, generator=gen) * 2 - 1
    frame = torch.tanh(pre).view(N, H, H, 3).permute(0, 3, 1, 2)
    loss = (frame - x).abs().sum() * 0.125
    loss.backward()
    pd = pre.detach().to(DEV).requires_grad_(True)
    lt, fr = FT._L1TanhFn.apply(pd, x.to(DEV), 0.125)
    assert abs(lt.item() - loss.item()) <= 1e-4 * abs(loss.item())
    lt.backward()
    assert (pd.grad.cpu() - pre.grad).abs().max().item() <= 1e-6
    # reparameterisation: z = mu + eps * exp(logvar / 2) with gradients into [mu | logvar]
    Z, M = 8, 64
    mulv = torch.randn(M, 2 * Z, generator=gen, requires_grad=True)
    eps = torch.randn(M, Z, generator=gen)
    mu, lv = mulv[:, :Z], mulv[:, Z:]
    z = mu + eps * torch.exp(0.5 * lv)
    w1, w2, w3 = torch.randn(M, Z, generator=gen), torch.randn(M, Z, generator=gen), torch.randn(M, Z, generator=gen)
    ((z * w1).sum() + (mu * w2).sum() + (lv * w3).sum()).backward()
    md = mulv.detach().to(DEV).requires_grad_(True)
    zz, m2, l2 = FT._ReparamFn.apply(md, eps.to(DEV), Z, "f32")
    ((zz * w1.to(DEV)).sum() + (m2 * w2.to(DEV)).sum() + (l2 * w3.to(DEV)).sum()).backward()
    assert (zz.detach().cpu() - z.detach()).abs().max().item() <= 1e-5
    assert (md.grad.cpu() - mulv.grad).abs().max().item() <= 1e-5


@pytest.mark.parametrize("mode", ["slabs", "capped"])
def test_wgrad_launch_modes_match_single_launch(mode):
    """Deterministic split-M slabs (+ ipoke_reduce_rows) and workgroup-capped launches give the single-launch result."""
    gen = torch.Generator().manual_seed(5)
    N, H, cin, cout = 4, 32, 64, 192
    dt = "bf16"
    x = torch.randn(N * H * H, cin, generator=gen).to(DEV).to(torch.bfloat16)
    g = torch.randn(N * H * H, cout, generator=gen).to(DEV).to(torch.bfloat16)

    def run(splitm=1, slabs=False, cap=0):
        wd = _lib.WgradDesc()
        wd.NB = N; wd.Di, wd.Hi, wd.Wi = 1, H, H; wd.Do, wd.Ho, wd.Wo = 1, H, H
        wd.kd, wd.kh, wd.kw = 1, 3, 3; wd.sd = wd.sh = wd.sw = 1; wd.ph = wd.pw = 1
        wd.A = x.data_ptr(); wd.a_sn = H * H * cin; wd.a_sd = H * H * cin; wd.a_sh = H * cin; wd.a_sw = cin; wd.a_sc = 1
        wd.Kc_real = cin; wd.Kc = cin; wd.Kc_store = cin
        wd.dY = g.data_ptr(); wd.ldy = cout; wd.Nout = cout
        wd.w_sn = cin * 9; wd.w_sc = 9; wd.w_st = 1
        wd.max_workgroups = cap
        dW = torch.zeros(cout, cin, 3, 3, device=DEV)
        if slabs:
            buf = torch.empty(splitm, dW.numel(), device=DEV)
            wd.splitm = splitm; wd.split_stride = dW.numel(); wd.dW = buf.data_ptr()
            check(_lib.lib().ipoke_conv_wgrad(byref(wd), _lib.BF16, _lib.current_stream()))
            check(_lib.lib().ipoke_reduce_rows(ptr(buf), ptr(dW), splitm, dW.numel(), _lib.current_stream()))
        else:
            wd.dW = dW.data_ptr()
            check(_lib.lib().ipoke_conv_wgrad(byref(wd), _lib.BF16, _lib.current_stream()))
        torch.cuda.synchronize()
        return dW

    ref = run()
    xr = x.float().view(N, H, H, cin).permute(0, 3, 1, 2).cpu().requires_grad_(False)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(xr, w, padding=1).backward(g.float().view(N, H, H, cout).permute(0, 3, 1, 2).cpu())
    assert _rel(ref.cpu(), w.grad) <= 2e-3            # bf16 operands are exact here, fp32 accumulation order differs
    got = run(splitm=8, slabs=True) if mode == "slabs" else run(cap=3)
    assert _rel(got, ref) <= 1e-5
    if mode == "slabs":
        assert torch.equal(run(splitm=8, slabs=True), got)       # deterministic


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("B,T,L,Z", [(2, 5, 3, 32), (3, 4, 4, 64), (1, 1, 1, 16)])
def test_native_gru_unroll_vs_autograd(B, T, L, Z, dtype):
    """ipoke_gru_unroll_forward / _backward (csrc/gru.hip: T steps x L stacked cells, the concatenations written in place by the producer
    kernels, one weight-gradient GEMM per convolution over all steps) against the oracle's ConvGRU unrolled under torch autograd exactly
    as SpadeCondMotionModel.forward drives it (first_stage_motion_model.py:503-514: every cell starts from the motion code, cell 0 sees a
    constant input): the output sequence, d input, d initial state and every weight / bias gradient."""
    from ipoke_amd.first_stage import ConvGRU
    from oracle import vae_ref
    gen = torch.Generator().manual_seed(B * 100 + T * 10 + L)
    rnn = ConvGRU(Z, Z, 3, L, dtype=dtype).to(DEV)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.3 if p.dim() == 1 else 1.5 / (9 * 2 * Z) ** 0.5))
    ref = vae_ref.ConvGRU(Z, Z, L)
    ref.load_state_dict({k: v.cpu() for k, v in rnn.state_dict().items()})
    x = torch.randn(B, Z, 8, 8, generator=gen, requires_grad=True)
    h0 = torch.randn(B, Z, 8, 8, generator=gen, requires_grad=True)
    hidden = [h0] * L
    outs = []
    for _ in range(T):
        hidden = ref(x, hidden)
        outs.append(hidden[-1])
    seq = torch.stack(outs, 0)                                        # [T, B, Z, 8, 8]
    dseq = torch.randn(seq.shape, generator=gen)
    seq.backward(dseq)
    xc, hc = _cl(x.detach(), dtype), _cl(h0.detach(), dtype)
    xc.t.requires_grad_(True); hc.t.requires_grad_(True)
    assert FT.gru_native_ok(rnn, xc, hc, dtype)
    out = FT.gru_unroll(rnn, xc, hc, T, dtype)
    assert out.N == T * B and out.t.shape[0] == T * B * 64
    tol = TOL[dtype] * (1 if dtype == "f32" else 1.5)
    got = _nchw(out, dtype).view(T, B, Z, 8, 8)
    assert _rel(got.detach(), seq.detach()) <= tol
    dcl = _cl(dseq.view(T * B, Z, 8, 8), dtype)
    out.t.backward(dcl.t)
    assert _rel(_nchw(K.CL(xc.t.grad, B, (1, 8, 8), Z), dtype), x.grad) <= tol * 4
    assert _rel(_nchw(K.CL(hc.t.grad, B, (1, 8, 8), Z), dtype), h0.grad) <= tol * 4
    for (k, p), (k2, q) in zip(rnn.named_parameters(), ref.named_parameters()):
        e = _rel(p.grad.cpu(), q.grad)
        assert k == k2 and e <= tol * 6, (k, e)


@pytest.mark.parametrize("B,T,L,Z", [(2, 5, 3, 32), (3, 4, 4, 64), (5, 15, 4, 64), (20, 3, 2, 32)])
def test_fused_gru_forward_matches_the_launch_per_phase_form(B, T, L, Z):
    """gru_fused_fwd_kernel (one workgroup per sample runs the T x L recurrence) against the four-launches-per-cell form it replaces: the
    output sequence and, through the unchanged backward pass on the workspace each form filled, every gradient.  Both round to bf16 at
    the same places, so they differ by the order of the fp32 sums only."""
    from ipoke_amd.first_stage import ConvGRU
    gen = torch.Generator().manual_seed(B * 100 + T * 10 + L)
    rnn = ConvGRU(Z, Z, 3, L, dtype="bf16").to(DEV)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.3 if p.dim() == 1 else 1.5 / (9 * 2 * Z) ** 0.5))
    x = torch.randn(B, Z, 8, 8, generator=gen)
    h0 = torch.randn(B, Z, 8, 8, generator=gen)
    dseq = torch.randn(T * B, Z, 8, 8, generator=gen)
    res = {}
    try:
        for mode in (0, 1):
            check(_lib.lib().ipoke_gru_set_fused(mode))
            for p in rnn.parameters():
                p.grad = None
            xc, hc = _cl(x, "bf16"), _cl(h0, "bf16")
            xc.t.requires_grad_(True); hc.t.requires_grad_(True)
            out = FT.gru_unroll(rnn, xc, hc, T, "bf16")
            out.t.backward(_cl(dseq, "bf16").t)
            res[mode] = (out.t.detach().float().cpu(), xc.t.grad.float().cpu(), hc.t.grad.float().cpu(),
                         [p.grad.detach().cpu().clone() for p in rnn.parameters()])
    finally:
        check(_lib.lib().ipoke_gru_set_fused(-1))
    a, b = res[0], res[1]
    scale = a[0].abs().max().item()
    err = (a[0] - b[0]).abs().max().item() / scale
    print(f"fused GRU B={B} T={T} L={L} Z={Z}: output rel err {err:.2e}")
    assert err <= 2e-2                                       # a bf16 rounding flip early in the recurrence is carried forward
    ge = [(a[k] - b[k]).abs().max().item() / a[k].abs().max().item() for k in (1, 2)]
    we = max((ga - gb).abs().max().item() / (ga.abs().max().item() + 1e-12) for ga, gb in zip(a[3], b[3]))
    print(f"   gradients: d x0 {ge[0]:.2e}, d h0 {ge[1]:.2e}, weights / biases {we:.2e} (relative to each tensor's maximum)")
    assert max(ge) <= 4e-2 and we <= 4e-2
