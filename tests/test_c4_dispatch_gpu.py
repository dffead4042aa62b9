"""The first-stage kernels at the sizes ``bench.py --config c4`` reaches them, under the DEFAULT dispatch rule (no override, no
environment switch): the frames of a batch of clips are decoded as one batch, so the 128 x 128 layers see hundreds of
16 x 16 patches per launch -- conv3x3_c64's persistent workgroups (grid capped at 256) walk over SEVERAL patches each, forward, in the
data-gradient form (transposed = 1) and in the four scattered sub-pixel phases; conv3x3_halo16 runs its data-gradient form on >= 256
workgroups; the weight gradients split their rows over ~512 workgroups (deterministic slabs).  ``ipoke_last_conv_kernel`` confirms the
dispatch.  References: torch's fp32 convolutions and their autograd on the same bf16-rounded operands (util.py:52, 252 call sites).
Also the unit tests of the frame-batched spectral-norm helpers (ipoke_conv_desc.row_scale, ipoke_rowscale_bwd,
ipoke_spectral_bwd_frames, ipoke_sum_frames)."""
from ctypes import byref

import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, first_stage as FS, first_stage_train as FT, nn as K, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rows(x, dtype=torch.bfloat16):          # [N, C, H, W] -> channels-last rows
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).to(dtype).contiguous()


def _nchw(t, N, H, W, C):
    return t[:, :C].float().reshape(N, H, W, C).permute(0, 3, 1, 2)


def _last():
    return _lib.lib().ipoke_last_conv_kernel()


def test_c64_forward_and_data_gradient_many_patches():
    """64 -> 64 channels, 3 x 3, on 10 images of 128 x 128 = 640 patches (>= 512: the default rule takes conv3x3_c64; > 256: every
    workgroup walks two or three patches through its double-buffered image loop), forward with bias + ReLU and data gradient."""
    g = torch.Generator().manual_seed(11)
    N, H, W, cin, cout = 10, 128, 128, 64, 64
    x = torch.randn(N, cin, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 24).bfloat16().float()
    b = torch.randn(cout, generator=g) * 0.1
    pre = F.conv2d(x, w, b, padding=1)
    dy = torch.randn(N, cout, H, W, generator=g).bfloat16().float()
    pre.backward(dy)
    wop, kc = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16")
    y = K.conv(K.CL(_rows(x.detach()).to(DEV), N, (1, H, W), cin), wop, kc, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16", bias=b.to(DEV),
               act=_lib.ACT_RELU)
    assert _last() == _lib.KERNEL_C64
    ref = F.relu(pre.detach())
    assert (_nchw(y.t, N, H, W, cout).cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    wop_t, kc_t = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16", transposed_conv=True)
    dx = K.conv(K.CL(_rows(dy).to(DEV), N, (1, H, W), cout), wop_t, kc_t, cin, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16", transposed=True)
    assert _last() == _lib.KERNEL_C64
    assert (_nchw(dx.t, N, H, W, cin).cpu() - x.grad).abs().max().item() <= 2e-2 * x.grad.abs().max().item()


def test_c64_scatter_phases_many_patches():
    """The last up-convolution of the decoder (128 -> 64 channels, 64 x 64 -> 128 x 128) on 40 images: every sub-pixel phase is a stride-1
    convolution over 640 patches with scattered output rows -- conv3x3_c64 with one / two / two / four taps of two channel chunks."""
    torch.manual_seed(3)
    N, H, W, cin, cout = 40, 64, 64, 128, 64
    mod = FS._Conv(cin, cout, 3, 2, 1, transposed=True).to(DEV)
    with torch.no_grad():
        mod.bias.copy_(0.1 * torch.randn(cout))
    x = torch.randn(N, cin, H, W).bfloat16().float()
    xc = K.CL(_rows(x).to(DEV), N, (1, H, W), cin)
    got = mod.run(xc, "bf16", act=_lib.ACT_RELU)
    assert _last() == _lib.KERNEL_C64 and got.dhw == (1, 2 * H, 2 * W)
    w = mod.weight.detach().cpu().bfloat16().float()
    ref = F.relu(F.conv_transpose2d(x, w, mod.bias.detach().cpu(), stride=2, padding=1, output_padding=1))
    assert (_nchw(got.t, N, 2 * H, 2 * W, cout).cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_halo16_forward_and_data_gradient_full_chip():
    """128 -> 128 channels on 16 images of 64 x 64: 256 workgroups of 256 pixels x 128 channels -- conv3x3_halo16 by the default rule,
    forward and data gradient."""
    g = torch.Generator().manual_seed(4)
    N, H, W, cin, cout = 16, 64, 64, 128, 128
    x = torch.randn(N, cin, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 34).bfloat16().float()
    pre = F.conv2d(x, w, None, padding=1)
    dy = torch.randn(N, cout, H, W, generator=g).bfloat16().float()
    pre.backward(dy)
    wop, kc = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16")
    y = K.conv(K.CL(_rows(x.detach()).to(DEV), N, (1, H, W), cin), wop, kc, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16")
    assert _last() == _lib.KERNEL_HALO16
    assert (_nchw(y.t, N, H, W, cout).cpu() - pre.detach()).abs().max().item() <= 2e-2 * pre.abs().max().item()
    wop_t, kc_t = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16", transposed_conv=True)
    dx = K.conv(K.CL(_rows(dy).to(DEV), N, (1, H, W), cout), wop_t, kc_t, cin, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16", transposed=True)
    assert _last() == _lib.KERNEL_HALO16
    assert (_nchw(dx.t, N, H, W, cin).cpu() - x.grad).abs().max().item() <= 2e-2 * x.grad.abs().max().item()


@pytest.mark.parametrize("transposed", [False, True])
def test_conv_autograd_at_the_benchmarked_size(transposed):
    """One decoder convolution through first_stage_train.conv at the c4 size -- forward, bias gradient, the 510-workgroup split-M weight
    gradient (deterministic slabs + ipoke_reduce_rows) and the data gradient -- against torch autograd.  transposed: the stride-2
    up-convolution (sub-pixel phases forward, strided direct convolution backward)."""
    torch.manual_seed(7 + int(transposed))
    if transposed:
        N, H, W, cin, cout = 40, 64, 64, 128, 64
        mod = FS._Conv(cin, cout, 3, 2, 1, transposed=True).to(DEV)
    else:
        N, H, W, cin, cout = 10, 128, 128, 64, 64
        mod = FS._Conv(cin, cout, 3, 1, 1).to(DEV)
    with torch.no_grad():
        mod.bias.copy_(0.1 * torch.randn(cout))
        mod.weight.copy_(mod.weight.bfloat16().float())
    x = torch.randn(N, cin, H, W).bfloat16().float()
    xt = _rows(x).to(DEV).requires_grad_(True)
    y = FT.conv(mod, K.CL(xt, N, (1, H, W), cin), "bf16", act=_lib.ACT_RELU)
    Ho, Wo = y.dhw[1], y.dhw[2]
    dy = torch.randn(N, cout, Ho, Wo).bfloat16().float()
    y.t.backward(_rows(dy).to(DEV))
    xr = x.clone().requires_grad_(True)
    w = mod.weight.detach().cpu().clone().requires_grad_(True)
    b = mod.bias.detach().cpu().clone().requires_grad_(True)
    ref = F.relu(F.conv_transpose2d(xr, w, b, stride=2, padding=1, output_padding=1) if transposed else F.conv2d(xr, w, b, padding=1))
    # the device masks the gradient with ITS (bf16-rounded) output's sign: use the same mask on the reference side
    mask = (_nchw(y.t.detach(), N, Ho, Wo, cout).cpu() > 0).float()
    pre = ref  # relu output; d relu handled by passing dy * mask through the linear part
    lin = F.conv_transpose2d(xr, w, b, stride=2, padding=1, output_padding=1) if transposed else F.conv2d(xr, w, b, padding=1)
    lin.backward(dy * mask)
    assert (_nchw(y.t.detach(), N, Ho, Wo, cout).cpu() - ref.detach()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    for name, got, want in (("dx", _nchw(xt.grad, N, H, W, cin).cpu(), xr.grad), ("dw", mod.weight.grad.cpu(), w.grad),
                            ("db", mod.bias.grad.cpu(), b.grad)):
        err = (got - want).abs().max().item() / want.abs().max().item()
        print(f"conv autograd (transposed={transposed}) {name}: rel err {err:.3e}")
        assert err <= 2e-2, name


# ------------------------------------------------------------------------------------------------ frame-batched spectral norm
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", ["igemm", "halo", "phases", "c64"])
def test_row_scale_epilogue(case, dtype):
    """ipoke_conv_desc.row_scale: the images of group n // rs_images are scaled by table[group * stride] before bias and activation, in
    every epilogue (implicit GEMM, halo-staged, scattered sub-pixel phases, filter-resident) = conv(x, W / sigma_t) frame by frame."""
    if dtype == "f32" and case in ("halo", "c64"):
        pytest.skip("bf16-only kernels")
    torch.manual_seed(5)
    frames, clips = 3, 2
    N = frames * clips
    H, W, cin, cout, transposed = {"igemm": (8, 8, 32, 48, False), "halo": (32, 32, 64, 64, False), "phases": (16, 16, 64, 40, True),
                                   "c64": (32, 32, 64, 64, False)}[case]
    mod = FS._Conv(cin, cout, 3, 2 if transposed else 1, 1, transposed=transposed).to(DEV)
    with torch.no_grad():
        mod.bias.copy_(0.1 * torch.randn(cout))
    x = torch.randn(N, cin, H, W)
    xc = K.from_nchw(x.to(DEV), dtype)
    xr = K.to_nchw(xc, dtype).cpu()
    sig = torch.tensor([[2.0, 0.5], [0.8, 1.25], [1.6, 0.625]], device=DEV)            # {sigma_t, 1 / sigma_t}
    w = mod.weight.detach()
    rs = (sig.view(-1)[1:], clips, 2)
    wop, kc = FT._build_weight_operand(w if w.dim() == 5 else w.unsqueeze(2), dtype, transposed)
    bias = mod.bias.detach().float().contiguous()
    with _lib.dispatch_override("c64", 2 if case == "c64" else 0):
        if transposed:
            y = FT._conv_transpose_phases(xc, wop, kc, cout, dtype, bias, _lib.ACT_ELU, False, row_scale=rs)
        else:
            y = K.conv(xc, wop, kc, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1), dtype, bias=bias, act=_lib.ACT_ELU, row_scale=rs)
        if case == "c64":
            assert _last() == _lib.KERNEL_C64
        if case == "halo":
            assert _last() == _lib.KERNEL_HALO
    got = K.to_nchw(y, dtype).cpu()
    wc = w.cpu().bfloat16().float() if dtype == "bf16" else w.cpu()
    refs = []
    for f in range(frames):
        xs = xr[f * clips:(f + 1) * clips]
        lin = F.conv_transpose2d(xs, wc, None, stride=2, padding=1, output_padding=1) if transposed else F.conv2d(xs, wc, None, padding=1)
        refs.append(F.elu(lin * sig[f, 1].item() + mod.bias.detach().cpu().view(1, -1, 1, 1)))
    ref = torch.cat(refs, 0)
    tol = (2e-5 if dtype == "f32" else 2e-2) * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("act", [_lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_ELU, _lib.ACT_LRELU02])
def test_rowscale_bwd_kernel(act, dtype):
    """ipoke_rowscale_bwd: gs = dy * act'(y) * scale[group], dots[group] = sum dy * act'(y) * (pre(y) - bias), dbias = column sums, against
    float64 torch arithmetic; ragged channel count (C = 20 in a pitch of 24 / 32), several row blocks per group."""
    torch.manual_seed(act + 1)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    e16 = 8 if dtype == "bf16" else 4
    frames, rpg, C = 3, 1100, 21 if dtype == "f32" else 20            # ragged in both pitches: 21 of 24 floats, 20 of 24 bf16
    M, ld = frames * rpg, -(-C // e16) * e16
    bias = torch.randn(C) * 0.3
    pre = torch.randn(M, C) * 1.5
    yfun = {_lib.ACT_NONE: lambda t: t, _lib.ACT_RELU: F.relu, _lib.ACT_ELU: F.elu, _lib.ACT_LRELU02: lambda t: F.leaky_relu(t, 0.2)}[act]
    y = torch.zeros(M, ld)
    y[:, :C] = yfun(pre)
    y = y.to(td)
    dy = torch.zeros(M, ld)
    dy[:, :C] = torch.randn(M, C)
    dy = dy.to(td)
    scale = torch.tensor([0.5, 7.0, 1.25, 7.0, 0.625], device=DEV)            # stride 2: entries 0, 2, 4
    gs = torch.full((M, ld), 7.0, dtype=td, device=DEV)
    dots = torch.empty(frames, device=DEV)
    dbias = torch.empty(C, device=DEV)
    L = _lib.lib()
    ws = torch.empty(int(L.ipoke_rowscale_bwd_workspace_floats(M, C, rpg)), device=DEV)
    yd, dyd, bd = y.to(DEV), dy.to(DEV), bias.to(DEV)
    d = _lib.RowScaleBwdDesc()
    d.dy = dyd.data_ptr(); d.lddy = ld; d.y = yd.data_ptr(); d.ldy = ld; d.M = M; d.C = C; d.Cpad = ld; d.act = act
    d.bias = bd.data_ptr(); d.scale = scale.data_ptr(); d.scale_stride = 2; d.rows_per_group = rpg
    d.gs = gs.data_ptr(); d.ldgs = ld; d.dots = dots.data_ptr(); d.dbias = dbias.data_ptr(); d.workspace = ws.data_ptr()
    _lib.check(L.ipoke_rowscale_bwd(byref(d), _lib.DTYPES[dtype], _lib.current_stream()))
    y64, dy64 = y[:, :C].double(), dy[:, :C].double()
    dact = {_lib.ACT_NONE: torch.ones_like(y64), _lib.ACT_RELU: (y64 > 0).double(), _lib.ACT_ELU: torch.where(y64 > 0, torch.ones_like(y64), y64 + 1),
            _lib.ACT_LRELU02: torch.where(y64 > 0, torch.ones_like(y64), torch.full_like(y64, 0.2))}[act]
    pre64 = {_lib.ACT_NONE: y64, _lib.ACT_RELU: y64, _lib.ACT_ELU: torch.where(y64 > 0, y64, torch.log1p(y64.clamp(min=-0.99999994))),
             _lib.ACT_LRELU02: torch.where(y64 > 0, y64, 5 * y64)}[act]
    g64 = dy64 * dact
    sc = torch.tensor([0.5, 1.25, 0.625], dtype=torch.float64).repeat_interleave(rpg).view(M, 1)
    want_gs = g64 * sc
    want_dots = (g64 * (pre64 - bias.double())).view(frames, -1).sum(1)
    want_db = g64.sum(0)
    got_gs = gs.float().cpu()
    assert ld == C or float(got_gs[:, C:].abs().max()) == 0.0
    e_gs = (got_gs[:, :C].double() - want_gs).abs().max().item() / want_gs.abs().max().item()
    e_dot = ((dots.cpu().double() - want_dots).abs() / want_dots.abs().clamp(min=1.0)).max().item()
    e_db = ((dbias.cpu().double() - want_db).abs() / want_db.abs().clamp(min=1.0)).max().item()
    print(f"rowscale_bwd act={act}[{dtype}]: gs {e_gs:.2e} dots {e_dot:.2e} dbias {e_db:.2e}")
    assert e_gs <= (1e-6 if dtype == "f32" else 5e-3) and e_dot <= 2e-4 and e_db <= 2e-4


@pytest.mark.parametrize("transposed", [0, 1])
def test_spectral_bwd_frames_and_sum_frames(transposed):
    """ipoke_spectral_bwd_frames: grad -= sum_t dots[t] / sigma_t * u_t v_t^T in PyTorch weight layout (Conv and ConvTranspose storage);
    ipoke_sum_frames: the sum of per-frame maps in fp32."""
    torch.manual_seed(2)
    cout, cin, taps, frames = 24, 40, 9, 5
    shape = (cin, cout, 3, 3) if transposed else (cout, cin, 3, 3)
    w = torch.randn(*shape, device=DEV)
    grad = torch.randn(*shape, device=DEV)
    g0 = grad.clone()
    snaps = torch.randn(frames, cout + cin * taps, device=DEV)
    sig = torch.rand(frames, 2, device=DEV) + 0.5
    dots = torch.randn(frames, device=DEV)
    L = _lib.lib()
    _lib.check(L.ipoke_spectral_bwd_frames(_lib.ptr(w), cout, cin, taps, transposed, _lib.ptr(grad), _lib.ptr(snaps), snaps.stride(0), _lib.ptr(sig),
                                           sig.stride(0), _lib.ptr(dots), frames, _lib.current_stream()))
    upd = torch.zeros(cout, cin * taps, dtype=torch.float64)
    for f in range(frames):
        upd += (dots[f] * sig[f, 1]).double().cpu() * torch.outer(snaps[f, :cout].double().cpu(), snaps[f, cout:].double().cpu())
    upd = upd.view(cout, cin, 3, 3)
    if transposed:
        upd = upd.transpose(0, 1)
    assert (grad.double().cpu() - (g0.double().cpu() - upd)).abs().max().item() <= 1e-5
    for td, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        src = torch.randn(frames, 4096 + 64, device=DEV).to(td)
        dst = torch.empty(4096 + 64, dtype=td, device=DEV)
        _lib.check(L.ipoke_sum_frames(_lib.ptr(src), _lib.ptr(dst), frames, src.shape[1], _lib.DTYPES[name], _lib.current_stream()))
        want = src.float().sum(0)
        assert (dst.float() - want).abs().max().item() <= (1e-6 if name == "f32" else 2e-2) * want.abs().max().item()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [("inst", 6, 64, 40, 0, False, False), ("group_res", 6, 32, 16, 16, True, True), ("inst_wide", 4, 256, 8, 0, False, False)],
                         ids=lambda c: c[0])
def test_norm_backward_folds_the_row_scale_pass(case, dtype):
    """ipoke_norm_bwd_desc.rs_*: the norm's last backward pass writes gs = round(dx) / sigma_t, the frames' <dx, x - b> and the bias
    gradient itself -- against the two-launch sequence (ipoke_groupnorm_bwd, then ipoke_rowscale_bwd on its dx): gs BIT-identical (the
    same rounded dx, the same product), dots / dbias equal up to the order of the partial sums; the residual gradient is untouched.
    Samples ordered (frame, clip); several 512-position blocks per sample in the first case (S = 1600)."""
    from ipoke_amd._lib import NormBwdDesc, NormDesc
    name, N, C, H, G, affine, use_res = case
    frames = 3 if N % 3 == 0 else 2
    S = H * H
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(N + C)
    x = (torch.randn(N * S, C, generator=gen) * 1.3 + 0.2).to(td).to(DEV)
    dy = torch.randn(N * S, C, generator=gen).to(td).to(DEV)
    res = torch.randn(N * S, C, generator=gen).to(td).to(DEV) if use_res else None
    gamma = (1 + 0.3 * torch.randn(C, generator=gen)).to(DEV) if affine else None
    beta = (0.2 * torch.randn(C, generator=gen)).to(DEV) if affine else None
    bias = (0.3 * torch.randn(C, generator=gen)).to(DEV)
    groups = C if G == 0 else G
    sig = torch.tensor([[2.0, 0.5], [0.8, 1.25], [1.6, 0.625]], device=DEV)[:frames].contiguous()
    L = _lib.lib()
    s = _lib.current_stream()
    # forward (for y and the statistics)
    y = torch.empty_like(x)
    fd = NormDesc()
    fd.x = x.data_ptr(); fd.ldx = C; fd.y = y.data_ptr(); fd.ldy = C; fd.N, fd.S, fd.C, fd.G, fd.eps = N, S, C, groups, 1e-5
    if affine:
        fd.gamma = gamma.data_ptr(); fd.beta = beta.data_ptr()
    if use_res:
        fd.res = res.data_ptr(); fd.ld_res = C
    fd.act = _lib.ACT_RELU
    ws_f = torch.empty(int(L.ipoke_groupnorm_workspace_floats(N, S, groups)), device=DEV)
    fd.workspace = ws_f.data_ptr()
    _lib.check(L.ipoke_groupnorm(byref(fd), _lib.DTYPES[dtype], s))

    def run(fold):
        dx = torch.full_like(x, 3.0)
        dres = torch.empty_like(x) if use_res else None
        d = NormBwdDesc()
        d.x = x.data_ptr(); d.ldx = C; d.y = y.data_ptr(); d.ldy = C; d.dy = dy.data_ptr(); d.lddy = C; d.dx = dx.data_ptr(); d.lddx = C
        d.N, d.S, d.C, d.G, d.eps = N, S, C, groups, 1e-5
        d.act = _lib.ACT_RELU
        if use_res:
            d.dres = dres.data_ptr(); d.lddres = C
        dg = db = None
        if affine:
            dg = torch.empty(C, device=DEV); db = torch.empty(C, device=DEV)
            d.gamma = gamma.data_ptr(); d.beta = beta.data_ptr(); d.dgamma = dg.data_ptr(); d.dbeta = db.data_ptr()
        ws = torch.empty(int(L.ipoke_groupnorm_bwd_workspace_floats(N, S, C, groups)), device=DEV)
        d.workspace = ws.data_ptr()
        dots = torch.empty(frames, device=DEV); dbias = torch.empty(C, device=DEV)
        if fold:
            ws_rs = torch.empty(int(L.ipoke_groupnorm_bwd_rs_workspace_floats(N, S, C)), device=DEV)
            d.rs_scale = sig.view(-1)[1:].data_ptr(); d.rs_scale_stride = 2; d.rs_rows_per_group = (N // frames) * S
            d.rs_bias = bias.data_ptr(); d.rs_dots = dots.data_ptr(); d.rs_dbias = dbias.data_ptr(); d.rs_workspace = ws_rs.data_ptr()
            _lib.check(L.ipoke_groupnorm_bwd(byref(d), _lib.DTYPES[dtype], s))
            gs = dx
        else:
            _lib.check(L.ipoke_groupnorm_bwd(byref(d), _lib.DTYPES[dtype], s))
            gs = torch.empty_like(x)
            M, rpg = N * S, (N // frames) * S
            ws_r = torch.empty(int(L.ipoke_rowscale_bwd_workspace_floats(M, C, rpg)), device=DEV)
            r = _lib.RowScaleBwdDesc()
            r.dy = dx.data_ptr(); r.lddy = C; r.y = x.data_ptr(); r.ldy = C; r.M = M; r.C = C; r.Cpad = C; r.act = _lib.ACT_NONE
            r.bias = bias.data_ptr(); r.scale = sig.view(-1)[1:].data_ptr(); r.scale_stride = 2; r.rows_per_group = rpg
            r.gs = gs.data_ptr(); r.ldgs = C; r.dots = dots.data_ptr(); r.dbias = dbias.data_ptr(); r.workspace = ws_r.data_ptr()
            _lib.check(L.ipoke_rowscale_bwd(byref(r), _lib.DTYPES[dtype], s))
        torch.cuda.synchronize()
        return gs, dots, dbias, dres, dg, db

    gs0, dots0, db0, dres0, dg0, dbt0 = run(False)
    gs1, dots1, db1, dres1, dg1, dbt1 = run(True)
    assert torch.equal(gs0, gs1)
    if use_res:
        assert torch.equal(dres0, dres1)
    if affine:
        assert torch.equal(dg0, dg1) and torch.equal(dbt0, dbt1)
    yard_d = float((gs0.float().abs().sum() / frames) * x.float().abs().mean()) * 1e-5 + 1e-4       # the dots cancel: yardstick = sum |dx| * mean |x|
    e_dot = (dots0 - dots1).abs().max().item()
    e_db = (db0 - db1).abs().max().item()
    print(f"norm bwd + row scale {name}[{dtype}]: dots {dots0.tolist()} diff {e_dot:.2e} (yard {yard_d:.2e}), dbias diff {e_db:.2e}")
    assert e_dot <= yard_d and e_db <= 1e-3 * max(1.0, db0.abs().max().item())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [("c64", 5, 3, 64, 24), ("c256", 4, 2, 256, 8), ("c96_ragged", 3, 2, 96, 9)], ids=lambda c: c[0])
def test_norm_backward_sums_the_shared_modulation_gradients_over_the_frames(case, dtype):
    """ipoke_norm_bwd_desc.dmod_summed: samples ordered (frame, clip) share the clip's SPADE maps; the backward pass walks the frames and
    writes d(mod_gamma), d(mod_beta) once per clip.  dx is bit-identical to the per-sample form; the summed maps are compared with float64
    arithmetic on the same inputs -- and must be at least as close as the per-sample form + ipoke_sum_frames (which rounds every frame's
    term to the storage type before adding)."""
    from ipoke_amd._lib import NormBwdDesc, NormDesc
    name, frames, clips, C, H = case
    N, S, G = frames * clips, H * H, 16 if C % 16 == 0 else 8
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N * S, C, generator=gen) * 1.3 + 0.2).to(td).to(DEV)
    dy = torch.randn(N * S, C, generator=gen).to(td).to(DEV)
    mg = (0.5 * torch.randn(clips * S, C, generator=gen)).to(td).to(DEV)
    mb = (0.5 * torch.randn(clips * S, C, generator=gen)).to(td).to(DEV)
    L = _lib.lib()
    s = _lib.current_stream()
    y = torch.empty_like(x)
    fd = NormDesc()
    fd.x = x.data_ptr(); fd.ldx = C; fd.y = y.data_ptr(); fd.ldy = C; fd.N, fd.S, fd.C, fd.G, fd.eps = N, S, C, G, 1e-5
    fd.mod_gamma = mg.data_ptr(); fd.mod_beta = mb.data_ptr(); fd.ld_mod = C; fd.mod_samples = clips; fd.act = _lib.ACT_NONE
    ws_f = torch.empty(int(L.ipoke_groupnorm_workspace_floats(N, S, G)), device=DEV)
    fd.workspace = ws_f.data_ptr()
    _lib.check(L.ipoke_groupnorm(byref(fd), _lib.DTYPES[dtype], s))

    def run(summed):
        dx = torch.full_like(x, 3.0)
        rows = clips * S if summed else N * S
        dmg = torch.empty(rows, C, dtype=td, device=DEV); dmb = torch.empty(rows, C, dtype=td, device=DEV)
        d = NormBwdDesc()
        d.x = x.data_ptr(); d.ldx = C; d.y = y.data_ptr(); d.ldy = C; d.dy = dy.data_ptr(); d.lddy = C; d.dx = dx.data_ptr(); d.lddx = C
        d.N, d.S, d.C, d.G, d.eps = N, S, C, G, 1e-5
        d.act = _lib.ACT_NONE
        d.dmod_gamma = dmg.data_ptr(); d.dmod_beta = dmb.data_ptr(); d.ld_dmod = C
        d.mod_gamma = mg.data_ptr(); d.ld_mod = C; d.mod_samples = clips; d.dmod_summed = int(summed)
        ws = torch.empty(int(L.ipoke_groupnorm_bwd_workspace_floats(N, S, C, G)), device=DEV)
        d.workspace = ws.data_ptr()
        _lib.check(L.ipoke_groupnorm_bwd(byref(d), _lib.DTYPES[dtype], s))
        if not summed:
            outs = []
            for t_ in (dmg, dmb):
                red = torch.empty(clips * S, C, dtype=td, device=DEV)
                _lib.check(L.ipoke_sum_frames(t_.data_ptr(), red.data_ptr(), frames, red.numel(), _lib.DTYPES[dtype], s))
                outs.append(red)
            dmg, dmb = outs
        torch.cuda.synchronize()
        return dx, dmg, dmb

    dx0, dmg0, dmb0 = run(False)
    dx1, dmg1, dmb1 = run(True)
    assert torch.equal(dx0, dx1)
    x64 = x.double().view(N, S, G, C // G)
    mean = x64.mean(dim=(1, 3), keepdim=True)
    var = x64.var(dim=(1, 3), unbiased=False, keepdim=True)
    xh = ((x64 - mean) / torch.sqrt(var + 1e-5)).view(frames, clips * S, C)
    # the forward pass over the shared maps (the frame-walking apply kernel): y = xhat * (1 + mod_gamma) + mod_beta
    want_y = xh * (1 + mg.double().view(1, clips * S, C)) + mb.double().view(1, clips * S, C)
    e_y = ((y.double().view(frames, clips * S, C) - want_y).abs().max() / want_y.abs().max()).item()
    assert e_y <= (2e-5 if dtype == "f32" else 6e-3), e_y
    dw = dy.double().view(frames, clips * S, C)
    want_g, want_b = (dw * xh).sum(0), dw.sum(0)
    e = lambda got, want: ((got.double() - want).abs().max() / want.abs().max()).item()
    e1g, e1b, e0g, e0b = e(dmg1, want_g), e(dmb1, want_b), e(dmg0, want_g), e(dmb0, want_b)
    print(f"summed modulation gradients {name}[{dtype}]: d gamma {e1g:.2e} (per-sample form {e0g:.2e}), d beta {e1b:.2e} ({e0b:.2e})")
    tol = 2e-5 if dtype == "f32" else 4e-3
    assert e1g <= tol and e1b <= tol and e1g <= e0g * 1.05 + 1e-7 and e1b <= e0b * 1.05 + 1e-7
