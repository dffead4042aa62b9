"""The halo-staged 3x3 convolution kernels (csrc/gemm.hip: conv3x3_halo_kernel, and conv3x3_halo16_kernel for wide layers -- every
test runs twice: with the dispatcher's own choice and with the halo16 dispatch switch at 2, which sends every shape the wide kernel can run to it)
-- conv3x3_halo_kernel -- 2-D 3x3 / stride 1 / pad 1 convolutions with
>= 64 dense input channels on maps of 16 x 16 and larger) through ipoke_conv_forward, against torch's fp32 convolution of the
same bf16-rounded operands: forward with bias + activation, narrow fp32 outputs (the decoder's 3-channel head), several output
tiles, channel offsets into a wider output, and the data-gradient form (mirrored taps, activation-derivative mask).  The
dispatcher sends 64-channel inputs and maps of <= 256 pixels to it (csrc/gemm.hip: halo_applicable); the other cases of this file
run the implicit-GEMM kernel on the same checks."""
import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, nn as K, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=["dispatch", "halo16", "c64"])
def _kernel_choice(request):
    """halo16 / c64: the dispatch switches at 2 (ipoke_set_dispatch_override) send every shape conv3x3_halo16_kernel / conv3x3_c64_kernel
    (persistent workgroups, filter resident in LDS: 64 -> <= 64 channels) can run to it; the dispatcher's own rule needs maps far larger
    than most of this file's (tests/test_c4_dispatch_gpu.py covers the default rule at the benchmarked sizes)."""
    if request.param == "dispatch":
        yield
    else:
        with _lib.dispatch_override(request.param, 2):
            yield


def _rows(x):          # [N, C, H, W] -> channels-last bf16 rows
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).to(torch.bfloat16).contiguous()


def _nchw(t, N, H, W, C):
    return t[:, :C].float().reshape(N, H, W, C).permute(0, 3, 1, 2)


@pytest.mark.parametrize("N,H,W,cin,cout,act,out_f32", [(4, 32, 32, 64, 64, _lib.ACT_ELU, False), (2, 64, 48, 128, 128, _lib.ACT_NONE, False),
                                                      (3, 16, 32, 256, 72, _lib.ACT_RELU, False), (2, 128, 128, 64, 3, _lib.ACT_TANH, True),
                                                      (1, 64, 64, 192, 320, _lib.ACT_LRELU02, False), (16, 16, 16, 256, 256, _lib.ACT_ELU, False),
                                                      (4, 16, 16, 512, 200, _lib.ACT_NONE, False), (3, 48, 16, 64, 20, _lib.ACT_RELU, False),
                                                      (2, 16, 32, 64, 40, _lib.ACT_NONE, True)])
def test_halo_conv_forward(N, H, W, cin, cout, act, out_f32):
    g = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(N, cin, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, w, b, padding=1)
    ref = {_lib.ACT_NONE: lambda t: t, _lib.ACT_ELU: F.elu, _lib.ACT_RELU: F.relu, _lib.ACT_TANH: torch.tanh,
           _lib.ACT_LRELU02: lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    wop, kc = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16")
    xcl = K.CL(_rows(x).to(DEV), N, (1, H, W), cin)
    y = K.conv(xcl, wop, kc, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16", bias=b.to(DEV), act=act, out_f32=out_f32)
    got = _nchw(y.t, N, H, W, cout).cpu()
    err = (got - ref).abs().max().item()
    tol = 2e-3 if out_f32 else 2e-2 * max(1.0, ref.abs().max().item())      # fp32 accumulation; bf16 output rounding
    assert err <= tol, (err, tol)
    if not out_f32 and y.t.shape[1] > cout:                                 # padded columns are written as zeros
        assert float(y.t[:, cout:].float().abs().max()) == 0.0


def test_halo_conv_into_channel_range():
    """c_coff / ldc: the result lands in a channel range of a wider row (concatenation-free outputs)."""
    from ctypes import byref
    g = torch.Generator().manual_seed(5)
    N, H, W, cin, cout = 2, 32, 32, 64, 64
    x = torch.randn(N, cin, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 24).to(torch.bfloat16).float()
    wop, kc = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16")
    rows = _rows(x).to(DEV)
    out = torch.full((N * H * W, 160), 7.0, dtype=torch.bfloat16, device=DEV)
    d = ops.conv_desc(N, (1, H, W), (1, H, W), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = rows.data_ptr(); d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = H * W * cin, H * W * cin, W * cin, cin, 1
    d.Kc_real = d.Kc = kc; d.W = wop.data_ptr(); d.ldw = wop.shape[1]; d.Nout = cout
    d.C = out.data_ptr(); d.ldc = 160; d.c_coff = 32
    ops.conv_forward(d, "bf16")
    ref = F.conv2d(x, w, None, padding=1)
    assert (_nchw(out[:, 32:96], N, H, W, cout).cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert float((out[:, :32].float() - 7).abs().max()) == 0 and float((out[:, 96:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("N,H,W,cin,cout", [(2, 32, 64, 128, 64), (3, 16, 48, 64, 64), (2, 32, 32, 40, 64)])
def test_halo_conv_data_gradient(N, H, W, cin, cout):
    """transposed = 1 (the adjoint of a stride-1 3x3 convolution) with the activation-derivative mask of the saved output."""
    g = torch.Generator().manual_seed(9)                      # forward conv: cin -> cout; its data gradient: cout -> cin
    x = torch.randn(N, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 34).to(torch.bfloat16).float()
    dy = torch.randn(N, cout, H, W, generator=g).to(torch.bfloat16).float()
    F.conv2d(x, w, None, padding=1).backward(dy)
    wop, kc = K.weight_operand(w.unsqueeze(2).to(DEV), "bf16", transposed_conv=True)       # [cout, cin, k] read as ConvTranspose weight
    g_cl = K.CL(_rows(dy).to(DEV), N, (1, H, W), cout)
    dx = K.conv(g_cl, wop, kc, cin, (1, 3, 3), (1, 1, 1), (0, 1, 1), "bf16", transposed=True)
    got = _nchw(dx.t, N, H, W, cin).cpu()
    assert (got - x.grad).abs().max().item() <= 2e-2 * x.grad.abs().max().item()


@pytest.mark.parametrize("N,D,H,W,cin,cout", [(2, 4, 32, 32, 64, 64), (1, 8, 16, 32, 128, 192), (2, 3, 16, 16, 256, 64)])
def test_halo_conv3d_forward_and_data_gradient(N, D, H, W, cin, cout):
    """The 3 x 3 x 3 form (depth halo by "virtual chunks": one staged input image per (channel chunk, depth tap)): forward with
    bias + ReLU, and the data gradient (all three tap axes mirrored), against torch's fp32 conv3d on the same bf16-rounded operands."""
    g = torch.Generator().manual_seed(D * H + cin)
    x = torch.randn(N, cin, D, H, W, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (5.2 * cin ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(cout, generator=g) * 0.1
    pre = F.conv3d(x, w, b, padding=1)
    ref = F.relu(pre)
    dy = torch.randn(N, cout, D, H, W, generator=g).to(torch.bfloat16).float()
    pre.backward(dy)
    rows = x.detach().permute(0, 2, 3, 4, 1).reshape(-1, cin).to(torch.bfloat16).contiguous().to(DEV)
    wop, kc = K.weight_operand(w.to(DEV), "bf16")
    y = K.conv(K.CL(rows, N, (D, H, W), cin), wop, kc, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1), "bf16", bias=b.to(DEV), act=_lib.ACT_RELU)
    got = y.t[:, :cout].float().reshape(N, D, H, W, cout).permute(0, 4, 1, 2, 3).cpu()
    assert (got - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    wop_t, kc_t = K.weight_operand(w.to(DEV), "bf16", transposed_conv=True)
    g_rows = dy.permute(0, 2, 3, 4, 1).reshape(-1, cout).to(torch.bfloat16).contiguous().to(DEV)
    dx = K.conv(K.CL(g_rows, N, (D, H, W), cout), wop_t, kc_t, cin, (3, 3, 3), (1, 1, 1), (1, 1, 1), "bf16", transposed=True)
    got_dx = dx.t[:, :cin].float().reshape(N, D, H, W, cin).permute(0, 4, 1, 2, 3).cpu()
    assert (got_dx - x.grad).abs().max().item() <= 2e-2 * x.grad.abs().max().item()


@pytest.mark.parametrize("N,D,H,W,cin,cout", [(2, 6, 16, 32, 64, 128), (1, 5, 32, 16, 128, 96)])
def test_halo_conv3d_depth_stride_2(N, D, H, W, cin, cout):
    """layer1[0].conv1 of the 3-D encoder (motion_encoder.py:165: stride (2, 1, 1)): the depth halo reads slices 2 dz - 1 + kd."""
    g = torch.Generator().manual_seed(D + cin)
    x = torch.randn(N, cin, D, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (5.2 * cin ** 0.5)).to(torch.bfloat16).float()
    ref = F.conv3d(x, w, None, stride=(2, 1, 1), padding=1)
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, cin).to(torch.bfloat16).contiguous().to(DEV)
    wop, kc = K.weight_operand(w.to(DEV), "bf16")
    y = K.conv(K.CL(rows, N, (D, H, W), cin), wop, kc, cout, (3, 3, 3), (2, 1, 1), (1, 1, 1), "bf16")
    Do = ref.shape[2]
    assert y.dhw == (Do, H, W)
    got = y.t[:, :cout].float().reshape(N, Do, H, W, cout).permute(0, 4, 1, 2, 3).cpu()
    assert (got - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
