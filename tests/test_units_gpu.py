"""Per-kernel GPU parity against the reference's unit goldens (G1) and torch CPU ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, ops
from ipoke_amd._lib import check
from ipoke_amd.utils.detfill import deterministic_fill_
from oracle import flow_ref
from tests.conftest import t
from tests.helpers import mcf_shadows, shadow_nt, shadow_t, tdt

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOLS = {"f32": 3e-5, "bf16": 3e-2}


@pytest.mark.parametrize("C", [8, 32])
def test_actnorm_and_shuffle(golden, C):
    g = golden("g1_flow_units")
    x = t(g[f"actnorm_{C}_init_x"], DEV)
    ls, b = t(g[f"actnorm_{C}_post_log_scale"], DEV).flatten(), t(g[f"actnorm_{C}_post_bias"], DEV).flatten()
    B = x.shape[0]
    s = ops.to_state(x)
    y = ops.from_state(ops.actnorm_fwd(s, 0, C, ls, b), B, C)
    assert (y.cpu() - t(g[f"actnorm_{C}_y"])).abs().max() <= 2e-6
    xi = ops.from_state(ops.actnorm_inv(ops.to_state(y), 0, C, ls, b), B, C)
    assert (xi.cpu() - t(g[f"actnorm_{C}_inv"])).abs().max() <= 2e-6
    # data-dependent init reproduces the reference's post-init parameters from its pre-init draw
    ls0 = t(g[f"actnorm_{C}_pre_log_scale"], DEV).flatten().clone()
    b0 = torch.zeros(C, device=DEV)
    ops.actnorm_init_(s, 0, C, ls0, b0)
    assert (ls0.cpu() - ls.cpu()).abs().max() <= 2e-6 and (b0.cpu() - b.cpu()).abs().max() <= 2e-6
    # permutation: bit exact
    xs = t(g[f"x_{C}"], DEV)
    fi, bi = t(g[f"shuffle_{C}_fwd_idx"], DEV), t(g[f"shuffle_{C}_bwd_idx"], DEV)
    ys = ops.from_state(ops.actnorm_fwd(ops.to_state(xs), 0, C, None, None, fi), 2, C)
    assert torch.equal(ys.cpu(), t(g[f"shuffle_{C}_y"]))
    xr = ops.from_state(ops.actnorm_inv(ops.to_state(ys), 0, C, None, None, bi), 2, C)
    assert torch.equal(xr.cpu(), xs.cpu())


@pytest.mark.parametrize("C", [8, 32])
def test_affine(golden, C):
    g = golden("g1_flow_units")
    x, raw = t(g[f"x_{C}"], DEV), t(g[f"affine_{C}_raw"], DEV)
    s = ops.to_state(x)
    raw_s = ops.to_state(raw)                       # [M][2C]: [mu | s]
    y, ld, scale = ops.affine_fwd(s, raw_s, None, C, 0, 1, 2)
    assert (ops.from_state(y, 2, C).cpu() - t(g[f"affine_{C}_y"])).abs().max() <= 2e-6
    assert (ld.cpu() - t(g[f"affine_{C}_logdet"])).abs().max() <= 1e-4
    xi = ops.affine_inv(y, raw_s, None, C, 0, 1, 2)
    assert (ops.from_state(xi, 2, C).cpu() - t(g[f"affine_{C}_inv"])).abs().max() <= 2e-6


def _run_conv(x, w, dtype, stride, pad, bias=None, act=_lib.ACT_NONE, transposed=False, out_pad=0, a_f32=False, splitk=1):
    """x [N,Cin,D,H,W] fp32 (cuda), w torch-layout weight; returns [N,Cout,Do,Ho,Wo] fp32."""
    N, Cin, Di, Hi, Wi = x.shape
    if transposed:
        Cout = w.shape[1]
        k = tuple(w.shape[2:])
        Do, Ho, Wo = [(i - 1) * s - 2 * p + kk + out_pad * (1 if kk > 1 else 0) for i, s, p, kk in zip((Di, Hi, Wi), stride, pad, k)]
        wt = w.transpose(0, 1)                     # -> [Cout][Cin][k]
    else:
        Cout = w.shape[0]
        k = tuple(w.shape[2:])
        Do, Ho, Wo = [(i + 2 * p - kk) // s + 1 for i, s, p, kk in zip((Di, Hi, Wi), stride, pad, k)]
        wt = w
    e16 = 8 if dtype == "bf16" else 4
    kc = -(-Cin // e16) * e16
    d = ops.conv_desc(N, (Di, Hi, Wi), (Do, Ho, Wo), k, stride, pad, transposed)
    if a_f32:
        xa = x.permute(0, 2, 3, 4, 1).contiguous()            # channels-last fp32
        d.a_f32 = 1
    else:
        xa = torch.zeros(N, Di, Hi, Wi, kc, device=DEV, dtype=tdt(dtype))
        xa[..., :Cin] = x.permute(0, 2, 3, 4, 1).to(tdt(dtype))
    ldx = xa.shape[-1]
    d.A = xa.data_ptr(); d.a_sn = Di * Hi * Wi * ldx; d.a_sd = Hi * Wi * ldx; d.a_sh = Wi * ldx; d.a_sw = ldx; d.a_sc = 1
    d.Kc_real = Cin if a_f32 else kc; d.Kc = kc
    ws = shadow_nt(wt.contiguous(), kc, dtype=dtype)
    d.W = ws.data_ptr(); d.ldw = ws.shape[1]; d.Nout = Cout
    M = N * Do * Ho * Wo
    if splitk > 1:
        ldc = -(-Cout // 4) * 4
        out = torch.zeros(splitk, M, ldc, device=DEV)
        d.C = out.data_ptr(); d.c_f32 = 1; d.ldc = ldc; d.splitk = splitk
        ops.conv_forward(d, dtype)
        res = out.sum(0)[:, :Cout]
        if bias is not None:
            res = res + bias
    else:
        out = torch.zeros(M, Cout, device=DEV)
        d.C = out.data_ptr(); d.c_f32 = 1; d.ldc = Cout
        d.bias = 0 if bias is None else bias.data_ptr(); d.act = act
        ops.conv_forward(d, dtype)
        res = out
    torch.cuda.synchronize()
    return res.view(N, Do, Ho, Wo, Cout).permute(0, 4, 1, 2, 3).contiguous()


CONV_CASES = [
    # name, N, Cin, (D,H,W), Cout, k, stride, pad, transposed
    ("nice_conv1", 3, 8, (1, 8, 8), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("conv1x1", 3, 64, (1, 8, 8), 96, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),
    ("nice_conv3", 5, 64, (1, 8, 8), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("conv3d_s2", 2, 16, (4, 16, 16), 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), False),
    ("conv3d_first", 1, 3, (4, 16, 16), 16, (3, 7, 7), (2, 2, 2), (1, 3, 3), False),
    ("conv3d_t211", 2, 16, (4, 8, 8), 32, (3, 3, 3), (2, 1, 1), (1, 1, 1), False),
    ("convT2d", 2, 32, (1, 8, 8), 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), True),
    ("big_m", 20, 64, (1, 8, 8), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),
    # 4x4 stride-1 padding-1 convolutions of the PatchGAN (patchgan.py:402-418): 16 -> 15 -> 14, not powers of two
    ("patch_s1", 3, 16, (1, 16, 16), 24, (1, 4, 4), (1, 1, 1), (0, 1, 1), False),
    ("patch_out", 2, 32, (1, 15, 15), 1, (1, 4, 4), (1, 1, 1), (0, 1, 1), False),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward_vs_torch(case, dtype):
    name, N, Cin, dhw, Cout, k, s, p, tr = case
    gen = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(N, Cin, *dhw, generator=gen)
    bias = torch.randn(Cout, generator=gen)
    if tr:
        w = torch.randn(Cin, Cout, *k, generator=gen) / (Cin * 9) ** 0.5
        ref = F.conv_transpose3d(x, w, bias, stride=s, padding=p, output_padding=(0, p[1], p[2]))
        got = _run_conv(x.to(DEV), w.to(DEV), dtype, s, p, bias.to(DEV), transposed=True, out_pad=1)
    else:
        w = torch.randn(Cout, Cin, *k, generator=gen) / (Cin * k[0] * k[1] * k[2]) ** 0.5
        ref = F.conv3d(x, w, bias, stride=s, padding=p)
        got = _run_conv(x.to(DEV), w.to(DEV), dtype, s, p, bias.to(DEV))
    err = (got.cpu() - ref).abs().max().item()
    print(f"{name}[{dtype}] max err {err:.3e} (ref max {ref.abs().max():.2f})")
    assert got.shape == ref.shape and err <= TOLS[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_forward_state_input_and_splitk(dtype):
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(4, 12, 1, 8, 8, generator=gen)
    w = torch.randn(40, 12, 1, 3, 3, generator=gen) / 10
    ref = F.conv3d(x, w, None, padding=(0, 1, 1))
    got = _run_conv(x.to(DEV), w.to(DEV), dtype, (1, 1, 1), (0, 1, 1), a_f32=True)
    assert (got.cpu() - ref).abs().max() <= TOLS[dtype] * 3
    x2 = torch.randn(4, 256, 1, 8, 8, generator=gen)
    w2 = torch.randn(24, 256, 1, 3, 3, generator=gen) / 48
    ref2 = F.conv3d(x2, w2, None, padding=(0, 1, 1))
    got2 = _run_conv(x2.to(DEV), w2.to(DEV), dtype, (1, 1, 1), (0, 1, 1), splitk=5)
    assert (got2.cpu() - ref2).abs().max() <= TOLS[dtype] * 3


@pytest.mark.parametrize("B,Cout", [(5, 60), (4, 32), (20, 64)])
def test_conv3x3_skinny_stationary_input(B, Cout):
    """The chunk-major 3x3 kernel for wide dense inputs and <= 64 outputs (conv3 of NICEConvBlock, macow_utils.py:281, and
    the data gradient of its conv1, :270): split-K partial slabs with the library's own split count (odd B = a half-empty
    last tile), and the transposed form accumulated atomically into a strided fp32 state."""
    Cin = 512
    gen = torch.Generator().manual_seed(B * 100 + Cout)
    x = torch.randn(B, Cin, 1, 8, 8, generator=gen)
    w = torch.randn(Cout, Cin, 1, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    sk = _lib.lib().ipoke_conv3x3_skinny_splitk(B * 64, Cin, _lib.BF16)
    assert 1 <= sk <= 32
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv3d(xb, wb, None, padding=(0, 1, 1))
    got = _run_conv(x.to(DEV), w.to(DEV), "bf16", (1, 1, 1), (0, 1, 1), splitk=sk)
    err = (got.cpu() - ref).abs().max().item()
    print(f"skinny 3x3 B={B} Cout={Cout} splitk={sk}: max err {err:.3e}")
    assert err <= 2e-3 * max(1.0, ref.abs().max().item())          # same bf16 operands, fp32 accumulation in another order
    # data-gradient form: transposed taps, accumulated into every second column of a 64-wide fp32 state
    wt = torch.randn(Cin, Cout, 1, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    reft = F.conv_transpose3d(xb, wt.bfloat16().float(), None, padding=(0, 1, 1))
    M = B * 64
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1), True)
    xa = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV).bfloat16()
    d.A = xa.data_ptr(); d.a_sn = 64 * Cin; d.a_sd = 0; d.a_sh = 8 * Cin; d.a_sw = Cin; d.a_sc = 1; d.Kc_real = Cin; d.Kc = Cin
    ws = shadow_nt(wt.transpose(0, 1).contiguous().to(DEV), Cin, dtype="bf16")
    d.W = ws.data_ptr(); d.ldw = ws.shape[1]; d.Nout = min(Cout, 32)
    state = torch.randn(M, 64, generator=gen).to(DEV)
    before = state.clone()
    d.C = state.data_ptr(); d.c_f32 = 1; d.c_accumulate = 1; d.ldc = 64; d.c_coff = 1; d.c_cstride = 2; d.splitk = sk
    ops.conv_forward(d, "bf16")
    torch.cuda.synchronize()
    add = (state - before).cpu()
    want = reft[:, :d.Nout, 0].permute(0, 2, 3, 1).reshape(M, d.Nout)
    assert (add[:, 1:2 * d.Nout:2] - want).abs().max() <= 2e-3 * max(1.0, want.abs().max().item())
    assert add[:, 0::2].abs().max() == 0 and (2 * d.Nout >= 64 or add[:, 2 * d.Nout + 1::2].abs().max() == 0)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_conv_weight_operand_as_a_column_slice_is_not_over_read(dtype):
    """Round 6 bug: the implicit-GEMM kernels masked the K columns of the weight operand by its row pitch (ldw) instead of the reduction
    length (Ktot).  With W given as a COLUMN SLICE of a wider operand -- the parity-class data gradients of a strided Conv3d,
    first_stage_train._dgrad_phases (motion_encoder.py:80-91): taps [first, first + ntap) of one permuted [cin][27 * kc] operand -- the
    last K-block ran past the slice, for the last row past the end of the allocation; those columns meet zero A chunks, but 0 x NaN = NaN
    whenever the memory behind held NaNs (test_strided_conv3d_data_gradient_by_parity_phases failed once in eight runs).  Here the operand
    is the head of a NaN-filled slab, so any over-read poisons the last output channel deterministically."""
    gen = torch.Generator().manual_seed(3)
    N, Cin, Cout, D, H, W = 2, 16, 24, 3, 8, 8                 # Cout = the "cin" of the data gradient: rows of the operand
    e16 = 8 if dtype == "bf16" else 4
    kc = Cin
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    wfull = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (Cin * 18) ** 0.5
    first, ntap = 9, 18                                          # depth taps 1 .. 2 of the 27: a (2, 3, 3) kernel
    wsl = wfull[:, :, 1:3]
    td = tdt(dtype)
    xr, wr = (x.to(td).float(), wsl.to(td).float())
    # depth: kernel 2, padding 0, output extent D (the last output reads one slice past the input: zero) -- as the parity class of odd positions
    xp = F.pad(xr, (0, 0, 0, 0, 0, 1))
    ref = F.conv3d(xp, wr, None, padding=(0, 1, 1))
    rows = N * D * H * W
    xa = x.permute(0, 2, 3, 4, 1).reshape(rows, Cin).to(DEV).to(td).contiguous()
    ld_full = 27 * kc
    slab = torch.full((Cout * ld_full + 8192,), float("nan"), dtype=td, device=DEV)
    op = slab[:Cout * ld_full].view(Cout, 27, kc)
    op.copy_(wfull.permute(0, 2, 3, 4, 1).reshape(Cout, 27, Cin).to(DEV).to(td))      # [n][tap][c], taps in (d, h, w) order
    wview = slab[first * kc:]                                    # the slice starts at column first * kc of row 0
    out = torch.full((rows, Cout), float("nan"), device=DEV)
    d = ops.conv_desc(N, (D, H, W), (D, H, W), (2, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = xa.data_ptr(); d.a_sn = D * H * W * kc; d.a_sd = H * W * kc; d.a_sh = W * kc; d.a_sw = kc; d.a_sc = 1; d.Kc_real = kc; d.Kc = kc
    d.W = wview.data_ptr(); d.ldw = ld_full; d.Nout = Cout
    d.C = out.data_ptr(); d.c_f32 = 1; d.ldc = Cout
    ops.conv_forward(d, dtype)
    torch.cuda.synchronize()
    got = out.view(N, D, H, W, Cout).permute(0, 4, 1, 2, 3).cpu()
    assert torch.isfinite(got).all(), f"{int((~torch.isfinite(got)).sum())} non-finite outputs: the operand was read past its K range"
    assert (got - ref).abs().max().item() <= TOLS[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N,H,W,Cin,Cout,transposed", [(6, 128, 128, 3, 64, True), (5, 128, 128, 8, 48, False), (9, 64, 128, 5, 16, True),
                                                       (17, 64, 64, 3, 64, True)])
def test_conv3x3_one_chunk_input(N, H, W, Cin, Cout, transposed):
    """conv3x3_k8 (round 6): 3x3 / stride 1 / 'same' with ONE 16-byte chunk of input channels and <= 64 outputs, nothing staged -- the data
    gradient of the decoder's last convolution (util.py Conv2dBlock out_conv: 64 -> 3 channels, its gradient 3 (padded to 8) -> 64) --
    against torch, direct and transposed (channels beyond the real ones are zero padding, as the interface demands of dtype activations)."""
    L = _lib.lib()
    gen = torch.Generator().manual_seed(N + H + Cin + Cout)
    M = N * H * W
    x = torch.randn(N, Cin, H, W, generator=gen)
    if transposed:
        w = torch.randn(Cin, Cout, 3, 3, generator=gen) / (Cin * 9) ** 0.5
        ref = F.conv_transpose2d(x.bfloat16().float(), w.bfloat16().float(), None, padding=1)
        wop = shadow_nt(w.transpose(0, 1).contiguous().unsqueeze(2).to(DEV), 8, dtype="bf16")
    else:
        w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
        ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), None, padding=1)
        wop = shadow_nt(w.unsqueeze(2).to(DEV), 8, dtype="bf16")
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    xa = torch.zeros(M, 8, dtype=torch.bfloat16, device=DEV)
    xa[:, :Cin] = x.permute(0, 2, 3, 1).reshape(M, Cin).to(DEV).bfloat16()
    d = ops.conv_desc(N, (1, H, W), (1, H, W), (1, 3, 3), (1, 1, 1), (0, 1, 1), transposed)
    d.A = xa.data_ptr(); d.a_sn = H * W * 8; d.a_sd = 0; d.a_sh = W * 8; d.a_sw = 8; d.a_sc = 1; d.Kc_real = 8; d.Kc = 8
    out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    d.W = wop.data_ptr(); d.ldw = wop.shape[1]; d.Nout = Cout
    d.C = out.data_ptr(); d.c_f32 = 0; d.ldc = Cout
    ops.conv_forward(d, "bf16")
    torch.cuda.synchronize()
    assert L.ipoke_last_conv_kernel() == _lib.KERNEL_K8
    err = (out.float().cpu() - ref).abs().max().item()
    print(f"one-chunk 3x3 N={N} {H}x{W} {Cin}->{Cout} transposed={transposed}: max err {err:.3e} (ref max {ref.abs().max():.2f})")
    assert torch.isfinite(out.float()).all() and err <= 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,Cin,Cout", [(20, 32, 2048), (3, 8, 256), (5, 24, 384), (4, 64, 512), (7, 40, 2048), (1, 16, 256)])
def test_conv3x3_narrow_input_stationary(B, Cin, Cout):
    """conv3x3_k64 (round 6): 3x3 on the 8x8 latent with a narrow dense input (<= 64 channels) and a wide output -- conv1 of
    NICEConvBlock (macow_utils.py:270: conditioning channels -> hidden, ELU) and, transposed with the ELU' mask of the saved activation,
    the data gradient of its conv3 (:281).  One staged input image, nine taps as shifted reads; dtype outputs as the engine uses them;
    odd batches (a half-empty last tile), channel counts that are no multiple of 32, outputs that do not fill the last column tile."""
    L = _lib.lib()
    gen = torch.Generator().manual_seed(11 * B + Cin + Cout)
    M = B * 64
    kc = -(-Cin // 8) * 8
    x = torch.randn(B, Cin, 1, 8, 8, generator=gen)
    w = torch.randn(Cout, Cin, 1, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=gen) * 0.1
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    # ---- forward: h = ELU(conv(x, W) + b), bf16 output
    ref = F.elu(F.conv3d(xb, wb, bias, padding=(0, 1, 1)))[:, :, 0].permute(0, 2, 3, 1).reshape(M, Cout)
    xa = torch.zeros(M, kc, dtype=torch.bfloat16, device=DEV)
    xa[:, :Cin] = x[:, :, 0].permute(0, 2, 3, 1).reshape(M, Cin).to(DEV).bfloat16()
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = xa.data_ptr(); d.a_sn = 64 * kc; d.a_sd = 0; d.a_sh = 8 * kc; d.a_sw = kc; d.a_sc = 1; d.Kc_real = kc; d.Kc = kc
    ws = shadow_nt(w.to(DEV), kc, dtype="bf16")
    bd = bias.to(DEV)
    out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    d.W = ws.data_ptr(); d.ldw = ws.shape[1]; d.Nout = Cout; d.bias = bd.data_ptr(); d.act = _lib.ACT_ELU
    d.C = out.data_ptr(); d.c_f32 = 0; d.ldc = Cout
    ops.conv_forward(d, "bf16")
    torch.cuda.synchronize()
    assert L.ipoke_last_conv_kernel() == 2, "expected the stationary-input family (IPOKE_KERNEL_S8)"
    err = (out.float().cpu() - ref).abs().max().item()
    print(f"narrow-input 3x3 B={B} {Cin}->{Cout}: forward max err {err:.3e} (ref max {ref.abs().max():.2f})")
    assert torch.isfinite(out.float()).all() and err <= 2e-2 * max(1.0, ref.abs().max().item())
    # ---- transposed + ELU' mask: dX = conv_transpose(dY, Wt) * ELU'(saved h), the data gradient of a 3x3 convolution Cout <- Cin ... here
    # the roles: gradient rows [M][kc] (narrow), result [M][Cout] (wide)
    dy = torch.randn(B, Cin, 1, 8, 8, generator=gen)
    wt = torch.randn(Cin, Cout, 1, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    hsaved = (torch.randn(M, Cout, generator=gen)).bfloat16()                      # saved ELU outputs (mask where <= 0: y + 1)
    dref = F.conv_transpose3d(dy.bfloat16().float(), wt.bfloat16().float(), None, padding=(0, 1, 1))[:, :, 0].permute(0, 2, 3, 1).reshape(M, Cout)
    hf = hsaved.float()
    dref = dref * torch.where(hf > 0, torch.ones_like(hf), hf + 1.0)
    ga = torch.zeros(M, kc, dtype=torch.bfloat16, device=DEV)
    ga[:, :Cin] = dy[:, :, 0].permute(0, 2, 3, 1).reshape(M, Cin).to(DEV).bfloat16()
    d2 = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1), True)
    d2.A = ga.data_ptr(); d2.a_sn = 64 * kc; d2.a_sd = 0; d2.a_sh = 8 * kc; d2.a_sw = kc; d2.a_sc = 1; d2.Kc_real = kc; d2.Kc = kc
    wts = shadow_nt(wt.transpose(0, 1).contiguous().to(DEV), kc, dtype="bf16")
    hd = hsaved.to(DEV)
    out2 = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    d2.W = wts.data_ptr(); d2.ldw = wts.shape[1]; d2.Nout = Cout; d2.dact = hd.data_ptr(); d2.ld_dact = Cout; d2.dact_act = _lib.ACT_ELU
    d2.C = out2.data_ptr(); d2.c_f32 = 0; d2.ldc = Cout
    ops.conv_forward(d2, "bf16")
    torch.cuda.synchronize()
    assert L.ipoke_last_conv_kernel() == 2
    err2 = (out2.float().cpu() - dref).abs().max().item()
    print(f"narrow-input 3x3 B={B} {Cin}->{Cout}: masked data gradient max err {err2:.3e} (ref max {dref.abs().max():.2f})")
    assert torch.isfinite(out2.float()).all() and err2 <= 2e-2 * max(1.0, dref.abs().max().item())


@pytest.mark.parametrize("dtype,B,Nout", [("bf16", 20, 32), ("bf16", 5, 30), ("bf16", 20, 64), ("bf16", 40, 16), ("f32", 4, 32), ("f32", 3, 60)])
def test_split_k_accumulation_through_the_scratch_is_deterministic(dtype, B, Nout):
    """ipoke_conv_desc.acc_scratch: the K slices of an accumulating launch (the conv1 data gradient of NICEConvBlock,
    macow_utils.py:270, added into the gradient of the conditioning channels) meet in a scratch and are summed in a fixed order by
    the workgroup that arrives last -- same value as the atomic form up to fp32 rounding, and BIT-identical from run to run while a
    copy stream perturbs the arrival order (the reference trains with deterministic=True, experiments/experiment.py:33, 86).
    bf16: conv3x3_s8n32 (<= 32 columns) / conv3x3_s8 (64); f32: the 64 x 64 skinny tile of the implicit GEMM; ragged last tiles."""
    L = _lib.lib()
    Cin = 512
    gen = torch.Generator().manual_seed(7 * B + Nout)
    x = torch.randn(B, Cin, 1, 8, 8, generator=gen)
    wt = torch.randn(Cin, Nout, 1, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    M = B * 64
    if dtype == "bf16":
        sk = L.ipoke_conv3x3_skinny_splitk(M, Cin, _lib.BF16)
        xr, wr = x.bfloat16().float(), wt.bfloat16().float()
    else:
        sk, xr, wr = 6, x, wt
    assert sk > 1
    want = F.conv_transpose3d(xr, wr, None, padding=(0, 1, 1))[:, :, 0].permute(0, 2, 3, 1).reshape(M, Nout)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1), True)
    xa = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV).to(tdt(dtype))
    d.A = xa.data_ptr(); d.a_sn = 64 * Cin; d.a_sd = 0; d.a_sh = 8 * Cin; d.a_sw = Cin; d.a_sc = 1; d.Kc_real = Cin; d.Kc = Cin
    ws = shadow_nt(wt.transpose(0, 1).contiguous().to(DEV), Cin, dtype=dtype)
    d.W = ws.data_ptr(); d.ldw = ws.shape[1]; d.Nout = Nout
    nbytes = L.ipoke_conv_acc_scratch_bytes(M, Nout, sk)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(L.ipoke_conv_acc_scratch_init(scratch.data_ptr(), _lib.current_stream()))
    base = torch.randn(M, 136, generator=gen).to(DEV)
    noise_src = torch.randn(64 << 20, device=DEV)
    side = torch.cuda.Stream()

    def run(with_scratch, disturb):
        state = base.clone()
        d.C = state.data_ptr(); d.c_f32 = 1; d.c_accumulate = 1; d.ldc = 136; d.c_coff = 3; d.c_cstride = 2; d.splitk = sk
        d.acc_scratch = scratch.data_ptr() if with_scratch else None
        d.acc_scratch_bytes = nbytes if with_scratch else 0
        if disturb:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(disturb):
                    noise_src.clone()
        ops.conv_forward(d, dtype)
        torch.cuda.synchronize()
        return state

    first = run(True, 0)
    add = (first - base).cpu()
    tol = (2e-3 if dtype == "bf16" else 2e-5) * max(1.0, want.abs().max().item())
    assert (add[:, 3:3 + 2 * Nout:2] - want).abs().max() <= tol
    untouched = torch.ones(136, dtype=torch.bool); untouched[3:3 + 2 * Nout:2] = False
    assert add[:, untouched].abs().max() == 0
    for k in range(1, 5):                                    # replays: the counters are back at zero, the sums in the same order
        assert torch.equal(run(True, k), first)
    atomic = run(False, 0)
    assert (atomic - first).abs().max().item() <= tol * 1e-2 + 1e-5
    assert int(scratch[:16384].view(torch.int32).abs().max()) == 0           # every tile's arrival counter was reset


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES[:6] + CONV_CASES[-2:], ids=[c[0] for c in CONV_CASES[:6] + CONV_CASES[-2:]])
def test_conv_wgrad_vs_torch(case, dtype):
    name, N, Cin, dhw, Cout, k, s, p, tr = case
    gen = torch.Generator().manual_seed(hash(name) % 1000 + 1)
    x = torch.randn(N, Cin, *dhw, generator=gen)
    w = (torch.randn(Cout, Cin, *k, generator=gen) / 10).requires_grad_(True)
    y = F.conv3d(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    Do, Ho, Wo = y.shape[2:]
    e16 = 8 if dtype == "bf16" else 4
    kc, ncp = -(-Cin // e16) * e16, -(-Cout // e16) * e16
    a_f32 = Cin % e16 != 0            # e.g. the RGB input of the first encoder conv: read as fp32, converted on load
    if a_f32:
        xa = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
        ldx = Cin
    else:
        xa = torch.zeros(N, *dhw, kc, device=DEV, dtype=tdt(dtype))
        xa[..., :Cin] = x.permute(0, 2, 3, 4, 1).to(DEV).to(tdt(dtype))
        ldx = kc
    dya = torch.zeros(N * Do * Ho * Wo, ncp, device=DEV, dtype=tdt(dtype))
    dya[:, :Cout] = dy.permute(0, 2, 3, 4, 1).reshape(-1, Cout).to(DEV).to(tdt(dtype))
    d = _lib.WgradDesc()
    d.NB = N; d.Di, d.Hi, d.Wi = dhw; d.Do, d.Ho, d.Wo = Do, Ho, Wo
    d.kd, d.kh, d.kw = k; d.sd, d.sh, d.sw = s; d.pd, d.ph, d.pw = p
    d.A = xa.data_ptr(); d.a_f32 = int(a_f32)
    d.a_sn = dhw[0] * dhw[1] * dhw[2] * ldx; d.a_sd = dhw[1] * dhw[2] * ldx; d.a_sh = dhw[2] * ldx; d.a_sw = ldx; d.a_sc = 1
    d.Kc_real = Cin if a_f32 else kc; d.Kc = kc
    d.dY = dya.data_ptr(); d.ldy = ncp; d.Nout = Cout
    taps = k[0] * k[1] * k[2]
    dW = torch.zeros(Cout, Cin, taps, device=DEV)
    d.dW = dW.data_ptr(); d.w_sn = Cin * taps; d.w_sc = taps; d.w_st = 1
    ops.conv_wgrad(d, dtype)
    torch.cuda.synchronize()
    ref = w.grad.reshape(Cout, Cin, taps)
    err = (dW.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"wgrad {name}[{dtype}] rel err {err:.3e}")
    assert err <= (2e-5 if dtype == "f32" else 2e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_engine_weight_shadows_match_torch_layouts(dtype):
    """ipoke_flow_prepare_weights (multi-tensor relayout + weight norm) against torch re-statements."""
    from ctypes import c_int64
    from ipoke_amd import configs
    from ipoke_amd.flow import SupervisedMacowTransformer
    from tests.helpers import wn_scale
    m = SupervisedMacowTransformer(configs.reduced_flow_arch(), dtype=dtype, device=DEV, init="none")
    deterministic_fill_(m, prefix="flow.")
    m.sync_buffers()
    eng = m.engine
    eng.prepare_weights()
    torch.cuda.synchronize()
    base = eng.lib.ipoke_flow_shadow_base(eng.handle)
    esz = 2 if dtype == "bf16" else 4
    flat = eng.shadow[base:].view(tdt(dtype))
    P = eng.params.detach()
    hid, Cc = 64, 128
    info = (c_int64 * 32)()
    seen = set()
    for i in range(eng.n_ops):
        _lib.check(eng.lib.ipoke_flow_op_info(eng.handle, i, info))
        v = list(info)
        typ, C = v[0], v[1]
        if typ == 1 and ("mcf", C) not in seen:
            seen.add(("mcf", C))
            d = ops.mcf_dims(C, Cc, dtype)
            H, K2 = 4 * C, 4 * C + Cc
            kh, kw = (2, 3) if v[8] < 2 else (3, 2)
            w1 = P[v[9]:v[9] + H * C * 6].view(H, C, kh, kw)
            g = P[v[11]:v[11] + 2 * C]; vv = P[v[12]:v[12] + 2 * C * K2].view(2 * C, K2, 1, 1)
            sd = {"net.shift_conv.weight": w1, "net.conv1x1.conv.weight_v": vv, "net.conv1x1.conv.weight_g": g,
                  "net.conv1x1.conv.bias": P[v[10]:v[10] + 2 * C]}
            sh = mcf_shadows(sd, "", C, Cc, dtype)
            for key, off in (("W1", v[13]), ("W1T", v[14]), ("W2", v[15]), ("W2T", v[16])):
                ref = sh[key]
                got = flat[off:off + ref.numel()].view_as(ref)
                err = (got.float() - ref.float()).abs().max().item()
                assert err <= (0 if dtype == "f32" else 1e-2) + 1e-6, (i, key, err)
        if typ == 2 and ("nice", C, v[18]) not in seen:
            seen.add(("nice", C, v[18]))
            cin, cout = v[18], v[19]
            e16 = 8 if dtype == "bf16" else 4
            kc1, kc3 = -(-cin // e16) * e16, -(-2 * cout // e16) * e16
            c1 = P[v[24]:v[24] + hid * cin * 9].view(hid, cin, 3, 3)
            c2 = P[v[25]:v[25] + hid * hid].view(hid, hid, 1, 1)
            g = P[v[11]:v[11] + 2 * cout]; vv = P[v[12]:v[12] + 2 * cout * hid * 9].view(2 * cout, hid, 3, 3)
            sc = wn_scale(g, vv)
            refs = {26: shadow_nt(c1, kc1, dtype=dtype), 27: shadow_t(c1, hid, dtype=dtype),
                    28: shadow_nt(c2, hid, dtype=dtype), 29: shadow_t(c2, hid, dtype=dtype),
                    30: shadow_nt(vv, hid, dtype=dtype, row_scale=sc), 31: shadow_t(vv, kc3, dtype=dtype, col_scale=sc)}
            assert (v[29] < 0) == (dtype == "bf16")        # bf16: conv2 keeps only its straight copy (the data gradient reads it K-major)
            for fld, ref in refs.items():
                if v[fld] < 0:
                    continue
                got = flat[v[fld]:v[fld] + ref.numel()].view_as(ref)
                err = (got.float() - ref.float()).abs().max().item()
                assert err <= (0 if dtype == "f32" else 1e-2) + 1e-6, (i, fld, err)
    assert len(seen) > 4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("variant", ["continuous_up", "continuous_down", "skip_up", "skip_down"])
@pytest.mark.parametrize("C", [8, 32])
def test_nice_coupling_composed(golden, C, variant, dtype):
    """conv3x3 -> ELU -> conv1x1 -> ELU -> weight-normed conv3x3 (split-K) -> affine, as the engine chains them."""
    g = golden("g1_flow_units")
    split, order = variant.split("_")
    tag = f"nice_{C}_{split}_{order}"
    o = flow_ref.NICE2d(C, 64, split, order)
    deterministic_fill_(o, prefix=tag + ".")
    sd = {k: v.detach().to(DEV) for k, v in o.state_dict().items()}
    hid, cout = 64, C // 2
    cin = C - cout
    up = order == "up"
    if split == "continuous":
        z1 = cin if up else cout
        z_off, t_off, stride = (0, z1, 1) if up else (z1, 0, 1)
    else:
        z_off, t_off, stride = (0, 1, 2) if up else (1, 0, 2)
    e16 = 8 if dtype == "bf16" else 4
    kc1 = -(-cin // e16) * e16
    x = t(g[f"x_{C}"], DEV); B = x.shape[0]; M = B * 64
    xs = ops.to_state(x)
    from tests.helpers import wn_scale
    w1 = shadow_nt(sd["net.conv1.weight"], kc1, dtype=dtype)
    w2 = shadow_nt(sd["net.conv2.weight"], hid, dtype=dtype)
    sc = wn_scale(sd["net.conv3.conv.weight_g"], sd["net.conv3.conv.weight_v"])
    w3 = shadow_nt(sd["net.conv3.conv.weight_v"], hid, dtype=dtype, row_scale=sc)
    h1 = torch.empty(M, hid, device=DEV, dtype=tdt(dtype)); h2 = torch.empty_like(h1)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = xs.data_ptr(); d.a_f32 = 1; d.a_sn = 64 * C; d.a_sh = 8 * C; d.a_sw = C; d.a_sc = stride; d.a_coff = z_off
    d.Kc_real = cin; d.Kc = kc1; d.W = w1.data_ptr(); d.ldw = 9 * kc1; d.Nout = hid; d.act = _lib.ACT_ELU
    d.C = h1.data_ptr(); d.ldc = hid
    ops.conv_forward(d, dtype)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d.A = h1.data_ptr(); d.a_sn = 64 * hid; d.a_sh = 8 * hid; d.a_sw = hid; d.a_sc = 1; d.Kc_real = hid; d.Kc = hid
    d.W = w2.data_ptr(); d.ldw = hid; d.Nout = hid; d.act = _lib.ACT_ELU; d.C = h2.data_ptr(); d.ldc = hid
    ops.conv_forward(d, dtype)
    nsplit = 3
    part = torch.zeros(nsplit, M, 64, device=DEV)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = h2.data_ptr(); d.a_sn = 64 * hid; d.a_sh = 8 * hid; d.a_sw = hid; d.a_sc = 1; d.Kc_real = hid; d.Kc = hid
    d.W = w3.data_ptr(); d.ldw = 9 * hid; d.Nout = 2 * cout; d.C = part.data_ptr(); d.c_f32 = 1; d.ldc = 64; d.splitk = nsplit
    ops.conv_forward(d, dtype)
    y, ld, scale = ops.affine_fwd(xs, part, sd["net.conv3.conv.bias"].float().contiguous(), cout, t_off, stride, B)
    torch.cuda.synchronize()
    e_y = (ops.from_state(y, B, C).cpu() - t(g[tag + "_y"])).abs().max().item()
    e_ld = (ld.cpu() - t(g[tag + "_logdet"])).abs().max().item()
    print(f"{tag}[{dtype}] y err {e_y:.3e} logdet err {e_ld:.3e}")
    assert e_y <= TOLS[dtype] * 4 and e_ld <= TOLS[dtype] * 200
    xi = ops.affine_inv(y, part, sd["net.conv3.conv.bias"].float().contiguous(), cout, t_off, stride, B)
    assert (ops.from_state(xi, B, C).cpu() - x.cpu()).abs().max() <= 1e-5


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("C,c0,Cn,stride,shuffle,params", [(64, 0, 64, 2, True, True), (60, 0, 60, 1, False, True), (32, 28, 4, 1, False, True),
                                                           (8, 0, 8, 2, True, False)])
def test_coupling_actnorm_pair_kernels(C, c0, Cn, stride, shuffle, params, dtype):
    """ipoke_affine_actnorm_fwd / ipoke_actnorm_affine_bwd (a coupling and the ActNorm (+ Shuffle) behind it in one launch per
    direction: MaCowStep's coupling -> actnorm, MultiScalePrior's coupling -> actnorm on a channel window) against the two separate
    launches they replace: both states, scales, log-det slots; gradient passed on, coupling-parameter gradients, bias and ActNorm
    partial sums."""
    from ctypes import byref
    from ipoke_amd import _lib, ops
    from ipoke_amd._lib import check, ptr
    from tests.helpers import tdt
    dev = "cuda"
    B, ld, P = 3, 64, 64
    Cp = C // 2
    t_off = 1 if stride == 2 else C - Cp
    gen = torch.Generator().manual_seed(C + c0 + Cn)
    x = torch.randn(B * P, ld, generator=gen).to(dev)
    raw = (torch.randn(2, B * P, 2 * Cp, generator=gen) * 0.5).to(dev)          # two split-K partial slabs
    bias = (torch.randn(2 * Cp, generator=gen) * 0.1).to(dev)
    ls = (torch.randn(Cn, generator=gen) * 0.2).to(dev) if params else None
    ab = torch.randn(Cn, generator=gen).to(dev) if params else None
    idx = torch.randperm(Cn, generator=gen).to(dev) if shuffle else None
    L = _lib.lib()
    # ---- forward: separate launches
    y1, ldet1, sc1 = ops.affine_fwd(x, raw, bias, Cp, t_off, stride, B)
    y2 = ops.actnorm_fwd(y1, c0, Cn, ls, ab, idx)
    # ---- forward: one launch
    d, keep = ops._affine_desc(raw, bias, Cp, t_off, stride, ld)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    sc = torch.empty(B * P, Cp, device=dev); slots = torch.zeros(B, device=dev)
    i32 = None if idx is None else idx.to(torch.int32).contiguous()
    check(L.ipoke_affine_actnorm_fwd(byref(d), ptr(x), ptr(o1), ptr(o2), ptr(sc), ptr(slots), 1, B, c0, Cn, ptr(ls), ptr(ab), ptr(i32),
                                     _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(o1, y1) and torch.equal(sc, sc1)
    assert (o2 - y2).abs().max().item() <= 1e-6 * max(1.0, y2.abs().max().item())
    assert (slots - ldet1).abs().max().item() <= 1e-4
    # the saved-nothing form (out == NULL) writes the same second state
    o2b = torch.empty_like(x)
    check(L.ipoke_affine_actnorm_fwd(byref(d), ptr(x), None, ptr(o2b), None, None, 1, B, c0, Cn, ptr(ls), ptr(ab), ptr(i32), _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(o2b, o2)
    # ---- backward: separate launches
    dy2 = torch.randn(B * P, ld, generator=gen).to(dev)
    dld = torch.randn(B, generator=gen).to(dev)
    e16 = 8 if dtype == "bf16" else 4
    ldp = -(-2 * Cp // e16) * e16 + e16                              # one extra chunk of K padding
    dx_a = torch.empty_like(x)
    part_a = torch.zeros(B, 2 * Cn, device=dev) if params else None
    check(L.ipoke_actnorm_bwd(ptr(dy2), ptr(y1), ptr(dx_a), B * P, ld, c0, Cn, ptr(ls), ptr(i32), ptr(dld), B, P, ptr(part_a),
                              _lib.current_stream()))
    g_ref = torch.empty_like(x)
    dprm_ref = torch.full((B * P, ldp), 7.0, device=dev, dtype=tdt(dtype))
    dbp_ref = torch.zeros(B, 2 * Cp, device=dev)
    check(L.ipoke_affine_bwd(Cp, t_off, stride, P, ld, ptr(dx_a), ptr(x), ptr(sc1), ptr(dld), ptr(g_ref), ptr(dprm_ref), ldp, ptr(dbp_ref), B,
                             _lib.DTYPES[dtype], _lib.current_stream()))
    # ---- backward: one launch
    g, dprm = torch.empty_like(x), torch.full((B * P, ldp), 7.0, device=dev, dtype=tdt(dtype))
    dbp = torch.zeros(B, 2 * Cp, device=dev)
    part = torch.zeros(B, 2 * Cn, device=dev) if params else None
    check(L.ipoke_actnorm_affine_bwd(c0, Cn, ptr(ls), ptr(i32), ptr(dy2), ptr(y1), ptr(part), Cp, t_off, stride, P, ld, ptr(x), ptr(sc1),
                                     ptr(dld), ptr(g), ptr(dprm), ldp, ptr(dbp), B, _lib.DTYPES[dtype], _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(g, g_ref) and torch.equal(dprm, dprm_ref)
    assert (dbp - dbp_ref).abs().max().item() <= 1e-5 * max(1.0, dbp_ref.abs().max().item())
    if params:
        assert (part - part_a).abs().max().item() <= 1e-5 * max(1.0, part_a.abs().max().item())


@pytest.mark.parametrize("M,N,K,mask", [(1280, 2048, 2048, True), (2560, 2048, 2048, False), (2048, 256, 192, True), (200, 128, 64, False),
                                        (1280, 200, 128, True)])
def test_kmajor_gemm(M, N, K, mask):
    """ipoke_conv_desc.w_kmajor (igemm_nn_glds): C = A W with W given as the row-major [K][N] matrix -- the data gradient of a 1 x 1
    convolution read from the weight's straight copy -- with the ELU'(saved output) mask of the coupling nets, at the c2 (80-row tiles,
    two K halves), c3 (160) and generic (128, ragged M / N) tile shapes, against torch's matmul of the same bf16 operands."""
    from ctypes import byref
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(K, N, generator=g) / K ** 0.5).bfloat16()
    h = (torch.randn(M, N, generator=g)).bfloat16()                       # a saved ELU output: > 0 -> derivative 1, <= 0 -> 1 + h
    S = 64 if M % 64 == 0 else 8
    B = M // S
    ad, wd, hd = a.to(DEV), w.to(DEV), h.to(DEV)
    c = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    d = ops.conv_desc(B, (1, 8, S // 8), (1, 8, S // 8), (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d.A = ad.data_ptr(); d.a_sn = S * K; d.a_sd = 0; d.a_sh = (S // 8) * K; d.a_sw = K; d.a_sc = 1; d.Kc_real = K; d.Kc = K
    d.W = wd.data_ptr(); d.ldw = N; d.Nout = N; d.w_kmajor = 1
    if mask:
        d.dact = hd.data_ptr(); d.ld_dact = N; d.dact_act = _lib.ACT_ELU
    d.C = c.data_ptr(); d.ldc = N
    ops.conv_forward(d, "bf16")
    ref = a.float() @ w.float()
    if mask:
        ref = ref * torch.where(h.float() > 0, torch.ones(()), h.float() + 1)
    err = (c.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"kmajor gemm {M}x{N}x{K} mask={mask}: rel err {err:.3e}")
    assert err <= 1.2e-2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_copy_cols(dtype):
    """ipoke_copy_cols: dst[m][0:C] = src[m][0:C] with different row pitches and a column offset on the destination (the activated
    conditioning map behind conv2's columns of a coupling net's hidden tile, condition_nice); everything else stays untouched."""
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    e16 = 4 if dtype == "f32" else 8
    g = torch.Generator().manual_seed(5)
    M, C, lds, ldd, off = 1280, 16 * e16, 20 * e16, 40 * e16, 8 * e16
    src = torch.randn(M, lds, generator=g).to(td).to(DEV)
    dst = torch.full((M, ldd), 3.0, dtype=td, device=DEV)
    before = dst.clone()
    esz = src.element_size()
    check(_lib.lib().ipoke_copy_cols(src.data_ptr(), lds, dst.data_ptr() + off * esz, ldd, C, M, ops._dt(dtype), _lib.current_stream()))
    assert torch.equal(dst[:, off:off + C], src[:, :C])
    assert torch.equal(dst[:, :off], before[:, :off]) and torch.equal(dst[:, off + C:], before[:, off + C:])
    # misaligned widths are refused, not rounded
    assert _lib.lib().ipoke_copy_cols(src.data_ptr(), lds, dst.data_ptr(), ldd, C - 1, M, ops._dt(dtype), _lib.current_stream()) != 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_output_scatter_with_depth(dtype):
    """ipoke_conv_desc.c_scatter with the depth stride c_sd: a stride-1 3-D convolution whose outputs go to every second position of a
    larger map (one parity class of a strided convolution's data gradient) -- against the dense result placed by index; the other
    positions keep their contents."""
    from ipoke_amd import nn as K
    g = torch.Generator().manual_seed(9)
    N, cin, cout, (D, H, W) = 2, 16, 24, (3, 5, 6)
    x = torch.randn(N * D * H * W, cin, generator=g)
    w = torch.randn(cout, cin, 2, 2, 1, generator=g) / (cin * 4) ** 0.5
    xc = K.CL(x.to(DEV).to(ops.torch_dtype(dtype)), N, (D, H, W), cin)
    wop, kc = K.weight_operand(w.to(DEV), dtype)
    dense = K.conv(xc, wop, kc, cout, (2, 2, 1), (1, 1, 1), (0, 0, 0), dtype, odhw=(D, H, W))      # windows overhanging the input read zeros
    Do, Ho, Wo = 2 * D, 2 * H, 2 * W
    out = torch.full((N * Do * Ho * Wo, dense.t.shape[1]), 5.0, dtype=dense.t.dtype, device=DEV)
    rd, rh, rw = 1, 0, 1
    K.conv(xc, wop, kc, cout, (2, 2, 1), (1, 1, 1), (0, 0, 0), dtype, out=out, odhw=(D, H, W),
           scatter=(Do * Ho * Wo, 2 * Ho * Wo, 2 * Wo, 2, (rd * Ho + rh) * Wo + rw))
    o5 = out.view(N, Do, Ho, Wo, -1)
    assert torch.equal(o5[:, rd::2, rh::2, rw::2, :cout], dense.t.view(N, D, H, W, -1)[..., :cout])
    mask = torch.ones(N, Do, Ho, Wo, dtype=torch.bool, device=DEV)
    mask[:, rd::2, rh::2, rw::2] = False
    assert (o5[mask] == 5.0).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("split", ["half", "skip"])
def test_actnorm_inv_with_conditioning_operand(dtype, split):
    """ipoke_actnorm_inv_ext: the ActNorm (+ Shuffle) inverse and, in the same launch, the conditioning operand of the coupling that is
    inverted next -- bit-identical to ipoke_actnorm_inv followed by ipoke_extract_cols (continuous and even / odd channel splits, a
    sub-range ActNorm as in the priors, padded operand columns zero, the columns beyond ext_ld untouched); ext == NULL is the plain
    inverse; padding wider than the state is refused."""
    g = torch.Generator().manual_seed(3)
    M, ld, c0, C = 2 * 64, 32, 8, 24
    s = torch.randn(M, ld, generator=g).to(DEV)
    ls, b = (0.3 * torch.randn(C, generator=g)).to(DEV), torch.randn(C, generator=g).to(DEV)
    inv_idx = torch.randperm(C, generator=g).to(torch.int32).to(DEV)
    e_off, e_stride, e_C = (16, 1, 12) if split == "half" else (1, 2, 16)
    ext_ld = 32
    td = ops.torch_dtype(dtype)
    want_state = ops.actnorm_inv(s, c0, C, ls, b, inv_idx)
    want_ext = torch.empty(M, ext_ld, dtype=td, device=DEV)
    check(_lib.lib().ipoke_extract_cols(want_state.data_ptr(), ld, e_off, e_stride, e_C, want_ext.data_ptr(), ext_ld, M, ops._dt(dtype),
                                        _lib.current_stream()))
    out = torch.empty_like(s)
    ext = torch.full((M, ext_ld), 7.0, dtype=td, device=DEV)
    check(_lib.lib().ipoke_actnorm_inv_ext(s.data_ptr(), out.data_ptr(), M, ld, c0, C, ls.data_ptr(), b.data_ptr(), inv_idx.data_ptr(),
                                           ext.data_ptr(), ext_ld, e_off, e_stride, e_C, ops._dt(dtype), _lib.current_stream()))
    assert torch.equal(out, want_state) and torch.equal(ext, want_ext)
    assert float(ext[:, e_C:].abs().max()) == 0.0
    out2 = torch.empty_like(s)
    check(_lib.lib().ipoke_actnorm_inv_ext(s.data_ptr(), out2.data_ptr(), M, ld, c0, C, ls.data_ptr(), b.data_ptr(), inv_idx.data_ptr(),
                                           None, 0, 0, 1, 0, ops._dt(dtype), _lib.current_stream()))
    assert torch.equal(out2, want_state)
    assert _lib.lib().ipoke_actnorm_inv_ext(s.data_ptr(), out.data_ptr(), M, ld, c0, C, ls.data_ptr(), b.data_ptr(), inv_idx.data_ptr(),
                                            ext.data_ptr(), e_C + ld + 1, e_off, e_stride, e_C, ops._dt(dtype), _lib.current_stream()) != 0
