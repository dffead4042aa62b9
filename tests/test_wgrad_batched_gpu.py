"""ipoke_conv_wgrad_batched (the flow engine's batched weight gradients of the coupling nets, macow_utils.py:270-281) against torch's
conv weight gradient on the same bf16 operands: the 128 x 128 tiles of wide outputs (conv1 / conv2) and the 64 x 256 tiles of narrow
ones (conv3: <= 64 output rows), several problems per launch, ragged output widths."""
import struct
from ctypes import byref

import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, ops
from ipoke_amd._lib import WgradDesc, check

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,Cin,Cout,k,nb", [(3, 256, 64, 3, 2), (5, 512, 60, 3, 3), (4, 256, 24, 3, 1), (2, 128, 64, 3, 2),
                                               (3, 64, 192, 3, 2), (3, 256, 256, 1, 2),
                                               (20, 2048, 64, 3, 2), (20, 32, 2048, 3, 2), (7, 320, 136, 3, 1), (1, 64, 64, 3, 1)])
def test_batched_weight_gradient_matches_torch(B, Cin, Cout, k, nb):
    """(the 3 x 3 cases run the stationary-input kernel wgrad3x3_lat8 of round 6 -- incl. the benchmarked conv3 / conv1 shapes of a
    coupling net at B = 20, odd batches (a half-empty last stage), ragged output and input widths; IPOKE_WGRAD_LAT8=0 sends them through
    the implicit-GEMM kernels again)"""
    lib = _lib.lib()
    M = B * 64
    gen = torch.Generator().manual_seed(B * 100 + Cout + k)
    pad = k // 2
    ldy = -(-Cout // 8) * 8
    xs = [(torch.randn(B, Cin, 8, 8, generator=gen) * 0.5).to(torch.bfloat16) for _ in range(nb)]
    dys = [(torch.randn(B, Cout, 8, 8, generator=gen) * 0.5).to(torch.bfloat16) for _ in range(nb)]
    # reference: d/dW of sum(conv(x, W) * dy) on the bf16-rounded operands, fp32 accumulation
    refs = []
    for x, dy in zip(xs, dys):
        w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
        y = F.conv2d(x.float(), w, padding=pad)
        (y * dy.float()).sum().backward()
        refs.append(w.grad.clone())
    # device operands: channels-last activations [M][Cin], gradients [M][ldy] (zero padded columns), one flat fp32 buffer of dW
    a_all = torch.stack([x.permute(0, 2, 3, 1).reshape(M, Cin) for x in xs]).contiguous().to(DEV)
    y_all = torch.zeros(nb, M, ldy, dtype=torch.bfloat16)
    for i, dy in enumerate(dys):
        y_all[i, :, :Cout] = dy.permute(0, 2, 3, 1).reshape(M, Cout)
    y_all = y_all.to(DEV)
    dw = torch.full((nb, Cout, Cin, k * k), float("nan"), device=DEV)
    es = lib.ipoke_wgrad_batch_entry_size()
    assert es == struct.calcsize("qqqiiiiq")
    raw = b"".join(struct.pack("qqqiiiiq", i * M * Cin * 2, i * M * ldy * 2, i * Cout * Cin * k * k, k, k, pad, pad, 0) for i in range(nb))
    entries = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(DEV)
    d = WgradDesc()
    d.NB, d.Di, d.Hi, d.Wi, d.Do, d.Ho, d.Wo = B, 1, 8, 8, 1, 8, 8
    d.kd, d.kh, d.kw, d.sd, d.sh, d.sw, d.pd, d.ph, d.pw = 1, k, k, 1, 1, 1, 0, pad, pad
    d.a_f32 = 0; d.a_sn = 64 * Cin; d.a_sh = 8 * Cin; d.a_sw = Cin; d.a_sc = 1; d.Kc_real = Cin; d.Kc = Cin
    d.ldy = ldy; d.Nout = Cout
    d.w_sn = Cin * k * k; d.w_sc = k * k; d.w_st = 1
    check(lib.ipoke_conv_wgrad_batched(byref(d), entries.data_ptr(), nb, a_all.data_ptr(), y_all.data_ptr(), dw.data_ptr(), _lib.BF16,
                                       ops._s()))
    torch.cuda.synchronize()
    got = dw.cpu().view(nb, Cout, Cin, k, k)
    assert not torch.isnan(got).any(), "unwritten weight-gradient elements"
    for i in range(nb):
        err = (got[i] - refs[i]).abs().max().item()
        assert err <= 2e-3 * max(1.0, refs[i].abs().max().item()), (i, err)


@pytest.mark.parametrize("B,C,nb", [(20, 256, 2), (3, 128, 3), (5, 384, 1)])
def test_adam_in_the_weight_gradient_epilogue_is_bit_identical(B, C, nb):
    """ipoke_wgrad_desc.adam: Adam-amsgrad of dense 1 x 1 weights (conv2 of NICEConvBlock, macow_utils.py:270-281, under
    torch.optim.Adam(amsgrad=True, weight_decay), second_stage_video.py:648-650) applied in the epilogue of the batched weight-gradient
    launch -- against the two-pass form it replaces: the same launch writing the gradient, then ipoke_adam_amsgrad_step over it.
    Parameters, both moments and the running maximum BIT-identical over two optimizer steps (carried state); the operand copy is the
    bf16 cast of the new parameters; keep_grad = 1 also leaves the gradient in w_base."""
    from ipoke_amd._lib import WgradAdam
    lib = _lib.lib()
    M = B * 64
    gen = torch.Generator().manual_seed(31 * B + C)
    n = C * C
    pad_front = 64                                           # problems do not start at the buffers' bases
    flat = pad_front + nb * n
    p0 = (torch.randn(flat, generator=gen) * 0.05).to(DEV)
    state = {k_: v_.to(DEV) for k_, v_ in (("m", torch.randn(flat, generator=gen) * 1e-3), ("v", torch.rand(flat, generator=gen) * 1e-5),
                                           ("vmax", torch.rand(flat, generator=gen) * 1e-5))}
    hyp = dict(lr=3e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5, gs=0.5)

    def entries(sh):
        raw = b"".join(struct.pack("qqqiiiiq", i * M * C * 2, i * M * C * 2, pad_front + i * n, 1, 1, 0, 0, (8 + i * n) if sh else 0) for i in range(nb))
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(DEV)

    def desc():
        d = WgradDesc()
        d.NB, d.Di, d.Hi, d.Wi, d.Do, d.Ho, d.Wo = B, 1, 8, 8, 1, 8, 8
        d.kd, d.kh, d.kw, d.sd, d.sh, d.sw, d.pd, d.ph, d.pw = 1, 1, 1, 1, 1, 1, 0, 0, 0
        d.a_f32 = 0; d.a_sn = 64 * C; d.a_sh = 8 * C; d.a_sw = C; d.a_sc = 1; d.Kc_real = C; d.Kc = C
        d.ldy = C; d.Nout = C; d.w_sn = C; d.w_sc = 1; d.w_st = 0
        return d

    two = {k_: [p0.clone()] + [state[q].clone() for q in ("m", "v", "vmax")] for k_ in ("plain", "fused", "keep")}
    operand = {k_: torch.zeros(8 + nb * n, dtype=torch.bfloat16, device=DEV) for k_ in ("fused", "keep")}
    for step in (1, 2):
        a_all = (torch.randn(nb, M, C, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
        y_all = (torch.randn(nb, M, C, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
        # two passes: gradient, then the stand-alone optimizer kernel over the whole flat buffer
        grad = torch.zeros(flat, device=DEV)
        d = desc()
        e_plain = entries(False)
        check(lib.ipoke_conv_wgrad_batched(byref(d), e_plain.data_ptr(), nb, a_all.data_ptr(), y_all.data_ptr(), grad.data_ptr(), _lib.BF16, ops._s()))
        P, Mm, V, X = two["plain"]
        check(lib.ipoke_adam_amsgrad_step(P.data_ptr(), grad.data_ptr(), Mm.data_ptr(), V.data_ptr(), X.data_ptr(), flat, hyp["lr"], hyp["b1"],
                                          hyp["b2"], hyp["eps"], hyp["wd"], step, hyp["gs"], ops._s()))
        for mode in ("fused", "keep"):
            P2, M2, V2, X2 = two[mode]
            wa = WgradAdam()
            wa.params, wa.m, wa.v, wa.vmax, wa.operand = P2.data_ptr(), M2.data_ptr(), V2.data_ptr(), X2.data_ptr(), operand[mode].data_ptr()
            wa.lr, wa.beta1, wa.beta2, wa.eps, wa.weight_decay, wa.grad_scale, wa.step = hyp["lr"], hyp["b1"], hyp["b2"], hyp["eps"], hyp["wd"], hyp["gs"], step
            wa.keep_grad = int(mode == "keep")
            d2 = desc()
            from ctypes import addressof
            d2.adam = addressof(wa)
            g2 = torch.full((flat,), float("nan"), device=DEV)
            e_f = entries(True)
            check(lib.ipoke_conv_wgrad_batched(byref(d2), e_f.data_ptr(), nb, a_all.data_ptr(), y_all.data_ptr(), g2.data_ptr(), _lib.BF16, ops._s()))
            torch.cuda.synchronize()
            sl = slice(pad_front, flat)
            for name, a_, b_ in (("p", P, P2), ("m", Mm, M2), ("v", V, V2), ("vmax", X, X2)):
                assert torch.equal(a_[sl], b_[sl]), (mode, step, name, (a_[sl] - b_[sl]).abs().max().item())
                assert torch.equal(b_[:pad_front], (p0 if name == "p" else state[name])[:pad_front]), "wrote outside the problems"
            assert torch.equal(operand[mode][8:], P2[sl].to(torch.bfloat16)) and float(operand[mode][:8].abs().max()) == 0
            if mode == "keep":
                assert torch.equal(g2[sl], grad[sl])
            else:
                assert torch.isnan(g2).all(), "the fused launch must not write the gradient"
        assert not torch.equal(two["plain"][0], p0)


@pytest.mark.parametrize("N,Cin,Cout,D,HW,kd", [(4, 64, 64, 1, 64, 1), (4, 128, 96, 1, 64, 1), (4, 64, 128, 4, 32, 3), (4, 40, 72, 2, 48, 3),
                                                 (20, 64, 64, 1, 128, 1), (1, 256, 256, 1, 128, 1)])
def test_halo_staged_weight_gradient_matches_torch(N, Cin, Cout, D, HW, kd):
    """wgrad3x3_halo (round 6): weight gradient of stride-1 3x3 (kd = 1) / 3x3x3 (kd = 3) "same" convolutions on large maps with the input
    patch + halo stationary in LDS -- BasicBlock of the 3-D encoder (motion_encoder.py:45-74) and the SPADE decoder's 3x3 layers
    (autoencoders/util.py:106-192) in first-stage training.  Against torch's autograd on the same bf16 operands: split-M slabs with the
    split count the library names (ipoke_conv_wgrad_splitm), summed by ipoke_reduce_rows as the trainer does; and one split (direct
    store).  Ragged channel counts, maps that are not square multiples of the patch grid's power of two, depth taps that leave the volume."""
    lib = _lib.lib()
    gen = torch.Generator().manual_seed(N * 1000 + Cin + Cout + kd)
    H = W = HW
    x = (torch.randn(N, Cin, D, H, W, generator=gen) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(N, Cout, D, H, W, generator=gen) * 0.5).to(torch.bfloat16)
    xg, dyg = x.to(DEV).float(), dy.to(DEV).float()
    w = torch.zeros(Cout, Cin, kd, 3, 3, device=DEV, requires_grad=True)
    y = F.conv3d(xg, w, padding=(kd // 2, 1, 1))
    (y * dyg).sum().backward()
    ref = w.grad.detach()
    M = N * D * H * W
    kc, ldy = -(-Cin // 8) * 8, -(-Cout // 8) * 8
    xa = torch.zeros(M, kc, dtype=torch.bfloat16, device=DEV); xa[:, :Cin] = x.to(DEV).permute(0, 2, 3, 4, 1).reshape(M, Cin)
    ya = torch.zeros(M, ldy, dtype=torch.bfloat16, device=DEV); ya[:, :Cout] = dy.to(DEV).permute(0, 2, 3, 4, 1).reshape(M, Cout)
    taps = kd * 9

    def desc():
        d = WgradDesc()
        d.NB, d.Di, d.Hi, d.Wi, d.Do, d.Ho, d.Wo = N, D, H, W, D, H, W
        d.kd, d.kh, d.kw, d.sd, d.sh, d.sw, d.pd, d.ph, d.pw = kd, 3, 3, 1, 1, 1, kd // 2, 1, 1
        d.A = xa.data_ptr(); d.a_f32 = 0; d.a_sn = D * H * W * kc; d.a_sd = H * W * kc; d.a_sh = W * kc; d.a_sw = kc; d.a_sc = 1
        d.Kc_real = kc; d.Kc = kc; d.Kc_store = Cin
        d.dY = ya.data_ptr(); d.ldy = ldy; d.Nout = Cout
        d.w_sn = Cin * taps; d.w_sc = taps; d.w_st = 1
        return d

    d = desc()
    sp = lib.ipoke_conv_wgrad_splitm(byref(d), _lib.BF16, 256)
    assert sp >= 1, "the halo-staged kernel should claim this problem"
    nel = Cout * Cin * taps
    slabs = torch.full((sp, nel), float("nan"), device=DEV)
    d.splitm = sp; d.split_stride = nel; d.dW = slabs.data_ptr()
    check(lib.ipoke_conv_wgrad(byref(d), _lib.BF16, ops._s()))
    got = torch.empty(nel, device=DEV)
    if sp > 1:
        check(lib.ipoke_reduce_rows(slabs.data_ptr(), got.data_ptr(), sp, nel, ops._s()))
    else:
        got = slabs[0]
    torch.cuda.synchronize()
    assert not torch.isnan(slabs).any(), "unwritten slab elements"
    scale = max(1.0, ref.abs().max().item())
    err = (got.view_as(ref) - ref).abs().max().item()
    print(f"halo wgrad N={N} {Cin}->{Cout} D={D} {H}x{W} kd={kd}: {sp} splits, max err {err:.3e} (ref max {ref.abs().max().item():.1f})")
    assert err <= 3e-3 * scale
    one = torch.full((nel,), float("nan"), device=DEV)
    d1 = desc(); d1.splitm = 1; d1.dW = one.data_ptr()
    check(lib.ipoke_conv_wgrad(byref(d1), _lib.BF16, ops._s()))
    torch.cuda.synchronize()
    assert not torch.isnan(one).any() and (one.view_as(ref) - ref).abs().max().item() <= 3e-3 * scale
