"""conv3 of a coupling net fused with the coupling transform (ipoke_conv3x3_coupling: the K splits of a row tile meet inside
the launch) against the two launches it replaces -- ipoke_conv_forward with the same split count, then ipoke_affine_fwd_ext /
ipoke_affine_actnorm_fwd / ipoke_affine_inv_ext (reference: NICEConvBlock's conv3 + the affine transform + ActNorm2dFlow,
macow_utils.py:42-66, 270-281, macow2.py:569-593).  Same arithmetic in the same order: every output must be BIT-identical."""
from ctypes import byref

import pytest
import torch

from ipoke_amd import _lib, ops
from ipoke_amd._lib import AffineDesc, CouplingEpi, check
from tests.helpers import shadow_nt

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scratch():
    lib = _lib.lib()
    nb = lib.ipoke_conv3x3_coupling_xchg_bytes()
    x = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib.ipoke_conv3x3_coupling_xchg_init(x.data_ptr(), ops._s()))
    return x


def _scratch_is_clean(x):
    head = x[:256].view(torch.int32)
    return int(head[0].item()) == 0 and bool((x[256:] == 255).all().item())


def _problem(B, Kc, Cp, ld, t_off, t_stride, seed):
    gen = torch.Generator().manual_seed(seed)
    M = B * 64
    h2 = (torch.randn(M, Kc, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
    w = torch.randn(2 * Cp, Kc, 1, 3, 3, generator=gen) / (Kc * 9) ** 0.5 * 2
    ws = shadow_nt(w.to(DEV).contiguous(), Kc, dtype="bf16")
    bias = (torch.randn(2 * Cp, generator=gen) * 0.3).to(DEV)
    state = torch.randn(M, ld, generator=gen).to(DEV)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    d.A = h2.data_ptr(); d.a_sn = 64 * Kc; d.a_sd = 64 * Kc; d.a_sh = 8 * Kc; d.a_sw = Kc; d.a_sc = 1
    d.Kc_real = Kc; d.Kc = Kc
    d.W = ws.data_ptr(); d.ldw = ws.shape[1]; d.Nout = 2 * Cp
    a = AffineDesc()
    a.bias = bias.data_ptr(); a.Cp = Cp; a.t_off = t_off; a.t_stride = t_stride; a.P = 64; a.ld = ld
    return d, a, state, (h2, ws, bias)


CASES = [
    # B, Kc, Cp, ld, t_off, t_stride          -> splits
    (20, 2048, 32, 64, 0, 2),               # c2: 10 tiles x 16
    (20, 2048, 32, 64, 1, 2),
    (32, 2048, 16, 64, 32, 1),              # c5 batch: 16 tiles x 16
    (40, 1024, 16, 32, 0, 2),               # c3: 20 tiles x 8
    (5, 2048, 30, 60, 0, 2),                # 3 tiles (the last half empty) x 32, ragged widths
    (3, 512, 4, 8, 4, 1),                   # 2 tiles x 8 (nchunks bound)
    (100, 512, 8, 16, 1, 2),                # 50 tiles x 4: two 16-row slices per owner
]


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}_K{c[1]}_Cp{c[2]}" for c in CASES])
def test_fused_conv3_coupling_is_bit_identical(case, mode):
    B, Kc, Cp, ld, t_off, t_stride = case
    lib = _lib.lib()
    M = B * 64
    ns = lib.ipoke_conv3x3_coupling_splitk(M, Kc, _lib.BF16)
    assert ns in (4, 8, 16, 32)
    d, a, state, keep = _problem(B, Kc, Cp, ld, t_off, t_stride, seed=B * 7 + Cp + mode)
    gen = torch.Generator().manual_seed(99)
    an_C, an_c0 = ld, 0
    an_ls = (torch.randn(an_C, generator=gen) * 0.2).to(DEV)
    an_b = torch.randn(an_C, generator=gen).to(DEV)
    an_idx = torch.randperm(an_C, generator=gen).to(torch.int32).to(DEV)
    ext_ld = -(-Cp // 8) * 8 + 8

    def outputs():
        return dict(out=torch.full((M, ld), float("nan"), device=DEV), out2=torch.full((M, ld), float("nan"), device=DEV),
                    scale=torch.full((M, Cp), float("nan"), device=DEV), slots=torch.full((B, 4), float("nan"), device=DEV),
                    ext=torch.full((M, ext_ld), float("nan"), device=DEV).to(torch.bfloat16))

    # --- the two launches
    r = outputs()
    slabs = torch.zeros(ns, M, 64, device=DEV)
    d.C = slabs.data_ptr(); d.c_f32 = 1; d.ldc = 64; d.splitk = ns
    ops.conv_forward(d, "bf16")
    a.raw = slabs.data_ptr(); a.nsplit = ns; a.split_stride = M * 64; a.ldraw = 64
    if mode == 0:
        check(lib.ipoke_affine_fwd_ext(byref(a), state.data_ptr(), r["out"].data_ptr(), r["scale"].data_ptr(), r["slots"].data_ptr(), 4, B,
                                       r["ext"].data_ptr(), ext_ld, _lib.BF16, ops._s()))
    elif mode == 1:
        check(lib.ipoke_affine_actnorm_fwd(byref(a), state.data_ptr(), r["out"].data_ptr(), r["out2"].data_ptr(), r["scale"].data_ptr(),
                                           r["slots"].data_ptr(), 4, B, an_c0, an_C, an_ls.data_ptr(), an_b.data_ptr(), an_idx.data_ptr(),
                                           ops._s()))
    else:
        check(lib.ipoke_affine_inv_ext(byref(a), state.data_ptr(), r["out"].data_ptr(), B, r["ext"].data_ptr(), ext_ld, _lib.BF16, ops._s()))

    # --- one launch, twice (the scratch must be back in its initial state after each)
    xchg = _scratch()
    for rep in range(2):
        f = outputs()
        e = CouplingEpi()
        e.mode = mode; e.inp = state.data_ptr(); e.out = f["out"].data_ptr(); e.xchg = xchg.data_ptr()
        if mode != 2:
            e.scale_out = f["scale"].data_ptr(); e.logdet_slot = f["slots"].data_ptr(); e.slot_stride = 4
        if mode == 1:
            e.out2 = f["out2"].data_ptr(); e.an_c0 = an_c0; e.an_C = an_C
            e.an_log_scale = an_ls.data_ptr(); e.an_bias = an_b.data_ptr(); e.an_idx = an_idx.data_ptr()
        else:
            e.ext = f["ext"].data_ptr(); e.ext_ld = ext_ld
        d.C = 0; d.splitk = 1
        check(lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), B, _lib.BF16, ops._s()))
        torch.cuda.synchronize()
        assert _scratch_is_clean(xchg), f"exchange scratch not restored (rep {rep})"
        names = {0: ("out", "scale", "slots", "ext"), 1: ("out", "out2", "scale", "slots"), 2: ("out", "ext")}[mode]
        for k in names:
            x, y = r[k].float().cpu(), f[k].float().cpu()
            assert not torch.isnan(y).any(), f"{k}: unwritten elements"
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{k} differs: max {(x - y).abs().max().item():.3e} (rep {rep})"


def test_fused_conv3_coupling_under_uneven_load():
    """The hand-off does not depend on workgroup placement or timing: the fused launch beside a copy stream that keeps taking CUs."""
    B, Kc, Cp, ld = 20, 2048, 32, 64
    lib = _lib.lib()
    M = B * 64
    ns = lib.ipoke_conv3x3_coupling_splitk(M, Kc, _lib.BF16)
    d, a, state, keep = _problem(B, Kc, Cp, ld, 0, 2, seed=5)
    slabs = torch.zeros(ns, M, 64, device=DEV)
    d.C = slabs.data_ptr(); d.c_f32 = 1; d.ldc = 64; d.splitk = ns
    ops.conv_forward(d, "bf16")
    a.raw = slabs.data_ptr(); a.nsplit = ns; a.split_stride = M * 64; a.ldraw = 64
    ref = torch.empty(M, ld, device=DEV); slots = torch.empty(B, 4, device=DEV)
    check(lib.ipoke_affine_fwd_ext(byref(a), state.data_ptr(), ref.data_ptr(), 0, slots.data_ptr(), 4, B, 0, 0, _lib.BF16, ops._s()))
    xchg = _scratch()
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device=DEV)
    torch.cuda.synchronize()
    d.C = 0; d.splitk = 1
    for it in range(20):
        with torch.cuda.stream(side):
            for _ in range(3):
                big = big * 1.0001 + 0.5
        out = torch.full((M, ld), float("nan"), device=DEV); s2 = torch.empty(B, 4, device=DEV)
        e = CouplingEpi()
        e.mode = 0; e.inp = state.data_ptr(); e.out = out.data_ptr(); e.logdet_slot = s2.data_ptr(); e.slot_stride = 4; e.xchg = xchg.data_ptr()
        check(lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), B, _lib.BF16, ops._s()))
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and torch.equal(s2, slots), f"iteration {it}"
    assert _scratch_is_clean(xchg)


def test_fused_conv3_coupling_rejects_bad_arguments():
    lib = _lib.lib()
    d, a, state, keep = _problem(4, 512, 8, 16, 0, 2, seed=1)
    out = torch.empty_like(state)
    e = CouplingEpi()
    e.mode = 0; e.inp = state.data_ptr(); e.out = out.data_ptr()
    assert lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), 4, _lib.BF16, ops._s()) != 0          # no scratch
    xchg = _scratch()
    e.xchg = xchg.data_ptr()
    assert lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), 4, _lib.F32, ops._s()) != 0           # bf16 only
    e.mode = 1                                                                                             # ActNorm mode without out2
    assert lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), 4, _lib.BF16, ops._s()) != 0
    e.mode = 0
    assert lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), 5, _lib.BF16, ops._s()) != 0          # B does not match the maps
    assert lib.ipoke_conv3x3_coupling_splitk(200 * 64, 2048, _lib.BF16) == 0                                # too many tiles for one round
    assert lib.ipoke_conv3x3_coupling(byref(d), byref(a), byref(e), 4, _lib.BF16, ops._s()) == 0
    torch.cuda.synchronize()
    assert _scratch_is_clean(xchg)
