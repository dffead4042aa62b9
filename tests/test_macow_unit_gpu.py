"""Fused MaCowUnit kernels (csrc/mcf_unit.hip: conv1(A) -> conv2(B) -> actnorm1 -> conv3(C) -> conv4(D) -> actnorm2 in one
launch per direction) against (i) the oracle's MaCowUnit (plain PyTorch fp32 restatement of macow2.py:925-995, autograd for
the backward pass) and (ii) the chain of per-layer kernels, layer by layer, on every tensor the backward pass and the
weight-gradient GEMMs consume."""
import pytest
import torch

from ipoke_amd import _lib, ops
from ipoke_amd.utils.detfill import deterministic_fill_
from oracle import flow_ref
from tests.helpers import mcf_shadows, tdt

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = "bf16"


def _unit(C, ld, B, seed):
    o = flow_ref.MaCowUnit(C, (2, 3), 128)
    deterministic_fill_(o, prefix=f"unit{C}.")
    with torch.no_grad():
        for an in (o.actnorm1, o.actnorm2):
            an.initialized.fill_(1)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, ld, 8, 8, generator=gen)
    h = torch.randn(B, 128, 8, 8, generator=gen)
    layers = [o.conv1, o.conv2, o.conv3, o.conv4]
    shs = [mcf_shadows({k: v.detach().to(DEV) for k, v in l.state_dict().items()}, "", C, 128, DT) for l in layers]
    posts = [None, o.actnorm1, None, o.actnorm2]
    return o, x, h, shs, posts


def _descs(C, ld, B, cond, shs, posts, keep):
    d4 = (_lib.McfDesc * 4)()
    for k in range(4):
        d = d4[k]
        d.ld, d.C, d.B = ld, C, B
        d.cond, d.Cc = cond.data_ptr(), cond.shape[1]
        sh = shs[k]
        d.W1, d.W2, d.bias2 = sh["W1"].data_ptr(), sh["W2"].data_ptr(), sh["bias"].data_ptr()
        d.W1T, d.W2T = sh["W1T"].data_ptr(), sh["W2T"].data_ptr()
        d.order = k
        d.rows_per_block = 16
        if posts[k] is not None:
            ls = posts[k].log_scale.detach().flatten().to(DEV).contiguous()
            pb = posts[k].bias.detach().flatten().to(DEV).contiguous()
            keep += [ls, pb]
            d.post_log_scale, d.post_bias = ls.data_ptr(), pb.data_ptr()
    return d4


@pytest.mark.parametrize("C,ld", [(8, 8), (30, 32), (32, 64), (60, 64), (64, 64)])
def test_macow_unit_forward_backward(C, ld):
    B = 3
    M = B * 64
    o, x, h, shs, posts = _unit(C, ld, B, 100 + C)
    dims = shs[0]["dims"]
    xs = ops.to_state(x.to(DEV))                      # [M][ld]: channels >= C pass through
    cond = ops.cond_prepare(h.to(DEV), DT)
    keep = []
    L = _lib.lib()
    # ---------------- fused forward
    d4 = _descs(C, ld, B, cond, shs, posts, keep)
    ys = [torch.full((M, ld), float("nan"), device=DEV) for _ in range(4)]
    a2 = [torch.zeros(M, dims["K2p"], device=DEV, dtype=tdt(DT)) for _ in range(4)]
    sc = [torch.zeros(M, C, device=DEV) for _ in range(4)]
    slots = torch.zeros(4, B, 4, device=DEV)
    d4[0].x = xs.data_ptr()
    for k in range(4):
        d4[k].y = ys[k].data_ptr(); d4[k].a2_save = a2[k].data_ptr(); d4[k].scale_save = sc[k].data_ptr()
        d4[k].logdet_slot = slots[k].data_ptr()
    # the coupling behind the unit conditions on every other channel ("skip") / on a contiguous half: the unit writes that operand
    zc_cases = []
    for (off, stride, cin) in [(1, 2, C // 2), (C // 2, 1, C - C // 2)]:
        zld = -(-cin // 8) * 8 + 8
        zc_cases.append((off, stride, cin, zld, torch.full((M, zld), 3.0, device=DEV, dtype=tdt(DT))))
    for off, stride, cin, zld, buf in zc_cases:
        d4[3].zc_out, d4[3].zc_off, d4[3].zc_stride, d4[3].zc_cin, d4[3].zc_ld = buf.data_ptr(), off, stride, cin, zld
        _lib.check(L.ipoke_macow_unit_fwd(d4, _lib.DTYPES[DT], _lib.current_stream()))
    torch.cuda.synchronize()
    for off, stride, cin, zld, buf in zc_cases:          # bit-identical to the rounded output state, zero padding
        want = ys[3][:, off:off + (cin - 1) * stride + 1:stride].to(tdt(DT))
        assert torch.equal(buf[:, :cin], want), (off, stride)
        assert float(buf[:, cin:].float().abs().max()) == 0.0
    # ---------------- per-layer chain (the ActNorms as their own launches)
    cur = xs
    ref_states, ref_a2, ref_sc, ref_ld = [], [], [], []
    for k in range(4):
        y = torch.empty_like(cur)
        a2k = torch.zeros(M, dims["K2p"], device=DEV, dtype=tdt(DT)); sck = torch.zeros(M, C, device=DEV)
        ldk = torch.zeros(B, 4, device=DEV)
        d = ops.mcf_desc(cur, C, B, cond, shs[k]["W1"], shs[k]["W2"], shs[k]["bias"], k)
        d.y = y.data_ptr(); d.logdet_slot = ldk.data_ptr(); d.rows_per_block = 16
        d.a2_save = a2k.data_ptr(); d.scale_save = sck.data_ptr()
        _lib.check(L.ipoke_mcf_fwd(d, _lib.DTYPES[DT], _lib.current_stream()))
        if posts[k] is not None:
            y = ops.actnorm_fwd(y, 0, C, posts[k].log_scale.detach().flatten().to(DEV), posts[k].bias.detach().flatten().to(DEV))
        ref_states.append(y); ref_a2.append(a2k); ref_sc.append(sck); ref_ld.append(ldk.sum(1))
        cur = y
    torch.cuda.synchronize()
    for k in range(4):
        got, ref = ys[k][:, :C] if k < 3 else ys[k], ref_states[k][:, :C] if k < 3 else ref_states[k]
        # The fused kernels use the hardware exp / log / rcp for the fp32 transforms and exp(x) - 1 for the (bf16-rounded)
        # ELU; the per-layer kernels use libm-grade tanhf / logf / expm1f.  Differences: ~1e-6 relative in the scales, an
        # occasional bf16 ulp in a hidden activation (4e-3 relative), which moves (mu, s) of that position by ~1e-3.
        e = (got - ref).abs().max().item()
        assert e <= 1e-2, (k, e)
        assert (a2[k].float() - ref_a2[k].float()).abs().max().item() <= 3e-2, k
        assert (sc[k] - ref_sc[k]).abs().max().item() <= 5e-3, k
        assert (slots[k].sum(1) - ref_ld[k]).abs().max().item() <= 5e-2, k
    # ---------------- oracle (fp32 CPU): output and log-det
    xo = x[:, :C].clone().requires_grad_(True)
    yo, ldo = o(xo, h=h)
    got_y = ops.from_state(ys[3], B, ld).cpu()
    e_y = (got_y[:, :C] - yo.detach()).abs().max().item()
    e_pass = (got_y[:, C:] - x[:, C:]).abs().max().item() if ld > C else 0.0
    const_ld = 64.0 * (o.actnorm1.log_scale.sum() + o.actnorm2.log_scale.sum()).item()
    e_ld = (slots.sum(dim=(0, 2)).cpu() + const_ld - ldo.detach()).abs().max().item()
    print(f"unit C={C} ld={ld}: y err {e_y:.3e} pass-through err {e_pass:.1e} logdet err {e_ld:.3e}")
    assert e_y <= 8e-2 and e_pass == 0.0 and e_ld <= 0.5
    # ---------------- fused inverse (sampling direction): reconstructs the input; same result as the oracle's reverse of y
    i4 = _descs(C, ld, B, cond, shs, posts, keep)
    xrec = torch.full((M, ld), float("nan"), device=DEV)
    i4[3].x = ys[3].data_ptr(); i4[0].y = xrec.data_ptr(); i4[0].x = ys[3].data_ptr(); i4[3].y = xrec.data_ptr()
    _lib.check(L.ipoke_macow_unit_inv(i4, _lib.DTYPES[DT], _lib.current_stream()))
    torch.cuda.synchronize()
    got_x = ops.from_state(xrec, B, ld).cpu()
    e_rt = (got_x[:, :C] - x[:, :C]).abs().max().item()
    with torch.no_grad():
        xo_rev = o(got_y[:, :C], h=h, reverse=True)
    e_or = (got_x[:, :C] - xo_rev).abs().max().item()
    e_pt = (got_x[:, C:] - got_y[:, C:]).abs().max().item() if ld > C else 0.0
    print(f"unit C={C} ld={ld}: inverse round trip {e_rt:.3e}, vs oracle reverse of the same y {e_or:.3e}")
    assert torch.isfinite(got_x).all() and e_rt <= 5e-2 and e_or <= 5e-2 and e_pt == 0.0
    # ---------------- fused backward of 0.5*sum(y^2) - sum(logdet) : dy = y on the active channels
    gen = torch.Generator().manual_seed(7)
    dy = ys[3].clone()
    if ld > C:
        dy[:, C:] = torch.randn(M, ld - C, generator=gen).to(DEV)
    dld = torch.full((B,), -1.0, device=DEV)
    dx = torch.full((M, ld), float("nan"), device=DEV)
    dprm = [torch.full((M, dims["K3p"]), float("nan"), device=DEV, dtype=tdt(DT)) for _ in range(4)]
    dc = [torch.full((M, dims["Hq"]), float("nan"), device=DEV, dtype=tdt(DT)) for _ in range(4)]
    dbp = [torch.zeros(B, 2 * C, device=DEV) for _ in range(4)]
    ppart = [torch.zeros(B, 2 * C, device=DEV) for _ in range(4)]
    b4 = _descs(C, ld, B, cond, shs, posts, keep)
    ins = [xs, ys[0], ys[1], ys[2]]
    for k in range(4):
        b4[k].x = ins[k].data_ptr(); b4[k].a2_save = a2[k].data_ptr(); b4[k].scale_save = sc[k].data_ptr()
        b4[k].dparams_save = dprm[k].data_ptr(); b4[k].dc_save = dc[k].data_ptr(); b4[k].dbias_part = dbp[k].data_ptr()
        if posts[k] is not None:
            b4[k].y_post = ys[k].data_ptr(); b4[k].post_part = ppart[k].data_ptr()
    b4[3].dy = dy.data_ptr(); b4[0].dx = dx.data_ptr(); b4[0].dld = dld.data_ptr()
    _lib.check(L.ipoke_macow_unit_bwd(b4, _lib.DTYPES[DT], _lib.current_stream()))
    torch.cuda.synchronize()
    # per-layer backward chain on the same saved tensors
    g = dy
    for k in (3, 2, 1, 0):
        if posts[k] is not None:
            g, dls_ref, db_ref = ops.actnorm_bwd(g, _pre_actnorm(ys[k], posts[k], C), 0, C,
                                                 posts[k].log_scale.detach().flatten().to(DEV), None, dld, B)
            got = ppart[k].sum(0)
            s = max(dls_ref.abs().max().item(), 1.0)
            assert (got[:C] - dls_ref).abs().max().item() <= 2e-4 * s and (got[C:] - db_ref).abs().max().item() <= 2e-4 * s, k
        gx = torch.empty_like(g)
        rp = torch.zeros(M, dims["K3p"], device=DEV, dtype=tdt(DT)); rc = torch.zeros(M, dims["Hq"], device=DEV, dtype=tdt(DT))
        rb = torch.zeros(B, 2 * C, device=DEV)
        d = ops.mcf_desc(ins[k], C, B, cond, shs[k]["W1"], shs[k]["W2"], shs[k]["bias"], k)
        d.y = gx.data_ptr(); d.dy = g.data_ptr(); d.dx = gx.data_ptr(); d.dld = dld.data_ptr()
        d.W2T = shs[k]["W2T"].data_ptr(); d.W1T = shs[k]["W1T"].data_ptr()
        d.a2_save = a2[k].data_ptr(); d.scale_save = sc[k].data_ptr()
        d.dparams_save = rp.data_ptr(); d.dc_save = rc.data_ptr(); d.dbias_part = rb.data_ptr()
        _lib.check(L.ipoke_mcf_bwd(d, _lib.DTYPES[DT], _lib.current_stream()))
        torch.cuda.synchronize()
        for name, got, ref in (("dparams", dprm[k], rp), ("dc", dc[k], rc)):
            assert torch.isfinite(got.float()).all(), (k, name)
            e = (got.float() - ref.float()).abs().max().item() / max(ref.float().abs().max().item(), 1e-6)
            assert e <= 2e-2, (k, name, e)
        e = (dbp[k].sum(0) - rb.sum(0)).abs().max().item() / max(rb.sum(0).abs().max().item(), 1e-6)
        assert e <= 1e-3, (k, "dbias", e)
        g = gx
    e_dx = (dx - g).abs().max().item() / g.abs().max().item()
    assert torch.isfinite(dx).all() and e_dx <= 2e-2, e_dx
    # oracle autograd: gradient with respect to the unit's input
    (0.5 * (yo ** 2).sum() - ldo.sum()).backward()
    got_dx = ops.from_state(dx, B, ld).cpu()
    e_o = (got_dx[:, :C] - xo.grad).abs().max().item() / xo.grad.abs().max().item()
    e_p = (got_dx[:, C:] - ops.from_state(dy, B, ld).cpu()[:, C:]).abs().max().item() if ld > C else 0.0
    print(f"unit C={C} ld={ld}: dx vs per-layer chain {e_dx:.3e}, vs oracle autograd {e_o:.3e}")
    assert e_o <= 5e-2 and e_p == 0.0


def _pre_actnorm(y_post, an, C):
    """input of an ActNorm from its output (the per-layer backward wants the saved input)"""
    ls = an.log_scale.detach().flatten().to(y_post.device)
    b = an.bias.detach().flatten().to(y_post.device)
    x = y_post.clone()
    x[:, :C] = (y_post[:, :C] - b) / ls.exp()
    return x


# ---------------------------------------------------------------------------------------------------------------------------
# Row-split launches (csrc/mcf_unit_split.hip): a sample's 8x8 latent on 2 / 4 workgroups that hand the halo rows of every layer
# to each other inside the launch.  The arithmetic per row is unchanged, so every state, save and data gradient must be
# BIT-IDENTICAL to the one-workgroup launch; log-dets and parameter-gradient partials are summed per part.
def _split_buffers(C, ld, B, S, dims):
    M = B * 64
    o = dict(ys=[torch.full((M, ld), float("nan"), device=DEV) for _ in range(4)],
             a2=[torch.zeros(M, dims["K2p"], device=DEV, dtype=tdt(DT)) for _ in range(4)],
             sc=[torch.zeros(M, C, device=DEV) for _ in range(4)],
             dps=[torch.zeros(M, dims["K3p"], device=DEV, dtype=tdt(DT)) for _ in range(4)],
             dcs=[torch.zeros(M, dims["Hq"], device=DEV, dtype=tdt(DT)) for _ in range(4)],
             xop=[torch.zeros(M, dims["Cp"], device=DEV, dtype=tdt(DT)) for _ in range(4)],
             dbp=[torch.zeros(B * S, 2 * C, device=DEV) for _ in range(4)],
             pp=[torch.zeros(B * S, 2 * C, device=DEV) for _ in range(4)],
             slot=torch.zeros(4, B, 4, device=DEV), dx=torch.full((M, ld), float("nan"), device=DEV))
    nb = _lib.lib().ipoke_macow_unit_xchg_bytes(B, S)
    o["xchg"] = torch.zeros(max(nb, 8) // 4, dtype=torch.int32, device=DEV)
    return o


def _split_descs(C, ld, B, S, xs, cond, shs, posts, keep, o, saved, dy, dld):
    d4 = _descs(C, ld, B, cond, shs, posts, keep)
    ins = [xs, saved["ys"][0], saved["ys"][1], saved["ys"][2]]
    for k in range(4):
        d = d4[k]
        d.x = ins[k].data_ptr(); d.y = o["ys"][k].data_ptr()
        d.a2_save = saved["a2"][k].data_ptr(); d.scale_save = saved["sc"][k].data_ptr(); d.logdet_slot = o["slot"][k].data_ptr()
        d.dparams_save = o["dps"][k].data_ptr(); d.dc_save = o["dcs"][k].data_ptr(); d.dbias_part = o["dbp"][k].data_ptr()
        d.x_op_save = o["xop"][k].data_ptr()
        if posts[k] is not None:
            d.y_post = saved["ys"][k].data_ptr(); d.post_part = o["pp"][k].data_ptr()
    d4[3].dy = dy.data_ptr(); d4[0].dx = o["dx"].data_ptr(); d4[0].dld = dld.data_ptr()
    if S > 1:
        d4[0].split = S; d4[0].xchg = o["xchg"].data_ptr()
    return d4


# (64, 64, 100): 400 workgroups at S = 4 on 256 CUs -- a grid that is not resident at once must not dead-lock (a workgroup only waits
# for its neighbours, whose block ids are adjacent) nor time out
@pytest.mark.parametrize("C,ld,B", [(64, 64, 5), (60, 64, 3), (32, 64, 4), (30, 32, 3), (8, 8, 2), (64, 64, 20), (64, 64, 100)])
def test_macow_unit_row_split_is_bit_identical(C, ld, B):
    o_, x, h, shs, posts = _unit(C, ld, B, 300 + C)
    dims = shs[0]["dims"]
    xs = ops.to_state(x.to(DEV))
    cond = ops.cond_prepare(h.to(DEV), DT)
    gen = torch.Generator(device=DEV).manual_seed(11)
    dy = torch.randn(B * 64, ld, device=DEV, generator=gen)
    dld = torch.randn(B, device=DEV, generator=gen)
    keep = []
    L = _lib.lib()
    ref = _split_buffers(C, ld, B, 1, dims)
    d = _split_descs(C, ld, B, 1, xs, cond, shs, posts, keep, ref, ref, dy, dld)
    _lib.check(L.ipoke_macow_unit_fwd(d, _lib.DTYPES[DT], _lib.current_stream()))
    _lib.check(L.ipoke_macow_unit_bwd(d, _lib.DTYPES[DT], _lib.current_stream()))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(128 << 20, dtype=torch.uint8, device=DEV); big2 = torch.empty_like(big)
    for S in (2, 4):
        for rep in range(4):
            o = _split_buffers(C, ld, B, S, dims)
            if rep % 2:                                   # uneven load beside the hand-offs (a copy stream on the other CUs)
                with torch.cuda.stream(side):
                    for _ in range(3):
                        big2.copy_(big)
            d = _split_descs(C, ld, B, S, xs, cond, shs, posts, keep, o, o, dy, dld)
            _lib.check(L.ipoke_macow_unit_fwd(d, _lib.DTYPES[DT], _lib.current_stream()))
            # the backward call reads the saves of the SPLIT forward (they are bit-identical, checked below)
            _lib.check(L.ipoke_macow_unit_bwd(d, _lib.DTYPES[DT], _lib.current_stream()))
            torch.cuda.synchronize()
            for nme in ("ys", "a2", "sc", "dps", "dcs", "xop"):
                for k in range(4):
                    cols = C if nme == "ys" and k < 3 else None      # pass-through columns of intermediate states are not written
                    assert torch.equal(o[nme][k][:, :cols], ref[nme][k][:, :cols]), (S, rep, nme, k)
            assert torch.equal(o["dx"], ref["dx"]), (S, rep)
            assert (o["slot"].sum(2) - ref["slot"].sum(2)).abs().max().item() <= 1e-3, (S, rep)
            assert (o["slot"][:, :, S:] == 0).all()
            for k in range(4):
                for nme in ("dbp", "pp"):
                    r = ref[nme][k].sum(0)
                    assert ((o[nme][k].sum(0) - r).abs().max() / r.abs().max().clamp_min(1e-6)).item() <= 2e-6, (S, rep, nme, k)
            assert int(o["xchg"][0].item()) == 0, "hand-off time-outs"
            assert int((o["xchg"][64:] != 0).sum().item()) == 0, "the exchange scratch must be all-zero again after a launch"


def test_macow_unit_row_split_rejects_bad_arguments():
    C, ld, B = 8, 8, 2
    o_, x, h, shs, posts = _unit(C, ld, B, 5)
    xs = ops.to_state(x.to(DEV)); cond = ops.cond_prepare(h.to(DEV), DT)
    keep = []
    o = _split_buffers(C, ld, B, 2, shs[0]["dims"])
    dy = torch.zeros(B * 64, ld, device=DEV); dld = torch.zeros(B, device=DEV)
    d = _split_descs(C, ld, B, 2, xs, cond, shs, posts, keep, o, o, dy, dld)
    d[0].xchg = None
    assert _lib.lib().ipoke_macow_unit_fwd(d, _lib.DTYPES[DT], _lib.current_stream()) != 0        # no scratch
    d[0].xchg = o["xchg"].data_ptr(); d[0].split = 3
    assert _lib.lib().ipoke_macow_unit_fwd(d, _lib.DTYPES[DT], _lib.current_stream()) != 0        # 2 or 4 only


@pytest.mark.parametrize("C,ld,B,Cp,t_off,t_stride,with_ls,with_idx", [(64, 64, 5, 32, 0, 2, True, True), (64, 64, 20, 32, 1, 2, True, False),
                                                                        (64, 64, 3, 16, 32, 1, False, True), (64, 64, 4, 8, 3, 4, True, True),
                                                                        (44, 64, 3, 22, 0, 2, True, True), (30, 32, 4, 15, 15, 1, True, True),
                                                                        (8, 64, 2, 4, 1, 2, True, False)])
def test_macow_unit_bwd_differentiates_the_pair_in_front(C, ld, B, Cp, t_off, t_stride, with_ls, with_idx):
    """ipoke_mcf_desc.pair: the ActNorm (+ shuffle) and the coupling in front of the unit (forward order), differentiated by the unit's
    row-split backward launch, against the unit's launch followed by ipoke_actnorm_affine_bwd (reference: MaCowStep's
    coupling -> actnorm -> unit, macow2.py:1066-1117; ActNorm2dFlow.backward / Affine.backward through autograd)."""
    from ctypes import byref, cast, c_void_p, pointer
    from ipoke_amd._lib import UnitPairDesc
    S = 4
    o_, x, h, shs, posts = _unit(C, ld, B, 900 + Cp)
    dims = shs[0]["dims"]
    xs = ops.to_state(x.to(DEV)); cond = ops.cond_prepare(h.to(DEV), DT)
    gen = torch.Generator(device=DEV).manual_seed(21)
    dy = torch.randn(B * 64, ld, device=DEV, generator=gen)
    dld = torch.randn(B, device=DEV, generator=gen)
    M = B * 64
    an_ls = (torch.randn(C, device=DEV, generator=gen) * 0.2) if with_ls else None
    an_idx = torch.randperm(C, device=DEV, generator=gen).to(torch.int32) if with_idx else None
    x1 = torch.randn(M, ld, device=DEV, generator=gen)          # the ActNorm's saved input
    x0 = torch.randn(M, ld, device=DEV, generator=gen)          # the coupling's saved input
    scale = torch.rand(M, Cp, device=DEV, generator=gen) * 1.5 + 0.2
    ldp = 2 * Cp + 8
    keep = []
    L = _lib.lib()
    # --- two launches
    o = _split_buffers(C, ld, B, S, dims)
    d = _split_descs(C, ld, B, S, xs, cond, shs, posts, keep, o, o, dy, dld)
    _lib.check(L.ipoke_macow_unit_fwd(d, _lib.DTYPES[DT], _lib.current_stream()))
    _lib.check(L.ipoke_macow_unit_bwd(d, _lib.DTYPES[DT], _lib.current_stream()))
    ref_dx = torch.full((M, ld), float("nan"), device=DEV)
    ref_dp = torch.full((M, ldp), float("nan"), device=DEV).to(torch.bfloat16)
    ref_part = torch.zeros(B, 2 * C, device=DEV); ref_db = torch.zeros(B, 2 * Cp, device=DEV)
    _lib.check(L.ipoke_actnorm_affine_bwd(0, C, _lib.ptr(an_ls), _lib.ptr(an_idx), o["dx"].data_ptr(), x1.data_ptr(),
                                          ref_part.data_ptr() if with_ls else None, Cp, t_off, t_stride, 64, ld, x0.data_ptr(),
                                          scale.data_ptr(), dld.data_ptr(), ref_dx.data_ptr(), ref_dp.data_ptr(), ldp, ref_db.data_ptr(), B,
                                          _lib.DTYPES[DT], _lib.current_stream()))
    # --- one launch
    o2 = _split_buffers(C, ld, B, S, dims)
    d2 = _split_descs(C, ld, B, S, xs, cond, shs, posts, keep, o2, o, dy, dld)          # (saves of the first forward pass)
    got_dx = torch.full((M, ld), float("nan"), device=DEV)
    got_dp = torch.full((M, ldp), float("nan"), device=DEV).to(torch.bfloat16)
    got_part = torch.zeros(B * S, 2 * C, device=DEV); got_db = torch.zeros(B * S, 2 * Cp, device=DEV)
    q = UnitPairDesc()
    q.an_log_scale = _lib.ptr(an_ls); q.an_idx = _lib.ptr(an_idx); q.an_x = x1.data_ptr(); q.an_part = got_part.data_ptr() if with_ls else None
    q.Cp, q.t_off, q.t_stride = Cp, t_off, t_stride
    q.x0 = x0.data_ptr(); q.scale = scale.data_ptr(); q.dparams = got_dp.data_ptr(); q.ldp = ldp; q.dbias_part = got_db.data_ptr()
    q.dx = got_dx.data_ptr(); d2[0].dx = got_dx.data_ptr()          # one buffer: columns >= C pass through the launch's own dx
    d2[0].pair = cast(pointer(q), c_void_p).value
    _lib.check(L.ipoke_macow_unit_bwd(d2, _lib.DTYPES[DT], _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(got_dx, ref_dx)
    assert torch.equal(got_dp.float(), ref_dp.float())
    for got, ref in ((got_db, ref_db),) + (((got_part, ref_part),) if with_ls else ()):
        g = got.view(B, S, -1).sum(1)
        assert ((g - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item() <= 5e-6
    for k in range(4):                                    # the unit's own outputs are untouched by the extra work
        for nme in ("dps", "dcs", "dbp", "pp"):
            assert torch.equal(o2[nme][k], o[nme][k]), (nme, k)
    # rejected: without the row split, or on a unit that does not cover the whole state
    d2[0].split = 1
    assert L.ipoke_macow_unit_bwd(d2, _lib.DTYPES[DT], _lib.current_stream()) != 0
