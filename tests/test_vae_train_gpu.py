"""First-stage training helpers (vae_train.hip) against plain PyTorch fp32 references of the same ops: the weight operand
writer, spectral-norm power iteration / sigma / backward (torch.nn.utils.spectral_norm semantics, reference
models/modules/autoencoders/util.py:52,252), multi-tensor Adam (first_stage_motion_model.py:283-300) and the KL term
(utils/losses.py:47-48)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, nn as K, ops
from ipoke_amd._lib import check, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape,transposed", [((40, 12, 3, 3), False), ((12, 40, 3, 3), True), ((16, 3, 3, 7, 7), False), ((64, 64, 1, 1), False)])
def test_weight_operand_matches_the_torch_chain(shape, transposed, dtype):
    from ipoke_amd.first_stage_train import _weight_operand
    g = torch.Generator().manual_seed(1)
    w = torch.randn(*shape, generator=g).to(DEV)
    inv = torch.tensor([0.37], device=DEV)
    w5 = w.unsqueeze(2) if w.dim() == 4 else w
    for scale in (None, inv):
        want, kc0 = K.weight_operand(w5, dtype, transposed, scale=None if scale is None else scale)
        got, kc = _weight_operand(w, dtype, transposed, scale)
        torch.cuda.synchronize()
        assert kc == kc0 and got.shape == want.shape and got.dtype == want.dtype
        assert torch.equal(got, want)


def _sn_case(cout, cin, k, transposed, seed):
    g = torch.Generator().manual_seed(seed)
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    w = torch.randn(*shape, generator=g).to(DEV)
    u = F.normalize(torch.randn(cout, generator=g), dim=0).to(DEV)
    v = F.normalize(torch.randn(cin * k * k, generator=g), dim=0).to(DEV)
    return w, u, v


@pytest.mark.parametrize("cout,cin,k,transposed", [(64, 32, 3, False), (48, 96, 3, True), (512, 512, 3, False), (3, 64, 7, False)])
@pytest.mark.parametrize("iterate", [1, 0])
def test_spectral_sigma_and_backward(cout, cin, k, transposed, iterate):
    w, u, v = _sn_case(cout, cin, k, transposed, cout + cin)
    wm = (w.transpose(0, 1) if transposed else w).reshape(cout, -1)
    u0, v0 = u.clone(), v.clone()
    if iterate:
        v0 = F.normalize(torch.mv(wm.t(), u0), dim=0, eps=1e-12)
        u0 = F.normalize(torch.mv(wm, v0), dim=0, eps=1e-12)
    wr = w.clone().requires_grad_(True)
    wmr = (wr.transpose(0, 1) if transposed else wr).reshape(cout, -1)
    sigma = torch.dot(u0, torch.mv(wmr, v0))
    w_eff = wr / sigma
    G = torch.randn(w.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    (w_eff * G).sum().backward()

    lib = _lib.lib()
    taps = k * k
    ws = torch.zeros(int(lib.ipoke_spectral_workspace_floats(cout, cin, taps)), device=DEV)
    sig = torch.empty(2, device=DEV); snap = torch.empty(cout + cin * taps, device=DEV)
    for rep in range(2):                                   # second call: the workspace accumulators were left clean
        uu, vv = u.clone(), v.clone()
        check(lib.ipoke_spectral_sigma(ptr(w), cout, cin, taps, int(transposed), ptr(uu), ptr(vv), iterate, 1e-12, ptr(sig), ptr(snap), ptr(ws),
                                       _lib.current_stream()))
        torch.cuda.synchronize()
        assert (uu - u0).abs().max() <= 2e-6 and (vv - v0).abs().max() <= 2e-6
        assert abs(sig[0].item() - sigma.item()) <= 2e-5 * abs(sigma.item())
        assert abs(sig[1].item() * sig[0].item() - 1.0) <= 1e-6
        assert torch.equal(snap[:cout], uu) and torch.equal(snap[cout:], vv)
    bws = torch.zeros(int(lib.ipoke_spectral_bwd_workspace_floats()), device=DEV)
    for rep in range(2):
        d = G.clone()
        check(lib.ipoke_spectral_bwd(ptr(w), cout, cin, taps, int(transposed), ptr(d), ptr(snap), ptr(sig), ptr(bws), _lib.current_stream()))
        torch.cuda.synchronize()
        err = (d - wr.grad).abs().max().item()
        assert err <= 1e-4 * max(1.0, wr.grad.abs().max().item()), err


def test_adam_multi_matches_torch_adam():
    g = torch.Generator().manual_seed(5)
    shapes = [(7,), (64, 3, 3, 3), (1, 5), (128, 64, 3, 3)] * 15                    # 60 tensors: two launches
    ps = [torch.randn(*s, generator=g).to(DEV) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.9), weight_decay=1e-5)
    m = [torch.zeros_like(p) for p in ps]; v = [torch.zeros_like(p) for p in ps]
    n = len(ps)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
    for step in range(1, 4):
        grads = [torch.randn(*s, generator=g).to(DEV) for s in shapes]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
        check(_lib.lib().ipoke_adam_multi(arr(ps), arr(grads), arr(m), arr(v), sizes, n, 2e-4, 0.5, 0.9, 1e-8, 1e-5, step, 1.0,
                                          _lib.current_stream()))
        torch.cuda.synchronize()
        worst = max((a - b.detach()).abs().max().item() for a, b in zip(ps, ref))
        assert worst <= 1e-6, (step, worst)                # 1-2 ulp of parameters of magnitude <= 4 (ulp 2.4e-7 .. 4.8e-7)


def test_kl_value_and_gradient():
    g = torch.Generator().manual_seed(9)
    B, Z, H = 3, 32, 8
    mu4 = torch.randn(B, Z, H, H, generator=g).to(DEV).requires_grad_(True)
    lv4 = (0.3 * torch.randn(B, Z, H, H, generator=g)).to(DEV).requires_grad_(True)
    kl = -0.5 * torch.mean(torch.sum(1 + lv4 - mu4.pow(2) - lv4.exp(), dim=1))
    kl.backward()
    mu = mu4.detach().permute(0, 2, 3, 1).reshape(-1, Z).contiguous(); lv = lv4.detach().permute(0, 2, 3, 1).reshape(-1, Z).contiguous()
    loss = torch.zeros(1, device=DEV); dmu = torch.empty_like(mu); dlv = torch.empty_like(lv)
    check(_lib.lib().ipoke_kl_loss(ptr(mu), ptr(lv), mu.shape[0], Z, ptr(loss), ptr(dmu), ptr(dlv), _lib.current_stream()))
    torch.cuda.synchronize()
    assert abs(loss.item() - kl.item()) <= 1e-5 * max(1.0, abs(kl.item()))
    assert (dmu - mu4.grad.permute(0, 2, 3, 1).reshape(-1, Z)).abs().max() <= 1e-7
    assert (dlv - lv4.grad.permute(0, 2, 3, 1).reshape(-1, Z)).abs().max() <= 1e-7


def test_weight_operand_multi_matches_single_calls_and_the_refreshed_cache():
    """ipoke_conv_weight_operand_multi (every cached operand rebuilt in ONE launch behind the optimizer step) against one
    ipoke_conv_weight_operand call per weight -- conv and ConvTranspose storage, 2-D / 3-D taps, channel padding, more weights than one
    launch takes (64) -- and the trainer-side cache: after ``take_operand_cache`` / raw-pointer update / ``refresh_operand_cache`` the
    cache hands out the SAME tensors holding the operands of the NEW values."""
    import ctypes as ct
    from ipoke_amd import first_stage_train as FT
    lib = _lib.lib()
    gen = torch.Generator().manual_seed(3)
    shapes = [((64, 3, 3, 3), 0), ((128, 64, 3, 3), 0), ((64, 128, 3, 3), 1), ((32, 20, 1, 1), 0), ((16, 8, 3, 3, 3), 0)] * 14     # 70 weights
    ws = [torch.randn(*sh, generator=gen).to(DEV) for sh, _ in shapes]
    for dt in ("bf16", "f32"):
        e16 = 8 if dt == "bf16" else 4
        singles, outs, dims = [], [], []
        for w, (sh, tr) in zip(ws, shapes):
            taps = 1
            for kk in sh[2:]:
                taps *= kk
            rows, cols = (sh[1], sh[0]) if tr else (sh[0], sh[1])
            kc = -(-cols // e16) * e16
            ref = torch.empty(rows, taps * kc, device=DEV, dtype=torch.bfloat16 if dt == "bf16" else torch.float32)
            check(lib.ipoke_conv_weight_operand(ptr(w), rows, cols, taps, tr, None, ptr(ref), kc, _lib.DTYPES[dt], _lib.current_stream()))
            singles.append(ref); outs.append(torch.full_like(ref, 7.0)); dims += [rows, cols, taps, tr, kc]
        n = len(ws)
        check(lib.ipoke_conv_weight_operand_multi((ct.c_void_p * n)(*[w.data_ptr() for w in ws]), (ct.c_void_p * n)(*[o.data_ptr() for o in outs]),
                                                  (ct.c_int32 * (5 * n))(*dims), n, _lib.DTYPES[dt], _lib.current_stream()))
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(singles, outs)), dt
    # the cache: operands of parameters are refreshed in place, derived (scoped) entries are dropped
    FT.clear_operand_cache()
    p1 = torch.nn.Parameter(ws[1].clone()); p2 = torch.nn.Parameter(ws[2].clone())
    o1, kc1 = FT._weight_operand(p1.detach(), "bf16", False, cacheable=True, owner=p1)
    o2, kc2 = FT._weight_operand(p2.detach(), "bf16", True, cacheable=True, owner=p2)
    od, _ = FT._weight_operand(ws[0], "bf16", False, cacheable=True, scope=5, owner=ws[0])
    snap = FT.take_operand_cache()
    assert len(snap) == 3 and not FT._OPCACHE
    v_before = p1._version
    p1.data.mul_(2.0)                                               # as the fused optimizer does: through the storage, no version bump
    assert p1._version == v_before
    FT.refresh_operand_cache(snap)
    torch.cuda.synchronize()
    assert len(FT._OPCACHE) == 2                                    # the scoped (derived) entry is gone
    fresh, _ = FT._build_weight_operand(p1.detach(), "bf16", False)
    hit, _ = FT._weight_operand(p1.detach(), "bf16", False, cacheable=True, owner=p1)
    assert hit is o1 and torch.equal(o1, fresh)                     # the SAME tensor, holding the operand of the new values
    hit2, _ = FT._weight_operand(p2.detach(), "bf16", True, cacheable=True, owner=p2)
    assert hit2 is o2
    FT.clear_operand_cache()
