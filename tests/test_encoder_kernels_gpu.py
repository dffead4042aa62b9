"""GPU parity of the kernels of the frozen-encoder phase: the one-launch GroupNorm of small maps (csrc/norm.hip gn_fused_kernel)
against torch.nn.functional.group_norm, the four-channel clip rewrite (ipoke_clip_to_cl4) and the folded (3, 7, 7) stem of the 3-D
encoder (motion_encoder.py:161) against the same convolution read in place and against torch's conv3d."""
import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, configs, nn as K, ops
from ipoke_amd import first_stage as FS
from ipoke_amd._lib import check, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"

# N, C, H, W, groups (0: InstanceNorm), affine, residual, act -- the shapes of the encoders' 32 x 32 ... 8 x 8 stages and ragged ones
GN_CASES = [
    (20, 256, 16, 16, 16, True, True, "relu"),
    (4, 256, 8, 8, 16, True, True, "relu"),
    (3, 128, 32, 64, 16, True, False, "relu"),       # S = 2048, 8 channels per group
    (2, 64, 32, 32, 16, True, True, "none"),         # 4 channels per group: a slab spans several groups
    (2, 64, 16, 16, 0, False, False, "elu"),         # InstanceNorm
    (2, 48, 5, 7, 16, True, False, "none"),          # 3 channels per group, ragged position count
    (1, 512, 8, 8, 16, True, False, "relu"),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", GN_CASES, ids=[f"n{c[0]}c{c[1]}s{c[2] * c[3]}g{c[4]}" for c in GN_CASES])
def test_groupnorm_small_maps_vs_torch(case, dtype):
    N, C, H, W, G, affine, use_res, act = case
    gen = torch.Generator().manual_seed(C + H)
    x = 2.0 * torch.randn(N, C, H, W, generator=gen) + 0.7          # a mean far from 0: the variance must not cancel
    gamma = 1 + 0.3 * torch.randn(C, generator=gen) if affine else None
    beta = 0.2 * torch.randn(C, generator=gen) if affine else None
    res = torch.randn(N, C, H, W, generator=gen) if use_res else None
    groups = C if G == 0 else G
    xc = K.from_nchw(x.to(DEV), dtype)
    rc = K.from_nchw(res.to(DEV), dtype) if use_res else None
    xr = K.to_nchw(xc, dtype).cpu()                                   # the (bf16-rounded) values the kernel sees
    rr = K.to_nchw(rc, dtype).cpu() if use_res else None
    y = F.group_norm(xr, groups, gamma, beta, eps=1e-5)
    if use_res:
        y = y + rr
    y = {"relu": torch.relu, "elu": F.elu, "none": lambda v: v}[act](y)
    actc = {"relu": _lib.ACT_RELU, "elu": _lib.ACT_ELU, "none": _lib.ACT_NONE}[act]
    out = K.group_norm(xc, groups, dtype, None if gamma is None else gamma.to(DEV), None if beta is None else beta.to(DEV), act=actc, res=rc)
    got = K.to_nchw(out, dtype).cpu()
    err = (got - y).abs().max().item()
    assert err <= (2e-5 if dtype == "f32" else 4e-2) * max(1.0, y.abs().max().item()), err


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(3, 64, 16, 16, 0, "relu"), (2, 64, 96, 96, 0, "relu"), (2, 128, 64, 64, 16, "elu")],
                         ids=["inst_small", "inst_large", "group_large"])
def test_groupnorm_residual_behind_the_activation(case, dtype):
    """ipoke_norm_desc.res_post: y = act(norm(x)) + res -- ResBlock's sum riding on the skip path's norm pass (util.py:106-192); the
    one-launch kernel of small maps and the stats / apply pair of large ones."""
    N, C, H, W, G, act = case
    gen = torch.Generator().manual_seed(C + H)
    x = 2.0 * torch.randn(N, C, H, W, generator=gen) + 0.7
    res = torch.randn(N, C, H, W, generator=gen)
    groups = C if G == 0 else G
    xc, rc = K.from_nchw(x.to(DEV), dtype), K.from_nchw(res.to(DEV), dtype)
    xr, rr = K.to_nchw(xc, dtype).cpu(), K.to_nchw(rc, dtype).cpu()
    y = {"relu": torch.relu, "elu": F.elu}[act](F.group_norm(xr, groups, eps=1e-5)) + rr
    actc = {"relu": _lib.ACT_RELU, "elu": _lib.ACT_ELU}[act]
    got = K.to_nchw(K.group_norm(xc, groups, dtype, act=actc, res=rc, res_post=True), dtype).cpu()
    err = (got - y).abs().max().item()
    assert err <= (2e-5 if dtype == "f32" else 4e-2) * max(1.0, y.abs().max().item()), err
    pre = K.to_nchw(K.group_norm(xc, groups, dtype, act=actc, res=rc), dtype).cpu()          # the other order is a different function
    assert (pre - y).abs().max().item() > 0.1


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(4, 64, 80, 80, 16), (3, 256, 16, 16, 16), (2, 128, 33, 31, 8)], ids=["large", "small_one_launch", "ragged"])
def test_groupnorm_hands_the_statistics_of_its_output_to_the_next_norm(case, dtype):
    """ipoke_norm_desc.next_part / part_chunks: an InstanceNorm pass (with the residual behind its activation) leaves the chunk statistics
    of its output; the SPADE norm that reads the output next skips its own statistics pass.  Same result as the stand-alone sequence up to
    the order of the partial sums."""
    N, C, H, W, G = case
    gen = torch.Generator().manual_seed(C + H)
    x = 2.0 * torch.randn(N, C, H, W, generator=gen) + 0.7
    res = torch.randn(N, C, H, W, generator=gen)
    mg, mb = 0.3 * torch.randn(N, C, H, W, generator=gen), 0.3 * torch.randn(N, C, H, W, generator=gen)
    xc, rc = K.from_nchw(x.to(DEV), dtype), K.from_nchw(res.to(DEV), dtype)
    mod = (K.from_nchw(mg.to(DEV), dtype), K.from_nchw(mb.to(DEV), dtype))
    mid0 = K.group_norm(xc, C, dtype, act=_lib.ACT_RELU, res=rc, res_post=True)
    mid1 = K.group_norm(xc, C, dtype, act=_lib.ACT_RELU, res=rc, res_post=True, next_groups=G)
    assert torch.equal(mid0.t, mid1.t) and mid1.stats_part[0] == G
    out0 = K.to_nchw(K.group_norm(mid0, G, dtype, mod=mod), dtype)
    out1 = K.to_nchw(K.group_norm(mid1, G, dtype, mod=mod), dtype)
    torch.cuda.synchronize()
    err = (out0 - out1).abs().max().item()
    print(f"handed statistics {case}[{dtype}]: max difference {err:.2e} (|out| <= {out0.abs().max().item():.2f})")
    assert err <= (2e-5 if dtype == "f32" else 1.6e-2) * max(1.0, out0.abs().max().item())
    other = K.group_norm(mid1, G // 2, dtype)                 # a norm of another group count ignores the hand-over
    assert torch.isfinite(other.t.float()).all()


def test_clip_to_cl4():
    gen = torch.Generator().manual_seed(3)
    B, T, H, W = 2, 3, 6, 10
    x = torch.randn(B, T + 1, 3, H, W, generator=gen).to(DEV)[:, 1:].transpose(1, 2)       # strided [B, 3, T, H, W] view, as the model passes it
    assert not x.is_contiguous()
    for dtype in ("f32", "bf16"):
        Wp = W + 6
        dst = torch.full((B * T * H * Wp, 4), 7.0, dtype=ops.torch_dtype(dtype), device=DEV)
        check(_lib.lib().ipoke_clip_to_cl4(ptr(x), x.stride(0), x.stride(1), x.stride(2), x.stride(3), x.stride(4), B, T, H, W, 3, 3,
                                           ptr(dst), ops._dt(dtype), _lib.current_stream()))
        want = torch.zeros(B, T, H, Wp, 4, device=DEV)
        want[:, :, :, 3:3 + W, :3] = x.permute(0, 2, 3, 4, 1)
        assert torch.equal(dst.float().view(B, T, H, Wp, 4), want.to(ops.torch_dtype(dtype)).float())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_folded_stem_matches_conv3d(dtype, monkeypatch):
    cfg = configs.first_stage_config(64, 32, 16)["architecture"]
    dic = dict(cfg); dic.update(img_size=64, max_frames=15, full_seq=True)
    torch.manual_seed(11)
    enc = FS.ResNetMotionEncoder(dic, dtype=dtype).to(DEV)
    x = torch.randn(2, 16, 3, 64, 64, device=DEV)[:, 1:].transpose(1, 2)                  # [2, 3, 15, 64, 64], strided
    w = enc.conv1.weight.detach()
    if dtype == "bf16":
        ref = F.conv3d(x.bfloat16().float().cpu(), w.bfloat16().float().cpu(), None, 2, (1, 3, 3))
    else:
        ref = F.conv3d(x.cpu(), w.cpu(), None, 2, (1, 3, 3))
    monkeypatch.setattr(FS, "_STEM_FOLD", True)
    folded = enc._stem(x.float())
    monkeypatch.setattr(FS, "_STEM_FOLD", False)
    plain = enc._stem(x.float())
    assert folded.dhw == plain.dhw == tuple(ref.shape[2:]) and folded.C == plain.C == 64
    a, b = K.to_nchw(folded, dtype).cpu(), K.to_nchw(plain, dtype).cpu()
    r = ref
    assert a.shape == r.shape
    tol = 2e-5 if dtype == "f32" else 2e-2
    scale = r.abs().max().item()
    assert (a - r).abs().max().item() <= tol * scale, ((a - r).abs().max().item(), scale)
    assert (a - b).abs().max().item() <= tol * scale


@pytest.mark.parametrize("N,H,W,cin,cout,snorm", [(3, 32, 32, 256, 128, True), (2, 16, 48, 128, 256, False), (5, 16, 16, 64, 160, True)])
def test_conv_transpose_four_tap_phase_on_the_halo_staged_kernel(N, H, W, cin, cout, snorm, request):
    """The four-tap sub-pixel phase (output pixels (2i + 1, 2j + 1): a 2 x 2 window without padding, scattered rows) of a stride-2
    ConvTranspose2d with a WIDE output runs on conv3x3_halo16 (round 6): the window's offsets 0 / +1 lie inside the staged patch's halo.
    Forced onto that kernel (the dispatch rule wants >= 256 workgroups), against torch and against the one-launch 9-tap form; the other
    three phases of the same call keep their kernels."""
    dtype = "bf16"
    _lib.check(_lib.lib().ipoke_set_dispatch_override(b"halo16", 2))
    request.addfinalizer(lambda: _lib.lib().ipoke_set_dispatch_override(b"halo16", -1))
    torch.manual_seed(H * W + cin)
    mod = FS._Conv(cin, cout, 3, 2, 1, transposed=True, snorm=snorm).to(DEV)
    with torch.no_grad():
        mod.bias.copy_(0.1 * torch.randn(cout))
    x = torch.randn(N, cin, H, W)
    xc = K.from_nchw(x.to(DEV), dtype)
    xr = K.to_nchw(xc, dtype).cpu()
    w = (mod.weight_orig / K.spectral_sigma(mod.weight_orig, mod.weight_u, mod.weight_v, True) if snorm else mod.weight).detach().cpu()
    w = w.bfloat16().float()
    ref = F.elu(F.conv_transpose2d(xr, w, mod.bias.detach().cpu(), stride=2, padding=1, output_padding=1))
    got = mod.run(xc, dtype, act=_lib.ACT_ELU)                 # the (1, 1) phase is the last launch of the call
    assert _lib.lib().ipoke_last_conv_kernel() == _lib.KERNEL_HALO16
    a = K.to_nchw(got, dtype).cpu()
    tol = 2e-2 * max(1.0, ref.abs().max().item())
    assert (a - ref).abs().max().item() <= tol
    assert (a[:, :, 1::2, 1::2] - ref[:, :, 1::2, 1::2]).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,cin,cout,snorm", [(2, 8, 8, 32, 24, False), (3, 16, 32, 64, 64, True), (1, 5, 7, 16, 40, False),
                                                  (2, 16, 48, 128, 64, True), (1, 32, 16, 128, 20, False)])
@pytest.mark.parametrize("c64", [False, True], ids=["dispatch", "c64"])
def test_conv_transpose_phases(N, H, W, cin, cout, snorm, dtype, c64, monkeypatch, request):
    """Stride-2 ConvTranspose2d (util.py:52-55) as four sub-pixel stride-1 convolutions with scattered output rows
    (ipoke_conv_desc.c_scatter) against torch.nn.functional.conv_transpose2d and against the one-launch 9-tap form."""
    # c64: the phases of 64- / 128-channel inputs on 16-aligned maps through conv3x3_c64_kernel (filter resident in LDS)
    _lib.check(_lib.lib().ipoke_set_dispatch_override(b"c64", 2 if c64 else -1))
    request.addfinalizer(lambda: _lib.lib().ipoke_set_dispatch_override(b"c64", -1))
    torch.manual_seed(H * W + cin)
    mod = FS._Conv(cin, cout, 3, 2, 1, transposed=True, snorm=snorm).to(DEV)
    with torch.no_grad():
        mod.bias.copy_(0.1 * torch.randn(cout))
    x = torch.randn(N, cin, H, W)
    xc = K.from_nchw(x.to(DEV), dtype)
    xr = K.to_nchw(xc, dtype).cpu()
    w = (mod.weight_orig / K.spectral_sigma(mod.weight_orig, mod.weight_u, mod.weight_v, True) if snorm else mod.weight).detach().cpu()
    if dtype == "bf16":
        w = w.bfloat16().float()
    ref = F.elu(F.conv_transpose2d(xr, w, mod.bias.detach().cpu(), stride=2, padding=1, output_padding=1))
    monkeypatch.setattr(FS, "_CT_PHASES", True)
    got = mod.run(xc, dtype, act=_lib.ACT_ELU)
    monkeypatch.setattr(FS, "_CT_PHASES", False)
    one = mod.run(xc, dtype, act=_lib.ACT_ELU)
    assert got.dhw == one.dhw == (1, 2 * H, 2 * W)
    a, b = K.to_nchw(got, dtype).cpu(), K.to_nchw(one, dtype).cpu()
    tol = (2e-5 if dtype == "f32" else 2e-2) * max(1.0, ref.abs().max().item())
    assert (a - ref).abs().max().item() <= tol
    assert (a - b).abs().max().item() <= tol
    if got.t.shape[1] > cout:                                  # padded columns are written as zeros by every phase
        assert float(got.t[:, cout:].float().abs().max()) == 0.0
