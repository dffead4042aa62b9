"""Channels-last building blocks for the first-stage VAE, executed by the HIP kernels of libipoke_hip.

Activations are ``CL`` records: a dense ``[N*D*H*W, C]`` tensor of the compute dtype (bf16 or f32) plus its
logical extents.  Weights stay fp32 ``nn.Parameter``s in PyTorch layout under the reference's state-dict
names; their matrix-core operand form (``[Cout][tap*Kc + c]``, K contiguous, zero padded, spectral norm
folded) is derived once per weight version with torch ops and cached.
"""
from ctypes import byref
from dataclasses import dataclass

import torch

from . import _lib, ops
from ._lib import NormDesc, check, ptr


@dataclass
class CL:
    t: torch.Tensor          # [N*D*H*W, C]
    N: int
    dhw: tuple
    C: int

    @property
    def M(self):
        return self.t.shape[0]

    @property
    def S(self):
        return self.dhw[0] * self.dhw[1] * self.dhw[2]


def e16(dtype):
    return 8 if ops._dt(dtype) == _lib.BF16 else 4


def round_up(a, b):
    return -(-a // b) * b


def weight_operand(w, dtype, transposed_conv=False, scale=None):
    """Conv weight (PyTorch layout) -> [Cout][taps*Kc] operand of the compute dtype."""
    if transposed_conv:
        w = w.transpose(0, 1)
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    kc = round_up(cin, e16(dtype))
    w3 = w.reshape(cout, cin, taps).permute(0, 2, 1).float()
    if scale is not None:
        w3 = w3 * scale
    buf = torch.zeros(cout, taps, kc, dtype=torch.float32, device=w.device)
    buf[:, :, :cin] = w3
    return buf.reshape(cout, taps * kc).to(ops.torch_dtype(dtype)).contiguous(), kc


def spectral_sigma(weight_orig, u, v, transposed_conv=False):
    """sigma of torch.nn.utils.spectral_norm in eval mode: u^T W v with W = weight_orig flattened over dim 0
    (dim 1 for ConvTranspose)."""
    w = weight_orig.transpose(0, 1) if transposed_conv else weight_orig
    return torch.dot(u, torch.mv(w.reshape(w.shape[0], -1), v))


def out_extent(i, k, s, p, transposed, out_pad=0):
    if transposed:
        return (i - 1) * s - 2 * p + k + out_pad
    return (i + 2 * p - k) // s + 1


def conv(x, w_op, kc, cout, k, stride, pad, dtype, bias=None, act=_lib.ACT_NONE, transposed=False, out_pad=(0, 0, 0),
         out_f32=False, src_f32=None, out=None, odhw=None, scatter=None, row_scale=None):
    """Implicit-GEMM convolution.  ``x`` is a CL, or ``src_f32`` = (tensor, N, C, dhw, strides(n,c,d,h,w)) for an
    fp32 source read in place (input images).  ``odhw``: output extent when it is not the symmetric-padding formula's (windows
    that overhang the input read zeros).  ``scatter`` = (c_sn, c_sh, c_sw, c_row0): rows of ``out`` the positions are written to
    (ipoke_conv_desc.c_scatter); the returned CL then describes this launch's positions only.  ``row_scale`` = (fp32 tensor, rs_images,
    rs_stride): the accumulators of image n are multiplied by tensor[(n // rs_images) * rs_stride] before the bias
    (ipoke_conv_desc.row_scale: 1 / sigma_t of the frame an image belongs to)."""
    if src_f32 is not None:
        src, N, cin, dhw, st = src_f32
    else:
        N, cin, dhw = x.N, x.C, x.dhw
    if odhw is None:
        odhw = tuple(out_extent(i, kk, s, p, transposed, op) for i, kk, s, p, op in zip(dhw, k, stride, pad, out_pad))
    M = N * odhw[0] * odhw[1] * odhw[2]
    d = ops.conv_desc(N, dhw, odhw, k, stride, pad, transposed)
    if src_f32 is not None:
        d.A = src.data_ptr(); d.a_f32 = 1
        d.a_sn, d.a_sc, d.a_sd, d.a_sh, d.a_sw = st
        d.a_coff = 0; d.Kc_real = cin; d.Kc = kc
    else:
        ld = x.t.shape[1]
        d.A = x.t.data_ptr(); d.a_f32 = 0
        d.a_sn = dhw[0] * dhw[1] * dhw[2] * ld; d.a_sd = dhw[1] * dhw[2] * ld; d.a_sh = dhw[2] * ld; d.a_sw = ld; d.a_sc = 1
        d.Kc_real = kc; d.Kc = kc
        assert ld >= kc, "activation pitch must cover the padded channel count"
    d.W = w_op.data_ptr(); d.ldw = w_op.stride(0); d.Nout = cout      # (a column range of a wider operand keeps its pitch)
    d.bias = 0 if bias is None else bias.data_ptr(); d.act = act
    if out_f32:
        y = torch.empty(M, cout, dtype=torch.float32, device=w_op.device) if out is None else out
        d.c_f32 = 1
    else:
        ldc = round_up(cout, e16(dtype))
        y = torch.empty(M, ldc, dtype=ops.torch_dtype(dtype), device=w_op.device) if out is None else out
    d.C = y.data_ptr(); d.ldc = y.shape[1]
    if scatter is not None:
        assert out is not None
        d.c_scatter = 1
        if len(scatter) == 5:                      # maps with depth: (c_sn, c_sd, c_sh, c_sw, c_row0)
            d.c_sn, d.c_sd, d.c_sh, d.c_sw, d.c_row0 = scatter
        else:
            d.c_sn, d.c_sh, d.c_sw, d.c_row0 = scatter
    if row_scale is not None:
        d.row_scale = row_scale[0].data_ptr(); d.rs_images = int(row_scale[1]); d.rs_stride = int(row_scale[2])
    ops.conv_forward(d, dtype)
    return CL(y, N, odhw, cout)


_norm_ws = {}


def _workspace(N, S, G, device):
    n = _lib.lib().ipoke_groupnorm_workspace_floats(N, S, G)
    key = (device, torch.cuda.current_stream().cuda_stream if device.type == "cuda" else 0)   # one scratch buffer per stream
    buf = _norm_ws.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 16), dtype=torch.float32, device=device)
        _norm_ws[key] = buf
    return buf


def group_norm(x, groups, dtype, gamma=None, beta=None, act=_lib.ACT_NONE, res=None, mod=None, eps=1e-5, res_post=False, next_groups=None):
    """y = act(GN(x)*gamma+beta [*(1+mod_gamma)+mod_beta] [+res]); InstanceNorm = groups == C, no affine.  ``res_post``: the residual
    joins behind the activation, y = act(...) + res.  ``next_groups``: the pass also leaves the chunk statistics of y for a following
    norm of that many groups (returned CL carries ``stats_part``); a CL with ``stats_part`` for ``groups`` skips its statistics pass."""
    y = torch.empty_like(x.t)
    d = NormDesc()
    d.x = x.t.data_ptr(); d.ldx = x.t.shape[1]; d.y = y.data_ptr(); d.ldy = y.shape[1]; d.y_f32 = 0
    d.N, d.S, d.C, d.G, d.eps = x.N, x.S, x.C, groups, eps
    d.gamma = 0 if gamma is None else gamma.data_ptr(); d.beta = 0 if beta is None else beta.data_ptr()
    if mod is not None:
        d.mod_gamma = mod[0].t.data_ptr(); d.mod_beta = mod[1].t.data_ptr(); d.ld_mod = mod[0].t.shape[1]
        if mod[0].N != x.N:                     # frames of a clip decoded as one batch ordered (frame, clip): shared SPADE maps
            assert x.N % mod[0].N == 0 and mod[0].S == x.S
            d.mod_samples = mod[0].N
    if res is not None:
        d.res = res.t.data_ptr(); d.ld_res = res.t.shape[1]; d.res_post = int(bool(res_post))
    d.act = act
    handed = getattr(x, "stats_part", None)
    if handed is not None and handed[0] == groups:          # the producer of x left its chunk statistics (for this group count)
        ws = handed[1]
        d.part_chunks = -(-x.S // 256)
    else:
        ws = _workspace(x.N, x.S, groups, x.t.device)
    d.workspace = ws.data_ptr()
    nxt = None
    if next_groups is not None and x.C % next_groups == 0 and x.C <= 2048:
        nxt = torch.empty(int(_lib.lib().ipoke_groupnorm_workspace_floats(x.N, x.S, next_groups)), dtype=torch.float32, device=x.t.device)
        d.next_part = nxt.data_ptr(); d.next_G = next_groups
    check(_lib.lib().ipoke_groupnorm(byref(d), ops._dt(dtype), _lib.current_stream()))
    out = CL(y, x.N, x.dhw, x.C)
    if nxt is not None:
        out.stats_part = (next_groups, nxt)
    return out


def add_act(a, b, dtype, act=_lib.ACT_NONE):
    y = torch.empty_like(a.t)
    check(_lib.lib().ipoke_add_act(ptr(a.t), a.t.shape[1], ptr(None if b is None else b.t), 0 if b is None else b.t.shape[1],
                                   ptr(y), y.shape[1], a.M, a.C, act, ops._dt(dtype), _lib.current_stream()))
    return CL(y, a.N, a.dhw, a.C)


def to_nchw(x, dtype):
    """CL -> fp32 [N, C, (D,) H, W]."""
    y = torch.empty(x.N, x.C, x.S, dtype=torch.float32, device=x.t.device)
    check(_lib.lib().ipoke_cl_to_nchw(ptr(x.t), x.t.shape[1], ptr(y), x.N, x.C, x.S, ops._dt(dtype), _lib.current_stream()))
    d, h, w = x.dhw
    return y.view(x.N, x.C, h, w) if d == 1 else y.view(x.N, x.C, d, h, w)


def from_nchw(t, dtype):
    """fp32 [N, C, H, W] -> CL of the compute dtype (channel pitch padded to 16 bytes)."""
    N, C, H, W = t.shape
    ld = round_up(C, e16(dtype))
    y = torch.empty(N * H * W, ld, dtype=ops.torch_dtype(dtype), device=t.device)
    check(_lib.lib().ipoke_nchw_to_cl(ptr(t.contiguous().float()), ptr(y), ld, N, C, H * W, ops._dt(dtype), _lib.current_stream()))
    return CL(y, N, (1, H, W), C)


def bilinear_cl(x_nchw, size):
    """align_corners=True bilinear resize of an fp32 NCHW image to channels-last fp32 [N*Ho*Wo, C]."""
    N, C, Hi, Wi = x_nchw.shape
    Ho, Wo = size
    y = torch.empty(N * Ho * Wo, C, dtype=torch.float32, device=x_nchw.device)
    check(_lib.lib().ipoke_bilinear_cl(ptr(x_nchw.contiguous().float()), ptr(y), N, C, Hi, Wi, Ho, Wo, _lib.current_stream()))
    return y
