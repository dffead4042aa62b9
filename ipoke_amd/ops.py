"""Per-layer Python entry points over the C ABI (one function per kernel family of include/ipoke_hip.h).

These are thin: they allocate outputs with torch, pass raw device pointers and the current HIP stream,
and raise on any non-zero status.  The flow engine (``ipoke_amd.flow``) chains the same kernels natively;
these wrappers exist for the first-stage VAE modules, for the per-layer parity tests and for users who
want a single layer.  State tensors are fp32 ``[B*64, ld]`` (see ``to_state``).
"""
from ctypes import byref, c_int32

import torch

from . import _lib
from ._lib import AffineDesc, ConvDesc, McfDesc, WgradDesc, check, ptr

P8 = 64   # positions of the 8x8 latent


def _dt(dtype):
    return _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)


def torch_dtype(dtype):
    return torch.bfloat16 if _dt(dtype) == _lib.BF16 else torch.float32


def _s():
    return _lib.current_stream()


# ------------------------------------------------------------------ layout
def to_state(x, ld=None):
    """[B,C,8,8] fp32 -> state [B*64, ld] (columns >= C zero)."""
    B, C = x.shape[0], x.shape[1]
    ld = C if ld is None else ld
    s = torch.zeros(B * P8, ld, dtype=torch.float32, device=x.device)
    check(_lib.lib().ipoke_nchw_to_state(ptr(x.contiguous().float()), ptr(s), B, C, P8, ld, _s()))
    return s


def from_state(s, B, C):
    x = torch.empty(B, C, 8, 8, dtype=torch.float32, device=s.device)
    check(_lib.lib().ipoke_state_to_nchw(ptr(s), ptr(x), B, C, P8, s.shape[1], _s()))
    return x


def cond_prepare(cond, dtype, act=_lib.ACT_ELU):
    B, Cc = cond.shape[0], cond.shape[1]
    out = torch.empty(B * P8, Cc, dtype=torch_dtype(dtype), device=cond.device)
    check(_lib.lib().ipoke_cond_prepare(ptr(cond.contiguous().float()), ptr(out), B, Cc, P8, act, _dt(dtype), _s()))
    return out


# ------------------------------------------------------------------ ActNorm / Shuffle
def _i32(idx):
    return None if idx is None else idx.to(torch.int32).contiguous()


def actnorm_fwd(state, c0, C, log_scale=None, bias=None, idx=None):
    out = torch.empty_like(state)
    i = _i32(idx)
    check(_lib.lib().ipoke_actnorm_fwd(ptr(state), ptr(out), state.shape[0], state.shape[1], c0, C,
                                       ptr(None if log_scale is None else log_scale.contiguous()),
                                       ptr(None if bias is None else bias.contiguous()), ptr(i), _s()))
    return out


def actnorm_inv(state, c0, C, log_scale=None, bias=None, inv_idx=None):
    out = torch.empty_like(state)
    i = _i32(inv_idx)
    check(_lib.lib().ipoke_actnorm_inv(ptr(state), ptr(out), state.shape[0], state.shape[1], c0, C,
                                       ptr(None if log_scale is None else log_scale.contiguous()),
                                       ptr(None if bias is None else bias.contiguous()), ptr(i), _s()))
    return out


def actnorm_bwd(dy, x, c0, C, log_scale, idx, dld, B):
    dx = torch.empty_like(dy)
    part = torch.zeros(B, 2 * C, device=dy.device) if log_scale is not None else None
    i = _i32(idx)
    check(_lib.lib().ipoke_actnorm_bwd(ptr(dy), ptr(x), ptr(dx), dy.shape[0], dy.shape[1], c0, C,
                                       ptr(None if log_scale is None else log_scale.contiguous()), ptr(i), ptr(dld), B, P8,
                                       ptr(part), _s()))
    if part is None:
        return dx, None, None
    tot = part.sum(0)
    return dx, tot[:C], tot[C:]


def actnorm_init_(state, c0, C, log_scale, bias):
    check(_lib.lib().ipoke_actnorm_init(ptr(state), state.shape[0], state.shape[1], c0, C, ptr(log_scale), ptr(bias), _s()))


# ------------------------------------------------------------------ affine coupling
def _affine_desc(raw, bias, Cp, t_off, t_stride, ld):
    d = AffineDesc()
    if raw.dim() == 2:
        raw = raw.unsqueeze(0)
    raw = raw.contiguous().float()
    d.raw = raw.data_ptr(); d.nsplit = raw.shape[0]; d.split_stride = raw.shape[1] * raw.shape[2]; d.ldraw = raw.shape[2]
    d.bias = 0 if bias is None else bias.data_ptr()
    d.Cp, d.t_off, d.t_stride, d.P, d.ld = Cp, t_off, t_stride, P8, ld
    return d, raw


def affine_fwd(state, raw, bias, Cp, t_off, t_stride, B):
    d, keep = _affine_desc(raw, bias, Cp, t_off, t_stride, state.shape[1])
    out = torch.empty_like(state)
    scale = torch.empty(state.shape[0], Cp, device=state.device)
    logdet = torch.empty(B, device=state.device)
    check(_lib.lib().ipoke_affine_fwd(byref(d), ptr(state), ptr(out), ptr(scale), ptr(logdet), 1, B, _s()))
    return out, logdet, scale


def affine_inv(state, raw, bias, Cp, t_off, t_stride, B):
    d, keep = _affine_desc(raw, bias, Cp, t_off, t_stride, state.shape[1])
    out = torch.empty_like(state)
    check(_lib.lib().ipoke_affine_inv(byref(d), ptr(state), ptr(out), B, _s()))
    return out


# ------------------------------------------------------------------ implicit-GEMM convolution
def conv_desc(NB, in_dhw, out_dhw, k, s, p, transposed=False):
    d = ConvDesc()
    d.NB = NB
    d.Di, d.Hi, d.Wi = in_dhw
    d.Do, d.Ho, d.Wo = out_dhw
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = s
    d.pd, d.ph, d.pw = p
    d.transposed = int(transposed)
    d.c_cstride = 1
    d.splitk = 1
    return d


def conv_forward(d, dtype):
    check(_lib.lib().ipoke_conv_forward(byref(d), _dt(dtype), _s()))


def conv_wgrad(d, dtype):
    check(_lib.lib().ipoke_conv_wgrad(byref(d), _dt(dtype), _s()))


# ------------------------------------------------------------------ masked convolutional flow
def mcf_dims(C, Cc, dtype):
    dims = (c_int32 * 8)()
    check(_lib.lib().ipoke_mcf_shadow_dims(C, Cc, _dt(dtype), dims))
    return dict(Cp=dims[0], K1p=dims[1], K2p=dims[2], K3p=dims[3], Hq=dims[4], Hr=dims[5], N2r=dims[6], Cr=dims[7])


def mcf_desc(x_state, C, B, cond_act, W1, W2, bias2, order):
    d = McfDesc()
    d.x = x_state.data_ptr(); d.ld = x_state.shape[1]; d.C = C; d.B = B
    d.cond = cond_act.data_ptr(); d.Cc = cond_act.shape[1]
    d.W1 = W1.data_ptr(); d.W2 = W2.data_ptr(); d.bias2 = bias2.data_ptr(); d.order = order
    return d
