"""First-stage video VAE of iPOKE on the HIP kernels: 3-D ResNet-18 motion encoder, ConvGRU, SPADE decoder and
the small 2-D poke / image encoders.  Class names, constructor arguments, attributes and state-dict keys follow
the reference (models/first_stage_motion_model.py:469-522, modules/motion_models/{motion_encoder,rnn}.py,
modules/autoencoders/{fully_conv_models,util}.py) so that reference checkpoints load with ``strict=False``
exactly as ``PokeMotionModel.__initialize_first_stage`` does.

``forward`` / ``decode`` are the *inference* direction (what the second stage and sampling need: everything runs under
``torch.no_grad`` there, second_stage_video.py:269-303); spectral-normalised convolutions are then evaluated in eval
mode (frozen u, v), i.e. ``W / sigma`` is folded into the cached weight operand.  The differentiable pass used for
first-stage training (L1 + KL) lives in ``ipoke_amd.first_stage_train`` and is reached through
``SpadeCondMotionModel.training_loss``.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F   # noqa: F401  (normalize() for buffer init only)

from . import _lib, nn as K, ops
from ._lib import check, ptr

ACT = {"none": _lib.ACT_NONE, "elu": _lib.ACT_ELU, "relu": _lib.ACT_RELU, "tanh": _lib.ACT_TANH}


class _Cached(nn.Module):
    """Mixin: cache of weight operands, dropped whenever the state dict is (re)loaded."""

    def _cache(self):
        if "_opcache" not in self.__dict__:
            self.__dict__["_opcache"] = {}
        return self.__dict__["_opcache"]

    def invalidate(self):
        for m in self.modules():
            m.__dict__.pop("_opcache", None)
        from . import first_stage_train
        first_stage_train.clear_operand_cache()

    def _load_from_state_dict(self, *a, **k):
        self.__dict__.pop("_opcache", None)
        return super()._load_from_state_dict(*a, **k)


class _Conv(_Cached):
    """Parameter holder + executor of one convolution (nn.Conv2d / nn.Conv3d / nn.ConvTranspose2d naming)."""

    def __init__(self, cin, cout, k, stride, pad, bias=True, transposed=False, snorm=False, dims=2):
        super().__init__()
        k3 = (1, k, k) if (dims == 2 and isinstance(k, int)) else ((k, k, k) if isinstance(k, int) else tuple(k))
        st = (1, stride, stride) if (dims == 2 and isinstance(stride, int)) else (
            (stride,) * 3 if isinstance(stride, int) else tuple(stride))
        pd = (0, pad, pad) if (dims == 2 and isinstance(pad, int)) else ((pad,) * 3 if isinstance(pad, int) else tuple(pad))
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k3, st, pd
        self.transposed, self.snorm, self.dims = transposed, snorm, dims
        kshape = k3[1:] if dims == 2 else k3
        wshape = (cin, cout, *kshape) if transposed else (cout, cin, *kshape)
        fan_in = cin * int(np.prod(kshape))
        w = torch.empty(wshape).uniform_(-1, 1) / math.sqrt(fan_in)
        if snorm:
            self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
            self.weight_orig = nn.Parameter(w)
            rows = cout
            self.register_buffer("weight_u", F.normalize(torch.randn(rows), dim=0))
            self.register_buffer("weight_v", F.normalize(torch.randn(w.numel() // rows), dim=0))
        else:
            self.weight = nn.Parameter(w)
            self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def operand(self, dtype):
        c = self._cache()
        if dtype not in c:
            with torch.no_grad():
                if self.snorm:
                    w = self.weight_orig / K.spectral_sigma(self.weight_orig, self.weight_u, self.weight_v, self.transposed)
                else:
                    w = self.weight
                if self.dims == 2:
                    w = w.unsqueeze(2)
                wop, kc = K.weight_operand(w, dtype, self.transposed)
                c[dtype] = (wop, kc, None if self.bias is None else self.bias.detach().float().contiguous())
        return c[dtype]

    def _mid_operand(self, dtype):
        """The kd = 1 slice of a (3, k, k) filter as a 2-D operand: on an input of depth 1 (pad 1) the other two depth taps only
        ever see the zero padding -- two thirds of the multiplications of layer3 / layer4 of the 3-D encoder."""
        c = self._cache()
        key = ("mid", dtype)
        if key not in c:
            with torch.no_grad():
                wop, kc = K.weight_operand(self.weight[:, :, 1:2].contiguous(), dtype, self.transposed)
                c[key] = (wop, kc, None if self.bias is None else self.bias.detach().float().contiguous())
        return c[key]

    _PHASE_TAPS = ((1,), (2, 0))      # output parity 0: tap 1 on input pixel i; parity 1: tap 2 on pixel i, tap 0 on pixel i + 1

    def _phase_operands(self, dtype):
        """The four sub-pixel phases of a 3 x 3 / stride 2 / padding 1 / output_padding 1 ConvTranspose2d (util.py:52-55) as
        stride-1 convolutions: output pixel (2 i + a, 2 j + b) = sum over the taps of parity (a, b) -- 1, 2, 2, 4 of the 9."""
        c = self._cache()
        key = ("phases", dtype)
        if key not in c:
            with torch.no_grad():
                w = self.weight_orig / K.spectral_sigma(self.weight_orig, self.weight_u, self.weight_v, True) if self.snorm else self.weight
                wr = w.detach().transpose(0, 1)                                  # [cout, cin, kh, kw]
                ops_ = {}
                # wide layers (>= 128 input channels, >= 96 outputs): the two-tap phases as 2 x 2 windows with a zero row / column, so
                # that they run on the halo-staged kernel like the four-tap phase (twice the multiplications at 2.4x the rate)
                pad2 = _PHASE_PAD and self.cin >= 128 and self.cout >= 96 and ops._dt(dtype) == _lib.BF16
                for a in (0, 1):
                    for b in (0, 1):
                        wp = wr[:, :, list(self._PHASE_TAPS[a])][:, :, :, list(self._PHASE_TAPS[b])].contiguous()
                        if pad2 and a + b == 1:
                            full = torch.zeros(wp.shape[0], wp.shape[1], 2, 2, dtype=wp.dtype, device=wp.device)
                            full[:, :, :1 + a, :1 + b] = wp
                            wp = full
                        ops_[(a, b)] = K.weight_operand(wp.unsqueeze(2), dtype) + ((wp.shape[2], wp.shape[3]),)
                c[key] = (ops_, None if self.bias is None else self.bias.detach().float().contiguous())
        return c[key]

    def _run_phases(self, x, dtype, act, out_f32):
        ops_, b = self._phase_operands(dtype)
        N, (_, Hi, Wi) = x.N, x.dhw
        Ho, Wo = 2 * Hi, 2 * Wi
        ldc = self.cout if out_f32 else K.round_up(self.cout, K.e16(dtype))
        y = torch.empty(N * Ho * Wo, ldc, dtype=torch.float32 if out_f32 else ops.torch_dtype(dtype), device=x.t.device)
        for (a, bb), (wop, kc, (kh, kw)) in ops_.items():
            K.conv(x, wop, kc, self.cout, (1, kh, kw), (1, 1, 1), (0, 0, 0), dtype, bias=b, act=act, out_f32=out_f32, out=y,
                   odhw=(1, Hi, Wi), scatter=(Ho * Wo, 2 * Wo, 2, a * Wo + bb))
        return K.CL(y, N, (1, Ho, Wo), self.cout)

    def run(self, x, dtype, act=_lib.ACT_NONE, out_f32=False, src_f32=None):
        if (_CT_PHASES and self.transposed and self.dims == 2 and x is not None and self.k == (1, 3, 3) and self.stride == (1, 2, 2)
                and self.pad == (0, 1, 1)):
            return self._run_phases(x, dtype, act, out_f32)
        if (_DEPTH1_SLICE and self.dims == 3 and x is not None and x.dhw[0] == 1 and self.k[0] == 3 and self.pad[0] == 1
                and not self.transposed and not self.snorm):
            wop, kc, b = self._mid_operand(dtype)
            y = K.conv(x, wop, kc, self.cout, (1, self.k[1], self.k[2]), (1, self.stride[1], self.stride[2]), (0, self.pad[1], self.pad[2]),
                       dtype, bias=b, act=act, out_f32=out_f32)
            return y
        wop, kc, b = self.operand(dtype)
        out_pad = (0, self.pad[1], self.pad[2]) if self.transposed else (0, 0, 0)     # util.py:52 output_padding=padding
        return K.conv(x, wop, kc, self.cout, self.k, self.stride, self.pad, dtype, bias=b, act=act, transposed=self.transposed,
                      out_pad=out_pad, out_f32=out_f32, src_f32=src_f32)


class _Norm(nn.Module):
    """GroupNorm / InstanceNorm parameter holder."""

    def __init__(self, kind, ch):
        super().__init__()
        self.kind, self.ch = kind, ch
        if kind == "group":
            self.weight = nn.Parameter(torch.ones(ch))
            self.bias = nn.Parameter(torch.zeros(ch))
            self.groups = 16
        else:                      # "in": InstanceNorm2d(affine=False)
            self.groups = ch

    def run(self, x, dtype, act=_lib.ACT_NONE, res=None, res_post=False, next_groups=None):
        if self.kind == "group":
            return K.group_norm(x, self.groups, dtype, self.weight.detach(), self.bias.detach(), act=act, res=res, res_post=res_post,
                                next_groups=next_groups)
        return K.group_norm(x, self.groups, dtype, act=act, res=res, res_post=res_post, next_groups=next_groups)


_PHASE_PAD = os.environ.get("IPOKE_NO_PHASE_PAD", "0") != "1"        # developer A/B: two-tap phases of wide up-convolutions as they are
_RES_POST = os.environ.get("IPOKE_NO_RES_POST", "0") != "1"         # developer A/B: ResBlock's sum as its own element-wise pass
_NEXT_STATS = os.environ.get("IPOKE_NO_NEXT_STATS", "0") != "1"     # developer A/B: the SPADE norm makes its own statistics pass
_STEM_FOLD = os.environ.get("IPOKE_NO_STEM_FOLD", "0") != "1"      # developer A/B: conv1 of the 3-D encoder read in place
_CT_PHASES = os.environ.get("IPOKE_NO_CT_PHASES", "0") != "1"       # developer A/B: stride-2 ConvTranspose2d as one 9-tap launch
_DEPTH1_SLICE = os.environ.get("IPOKE_NO_DEPTH1_SLICE", "0") != "1"  # developer A/B: 3 x 3 x 3 filters on depth-1 inputs run all 27 taps


# ---------------------------------------------------------------------------------------------- 3-D encoder
class BasicBlock(nn.Module):
    """motion_encoder.py:45-74."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = _Conv(cin, cout, 3, stride, 1, bias=False, dims=3)
        self.bn1 = _Norm("group", cout)
        self.conv2 = _Conv(cout, cout, 3, 1, 1, bias=False, dims=3)
        self.bn2 = _Norm("group", cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_Conv(cin, cout, 1, stride, 0, bias=False, dims=3), _Norm("group", cout))

    def run(self, x, dtype):
        out = self.bn1.run(self.conv1.run(x, dtype), dtype, act=_lib.ACT_RELU)
        res = x if self.downsample is None else self.downsample[1].run(self.downsample[0].run(x, dtype), dtype)
        return self.bn2.run(self.conv2.run(out, dtype), dtype, act=_lib.ACT_RELU, res=res)


class ResNetMotionEncoder(nn.Module):
    """motion_encoder.py:150-241 (resnet18_alternative).  ``forward(x[B,3,T,H,W]) -> (z, mu, logvar)``."""

    def __init__(self, dic, dtype="bf16"):
        super().__init__()
        ch = list(dic["ENC_M_channels"])
        self.dtype = dtype
        self.be_determinstic = bool(dic.get("deterministic", False))      # [sic]
        self.spatial_size = dic["img_size"]
        max_frames = dic["max_frames"]
        self.min_ssize = dic.get("min_spatial_size", 8)
        self.z_dim = dic["z_dim"]
        self.conv1 = _Conv(3, ch[0], (3, 7, 7), 2, (1, 3, 3), bias=False, dims=3)
        self.bn1 = _Norm("group", ch[0])
        first_down = (len(ch) - 1 < int(np.ceil(np.log2(max_frames)))) or dic["full_seq"]
        self.layer1 = self._make(ch[0], ch[1], (2, 1, 1) if first_down else 1)
        self.layer2 = self._make(ch[1], ch[2], 2)
        self.layer3 = self._make(ch[2], ch[3], 2)
        last = ch[3]
        self.stride4 = (2, 1, 1) if dic["full_seq"] and max_frames >= 16 else None
        if self.spatial_size // 8 > self.min_ssize:
            self.stride4 = 2
        if self.stride4 is not None:
            if len(ch) < 5:
                ch.append(ch[-1])
            self.layer4 = self._make(ch[3], ch[4], self.stride4)
            last = ch[4]
        self.has5 = self.spatial_size // 16 > self.min_ssize
        if self.has5:
            self.layer5 = self._make(last, ch[5], 2)
            last = ch[5]
        self.conv_mu = _Conv(last, self.z_dim, 3, 1, 1)
        self.conv_var = _Conv(last, self.z_dim, 3, 1, 1)

    @staticmethod
    def _make(cin, cout, stride):
        return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))

    def _head_operand(self):
        """conv_mu and conv_var share their input: one GEMM with 2*z output channels."""
        c = self.conv_mu._cache()
        key = ("head", self.dtype)
        if key not in c:
            with torch.no_grad():
                w = torch.cat([self.conv_mu.weight, self.conv_var.weight], 0).unsqueeze(2)
                wop, kc = K.weight_operand(w, self.dtype)
                b = torch.cat([self.conv_mu.bias, self.conv_var.bias]).detach().float().contiguous()
            c[key] = (wop, kc, b)
        return c[key]

    def _stem_operand(self):
        """conv1 as 21 taps (kd, kh) of 32 channels: channel 4*kw + c of a tap is weight[:, c, kd, kh, kw] (kw < 7, c < 3), the rest 0."""
        c = self.conv1._cache()
        key = ("stem", self.dtype)
        if key not in c:
            with torch.no_grad():
                w = self.conv1.weight.detach().float()                       # [64][3][3][7][7]
                buf = torch.zeros(w.shape[0], 3, 7, 8, 4, dtype=torch.float32, device=w.device)
                buf[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
                c[key] = buf.reshape(w.shape[0], 21 * 32).to(ops.torch_dtype(self.dtype)).contiguous()
        return c[key]

    def _stem(self, x):
        """conv1 on the fp32 clip.  A (3, 7, 7) window over 3 channels read in place is 147 taps of 3 strided floats (653 us at
        B = 20, 16 x 128 x 128 through the register-staged kernel); with the clip rewritten once as padded channels-last pixels of four
        channels (ipoke_clip_to_cl4) a 7-tap run along x is one aligned 8-pixel read and the convolution 21 taps of 32 channels on
        the LDS-DMA kernel.  Descriptor: 'double pixels' of 8 channels along x (stride 1) so that every tap start is 16-byte aligned."""
        dt = self.dtype
        B, C, T, H, W = x.shape
        conv = self.conv1
        if (not _STEM_FOLD or C != 3 or conv.k != (3, 7, 7) or conv.stride != (2, 2, 2) or conv.pad != (1, 3, 3) or W % 2
                or conv.bias is not None):
            st = (x.stride(0), x.stride(1), x.stride(2), x.stride(3), x.stride(4))
            return conv.run(None, dt, src_f32=(x, B, C, (T, H, W), st))
        Wp = W + 6
        clip = torch.empty(B * T * H * Wp, 4, dtype=ops.torch_dtype(dt), device=x.device)
        check(_lib.lib().ipoke_clip_to_cl4(ptr(x), x.stride(0), x.stride(1), x.stride(2), x.stride(3), x.stride(4), B, T, H, W, 3, 3,
                                           ptr(clip), ops._dt(dt), _lib.current_stream()))
        odhw = ((T + 2 - 3) // 2 + 1, (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1)
        wop = self._stem_operand()
        d = ops.conv_desc(B, (T, H, odhw[2]), odhw, (3, 7, 1), (2, 2, 1), (1, 3, 0))
        d.A = clip.data_ptr(); d.a_f32 = 0
        d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = T * H * Wp * 4, H * Wp * 4, Wp * 4, 8, 1
        d.a_coff = 0; d.Kc_real = 32; d.Kc = 32
        d.W = wop.data_ptr(); d.ldw = wop.shape[1]; d.Nout = conv.cout
        d.bias = 0; d.act = _lib.ACT_NONE
        y = torch.empty(B * odhw[0] * odhw[1] * odhw[2], K.round_up(conv.cout, K.e16(dt)), dtype=ops.torch_dtype(dt), device=x.device)
        d.C = y.data_ptr(); d.ldc = y.shape[1]
        ops.conv_forward(d, dt)
        return K.CL(y, B, odhw, conv.cout)

    @torch.no_grad()
    def forward(self, x, eps=None):
        _lib.require_gpu()
        dt = self.dtype
        B, C, T, H, W = x.shape
        x = x.float()
        h = self._stem(x)
        h = self.bn1.run(h, dt, act=_lib.ACT_RELU)
        layers = [self.layer1, self.layer2, self.layer3] + ([self.layer4] if self.stride4 is not None else []) + (
            [self.layer5] if self.has5 else [])
        for layer in layers:
            for blk in layer:
                h = blk.run(h, dt)
        if h.dhw[0] != 1:
            raise ValueError(f"temporal extent after the encoder is {h.dhw[0]}, expected 1 (x.squeeze(2) in the reference)")
        wop, kc, b = self._head_operand()
        mulv = K.conv(h, wop, kc, 2 * self.z_dim, (1, 3, 3), (1, 1, 1), (0, 1, 1), dt, bias=b)
        M, Z = mulv.M, self.z_dim
        if not self.be_determinstic and eps is None:
            # the reference draws on the CPU generator: torch.FloatTensor(size).normal_() (motion_encoder.py:220)
            eps = torch.FloatTensor(B, Z, h.dhw[1], h.dhw[2]).normal_().to(x.device)
        eps_s = None if self.be_determinstic else ops.to_state(eps)
        z = torch.empty(M, Z, device=x.device); mu = torch.empty_like(z); lv = torch.empty_like(z)
        check(_lib.lib().ipoke_reparameterize(ptr(mulv.t), mulv.t.shape[1], ptr(eps_s), ptr(z), ptr(mu), ptr(lv), M, Z,
                                              ops._dt(dt), _lib.current_stream()))
        z, mu, lv = (ops.from_state(t_, B, Z) for t_ in (z, mu, lv))
        if self.be_determinstic:
            return mu, mu, mu
        return z, mu, lv


# ---------------------------------------------------------------------------------------------- ConvGRU
class ConvGRUCell(_Cached):
    """rnn.py:4-56."""

    def __init__(self, cin, hidden, k=3):
        super().__init__()
        self.cin, self.hidden = cin, hidden
        self.reset_gate = _Conv(cin + hidden, hidden, k, 1, k // 2)
        self.update_gate = _Conv(cin + hidden, hidden, k, 1, k // 2)
        self.out_gate = _Conv(cin + hidden, hidden, k, 1, k // 2)

    def _ur_operand(self, dtype):
        c = self._cache()
        if dtype not in c:
            with torch.no_grad():
                w = torch.cat([self.update_gate.weight, self.reset_gate.weight], 0).unsqueeze(2)
                wop, kc = K.weight_operand(w, dtype)
                b = torch.cat([self.update_gate.bias, self.reset_gate.bias]).detach().float().contiguous()
            c[dtype] = (wop, kc, b)
        return c[dtype]

    def run(self, x, h, dtype):
        """x, h: CL with C = cin / hidden.  Returns the new hidden state (CL)."""
        Ch = self.hidden
        xh = torch.cat([x.t[:, :x.C], h.t[:, :Ch]], dim=1).contiguous()          # [M][cin+hidden]  (torch.cat in rnn.py:48)
        xh_cl = K.CL(xh, x.N, x.dhw, x.C + Ch)
        wop, kc, b = self._ur_operand(dtype)
        ur = K.conv(xh_cl, wop, kc, 2 * Ch, (1, 3, 3), (1, 1, 1), (0, 1, 1), dtype, bias=b)
        xhr = xh.clone()
        u = torch.empty(x.M, Ch, dtype=xh.dtype, device=xh.device)
        check(_lib.lib().ipoke_gru_gates(ptr(ur.t), ptr(h.t), h.t.shape[1], ptr(xhr[:, x.C:]), xhr.shape[1], ptr(u), x.M, Ch,
                                         ops._dt(dtype), _lib.current_stream()))
        o = self.out_gate.run(K.CL(xhr, x.N, x.dhw, x.C + Ch), dtype)
        hn = torch.empty(x.M, Ch, dtype=xh.dtype, device=xh.device)
        check(_lib.lib().ipoke_gru_update(ptr(o.t), ptr(u), ptr(h.t), h.t.shape[1], ptr(hn), Ch, x.M, Ch, ops._dt(dtype),
                                          _lib.current_stream()))
        return K.CL(hn, x.N, x.dhw, Ch)


class ConvGRU(nn.Module):
    """rnn.py:59-133: ``forward(x, hidden) -> list of new hidden states`` (NCHW fp32 at the API)."""

    def __init__(self, input_size, hidden_sizes, kernel_sizes, n_layers, upsampling=None, dtype="bf16"):
        super().__init__()
        self.n_layers, self.dtype = n_layers, dtype
        self.cells = nn.Sequential(*[ConvGRUCell(input_size if i == 0 else hidden_sizes, hidden_sizes, kernel_sizes)
                                     for i in range(n_layers)])

    def run(self, x, hidden):
        out = []
        for cell, h in zip(self.cells, hidden):
            x = cell.run(x, h, self.dtype)
            out.append(x)
        return out

    @torch.no_grad()
    def forward(self, x, hidden):
        xs = K.from_nchw(x, self.dtype)
        hs = [K.from_nchw(h, self.dtype) for h in hidden]
        return [K.to_nchw(h, self.dtype) for h in self.run(xs, hs)]


# ---------------------------------------------------------------------------------------------- 2-D conv blocks
class Conv2dBlock(nn.Module):
    """util.py:195-273 (zero pad -> conv -> norm -> activation)."""

    def __init__(self, cin, cout, ks, st, padding=0, norm="none", activation="elu", snorm=False):
        super().__init__()
        self.activation = activation
        self.norm = None if norm == "none" else _Norm(norm, cout)
        self.conv = _Conv(cin, cout, ks, st, padding, snorm=snorm)

    def run(self, x, dtype, res=None, out_f32=False):
        if self.norm is None:
            y = self.conv.run(x, dtype, act=ACT[self.activation] if res is None else _lib.ACT_NONE, out_f32=out_f32)
            return y if res is None else K.add_act(y, res, dtype, ACT[self.activation])
        return self.norm.run(self.conv.run(x, dtype), dtype, act=ACT[self.activation], res=res)


class Conv2dTransposeBlock(nn.Module):
    """util.py:7-73.  NB the key "elu" selects nn.ReLU in this block (util.py:41-42)."""

    def __init__(self, cin, cout, ks, st, padding=0, norm="none", activation="elu", snorm=False):
        super().__init__()
        self.act = _lib.ACT_RELU if activation == "elu" else ACT[activation]
        self.norm = None if norm == "none" else _Norm(norm, cout)
        self.conv = _Conv(cin, cout, ks, st, padding, transposed=True, snorm=snorm)

    def run(self, x, dtype):
        if self.norm is None:
            return self.conv.run(x, dtype, act=self.act)
        return self.norm.run(self.conv.run(x, dtype), dtype, act=self.act)


class ResBlock(nn.Module):
    """util.py:106-192: out = conv2(conv1(x)) + res_conv(x) (skip conv with InstanceNorm + activation)."""

    def __init__(self, cin, cout, norm="in", activation="elu", upsampling=False, stride=1, snorm=False):
        super().__init__()
        if upsampling:
            self.conv1 = Conv2dTransposeBlock(cin, cout, 3, 2, 1, norm=norm, activation=activation, snorm=snorm)
        else:
            self.conv1 = Conv2dBlock(cin, cout, 3, stride, 1, norm=norm, activation=activation, snorm=snorm)
        self.conv2 = Conv2dBlock(cout, cout, 3, 1, 1, norm=norm, activation="none", snorm=snorm)
        self.convolve_res = cin != cout or upsampling or stride != 1
        if self.convolve_res:
            if upsampling:
                self.res_conv = Conv2dTransposeBlock(cin, cout, 3, 2, 1, norm="in", activation=activation, snorm=snorm)
            else:
                self.res_conv = Conv2dBlock(cin, cout, 3, stride, 1, norm="in", activation=activation, snorm=snorm)

    def run(self, x, dtype, next_groups=None):
        """``next_groups``: a norm of that many groups reads the block's output next (the decoder's SPADE norm): the pass that writes
        the output leaves its chunk statistics for it."""
        rc = self.res_conv if self.convolve_res else None
        if (_RES_POST and rc is not None and rc.norm is not None and self.conv2.norm is None and self.conv2.activation == "none"):
            # out = conv2(conv1(x)) + act(norm(res_conv(x))): the sum rides on the skip path's norm pass (the residual joins behind its
            # activation) instead of being a pass of its own over three tensors of the block's output size
            y2 = self.conv2.run(self.conv1.run(x, dtype), dtype)
            act = rc.act if isinstance(rc, Conv2dTransposeBlock) else ACT[rc.activation]
            return rc.norm.run(rc.conv.run(x, dtype), dtype, act=act, res=y2, res_post=True, next_groups=next_groups if _NEXT_STATS else None)
        res = rc.run(x, dtype) if rc is not None else x
        return self.conv2.run(self.conv1.run(x, dtype), dtype, res=res)


class Spade(_Cached):
    """util.py:473-500.  gamma/beta depend on the start frame only: ``modulation`` is computed once per clip and
    reused for every generated frame (the reference recomputes it T-1 times)."""

    def __init__(self, ch, groups=16):
        super().__init__()
        while ch % groups != 0:
            groups -= 1
        self.groups, self.ch = groups, ch
        self.conv = _Conv(3, 128, 3, 1, 1)
        self.conv_gamma = _Conv(128, ch, 3, 1, 1)
        self.conv_beta = _Conv(128, ch, 3, 1, 1)

    def modulation(self, y_nchw, size, dtype):
        N = y_nchw.shape[0]
        ycl = K.bilinear_cl(y_nchw, size)                                     # fp32 [N*H*W, 3]
        st = (size[0] * size[1] * 3, 1, 0, size[1] * 3, 3)
        h = self.conv.run(None, dtype, act=_lib.ACT_LRELU02, src_f32=(ycl, N, 3, (1, size[0], size[1]), st))
        return self.conv_gamma.run(h, dtype), self.conv_beta.run(h, dtype)

    def run(self, x, mod, dtype):
        return K.group_norm(x, self.groups, dtype, mod=mod)


class SpadeCondConvDecoder(nn.Module):
    """fully_conv_models.py:135-177: ``forward([h], start_frame) -> frame [B,3,H,W]``."""

    def __init__(self, config, stacked_input=False, dtype="bf16"):
        super().__init__()
        ch = config["dec_channels"]
        sn = config["spectral_norm"]
        self.dtype = dtype
        self.n_stages = len(ch) - 1
        self.blocks = nn.ModuleList()
        self.spade_blocks = nn.ModuleList()
        self.in_block = ResBlock(2 * config["z_dim"] if stacked_input else config["z_dim"], ch[0], snorm=sn, norm=config["norm"])
        for i, nf in enumerate(ch[1:]):
            self.blocks.append(ResBlock(ch[i], nf, norm="none", upsampling=True, snorm=sn))
            self.spade_blocks.append(Spade(nf))
        self.out_conv = Conv2dBlock(ch[-1], config.get("out_channels", 3), 3, 1, 1, norm="none", activation="tanh")

    def modulations(self, start_frame):
        size = 8
        mods = []
        for sp in self.spade_blocks:
            size *= 2
            mods.append(sp.modulation(start_frame, (size, size), self.dtype))
        return mods

    def run(self, h, mods, frames=1):
        """``frames`` > 1: ``h`` holds frames x clips samples ordered (frame, clip) -- all generated frames of the clips as one batch
        (the SPADE maps in ``mods`` are per clip and shared by the frames); the result is [clips, frames, 3, H, W]."""
        x = self.in_block.run(h, self.dtype)
        for blk, sp, mod in zip(self.blocks, self.spade_blocks, mods):
            x = sp.run(blk.run(x, self.dtype, next_groups=sp.groups), mod, self.dtype)
        y = self.out_conv.run(x, self.dtype, out_f32=True)                     # [M][3] fp32, tanh applied
        if frames > 1:
            return y.t.view(frames, y.N // frames, y.dhw[1], y.dhw[2], 3).permute(1, 0, 4, 2, 3).contiguous()
        return y.t.view(y.N, y.dhw[1], y.dhw[2], 3).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def forward(self, actual_frame, start_frame, del_shape=True):
        h = actual_frame.pop() if del_shape else actual_frame[-1]
        return self.run(K.from_nchw(h, self.dtype), self.modulations(start_frame.float()))


class ConvEncoder(nn.Module):
    """fully_conv_models.py:28-94, deterministic variant: returns (bottleneck(out), out, None)."""

    def __init__(self, nf_in, nf_max, n_stages, dtype="bf16"):
        super().__init__()
        self.dtype = dtype
        nf = 32
        blocks = [Conv2dBlock(nf_in, nf, 3, 2, 1, norm="group", activation="elu", snorm=True)]
        for _ in range(n_stages - 1):
            nxt = min(2 * nf, nf_max)
            blocks.append(ResBlock(nf, nxt, stride=2, norm="group", activation="elu", snorm=True))
            nf = nxt
        self.model = nn.Sequential(*blocks)
        self.bottleneck = nn.Sequential(ResBlock(nf, nf_max, activation="elu", norm="group"))
        self.variational = False

    @torch.no_grad()
    def forward(self, x, sample_prior=False):
        _lib.require_gpu()
        x = x.float()
        N, C, H, W = x.shape
        first = self.model[0]
        st = (x.stride(0), x.stride(1), 0, x.stride(2), x.stride(3))
        h = first.norm.run(first.conv.run(None, self.dtype, src_f32=(x, N, C, (1, H, W), st)), self.dtype, act=_lib.ACT_ELU)
        for blk in list(self.model)[1:]:
            h = blk.run(h, self.dtype)
        mean = h
        out = self.bottleneck[0].run(h, self.dtype)
        return K.to_nchw(out, self.dtype), K.to_nchw(mean, self.dtype), None


class FirstStageWrapper(nn.Module):
    """fully_conv_models.py:9-26.  Only the encoder half is on the hot path; decoder checkpoint keys are
    accepted and ignored (``load_state_dict(strict=False)``)."""

    def __init__(self, config, dtype="bf16"):
        super().__init__()
        self.config = config
        arch = config["architecture"]
        self.be_deterministic = arch["deterministic"]
        if not self.be_deterministic:
            raise NotImplementedError("the shipped poke / image encoders are deterministic")
        n_stages = int(np.log2(config["data"]["spatial_size"][0] // arch["min_spatial_size"]))
        nf_in = arch["nf_in"] + (3 if arch.get("poke_and_image", False) else 0)
        self.encoder = ConvEncoder(nf_in, arch["nf_max"], n_stages, dtype=dtype)


class SpadeCondMotionModel(nn.Module):
    """first_stage_motion_model.py:469-522 (inference).  ``forward(X[B,T,3,H,W]) -> (X_hat, mu, logvar)``."""

    def __init__(self, config, dirs=None, train=False, dtype="bf16"):
        super().__init__()
        # ``train=True`` in the reference additionally builds the GAN discriminators and the VGG perceptual loss
        # (first_stage_motion_model.py:171-263); here training covers the L1 + KL terms.
        self.config, self.dirs, self.dtype = config, dirs, dtype
        arch = dict(config["architecture"])
        self.full_sequence = bool(config["training"].get("full_sequence", False))
        arch.update(img_size=config["data"]["spatial_size"][0], max_frames=config["data"]["max_frames"], full_seq=self.full_sequence)
        self.use_motion_bias = bool(arch.get("motion_bias", False))
        self.enc_motion = ResNetMotionEncoder(arch, dtype=dtype)
        self.n_layers = arch["n_gru_layers"]
        self.rnn = ConvGRU(arch["z_dim"], arch["z_dim"], 3, self.n_layers, dtype=dtype)
        if self.use_motion_bias:
            s = arch["min_spatial_size"]
            self.motion_bias = nn.Parameter(torch.randn(1, arch["z_dim"], s, s))
        self.gen = SpadeCondConvDecoder(arch, dtype=dtype)

    @torch.no_grad()
    def decode(self, motion, start_frame, length):
        """GRU unroll + per-frame SPADE decoding (first_stage_motion_model.py:503-520, second_stage_video.py:361-382)."""
        B = start_frame.shape[0]
        dt = self.dtype
        m = K.from_nchw(motion.float(), dt)
        hidden = [m] * self.n_layers
        if self.use_motion_bias:
            in_rnn = K.from_nchw(self.motion_bias.detach().float().expand(B, -1, -1, -1).contiguous(), dt)
        else:
            in_rnn = m
        mods = self.gen.modulations(start_frame.float())
        # The ConvGRU is sequential in time; the decoder is not: every frame is a function of its own hidden state and of the
        # clip's start frame.  In evaluation mode (no power iteration between the reference's per-frame calls) the frames are
        # therefore decoded as ONE batch of length x B samples -- 15x fewer launches, and the 8x8 ... 32x32 stages become
        # GEMMs of a useful height (c5: 63.6 -> see DESIGN.md section 7).  Chunked so that row offsets stay below 2^31 elements.
        from . import first_stage_train as FT
        seq = None
        if FT._GRU_NATIVE and FT.gru_native_ok(self.rnn, in_rnn, m, dt):
            # the ConvGRU steps issued natively (csrc/gru.hip): one call instead of length x n_layers cells of Python launches
            weights = []
            for c in self.rnn.cells:
                weights += [torch.cat([c.update_gate.weight, c.reset_gate.weight], 0), torch.cat([c.update_gate.bias, c.reset_gate.bias]),
                            c.out_gate.weight, c.out_gate.bias]
            geom = (B, length, self.n_layers, in_rnn.C, m.C, m.dhw[1], m.dhw[2])
            seq, _ = FT.gru_unroll_forward(weights, in_rnn.t.contiguous(), m.t.contiguous(), geom, dt)
            rows = B * m.S
        else:
            hs = []
            for _ in range(length):
                hidden = self.rnn.run(in_rnn, hidden)
                hs.append(hidden[-1])
        size = self.config["data"]["spatial_size"][0]
        per = max(1, min(length, (1 << 30) // max(1, B * size * size * 64)))
        outs = []
        for t0 in range(0, length, per):
            n = min(per, length - t0)
            if seq is not None:
                h_all = K.CL(seq[t0 * rows:(t0 + n) * rows], B * n, m.dhw, m.C)
            else:
                part = hs[t0:t0 + per]
                h_all = K.CL(torch.cat([h.t for h in part], 0), B * len(part), part[0].dhw, part[0].C)
            part = [None] * n
            y = self.gen.run(h_all, mods, frames=len(part))
            outs.append(y if len(part) > 1 else y.unsqueeze(1))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    @torch.no_grad()
    def forward(self, X, eps=None):
        X_in = X if self.full_sequence else X[:, 1:]
        motion, mu, logvar = self.enc_motion(X_in.transpose(1, 2), eps=eps)
        return self.decode(motion, X[:, 0], X.shape[1] - 1), mu, logvar

    # ---- validation loop (first_stage_motion_model.py:303-367) -------------------------------------------
    def attach_fvd(self, i3d=None, dtype="f32", vgg_loss=None):
        """The reference builds ``self.FVD`` / ``self.vgg_loss`` in __init__ when ``train=True`` (:52-63); here they are attached
        explicitly (their checkpoints are separate files)."""
        from .fvd import FVD
        self.FVD = FVD(n_samples=self.config["logging"]["n_samples_fvd"], i3d=i3d, dtype=dtype)
        self.vgg_loss = vgg_loss
        self.features_fvd_fake, self.features_fvd_true, self.fvd_features_fake_x0, self.fvd_features_true_x0 = [], [], [], []
        self.logged = {}
        return self.FVD

    @torch.no_grad()
    def validation_step(self, batch, batch_id):
        """Reconstruction, ``val/rec_loss`` (mean |X[:, 1:] - X_hat|, ``ipoke_l1_pair``), ``val/vgg_loss`` when a VGGLoss is attached, and
        the clips kept (on the device) for the epoch's FVD; ``ssim-val`` / ``psnr-val`` (first_stage_motion_model.py:323-324) on the device
        (ipoke_amd/metrics.py).  lpips-val needs the lpips package's pretrained network and is not part of this path."""
        from ._lib import check, ptr
        X = batch["images"].float()
        X_hat, mu, logvar = self(X)
        tgt = X[:, 1:].contiguous()
        W = X.shape[-1]
        M = tgt.numel() // W
        loss = torch.zeros(1, device=X.device)
        scratch = torch.empty(M, W, device=X.device)
        check(_lib.lib().ipoke_l1_pair(ptr(X_hat.contiguous()), W, ptr(tgt), W, M, W, 1.0 / tgt.numel(), ptr(loss), ptr(scratch), W, _lib.F32,
                                       _lib.current_stream()))
        self.logged["val/rec_loss"] = loss[0]
        if getattr(self, "vgg_loss", None) is not None:
            self.logged["val/vgg_loss"] = self.vgg_loss(tgt.reshape(-1, *X.shape[2:]), X_hat.reshape(-1, *X_hat.shape[2:]))
        from . import metrics
        both = metrics.psnr_ssim(X_hat.reshape(-1, *X_hat.shape[2:]), tgt.reshape(-1, *X_hat.shape[2:]))
        # logged with on_epoch=True in the reference: the epoch value is the mean over the validation batches (running sums on the device)
        acc = self.__dict__.setdefault("_val_metric_acc", [0, None, None])
        acc[0] += 1
        acc[1] = both[1] if acc[1] is None else acc[1] + both[1]
        acc[2] = both[0] if acc[2] is None else acc[2] + both[0]
        self.logged["ssim-val"], self.logged["psnr-val"] = acc[1] / acc[0], acc[2] / acc[0]
        self.logged["ssim-val_step"], self.logged["psnr-val_step"] = both[1], both[0]
        if getattr(self, "FVD", None) is not None and batch_id <= int(self.config["logging"]["n_samples_fvd"] / X_hat.size(0)):
            self.features_fvd_fake.append(X_hat)
            self.features_fvd_true.append(tgt)
            self.fvd_features_fake_x0.append(torch.cat([X[:, 0].unsqueeze(1), X_hat], dim=1))
            self.fvd_features_true_x0.append(X)
        return X_hat

    def validation_epoch_end(self, outputs=None):
        from .fvd import calculate_FVD
        bs = self.config["logging"]["bs_i3d"]
        fvd = calculate_FVD(self.FVD.i3d, torch.cat(self.features_fvd_fake), torch.cat(self.features_fvd_true), batch_size=bs)
        fvd_x0 = calculate_FVD(self.FVD.i3d, torch.cat(self.fvd_features_fake_x0), torch.cat(self.fvd_features_true_x0), batch_size=bs)
        self.logged["FVD-val"], self.logged["FVD-val-x0"] = fvd, fvd_x0
        for lst in (self.features_fvd_fake, self.features_fvd_true, self.fvd_features_fake_x0, self.fvd_features_true_x0):
            lst.clear()
        self.__dict__.pop("_val_metric_acc", None)          # the next epoch's ssim-val / psnr-val means start afresh
        return fvd, fvd_x0

    def training_loss(self, X, eps, w_l1=10.0, w_kl=1e-7, power_iteration=None):
        """Differentiable forward + ``w_l1 * L1 + w_kl * KL`` (first_stage_motion_model.py:263-276 without the GAN / VGG
        terms).  Returns (loss, X_hat, mu, logvar); ``loss.backward()`` fills ``.grad`` of every parameter."""
        from . import first_stage_train
        return first_stage_train.first_stage_forward_loss(self, X, eps, w_l1, w_kl, power_iteration)

    def invalidate_operands(self):
        """Drop the cached inference weight operands (after an optimiser step or a state-dict load)."""
        for m in self.modules():
            m.__dict__.pop("_opcache", None)
        from . import first_stage_train
        first_stage_train.clear_operand_cache()
