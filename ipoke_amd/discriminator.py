"""First-stage temporal discriminator on the HIP kernels (first part): the 3-D ResNet-18 of
reference models/modules/discriminators/patchgan_3d.py:171-304 as ``first_stage_motion_model.py:66`` builds it
(``resnet(config=d_t, spatial_size, sequence_length)``), forward and backward.

Parameter names equal the reference's state dict (``conv1.weight_orig / weight_u / weight_v``, ``gn1.weight``,
``layer2.0.downsample.0.weight_orig``, ``fc.weight`` ...).  Every layer runs through the differentiable ops of
``first_stage_train`` (implicit-GEMM Conv3d with the spectral-norm kernels, GroupNorm + ReLU (+ residual)), plus
MaxPool3d / AvgPool3d from vae_train.hip.  Supported: predictions and the four feature maps, hinge / feature-matching /
generator losses, gradients w.r.t. the parameters and w.r.t. the input clip.

NOT yet supported: the gradient penalty ``gp2`` (:285-294) -- it differentiates the input gradient again
(``create_graph=True``), i.e. needs the backward kernels themselves as differentiable ops; DESIGN.md §9 holds the plan.
``gp2`` raises instead of silently falling back to PyTorch.
"""
import math

import torch
import torch.nn as nn

from . import _lib, first_stage as FS, first_stage_train as T, nn as K, ops
from ._lib import check, ptr


class _MaxPool3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_t, N, C, dhw, k, s, p, dtype):
        odhw = tuple((i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip(dhw, k, s, p))
        dims = (torch.tensor([N, C, *dhw, *odhw, *k, *s, *p], dtype=torch.int32)).numpy()
        Mo = N * odhw[0] * odhw[1] * odhw[2]
        y = torch.empty(Mo, x_t.shape[1], dtype=x_t.dtype, device=x_t.device)
        if x_t.shape[1] > C:
            y[:, C:].zero_()
        idx = torch.empty(Mo, C, dtype=torch.int32, device=x_t.device)
        check(_lib.lib().ipoke_maxpool3d_fwd(dims.ctypes.data, ptr(x_t), x_t.shape[1], ptr(y), y.shape[1], ptr(idx), ops._dt(dtype),
                                             _lib.current_stream()))
        ctx.save_for_backward(idx)
        ctx.args = (dims, x_t.shape, dtype)
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _):
        (idx,) = ctx.saved_tensors
        dims, xshape, dtype = ctx.args
        dy = dy.contiguous()
        dx = torch.empty(xshape, dtype=dy.dtype, device=dy.device)
        check(_lib.lib().ipoke_maxpool3d_bwd(dims.ctypes.data, ptr(dy), dy.shape[1], ptr(idx), ptr(dx), dx.shape[1], ops._dt(dtype),
                                             _lib.current_stream()))
        return dx, None, None, None, None, None, None, None


class _AvgPoolRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_t, G, S, C, dtype):
        y = torch.empty(G, x_t.shape[1], dtype=x_t.dtype, device=x_t.device)
        check(_lib.lib().ipoke_avgpool_rows(ptr(x_t), x_t.shape[1], ptr(y), y.shape[1], G, S, C, ops._dt(dtype), _lib.current_stream()))
        ctx.args = (G, S, C, dtype, x_t.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        G, S, C, dtype, xshape = ctx.args
        dy = dy.contiguous()
        dx = torch.empty(xshape, dtype=dy.dtype, device=dy.device)
        check(_lib.lib().ipoke_avgpool_rows_bwd(ptr(dy), dy.shape[1], ptr(dx), dx.shape[1], G, S, C, ops._dt(dtype), _lib.current_stream()))
        return dx, None, None, None, None


def max_pool3d(x, k, s, p, dtype):
    y, _ = _MaxPool3dFn.apply(x.t, x.N, x.C, tuple(x.dhw), tuple(k), tuple(s), tuple(p), dtype)
    odhw = tuple((i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip(x.dhw, k, s, p))
    return K.CL(y, x.N, odhw, x.C)


class _Block(nn.Module):
    """patchgan_3d.py:43-63."""

    def __init__(self, cin, planes, stride=1, stride_t=1, downsample=False):
        super().__init__()
        st = (stride_t, stride, stride)
        self.conv1 = FS._Conv(cin, planes, 3, st, 1, bias=False, snorm=True, dims=3)
        self.bn1 = FS._Norm("group", planes)
        self.conv2 = FS._Conv(planes, planes, 3, 1, 1, bias=False, snorm=True, dims=3)
        self.bn2 = FS._Norm("group", planes)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(FS._Conv(cin, planes, 3, st, 1, bias=False, snorm=True, dims=3), FS._Norm("group", planes))


def _block(blk, x, dt, pit):
    out = T.norm(blk.bn1, T.conv(blk.conv1, x, dt, w=T.effective_weight(blk.conv1, pit)), dt, act=_lib.ACT_RELU)
    out = T.conv(blk.conv2, out, dt, w=T.effective_weight(blk.conv2, pit))
    res = x
    if blk.downsample is not None:
        ds = blk.downsample
        res = T.norm(ds[1], T.conv(ds[0], x, dt, w=T.effective_weight(ds[0], pit)), dt)
    return T.norm(blk.bn2, out, dt, act=_lib.ACT_RELU, res=res)


class TemporalDiscriminator(nn.Module):
    """``resnet(config=d_t, spatial_size=..., sequence_length=...)`` (ResNet-18 layout, patchgan_3d.py:16-20, 171-258)."""

    def __init__(self, spatial_size, config, dtype="bf16", layers=(2, 2, 2, 2)):
        super().__init__()
        self.dtype = dtype
        self.bce_loss = bool(config.get("bce_loss", False))
        self.gp_weight = float(config.get("gp_weight", 0.0))
        self.num_classes = int(config.get("num_classes", 1))
        if self.bce_loss:
            raise NotImplementedError("bce_loss discriminators are not on the path (config/first_stage.yaml:71 uses the hinge loss)")
        stride_t = 1 if config.get("patch_temp_disc", False) else 2
        self.conv1 = FS._Conv(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), bias=False, snorm=True, dims=3)
        self.gn1 = FS._Norm("group", 64)
        spec = [(64, 1, 1), (128, 1, stride_t), (256, 2, stride_t), (512, 2, stride_t)]
        inplanes = 64
        for li, ((planes, s, st), n) in enumerate(zip(spec, layers), start=1):
            blocks = [_Block(inplanes, planes, s, st, downsample=(s != 1 or inplanes != planes))]
            inplanes = planes
            blocks += [_Block(planes, planes) for _ in range(1, n)]
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.last = int(math.ceil(spatial_size / 16))
        self.fc = nn.Linear(512, self.num_classes, bias=False)

    def forward(self, x, power_iteration=None):
        """x [B, 3, T, H, W] fp32 on the GPU -> (pred [B, T' * num_classes] fp32, [four feature maps as channels-last CL])."""
        _lib.require_gpu()
        dt = self.dtype
        pit = self.training if power_iteration is None else bool(power_iteration)
        B, C, Tn, H, W = x.shape
        w1 = T.effective_weight(self.conv1, pit)
        if x.requires_grad:          # generator side: the clip itself needs a gradient -> enter through a channels-last copy
            xcl = T._pad_cols(x.permute(0, 2, 3, 4, 1).reshape(-1, C), K.round_up(C, K.e16(dt)), dt)
            h = T.conv(self.conv1, K.CL(xcl, B, (Tn, H, W), C), dt, w=w1)
        else:
            x = x.float()
            st = (x.stride(0), x.stride(1), x.stride(2), x.stride(3), x.stride(4))
            h = T.conv(self.conv1, None, dt, src=(x, B, C, (Tn, H, W), st), w=w1)
        h = T.norm(self.gn1, h, dt, act=_lib.ACT_RELU)
        h = max_pool3d(h, (3, 3, 3), (1, 2, 2), (1, 1, 1), dt)
        fmaps = []
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                h = _block(blk, h, dt, pit)
            fmaps.append(h)
        D, Hh, Ww = h.dhw
        if Hh != self.last or Ww != self.last:
            raise ValueError(f"AvgPool3d((1, {self.last}, {self.last})) does not cover the {Hh}x{Ww} map: only the global case is built")
        p = _AvgPoolRowsFn.apply(h.t, B * D, Hh * Ww, h.C, dt)
        meta = dict(N=B, dhw=(D, 1, 1), cin=h.C, cout=self.num_classes, k=(1, 1, 1), stride=(1, 1, 1), pad=(0, 0, 0), transposed=False,
                    out_pad=(0, 0, 0), dtype=dt, act=_lib.ACT_NONE, out_f32=True, src=None)
        rows = T._ConvFn.apply(p, self.fc.weight.view(self.num_classes, h.C, 1, 1, 1), None, meta)       # [B*D, >= num_classes] fp32
        pred = rows[:, :self.num_classes].float().reshape(B, D * self.num_classes)
        return pred, fmaps

    # ---- losses (patchgan_3d.py:263-304): scalars over [B, T'] predictions; the feature-matching term runs over the maps
    @staticmethod
    def loss(pred, real):
        return torch.relu(1.0 - pred).mean() if real else torch.relu(1.0 + pred).mean()

    @staticmethod
    def fmap_loss(fmap1, fmap2):
        tot = 0.0
        for a, b in zip(fmap1, fmap2):
            tot = tot + (a.t[:, :a.C].float() - b.t[:, :b.C].float()).abs().mean()
        return tot / len(fmap1)

    def gp2(self, pred, x):
        raise NotImplementedError("gradient penalty (patchgan_3d.py:285-294) needs double backward through the HIP convolution / "
                                  "GroupNorm kernels: not built yet (DESIGN.md section 9)")


class PatchDiscriminator(nn.Module):
    """The first-stage 2-D PatchGAN ``PatchDiscriminator(config=d_s)`` (reference patchgan.py:368-470 with the default
    InstanceNorm2d): spectral-normalised 4x4 convolutions with bias -- stride 2 three times, then stride 1 and the 1-channel
    head (maps of 15 and 14 pixels at 128 px: the convolution kernels decode non-power-of-two extents by division) --,
    InstanceNorm + LeakyReLU(0.2) after the inner ones.  ``forward(x[N,3,H,W]) -> (pred [N,1,h,w] fp32, [feature maps])``."""

    def __init__(self, config, dtype="bf16"):
        super().__init__()
        self.dtype = dtype
        self.bce_loss = bool(config.get("bce_loss", False))
        self.gp_weight = float(config.get("gp_weight", 0.0))
        if self.bce_loss:
            raise NotImplementedError("bce_loss discriminators are not on the path (config/first_stage.yaml:79 uses the hinge loss)")
        if config.get("deep_disc", False):
            raise NotImplementedError("deep_disc is not used by the shipped configs")
        n_layers = int(config.get("n_layers", 3))
        ndf = 64
        self.in_conv = FS._Conv(3, ndf, 4, 2, 1, bias=True, snorm=True, dims=2)
        self.layers, self.norms = nn.ModuleList(), nn.ModuleList()
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            self.layers.append(FS._Conv(ndf * prev, ndf * mult, 4, 2, 1, bias=True, snorm=True, dims=2))
            self.norms.append(FS._Norm("in", ndf * mult))
        prev, mult = mult, min(2 ** n_layers, 8)
        self.layers.append(FS._Conv(ndf * prev, ndf * mult, 4, 1, 1, bias=True, snorm=True, dims=2))
        self.norms.append(FS._Norm("in", ndf * mult))
        self.out_conv = FS._Conv(ndf * mult, 1, 4, 1, 1, bias=True, snorm=True, dims=2)

    def forward(self, x, power_iteration=None):
        _lib.require_gpu()
        dt = self.dtype
        pit = self.training if power_iteration is None else bool(power_iteration)
        N, C, H, W = x.shape
        w0 = T.effective_weight(self.in_conv, pit)
        if x.requires_grad:
            xcl = T._pad_cols(x.permute(0, 2, 3, 1).reshape(-1, C), K.round_up(C, K.e16(dt)), dt)
            h = T.conv(self.in_conv, K.CL(xcl, N, (1, H, W), C), dt, act=_lib.ACT_LRELU02, w=w0)
        else:
            x = x.float()
            st = (x.stride(0), x.stride(1), 0, x.stride(2), x.stride(3))
            h = T.conv(self.in_conv, None, dt, act=_lib.ACT_LRELU02, src=(x, N, C, (1, H, W), st), w=w0)
        fmap = []
        for cv, nm in zip(self.layers, self.norms):
            h = T.norm(nm, T.conv(cv, h, dt, w=T.effective_weight(cv, pit)), dt, act=_lib.ACT_LRELU02)
            fmap.append(h)
        o = T.conv(self.out_conv, h, dt, out_f32=True, w=T.effective_weight(self.out_conv, pit))
        pred = o.t[:, :1].float().reshape(N, o.dhw[1], o.dhw[2], 1).permute(0, 3, 1, 2)
        return pred, fmap

    loss = staticmethod(TemporalDiscriminator.loss)
    fmap_loss = staticmethod(TemporalDiscriminator.fmap_loss)

    def gp(self, pred, x):
        raise NotImplementedError("gradient penalty (patchgan.py:438-447) needs double backward: not built (DESIGN.md section 9); "
                                  "config/first_stage.yaml:80 sets d_s.gp_weight = 0")
