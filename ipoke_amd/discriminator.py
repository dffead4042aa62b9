"""First-stage temporal discriminator on the HIP kernels (first part): the 3-D ResNet-18 of
reference models/modules/discriminators/patchgan_3d.py:171-304 as ``first_stage_motion_model.py:66`` builds it
(``resnet(config=d_t, spatial_size, sequence_length)``), forward and backward.

Parameter names equal the reference's state dict (``conv1.weight_orig / weight_u / weight_v``, ``gn1.weight``,
``layer2.0.downsample.0.weight_orig``, ``fc.weight`` ...).  Every layer runs through the differentiable ops of
``first_stage_train`` (implicit-GEMM Conv3d with the spectral-norm kernels, GroupNorm + ReLU (+ residual)), plus
MaxPool3d / AvgPool3d from vae_train.hip.  Supported: predictions and the four feature maps, hinge / feature-matching /
generator losses, gradients w.r.t. the parameters and w.r.t. the input clip.

The gradient penalty ``gp2`` (:285-294, ``create_graph=True`` in the reference) is evaluated forward-over-reverse instead of by
double backward: g = d sum(pred) / dx from an ordinary backward pass gives the value mean_b |g_b|^2; with v = (2/B) g held
constant, d reg / d theta = d/d theta [D_v sum(pred)], and D_v is propagated by a tangent pass that runs next to a primal pass
-- the same convolution kernels on the tangent (convolutions are linear), the primal ReLU masks and max-pool selections, and
``ipoke_groupnorm_jvp`` for the one second-order term (GroupNorm statistics).  One more ordinary backward pass then yields the
parameter gradients.  No PyTorch double backward is involved.
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


class _L1MeanFn(torch.autograd.Function):
    """mean |a - b| over the real channels of two channels-last maps; the gradient flows into ``a`` only (``b`` = the true
    clip's map, which the generator step does not differentiate)."""

    @staticmethod
    def forward(ctx, a_t, b_t, C, dtype):
        M = a_t.shape[0]
        loss = torch.zeros(1, device=a_t.device)
        grad = torch.empty_like(a_t)
        check(_lib.lib().ipoke_l1_pair(ptr(a_t), a_t.shape[1], ptr(b_t), b_t.shape[1], M, C, 1.0 / (M * C), ptr(loss), ptr(grad), grad.shape[1],
                                       ops._dt(dtype), _lib.current_stream()))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, d):
        (grad,) = ctx.saved_tensors
        return grad * d.to(grad.dtype), None, None, None


def max_pool3d(x, k, s, p, dtype):
    y, _ = _MaxPool3dFn.apply(x.t, x.N, x.C, tuple(x.dhw), tuple(k), tuple(s), tuple(p), dtype)
    odhw = tuple((i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip(x.dhw, k, s, p))
    return K.CL(y, x.N, odhw, x.C)


class _NormJvpFn(torch.autograd.Function):
    """Tangent of GroupNorm (+ residual, + ReLU mask of the primal output): see ipoke_groupnorm_jvp."""

    @staticmethod
    def forward(ctx, x_t, xd_t, y_t, resd_t, gamma, meta):
        dt = meta["dtype"]
        yd = torch.zeros_like(xd_t) if xd_t.shape[1] > meta["C"] else torch.empty_like(xd_t)
        g32 = None if gamma is None else gamma.detach().float().contiguous()
        ws = torch.empty(meta["N"] * meta["G"] * 8, dtype=torch.float32, device=x_t.device)
        check(_lib.lib().ipoke_groupnorm_jvp(ptr(x_t), x_t.shape[1], ptr(xd_t), xd_t.shape[1], ptr(y_t), y_t.shape[1],
                                             None if resd_t is None else ptr(resd_t), 0 if resd_t is None else resd_t.shape[1],
                                             ptr(yd), yd.shape[1], None if g32 is None else ptr(g32), meta["N"], meta["S"], meta["C"],
                                             meta["G"], meta["act"], 1e-5, ptr(ws), ops._dt(dt), _lib.current_stream()))
        ctx.save_for_backward(x_t, xd_t, y_t, g32)
        ctx.meta, ctx.has_res = meta, resd_t is not None
        return yd

    @staticmethod
    def backward(ctx, q):
        x_t, xd_t, y_t, g32 = ctx.saved_tensors
        m = ctx.meta
        q = q.contiguous()
        dxd = torch.zeros_like(xd_t) if xd_t.shape[1] > m["C"] else torch.empty_like(xd_t)
        dx = torch.zeros_like(x_t) if x_t.shape[1] > m["C"] else torch.empty_like(x_t)
        dres = (torch.zeros_like(xd_t) if xd_t.shape[1] > m["C"] else torch.empty_like(xd_t)) if ctx.has_res else None
        dgamma = None if g32 is None else torch.zeros(m["C"], dtype=torch.float32, device=q.device)
        ws = torch.empty(m["N"] * m["G"] * 8, dtype=torch.float32, device=q.device)
        check(_lib.lib().ipoke_groupnorm_jvp_bwd(ptr(x_t), x_t.shape[1], ptr(xd_t), xd_t.shape[1], ptr(y_t), y_t.shape[1], ptr(q), q.shape[1],
                                                 ptr(dxd), dxd.shape[1], ptr(dx), dx.shape[1], None if dres is None else ptr(dres),
                                                 0 if dres is None else dres.shape[1], None if dgamma is None else ptr(dgamma),
                                                 None if g32 is None else ptr(g32), m["N"], m["S"], m["C"], m["G"], m["act"], 1e-5, ptr(ws),
                                                 ops._dt(m["dtype"]), _lib.current_stream()))
        return dx, dxd, None, dres, dgamma, None


class _GatherRowsFn(torch.autograd.Function):
    """Tangent of MaxPool3d: the primal pass's selection applied to the tangent (backward = the pooling backward)."""

    @staticmethod
    def forward(ctx, xd_t, idx, dims, C, dtype):
        Mo = idx.shape[0]
        y = torch.empty(Mo, xd_t.shape[1], dtype=xd_t.dtype, device=xd_t.device)
        check(_lib.lib().ipoke_gather_rows(ptr(xd_t), xd_t.shape[1], ptr(idx), ptr(y), y.shape[1], Mo, C, ops._dt(dtype), _lib.current_stream()))
        ctx.save_for_backward(idx)
        ctx.args = (dims, xd_t.shape, dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dims, xshape, dtype = ctx.args
        dy = dy.contiguous()
        dx = torch.empty(xshape, dtype=dy.dtype, device=dy.device)
        check(_lib.lib().ipoke_maxpool3d_bwd(dims.ctypes.data, ptr(dy), dy.shape[1], ptr(idx), ptr(dx), dx.shape[1], ops._dt(dtype),
                                             _lib.current_stream()))
        return dx, None, None, None, None


def _norm_pair(mod, xp, xd, dt, act, res=None, resd=None):
    """(GroupNorm(+res)(+act) of the primal, its tangent)."""
    yp = T.norm(mod, xp, dt, act=act, res=res)
    meta = dict(N=xp.N, S=xp.S, C=xp.C, G=mod.groups, dtype=dt, act=act)
    yd = _NormJvpFn.apply(xp.t, xd.t, yp.t, None if resd is None else resd.t, mod.weight if mod.kind == "group" else None, meta)
    return yp, K.CL(yd, xp.N, xp.dhw, xp.C)


def _block_pair(blk, xp, xd, dt):
    w1, w2 = T.effective_weight(blk.conv1, False), T.effective_weight(blk.conv2, False)
    op, od = _norm_pair(blk.bn1, T.conv(blk.conv1, xp, dt, w=w1), T.conv(blk.conv1, xd, dt, w=w1), dt, _lib.ACT_RELU)
    op, od = T.conv(blk.conv2, op, dt, w=w2), T.conv(blk.conv2, od, dt, w=w2)
    rp, rd = xp, xd
    if blk.downsample is not None:
        ds = blk.downsample
        wd = T.effective_weight(ds[0], False)
        rp, rd = _norm_pair(ds[1], T.conv(ds[0], xp, dt, w=wd), T.conv(ds[0], xd, dt, w=wd), dt, _lib.ACT_NONE)
    return _norm_pair(blk.bn2, op, od, dt, _lib.ACT_RELU, res=rp, resd=rd)


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

    def fmap_loss(self, fmap1, fmap2):
        """mean over the maps of mean |f1 - f2| (:297-304); differentiated w.r.t. ``fmap1`` (the fake clip's maps) only, as the
        generator step uses it (the discriminator gradients of that step are discarded by the next ``zero_grad``)."""
        tot = 0.0
        for a, b in zip(fmap1, fmap2):
            tot = tot + _L1MeanFn.apply(a.t, b.t.detach(), a.C, self.dtype)
        return tot / len(fmap1)

    def forward_with_tangent(self, x, v):
        """Primal forward of ``x`` and, next to it, the derivative of every activation along the input direction ``v`` (both
        [B, 3, T, H, W] fp32, neither needs a gradient).  Returns (pred, pred_dot).  Spectral-norm buffers are not iterated."""
        _lib.require_gpu()
        dt = self.dtype
        B, C, Tn, H, W = x.shape
        x, v = x.float().contiguous(), v.float().contiguous()
        st = (x.stride(0), x.stride(1), x.stride(2), x.stride(3), x.stride(4))
        w1 = T.effective_weight(self.conv1, False)
        hp = T.conv(self.conv1, None, dt, src=(x, B, C, (Tn, H, W), st), w=w1)
        hd = T.conv(self.conv1, None, dt, src=(v, B, C, (Tn, H, W), st), w=w1)
        hp, hd = _norm_pair(self.gn1, hp, hd, dt, _lib.ACT_RELU)
        k, s_, p_ = (3, 3, 3), (1, 2, 2), (1, 1, 1)
        yp, idx = _MaxPool3dFn.apply(hp.t, hp.N, hp.C, tuple(hp.dhw), k, s_, p_, dt)
        odhw = tuple((i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip(hp.dhw, k, s_, p_))
        dims = torch.tensor([hp.N, hp.C, *hp.dhw, *odhw, *k, *s_, *p_], dtype=torch.int32).numpy()
        yd = _GatherRowsFn.apply(hd.t, idx, dims, hp.C, dt)
        hp, hd = K.CL(yp, B, odhw, hp.C), K.CL(yd, B, odhw, hp.C)
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                hp, hd = _block_pair(blk, hp, hd, dt)
        D, Hh, Ww = hp.dhw
        meta = dict(N=B, dhw=(D, 1, 1), cin=hp.C, cout=self.num_classes, k=(1, 1, 1), stride=(1, 1, 1), pad=(0, 0, 0), transposed=False,
                    out_pad=(0, 0, 0), dtype=dt, act=_lib.ACT_NONE, out_f32=True, src=None)
        wfc = self.fc.weight.view(self.num_classes, hp.C, 1, 1, 1)
        outs = []
        for h in (hp, hd):
            pooled = _AvgPoolRowsFn.apply(h.t, B * D, Hh * Ww, h.C, dt)
            rows = T._ConvFn.apply(pooled, wfc, None, dict(meta))
            outs.append(rows[:, :self.num_classes].float().reshape(B, D * self.num_classes))
        return outs[0], outs[1]

    def gp2(self, x):
        """Gradient penalty of patchgan_3d.py:285-294 on the clip ``x``: a scalar whose value is mean_b |d sum(pred)/dx|_b^2 and
        whose gradient w.r.t. the parameters is that of the penalty (forward-over-reverse, see the module docstring).  Unlike
        the reference's ``gp2(pred, x)`` it runs its own forward passes (the current spectral-norm buffers, no iteration)."""
        xg = x.detach().float().requires_grad_(True)
        pred, _ = self.forward(xg, power_iteration=False)
        (g,) = torch.autograd.grad(pred.sum(), xg)
        B = x.shape[0]
        reg = g.pow(2).reshape(B, -1).sum(1).mean()
        _, pdot = self.forward_with_tangent(x.detach(), (2.0 / B) * g)
        tsum = pdot.sum()
        return tsum + (reg - tsum).detach()


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
    fmap_loss = TemporalDiscriminator.fmap_loss

    def gp(self, pred, x):
        raise NotImplementedError("gradient penalty (patchgan.py:438-447) needs double backward: not built (DESIGN.md section 9); "
                                  "config/first_stage.yaml:80 sets d_s.gp_weight = 0")
