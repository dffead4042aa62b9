"""FVD evaluation on the MI355X: the I3D network, the 224x224 preprocess, the activation moments and the Frechet distance.

Mirrors reference utils/metrics.py (same function names, arguments and state-dict keys):
    I3D (:1000-1099), Unit3Dpy (:854-936), MaxPool3dTFPadding (:939-960), Mixed (:963-998)
    preprocess (:787-800), get_activations (:679-731), calculate_activation_statistics (:743-771),
    calculate_frechet_distance (:622-676), calculate_FVD (:774-781), compute_activations (:783-785), FVD (:335-380)

Execution (no PyTorch math on the path): every Unit3Dpy is one ``ipoke_conv_forward`` launch with the eval-mode BatchNorm folded
into the weight operand and the bias and the ReLU in the epilogue; the four branches of a Mixed block write straight into
their channel ranges of the block's output rows (no concatenation pass); TF-SAME padding is the convolution's front padding
plus the zero fill of out-of-range taps.  The stem's 7x7x7 window over 3 channels runs as 49 taps of 21 channels (a 7-pixel
run of a channels-last fp32 row is contiguous), reading the resized clip in place.  Pools, the resize, the conditional
de-normalisation and the float64 moments are the kernels of ``csrc/eval.hip``.  The matrix square root of the 400x400
covariance product stays ``scipy.linalg.sqrtm`` on the host, exactly as in the reference.
"""
import numpy as np
import torch
import torch.nn as nn
from scipy import linalg

from . import _lib, nn as hnn, ops
from ._lib import check, ptr
from .nn import CL

MIXED = (("mixed_3b", 192, (64, 96, 128, 16, 32, 32)), ("mixed_3c", 256, (128, 128, 192, 32, 96, 64)),
         ("mixed_4b", 480, (192, 96, 208, 16, 48, 64)), ("mixed_4c", 512, (160, 112, 224, 24, 64, 64)),
         ("mixed_4d", 512, (128, 128, 256, 24, 64, 64)), ("mixed_4e", 512, (112, 144, 288, 32, 64, 64)),
         ("mixed_4f", 528, (256, 160, 320, 32, 128, 128)), ("mixed_5b", 832, (256, 160, 320, 32, 128, 128)),
         ("mixed_5c", 832, (384, 192, 384, 48, 128, 128)))
TF_BN_EPS = 1e-3


def _same(extent, k, s, use_remainder):
    """(front, back) zero padding and the output extent of one axis under TF SAME as the reference applies it: the remainder
    rule only on the time axis (metrics.py:831-833), max(k - s, 0) elsewhere; pooling uses ceil_mode."""
    r = extent % s if use_remainder else 0
    along = max(k - (r if r else s), 0)
    front = along // 2
    return front, along - front


class Unit3Dpy(nn.Module):
    """Parameter container with the reference's names (conv3d.weight [, conv3d.bias], batch3d.*)."""

    def __init__(self, in_channels, out_channels, kernel_size=(1, 1, 1), stride=(1, 1, 1), activation="relu", use_bias=False, use_bn=True):
        super().__init__()
        self.kernel_size, self.stride, self.relu = tuple(kernel_size), tuple(stride), activation == "relu"
        self.conv3d = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=use_bias)
        if use_bn:
            self.batch3d = nn.BatchNorm3d(out_channels, eps=TF_BN_EPS)


class _Slot(nn.Module):
    """Parameter-free position 0 of branch_3 (the pool), so that the conv is ``branch_3.1`` as in the reference."""


class Mixed(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        c = out_channels
        self.branch_0 = Unit3Dpy(in_channels, c[0])
        self.branch_1 = nn.Sequential(Unit3Dpy(in_channels, c[1]), Unit3Dpy(c[1], c[2], (3, 3, 3)))
        self.branch_2 = nn.Sequential(Unit3Dpy(in_channels, c[3]), Unit3Dpy(c[3], c[4], (3, 3, 3)))
        self.branch_3 = nn.Sequential(_Slot(), Unit3Dpy(in_channels, c[5]))
        self.widths = (c[0], c[2], c[4], c[5])


class I3D(nn.Module):
    """Kinetics I3D (RGB), frozen, evaluation mode only.  ``dtype``: "f32" (the reference's arithmetic; default) or "bf16"."""

    def __init__(self, num_classes=400, modality="rgb", dropout_prob=0, name="inception", dtype="f32", device="cuda"):
        super().__init__()
        if modality != "rgb":
            raise ValueError("{} not among known modalities [rgb]".format(modality))
        if dropout_prob != 0:
            raise NotImplementedError("the FVD path runs the network in eval mode: dropout is the identity")
        self.name, self.num_classes, self.modality, self.dtype = name, num_classes, modality, dtype
        self.conv3d_1a_7x7 = Unit3Dpy(3, 64, (7, 7, 7), (2, 2, 2))
        self.conv3d_2b_1x1 = Unit3Dpy(64, 64)
        self.conv3d_2c_3x3 = Unit3Dpy(64, 192, (3, 3, 3))
        for nm, cin, c in MIXED:
            setattr(self, nm, Mixed(cin, c))
        self.conv3d_0c_1x1 = Unit3Dpy(1024, num_classes, activation=None, use_bias=True, use_bn=False)
        for p in self.parameters():
            p.requires_grad = False
        self._operands = None
        self.gflop = 0.0
        self.register_load_state_dict_post_hook(lambda m, _: setattr(m, "_operands", None))
        self.to(device)
        self.eval()

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("I3D is a frozen feature extractor here (reference: self.i3d.eval())")
        return super().train(False)

    # ------------------------------------------------------------------ operands
    @torch.no_grad()
    def _operand(self, unit):
        """[Cout][taps*Kc] weight of the compute dtype with the BatchNorm scale folded in, fp32 bias, Kc."""
        w = unit.conv3d.weight.float()
        cout = w.shape[0]
        if hasattr(unit, "batch3d"):
            bn = unit.batch3d
            inv = torch.rsqrt(bn.running_var.float() + TF_BN_EPS) * bn.weight.float()
            bias = bn.bias.float() - bn.running_mean.float() * inv
        else:
            inv = None
            bias = unit.conv3d.bias.float()
        if unit.kernel_size == (7, 7, 7):
            # stem: tap = (kd, kh), channel = kw*3 + c -- a 7-pixel run of the channels-last fp32 clip is contiguous
            kc = hnn.round_up(21, hnn.e16(self.dtype))
            w5 = w.permute(0, 2, 3, 4, 1).reshape(cout, 49, 21)
            if inv is not None:
                w5 = w5 * inv.view(-1, 1, 1)
            buf = torch.zeros(cout, 49, kc, dtype=torch.float32, device=w.device)
            buf[:, :, :21] = w5
            op = buf.reshape(cout, 49 * kc).to(ops.torch_dtype(self.dtype)).contiguous()
        else:
            op, kc = hnn.weight_operand(w, self.dtype, scale=None if inv is None else inv.view(-1, 1, 1))
        return op, bias.contiguous(), kc

    def _ops(self):
        if self._operands is None:
            self._operands = {name: self._operand(m) for name, m in self.named_modules() if isinstance(m, Unit3Dpy)}
        return self._operands

    # ------------------------------------------------------------------ layers
    def _unit(self, name, x, out=None, c_coff=0):
        unit = self.get_submodule(name)
        op, bias, kc = self._ops()[name]
        k, s = unit.kernel_size, unit.stride
        cout = unit.conv3d.out_channels
        pads = [_same(e, kk, ss, i == 0) for i, (e, kk, ss) in enumerate(zip(x.dhw, k, s))]
        odhw = tuple(-(-e // ss) for e, ss in zip(x.dhw, s))
        d = ops.conv_desc(x.N, x.dhw, odhw, k, s, tuple(p[0] for p in pads))
        ld = x.t.shape[1]
        D, H, W = x.dhw
        d.A = x.t.data_ptr(); d.a_f32 = 0
        d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = D * H * W * ld, H * W * ld, W * ld, ld, 1
        d.Kc_real = kc; d.Kc = kc
        assert x.C == unit.conv3d.in_channels and ld >= kc
        return self._launch(d, op, bias, cout, unit.relu, x.N, odhw, out, c_coff)

    def _launch(self, d, op, bias, cout, relu, N, odhw, out, c_coff, out_f32=False):
        M = N * odhw[0] * odhw[1] * odhw[2]
        self.gflop += 2e-9 * M * cout * d.kd * d.kh * d.kw * d.Kc_real       # algorithmic multiply-adds of this launch (bench.py)
        d.W = op.data_ptr(); d.ldw = op.shape[1]; d.Nout = cout
        d.bias = bias.data_ptr(); d.act = _lib.ACT_RELU if relu else _lib.ACT_NONE
        if out is None:
            ldc = cout if out_f32 else hnn.round_up(cout, hnn.e16(self.dtype))
            out = torch.empty(M, ldc, dtype=torch.float32 if out_f32 else ops.torch_dtype(self.dtype), device=op.device)
        d.C = out.data_ptr(); d.ldc = out.shape[1]; d.c_coff = c_coff; d.c_f32 = int(out_f32)
        ops.conv_forward(d, self.dtype)
        return CL(out, N, odhw, cout)

    def _stem(self, clip, N, T, H, W, pad_l, Wp):
        """conv3d_1a_7x7 over the padded channels-last fp32 clip [N*T*H*Wp, 3]."""
        op, bias, kc = self._ops()["conv3d_1a_7x7"]
        pt = _same(T, 7, 2, True)[0]
        ph = _same(H, 7, 2, False)[0]
        assert pad_l == _same(W, 7, 2, False)[0]
        odhw = (-(-T // 2), -(-H // 2), -(-W // 2))
        assert Wp >= 2 * (odhw[2] - 1) + 7
        d = ops.conv_desc(N, (T, H, Wp - 6), odhw, (7, 7, 1), (2, 2, 2), (pt, ph, 0))
        d.A = clip.data_ptr(); d.a_f32 = 1
        d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = T * H * Wp * 3, H * Wp * 3, Wp * 3, 3, 1
        d.a_coff = 0; d.Kc_real = 21; d.Kc = kc
        return self._launch(d, op, bias, 64, True, N, odhw, None, 0)

    def _pool(self, x, k, s):
        pads = [_same(e, kk, ss, i == 0) for i, (e, kk, ss) in enumerate(zip(x.dhw, k, s))]
        # MaxPool3d(ceil_mode=True) on the padded extent; a window may not start in the overhang (PyTorch's rule)
        odhw = []
        for e, kk, ss, (pf, pb) in zip(x.dhw, k, s, pads):
            pe = e + pf + pb
            o = -(-(pe - kk) // ss) + 1
            if (o - 1) * ss >= pe:
                o -= 1
            odhw.append(o)
        dims = torch.tensor([x.N, x.C, *x.dhw, *odhw, *k, *s, *(p[0] for p in pads), *(e + p[1] for e, p in zip(x.dhw, pads))], dtype=torch.int32)
        y = torch.empty(x.N * odhw[0] * odhw[1] * odhw[2], x.t.shape[1], dtype=x.t.dtype, device=x.t.device)
        check(_lib.lib().ipoke_pool3d_same(dims.numpy().ctypes.data, ptr(x.t), x.t.shape[1], ptr(y), y.shape[1], ops._dt(self.dtype),
                                           _lib.current_stream()))
        return CL(y, x.N, tuple(odhw), x.C)

    def _mixed(self, name, x):
        m = self.get_submodule(name)
        widths = m.widths
        M = x.M
        out = torch.empty(M, sum(widths), dtype=x.t.dtype, device=x.t.device)
        self._unit(name + ".branch_0", x, out, 0)
        self._unit(name + ".branch_1.1", self._unit(name + ".branch_1.0", x), out, widths[0])
        self._unit(name + ".branch_2.1", self._unit(name + ".branch_2.0", x), out, widths[0] + widths[1])
        self._unit(name + ".branch_3.1", self._pool(x, (3, 3, 3), (1, 1, 1)), out, widths[0] + widths[1] + widths[2])
        return CL(out, x.N, x.dhw, sum(widths))

    # ------------------------------------------------------------------ forward
    def _clip_rows(self, src, N, T, H, W, Ho, Wo, strides, minval=None):
        """Strided fp32 clips -> padded channels-last rows [N*T*Ho*Wp, 3] (resized when (Ho, Wo) != (H, W))."""
        pad_l, pad_r = _same(Wo, 7, 2, False)
        pad_r = max(pad_r, 2 * (-(-Wo // 2) - 1) + 7 - pad_l - Wo)
        Wp = pad_l + Wo + pad_r
        dst = torch.empty(N * T * Ho * Wp, 3, dtype=torch.float32, device=src.device)
        s_n, s_f, s_c, s_h, s_w = strides
        check(_lib.lib().ipoke_video_to_cl(ptr(src), s_n, s_f, s_c, s_h, s_w, N, T, 3, H, W, ptr(dst), Ho, Wo, pad_l, pad_r, ptr(minval),
                                           _lib.current_stream()))
        return dst, pad_l, pad_r, Wp

    def _trunk(self, clip, N, T, H, W, pad_l, Wp, taps=None):
        def tap(name, x):
            if taps is not None:
                taps[name] = hnn.to_nchw(x, self.dtype)
            return x
        x = tap("conv1a", self._stem(clip, N, T, H, W, pad_l, Wp))
        x = tap("pool2a", self._pool(x, (1, 3, 3), (1, 2, 2)))
        x = self._unit("conv3d_2c_3x3", self._unit("conv3d_2b_1x1", x))
        x = tap("pool3a", self._pool(x, (1, 3, 3), (1, 2, 2)))
        x = tap("mixed_3c", self._mixed("mixed_3c", tap("mixed_3b", self._mixed("mixed_3b", x))))
        x = tap("pool4a", self._pool(x, (3, 3, 3), (2, 2, 2)))
        for nm in ("mixed_4b", "mixed_4c", "mixed_4d", "mixed_4e", "mixed_4f"):
            x = self._mixed(nm, x)
        x = tap("mixed_4f", x)
        x = tap("pool5a", self._pool(x, (2, 2, 2), (2, 2, 2)))
        x = tap("mixed_5c", self._mixed("mixed_5c", self._mixed("mixed_5b", x)))
        # AvgPool3d((2, 7, 7), 1) + logits conv + mean over the remaining time steps: the conv is linear, so pool first
        Tf, Hf, Wf = x.dhw
        if Tf < 2 or (Hf, Wf) != (7, 7):
            raise ValueError(f"I3D head: AvgPool3d((2, 7, 7)) needs a final map of >= 2 x 7 x 7, got {x.dhw} (input must be 224 x 224)")
        wt = torch.zeros(Tf)
        for t in range(Tf - 1):
            wt[t] += 0.5; wt[t + 1] += 0.5
        w = (wt / (Tf - 1)).repeat_interleave(49) / 49.0
        w = w.to(x.t.device)
        pooled = torch.empty(N, x.t.shape[1], dtype=x.t.dtype, device=x.t.device)
        check(_lib.lib().ipoke_pool_rows_weighted(ptr(x.t), x.t.shape[1], ptr(pooled), pooled.shape[1], N, Tf * 49, x.C, ptr(w),
                                                  ops._dt(self.dtype), _lib.current_stream()))
        op, bias, kc = self._ops()["conv3d_0c_1x1"]
        d = ops.conv_desc(N, (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0))
        ld = pooled.shape[1]
        d.A = pooled.data_ptr(); d.a_f32 = 0
        d.a_sn, d.a_sd, d.a_sh, d.a_sw, d.a_sc = ld, ld, ld, ld, 1
        d.Kc_real = kc; d.Kc = kc
        return self._launch(d, op, bias, self.num_classes, False, N, (1, 1, 1), None, 0, out_f32=True).t

    @torch.no_grad()
    def forward(self, inp, taps=None):
        """inp fp32 [N, 3, T, H, W] (any strides) -> (softmax, logits), as the reference's I3D.forward."""
        _lib.require_gpu()
        N, C, T, H, W = inp.shape
        assert C == 3 and inp.dtype == torch.float32 and inp.is_cuda
        s = inp.stride()
        clip, pad_l, pad_r, Wp = self._clip_rows(inp, N, T, H, W, H, W, (s[0], s[2], s[1], s[3], s[4]))
        logits = self._trunk(clip, N, T, H, W, pad_l, Wp, taps)
        return torch.softmax(logits, 1), logits

    @torch.no_grad()
    def logits_of_videos(self, videos, size=(224, 224), minval=None):
        """videos fp32 [N, T, 3, h, w] in the data range: resize to ``size`` (+ de-normalise when *minval < 0) and run the trunk."""
        N, T, C, h, w = videos.shape
        s = videos.stride()
        clip, pad_l, pad_r, Wp = self._clip_rows(videos, N, T, h, w, size[0], size[1], (s[0], s[1], s[2], s[3], s[4]))
        if minval is not None:
            check(_lib.lib().ipoke_denorm_if_negative(ptr(clip), N * T * size[0], size[1], pad_l, pad_r, 3, ptr(minval), _lib.current_stream()))
        return self._trunk(clip, N, T, size[0], size[1], pad_l, Wp)


# ---------------------------------------------------------------------------------------------- reference-named functions
def denorm(x):
    return (x + 1.0) / 2.0


def _resized_min(videos, size=(224, 224), chunk=64):
    """Device scalar: min over the bilinearly resized data set (what ``data.min() < 0`` tests in the reference's preprocess)."""
    minval = torch.empty(1, dtype=torch.float32, device="cuda")
    check(_lib.lib().ipoke_min_reset(ptr(minval), _lib.current_stream()))
    for i in range(0, videos.shape[0], chunk):
        v = videos[i:i + chunk].cuda().float()
        N, T, C, h, w = v.shape
        s = v.stride()
        check(_lib.lib().ipoke_video_to_cl(ptr(v), s[0], s[1], s[2], s[3], s[4], N, T, C, h, w, None, size[0], size[1], 0, 0, ptr(minval),
                                           _lib.current_stream()))
    return minval


def preprocess(data_gen, data_orig):
    """metrics.py:787-800 -- materialises the resized [N, T, 3, 224, 224] tensors (use calculate_FVD for the streaming form)."""
    out = []
    for data in (data_gen, data_orig):
        data = data.cuda().float()
        N, T, C, h, w = data.shape
        minval = _resized_min(data)
        s = data.stride()
        dst = torch.empty(N * T * 224 * 224, C, dtype=torch.float32, device=data.device)
        check(_lib.lib().ipoke_video_to_cl(ptr(data), s[0], s[1], s[2], s[3], s[4], N, T, C, h, w, ptr(dst), 224, 224, 0, 0, None, _lib.current_stream()))
        check(_lib.lib().ipoke_denorm_if_negative(ptr(dst), N * T * 224, 224, 0, 0, C, ptr(minval), _lib.current_stream()))
        out.append(dst.view(N, T, 224, 224, C).permute(0, 1, 4, 2, 3))
    return out[0], out[1]


def get_activations(data, model, batch_size=50, cuda=True, verbose=False):
    """Logits [n_used, 400] (numpy float64, as the reference's ``pred_arr``) of preprocessed clips [n, T, 3, 224, 224]."""
    return _activations(model, data, batch_size, None).double().cpu().numpy()


def _activations(model, data, batch_size, minval, resize=None):
    n = data.size(0)
    batch_size = min(batch_size, n)
    n_batches = n // batch_size                                      # a trailing partial batch is ignored (metrics.py:709-711)
    out = torch.empty(n_batches * batch_size, model.num_classes, dtype=torch.float32, device="cuda")
    for i in range(n_batches):
        batch = data[i * batch_size:(i + 1) * batch_size].cuda().float()
        if resize is None:
            out[i * batch_size:(i + 1) * batch_size] = model(batch.permute(0, 2, 1, 3, 4))[1]
        else:
            out[i * batch_size:(i + 1) * batch_size] = model.logits_of_videos(batch, resize, minval)
    return out


def _moments_device(act):
    n, D = act.shape
    mu = torch.empty(D, dtype=torch.float64, device=act.device)
    sigma = torch.empty(D, D, dtype=torch.float64, device=act.device)
    ws = torch.empty(n + 1, dtype=torch.int32, device=act.device)
    check(_lib.lib().ipoke_activation_moments(ptr(act.contiguous()), n, D, ptr(mu), ptr(sigma), ptr(ws), _lib.current_stream()))
    return mu.cpu().numpy(), sigma.cpu().numpy()


def calculate_moments(data):
    return _moments_device(torch.as_tensor(data, dtype=torch.float32, device="cuda"))


def calculate_activation_statistics(data, model, batch_size=50, cuda=True, verbose=False):
    return _moments_device(_activations(model, data, batch_size, None))


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """metrics.py:622-676 (host, float64, scipy.linalg.sqrtm as in the reference)."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        print("fid calculation produces singular product; adding %s to diagonal of cov estimates" % eps)
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def calculate_FVD(model, data_gen, data_orig, batch_size, cuda=True):
    """metrics.py:774-781.  Streaming: the minimum of each resized data set is taken in a first pass (resize kernel without
    output), then every batch is resized, de-normalised and pushed through I3D without materialising the 224x224 data set."""
    stats = []
    for data in (data_gen, data_orig):
        minval = _resized_min(data)
        stats.append(_moments_device(_activations(model, data, batch_size, minval, resize=(224, 224))))
    return calculate_frechet_distance(stats[0][0], stats[0][1], stats[1][0], stats[1][1])


def compute_activations(model, data_gen, data_orig, batch_size, cuda=True):
    acts = []
    for data in (data_orig, data_gen):
        acts.append(_activations(model, data, batch_size, _resized_min(data), resize=(224, 224)).double().cpu().numpy())
    return acts[0], acts[1]


class FVD:
    """metrics.py:335-380 without the Lightning ``Metric`` machinery: update() with generated / target clips until n_samples are
    collected, compute() -> Frechet distance of the collected logits."""

    def __init__(self, n_samples, i3d=None, dtype="f32"):
        self.n_max_samples = n_samples
        self.i3d = I3D(400, "rgb", dtype=dtype) if i3d is None else i3d
        self.features_fake, self.features_real, self.n_samples = [], [], 0

    def load_i3d(self, path):
        self.i3d.load_state_dict(torch.load(path, map_location="cpu"))

    def update(self, pred, target, cuda=True):
        if self.n_samples < self.n_max_samples:
            bs = pred.size(0)
            self.features_fake.append(_activations(self.i3d, pred, bs, _resized_min(pred), resize=(224, 224)))
            self.features_real.append(_activations(self.i3d, target, bs, _resized_min(target), resize=(224, 224)))
            self.n_samples += bs

    def compute(self):
        real = torch.cat(self.features_real)[:self.n_max_samples]
        fake = torch.cat(self.features_fake)[:self.n_max_samples]
        m_real, s_real = _moments_device(real)
        m_fake, s_fake = _moments_device(fake)
        return calculate_frechet_distance(m_fake, s_fake, m_real, s_real)

    def reset(self):
        self.features_fake, self.features_real, self.n_samples = [], [], 0
