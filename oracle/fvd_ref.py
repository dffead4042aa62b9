"""CPU restatement of the reference's FVD evaluation (TEST INFRASTRUCTURE ONLY -- imported by tests/ and
oracle/make_goldens.py; the product path never touches it).

Follows reference utils/metrics.py:
    preprocess (:787-800)              bilinear resize (align_corners=True) of every frame to 224x224, then (x + 1) / 2 when
                                       the *whole tensor's* minimum is negative
    get_padding_shape (:813-842)       TensorFlow "SAME" padding: along an axis pad_along = max(k - s, 0) -- or
                                       max(k - (T mod s), 0) on the time axis when T is not a multiple of the stride --
                                       split as (pad_along // 2, rest)
    Unit3Dpy (:854-936)                zero pad -> Conv3d (no bias) -> BatchNorm3d(eps 1e-3) in eval mode -> ReLU
    MaxPool3dTFPadding (:939-960)      ZERO padding (not -inf) as above, then MaxPool3d(ceil_mode=True)
    Mixed (:963-998)                   four branches concatenated: 1x1 | 1x1 -> 3x3x3 | 1x1 -> 3x3x3 | pool(3, 1) -> 1x1
    I3D (:1000-1099)                   stem 7x7x7/2, pool (1,3,3)/(1,2,2), 1x1, 3x3x3, pool, Mixed 3b 3c, pool 3/2, Mixed 4b-4f,
                                       pool 2/2, Mixed 5b 5c, AvgPool3d((2,7,7), 1), 1x1 conv with bias to 400 logits,
                                       mean over the remaining time steps
    get_activations (:679-731)         logits of ``videos.permute(0, 2, 1, 3, 4)`` in batches; a trailing partial batch is dropped
    calculate_activation_statistics (:743-771)  rows with no finite entry dropped, mean and np.cov (float64, N - 1)
    calculate_frechet_distance (:622-676)       |mu1 - mu2|^2 + tr S1 + tr S2 - 2 tr sqrtm(S1 S2)
    calculate_FVD (:774-781)
State-dict keys equal the reference's (``conv3d_1a_7x7.conv3d.weight``, ``….batch3d.running_mean``,
``mixed_3b.branch_1.0.conv3d.weight``, ``mixed_3b.branch_3.1.…``, ``conv3d_0c_1x1.conv3d.bias``).  Parity is pinned by
oracle/make_goldens.py job g10 (logits, intermediate maps and the FVD value of this module asserted against the reference's
I3D / calculate_FVD on the same weights and inputs).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy import linalg

# name, out-channel sextet of the Mixed blocks, in order of execution between the pools
MIXED = (("mixed_3b", 192, (64, 96, 128, 16, 32, 32)), ("mixed_3c", 256, (128, 128, 192, 32, 96, 64)),
         ("mixed_4b", 480, (192, 96, 208, 16, 48, 64)), ("mixed_4c", 512, (160, 112, 224, 24, 64, 64)),
         ("mixed_4d", 512, (128, 128, 256, 24, 64, 64)), ("mixed_4e", 512, (112, 144, 288, 32, 64, 64)),
         ("mixed_4f", 528, (256, 160, 320, 32, 128, 128)), ("mixed_5b", 832, (256, 160, 320, 32, 128, 128)),
         ("mixed_5c", 832, (384, 192, 384, 48, 128, 128)))
BN_EPS = 1e-3


def same_pad(extent, k, s):
    """(front, back) zero padding of one axis under TF SAME."""
    r = extent % s
    along = max(k - (r if r else s), 0)
    return along // 2, along - along // 2


def pad_same(x, k, s, time_only_mod=True):
    """Zero-pad [N,C,T,H,W] for a window k / stride s.  The reference looks at the remainder on the time axis only
    (metrics.py:831-833: ``depth_mod = (idx == 0) and mod``); H and W always use max(k - s, 0)."""
    T = x.shape[2]
    pt = same_pad(T, k[0], s[0])
    ph = (max(k[1] - s[1], 0) // 2, max(k[1] - s[1], 0) - max(k[1] - s[1], 0) // 2)
    pw = (max(k[2] - s[2], 0) // 2, max(k[2] - s[2], 0) - max(k[2] - s[2], 0) // 2)
    return F.pad(x, (pw[0], pw[1], ph[0], ph[1], pt[0], pt[1]))


class Unit(nn.Module):
    def __init__(self, cin, cout, k=(1, 1, 1), s=(1, 1, 1), bn=True, bias=False, relu=True):
        super().__init__()
        self.k, self.s, self.relu = k, s, relu
        self.conv3d = nn.Conv3d(cin, cout, k, stride=s, bias=bias)
        if bn:
            self.batch3d = nn.BatchNorm3d(cout, eps=BN_EPS)

    def forward(self, x):
        y = self.conv3d(pad_same(x, self.k, self.s))
        if hasattr(self, "batch3d"):
            bn = self.batch3d
            inv = torch.rsqrt(bn.running_var + BN_EPS) * bn.weight
            y = y * inv.view(1, -1, 1, 1, 1) + (bn.bias - bn.running_mean * inv).view(1, -1, 1, 1, 1)
        return F.relu(y) if self.relu else y


def pool_same(x, k, s):
    return F.max_pool3d(pad_same(x, k, s), k, s, ceil_mode=True)


class Pool(nn.Module):
    """Parameter-free slot so that ``branch_3.1`` is the conv, as in the reference's nn.Sequential."""

    def forward(self, x):
        return pool_same(x, (3, 3, 3), (1, 1, 1))


class Mixed(nn.Module):
    def __init__(self, cin, c):
        super().__init__()
        self.branch_0 = Unit(cin, c[0])
        self.branch_1 = nn.Sequential(Unit(cin, c[1]), Unit(c[1], c[2], (3, 3, 3)))
        self.branch_2 = nn.Sequential(Unit(cin, c[3]), Unit(c[3], c[4], (3, 3, 3)))
        self.branch_3 = nn.Sequential(Pool(), Unit(cin, c[5]))

    def forward(self, x):
        return torch.cat([self.branch_0(x), self.branch_1(x), self.branch_2(x), self.branch_3(x)], 1)


class I3D(nn.Module):
    def __init__(self, num_classes=400):
        super().__init__()
        self.conv3d_1a_7x7 = Unit(3, 64, (7, 7, 7), (2, 2, 2))
        self.conv3d_2b_1x1 = Unit(64, 64)
        self.conv3d_2c_3x3 = Unit(64, 192, (3, 3, 3))
        for name, cin, c in MIXED:
            setattr(self, name, Mixed(cin, c))
        self.conv3d_0c_1x1 = Unit(1024, num_classes, bn=False, bias=True, relu=False)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x, taps=None):
        """x [N,3,T,224,224] -> logits [N, classes]; ``taps`` (a dict) receives named intermediate maps."""
        def tap(name, t):
            if taps is not None:
                taps[name] = t
            return t
        x = tap("conv1a", self.conv3d_1a_7x7(x))
        x = tap("pool2a", pool_same(x, (1, 3, 3), (1, 2, 2)))
        x = self.conv3d_2c_3x3(self.conv3d_2b_1x1(x))
        x = tap("pool3a", pool_same(x, (1, 3, 3), (1, 2, 2)))
        x = tap("mixed_3c", self.mixed_3c(tap("mixed_3b", self.mixed_3b(x))))
        x = tap("pool4a", pool_same(x, (3, 3, 3), (2, 2, 2)))
        for name in ("mixed_4b", "mixed_4c", "mixed_4d", "mixed_4e", "mixed_4f"):
            x = getattr(self, name)(x)
        x = tap("mixed_4f", x)
        x = tap("pool5a", pool_same(x, (2, 2, 2), (2, 2, 2)))
        x = tap("mixed_5c", self.mixed_5c(self.mixed_5b(x)))
        x = F.avg_pool3d(x, (2, 7, 7), (1, 1, 1))
        x = self.conv3d_0c_1x1(x)
        return x.squeeze(3).squeeze(3).mean(2)


def preprocess(videos):
    """[N,T,3,h,w] -> [N,T,3,224,224] in [0,1] (reference :787-800; the de-normalisation test is on the whole tensor)."""
    N, T = videos.shape[:2]
    v = F.interpolate(videos.reshape(-1, *videos.shape[2:]).float(), mode="bilinear", size=(224, 224), align_corners=True)
    v = v.reshape(N, T, 3, 224, 224)
    return (v + 1.0) / 2.0 if v.min() < 0 else v


def activations(net, videos, batch_size):
    n = videos.shape[0]
    batch_size = min(batch_size, n)
    out = np.empty(((n // batch_size) * batch_size, 400))
    with torch.no_grad():
        for i in range(n // batch_size):
            out[i * batch_size:(i + 1) * batch_size] = net(videos[i * batch_size:(i + 1) * batch_size].permute(0, 2, 1, 3, 4)).numpy()
    return out


def moments(act):
    act = act[np.flatnonzero(np.logical_not(np.isnan(act)).any(axis=-1))]
    return act.mean(axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, s1, mu2, s2, eps=1e-6):
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(s1.shape[0]) * eps
        covmean = linalg.sqrtm((s1 + off).dot(s2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean)


def fvd(net, videos_gen, videos_orig, batch_size):
    m1, s1 = moments(activations(net, preprocess(videos_gen), batch_size))
    m2, s2 = moments(activations(net, preprocess(videos_orig), batch_size))
    return frechet_distance(m1, s1, m2, s2)
