"""CPU oracle for the iPOKE conditional flow — TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch fp32 restatement of the second-stage normalizing flow of
CompVis/ipoke, written from the behaviour of the reference (file:line cited
per function) and pinned against golden vectors produced by the reference's own
modules (``oracle/make_goldens.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_vs_golden.py``).  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package ``ipoke_amd`` never does (it fails loudly without its HIP library).

State-dict keys are identical to the reference's, so a reference checkpoint
loads here with ``strict=True``.

Everything is expressed on NCHW tensors with stock torch ops; no attempt is
made to be fast.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# elementary invertible layers
# --------------------------------------------------------------------------
class ActNorm2dFlow(nn.Module):
    """Per-channel affine with data-dependent init.

    Reference: models/modules/INN/macow2.py:476-540.  Quirks kept on purpose:
    * ``reset_parameters`` draws ``log_scale ~ N(0, 0.05)`` (:485-487);
    * the init measures statistics of ``x*exp(log_scale)+bias`` but then
      *overwrites* ``log_scale``/``bias`` (:531-539), unbiased std, eps 1e-6;
    * inverse divides by ``exp(log_scale) + 1e-8`` (:520).
    """

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.log_scale = nn.Parameter(torch.empty(channels, 1, 1).normal_(0.0, 0.05))
        self.bias = nn.Parameter(torch.zeros(channels, 1, 1))
        self.register_buffer("initialized", torch.tensor(0, dtype=torch.uint8))

    @torch.no_grad()
    def data_init(self, x):
        y = x * self.log_scale.exp() + self.bias
        flat = y.transpose(0, 1).reshape(self.channels, -1)
        mean = flat.mean(dim=1).view(-1, 1, 1)
        std = flat.std(dim=1).view(-1, 1, 1)          # unbiased
        inv = 1.0 / (std + 1e-6)
        self.log_scale.copy_(inv.log())
        self.bias.copy_(-mean * inv)

    def forward(self, x, reverse=False):
        if reverse:
            return (x - self.bias) / (self.log_scale.exp() + 1e-8)
        if int(self.initialized) == 0:
            self.data_init(x)
            self.initialized.fill_(1)
        hw = x.shape[2] * x.shape[3]
        y = x * self.log_scale.exp() + self.bias
        logdet = self.log_scale.sum() * hw * torch.ones(x.shape[0], dtype=x.dtype, device=x.device)
        return y, logdet


class Shuffle(nn.Module):
    """Fixed random channel permutation (the flow's "invertible 1x1 conv").

    Reference: models/modules/INN/flow_blocks.py:314-326.  Index buffers are
    int64 and part of the state dict; the forward log-det is the integer 0.
    """

    def __init__(self, channels):
        super().__init__()
        idx = torch.randperm(channels)
        self.register_buffer("forward_shuffle_idx", idx)
        self.register_buffer("backward_shuffle_idx", torch.argsort(idx))

    def forward(self, x, reverse=False):
        if reverse:
            return x[:, self.backward_shuffle_idx]
        return x[:, self.forward_shuffle_idx], 0


class InvertibleConvLU1d(nn.Module):
    """LU-parametrised invertible 1x1 convolution.  Reference macow2.py:596-649 (selected by ``use1x1`` for the per-level
    ``shuffle_layers`` only: MultiScaleInternal :862; the priors' and steps' conv1x1 stay Shuffles, :551 is never passed
    use_1x1 and :1017 is hard-wired).  W = P (L * lmask + I) (U * umask + diag(sign_s exp(log_s))); log-det = H W sum(log_s);
    the inverse multiplies the three matrix inverses (torch.inverse in the reference)."""

    def __init__(self, nf):
        super().__init__()
        import numpy as np
        import scipy.linalg as alg
        self.nf = nf
        w_init = np.linalg.qr(np.random.randn(nf, nf))[0].astype(np.float32)
        p, l, u = alg.lu(w_init)
        s = np.diag(u)
        u = np.triu(u, k=1)
        lmask = np.tril(np.ones_like(w_init), -1)
        self.register_buffer("permutated", torch.FloatTensor(p))
        self.register_buffer("sign_s", torch.FloatTensor(np.sign(s)))
        self.register_buffer("lmask", torch.FloatTensor(lmask))
        self.register_buffer("umask", torch.FloatTensor(lmask.T.copy()))
        self.register_buffer("eye", torch.FloatTensor(np.eye(nf)))
        self.l = nn.Parameter(torch.FloatTensor(l))
        self.u = nn.Parameter(torch.FloatTensor(u))
        self.log_s = nn.Parameter(torch.FloatTensor(np.log(np.abs(s))))

    def matrices(self):
        wl = self.l * self.lmask + self.eye
        wu = self.u * self.umask + torch.diag(self.sign_s * torch.exp(self.log_s))
        return wl, wu

    def forward(self, x, reverse=False):
        wl, wu = self.matrices()
        if not reverse:
            w = self.permutated @ (wl @ wu)
            logdet = self.log_s.sum() * x.shape[2] * x.shape[3] * torch.ones(x.shape[0])
            return torch.einsum("ij,bjhw->bihw", w, x), logdet
        w = torch.inverse(wu) @ (torch.inverse(wl) @ torch.inverse(self.permutated))
        return torch.einsum("ij,bjhw->bihw", w, x)


def affine_params(raw):
    """(mu, scale) from the coupling net output; reference macow_utils.py:49-52 (alpha = 1)."""
    mu, s = raw.chunk(2, dim=1)
    return mu, torch.tanh(0.5 * s) + 1.0


def affine_fwd(z, mu, scale):
    """Reference macow_utils.py:54-59."""
    return scale * z + mu, scale.log().flatten(1).sum(dim=1)


def affine_inv(z, mu, scale):
    """Reference macow_utils.py:61-66 (note the +1e-12)."""
    return (z - mu) / (scale + 1e-12)


class _WNConvParams(nn.Module):
    """Holder giving the old-style ``weight_norm`` parameter names (bias, weight_g, weight_v)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        kh, kw = (k, k) if isinstance(k, int) else k
        self.bias = nn.Parameter(torch.zeros(cout))
        v = torch.empty(cout, cin, kh, kw).normal_(0.0, 0.05)
        # nn.utils.weight_norm initialises g to the per-output-channel norm of v
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(cout, 1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)

    def weight(self):
        v = self.weight_v
        return self.weight_g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1, 1)


class Conv2dWeightNorm(nn.Module):
    """Weight-normalised conv with data-dependent (here: zero) init.

    Reference: macow_utils.py:211-251.  With ``zero_init`` the init sets
    ``weight_g = 0/(std+1e-6) = 0`` and ``bias = -mean*0 = 0`` so every coupling
    starts as the identity.
    """

    def __init__(self, cin, cout, k, padding, zero_init=True):
        super().__init__()
        self.register_buffer("initialized", torch.tensor(0, dtype=torch.uint8))
        self.conv = _WNConvParams(cin, cout, k)
        self.padding = padding
        self.zero_init = zero_init

    @torch.no_grad()
    def data_init(self, x):
        y = F.conv2d(x, self.conv.weight(), self.conv.bias, padding=self.padding)
        flat = y.transpose(0, 1).reshape(y.shape[1], -1)
        mean, std = flat.mean(dim=1), flat.std(dim=1)
        inv = (0.0 if self.zero_init else 1.0) / (std + 1e-6)
        self.conv.weight_g.copy_(inv.view(-1, 1, 1, 1))
        self.conv.bias.copy_(-mean * inv)

    def forward(self, x):
        if int(self.initialized) == 0:
            self.data_init(x)
            self.initialized.fill_(1)
        return F.conv2d(x, self.conv.weight(), self.conv.bias, padding=self.padding)


# --------------------------------------------------------------------------
# masked convolutional flow
# --------------------------------------------------------------------------
class ShiftedConv2d(nn.Module):
    """Bias-free conv whose receptive field lies strictly above/below/left/right.

    Reference: macow_utils.py:446-499.  ``shifted=False`` is the plain valid
    conv used strip by strip in the analytic inverse.
    """

    def __init__(self, cin, cout, kernel_size, order):
        super().__init__()
        kh, kw = kernel_size
        self.order = order
        self.weight = nn.Parameter(torch.empty(cout, cin, kh, kw))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if order == "A":      # look at the kh rows above
            self.pad, self.crop = ((kw - 1) // 2, (kw - 1) // 2, kh, 0), (0, -1, 0, 0)
        elif order == "B":    # rows below
            self.pad, self.crop = ((kw - 1) // 2, (kw - 1) // 2, 0, kh), (1, 0, 0, 0)
        elif order == "C":    # kw columns to the left
            self.pad, self.crop = (kw, 0, (kh - 1) // 2, (kh - 1) // 2), (0, 0, 0, -1)
        elif order == "D":    # columns to the right
            self.pad, self.crop = (0, kw, (kh - 1) // 2, (kh - 1) // 2), (0, 0, 1, 0)
        else:
            raise ValueError(order)

    def forward(self, x, shifted=True):
        if shifted:
            x = F.pad(x, self.pad)
            t, b, l, r = self.crop
            x = x[:, :, t:x.shape[2] + b, l:x.shape[3] + r]
        return F.conv2d(x, self.weight)


class MCFBlock(nn.Module):
    """shift-conv -> concat cond -> ELU -> weight-normed 1x1.  Reference macow_utils.py:407-434."""

    def __init__(self, channels, out_channels, kernel_size, hidden, order, h_channels):
        super().__init__()
        self.shift_conv = ShiftedConv2d(channels, hidden, kernel_size, order)
        self.conv1x1 = Conv2dWeightNorm(hidden + h_channels, out_channels, 1, padding=0)

    def forward(self, x, h=None, shifted=True):
        c = self.shift_conv(x, shifted=shifted)
        if h is not None:
            c = torch.cat([c, h], dim=1)
        return self.conv1x1(F.elu(c))


class MaskedConvFlow(nn.Module):
    """Autoregressive affine flow over rows (A/B) or columns (C/D).

    Reference: macow2.py:25-288.  hidden = 4*C for C <= 96 (:36-40).  The inverse
    walks the 8 rows/columns sequentially (:174-288).
    """

    def __init__(self, channels, kernel_size, order, h_channels):
        super().__init__()
        hidden = 4 * channels if channels <= 96 else min(2 * channels, 512)
        self.kernel_size = tuple(kernel_size)
        self.order = order
        self.net = MCFBlock(channels, 2 * channels, self.kernel_size, hidden, order, h_channels)

    def forward(self, x, h=None, reverse=False):
        if reverse:
            return self.inverse(x, h)
        mu, scale = affine_params(self.net(x, h=h))
        return affine_fwd(x, mu, scale)

    def inverse(self, z, h):
        B, C, H, W = z.shape
        kh, kw = self.kernel_size
        along_rows = self.order in ("A", "B")
        backwards = self.order in ("B", "D")
        if along_rows:
            cw = kw // 2
            buf = z.new_zeros(B, C, H + kh, W + 2 * cw)
            for i in (reversed(range(H)) if backwards else range(H)):
                lo = i + 1 if backwards else i
                strip = buf[:, :, lo:lo + kh]
                hh = None if h is None else h[:, :, i:i + 1]
                raw = self.net(strip, h=hh, shifted=False).squeeze(2)
                mu, scale = affine_params(raw)
                buf[:, :, i if backwards else i + kh, cw:cw + W] = affine_inv(z[:, :, i], mu, scale)
            return buf[:, :, :H, cw:cw + W] if backwards else buf[:, :, kh:, cw:cw + W]
        ch = kh // 2
        buf = z.new_zeros(B, C, H + 2 * ch, W + kw)
        for j in (reversed(range(W)) if backwards else range(W)):
            lo = j + 1 if backwards else j
            strip = buf[:, :, :, lo:lo + kw]
            hh = None if h is None else h[:, :, :, j:j + 1]
            raw = self.net(strip, h=hh, shifted=False).squeeze(3)
            mu, scale = affine_params(raw)
            buf[:, :, ch:ch + H, j if backwards else j + kw] = affine_inv(z[:, :, :, j], mu, scale)
        return buf[:, :, ch:ch + H, :W] if backwards else buf[:, :, ch:ch + H, kw:]


# --------------------------------------------------------------------------
# NICE coupling
# --------------------------------------------------------------------------
class NICEConvBlock(nn.Module):
    """conv3x3 -> ELU -> conv1x1 -> [cat h] -> ELU -> weight-normed conv3x3.  Reference macow_utils.py:253-337
    (no norm, no attention, dropout p=0).  ``h_channels`` > 0 is ``condition_nice`` (macow2.py:1024-1060, 553): the
    conditioning map is concatenated behind conv2's output BEFORE the activation (macow_utils.py:328-332), so conv3 sees
    ELU(h) in its last ``h_channels`` input channels; ``cond_conv`` (a GatedConv2d on h first, :330-331) is not restated."""

    def __init__(self, cin, cout, hidden, h_channels=0):
        super().__init__()
        self.cond = h_channels > 0
        self.conv1 = nn.Conv2d(cin, hidden, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(hidden, hidden, 1, bias=False)
        self.conv3 = Conv2dWeightNorm(hidden + h_channels, cout, 3, padding=1)

    def forward(self, x, h=None):
        out = self.conv2(F.elu(self.conv1(x)))
        if h is not None and self.cond:
            out = torch.cat([out, h], dim=1)
        return self.conv3(F.elu(out))


class NICE2d(nn.Module):
    """Affine coupling with 'continuous' or 'skip' (even/odd) channel split.

    Reference: macow2.py:291-448.  out = C//factor channels are transformed,
    conditioned on the other C-out; ``up``: z1 (first/even part) conditions.
    'skip' with odd C falls back to 'continuous' (:304-307).
    """

    def __init__(self, channels, hidden, split_type="continuous", order="up", factor=2, h_channels=0):
        super().__init__()
        if split_type == "skip" and channels % factor == 1:
            split_type = "continuous"
        self.split_type = split_type
        self.up = order == "up"
        cout = channels // factor
        cin = channels - cout
        self.z1_channels = cin if self.up else cout
        self.net = NICEConvBlock(cin, 2 * cout, hidden, h_channels)

    def split(self, x):
        if self.split_type == "continuous":
            return x[:, :self.z1_channels], x[:, self.z1_channels:]
        return x[:, 0::2], x[:, 1::2]

    def unsplit(self, z1, z2):
        if self.split_type == "continuous":
            return torch.cat([z1, z2], dim=1)
        n = z1.shape[1]
        idx = torch.tensor([i // 2 if i % 2 == 0 else i // 2 + n for i in range(2 * n)], device=z1.device)
        return torch.cat([z1, z2], dim=1)[:, idx]

    def forward(self, x, h=None, reverse=False):
        z1, z2 = self.split(x)
        z, zp = (z1, z2) if self.up else (z2, z1)
        mu, scale = affine_params(self.net(z, h=h))
        if reverse:
            zp = affine_inv(zp, mu, scale)
            z1, z2 = (z, zp) if self.up else (zp, z)
            return self.unsplit(z1, z2)
        zp, logdet = affine_fwd(zp, mu, scale)
        z1, z2 = (z, zp) if self.up else (zp, z)
        return self.unsplit(z1, z2), logdet


# --------------------------------------------------------------------------
# composite blocks
# --------------------------------------------------------------------------
class MaCowUnit(nn.Module):
    """MCF(A) MCF(B) ActNorm MCF(C) MCF(D) ActNorm.  Reference macow2.py:925-995."""

    def __init__(self, channels, kernel_size, h_channels):
        super().__init__()
        kh, kw = kernel_size
        self.conv1 = MaskedConvFlow(channels, (kh, kw), "A", h_channels)
        self.conv2 = MaskedConvFlow(channels, (kh, kw), "B", h_channels)
        self.actnorm1 = ActNorm2dFlow(channels)
        self.conv3 = MaskedConvFlow(channels, (kw, kh), "C", h_channels)
        self.conv4 = MaskedConvFlow(channels, (kw, kh), "D", h_channels)
        self.actnorm2 = ActNorm2dFlow(channels)

    def forward(self, x, h=None, reverse=False):
        seq = [self.conv1, self.conv2, self.actnorm1, self.conv3, self.conv4, self.actnorm2]
        if reverse:
            for layer in reversed(seq):
                x = layer(x, reverse=True) if isinstance(layer, ActNorm2dFlow) else layer(x, h=h, reverse=True)
            return x
        total = 0
        for layer in seq:
            x, ld = layer(x) if isinstance(layer, ActNorm2dFlow) else layer(x, h=h)
            total = total + ld
        return x, total


class MaCowStep(nn.Module):
    """Reference macow2.py:999-1117."""

    def __init__(self, channels, kernel_size, hidden, h_channels, condition_nice=False):
        super().__init__()
        hn = h_channels if condition_nice else 0                   # macow2.py:1024-1060
        self.actnorm1 = ActNorm2dFlow(channels)
        self.conv1x1 = Shuffle(channels)
        self.units1 = nn.ModuleList([MaCowUnit(channels, kernel_size, h_channels) for _ in range(2)])
        self.coupling1_up = NICE2d(channels, hidden, "continuous", "up", h_channels=hn)
        self.coupling1_dn = NICE2d(channels, hidden, "continuous", "down", h_channels=hn)
        self.actnorm2 = ActNorm2dFlow(channels)
        self.units2 = nn.ModuleList([MaCowUnit(channels, kernel_size, h_channels) for _ in range(2)])
        self.coupling2_up = NICE2d(channels, hidden, "skip", "up", h_channels=hn)
        self.coupling2_dn = NICE2d(channels, hidden, "skip", "down", h_channels=hn)

    def _sequence(self):
        return ([self.actnorm1, self.conv1x1] + list(self.units1) + [self.coupling1_up, self.coupling1_dn,
                self.actnorm2] + list(self.units2) + [self.coupling2_up, self.coupling2_dn])

    def forward(self, x, h=None, reverse=False):
        seq = self._sequence()
        if reverse:
            for layer in reversed(seq):
                if isinstance(layer, (ActNorm2dFlow, Shuffle)):
                    x = layer(x, reverse=True)
                else:
                    x = layer(x, h=h, reverse=True)
            return x
        total = 0
        for layer in seq:
            if isinstance(layer, (ActNorm2dFlow, Shuffle)):
                x, ld = layer(x)
            else:
                x, ld = layer(x, h=h)
            total = total + ld
        return x, total


class MultiScalePrior(nn.Module):
    """Shuffle -> NICE(factor f, continuous, up) -> ActNorm on the last C/f channels.  Reference macow2.py:543-593."""

    def __init__(self, channels, hidden, factor, h_channels=0):
        super().__init__()
        self.conv1x1 = Shuffle(channels)
        self.coupling = NICE2d(channels, hidden, "continuous", "up", factor=factor, h_channels=h_channels)   # macow2.py:553
        self.z1_channels = self.coupling.z1_channels
        self.actnorm = ActNorm2dFlow(channels // factor)

    def forward(self, x, h=None, reverse=False):
        k = self.z1_channels
        if reverse:
            x = torch.cat([x[:, :k], self.actnorm(x[:, k:], reverse=True)], dim=1)
            x = self.coupling(x, h=h, reverse=True)
            return self.conv1x1(x, reverse=True)
        x, _ = self.conv1x1(x)
        x, ld = self.coupling(x, h=h)
        tail, ld2 = self.actnorm(x[:, k:])
        return torch.cat([x[:, :k], tail], dim=1), ld + ld2


class MultiScaleInternal(nn.Module):
    """Level loop with channel split-off.  Reference macow2.py:821-920."""

    def __init__(self, num_steps, channels, hidden, h_channels, factor, kernel_size, use_1x1=False, condition_nice=False):
        super().__init__()
        self.reshape = "none"
        self.layers = nn.ModuleList()
        self.priors = nn.ModuleList()
        self.shuffle_layers = nn.ModuleList()
        step = channels // factor
        for n in num_steps:
            self.layers.append(nn.ModuleList([MaCowStep(channels, kernel_size, hidden, h_channels, condition_nice) for _ in range(n)]))
            self.priors.append(MultiScalePrior(channels, hidden, factor, h_channels if condition_nice else 0))
            self.shuffle_layers.append(InvertibleConvLU1d(channels) if use_1x1 else Shuffle(channels))   # macow2.py:862
            channels -= step
            factor -= 1
        self.z_channels = channels

    def forward(self, x, h=None, reverse=False):
        if not reverse:
            logdet = x.new_zeros(x.shape[0])
            outs = []
            for steps, prior, shuffle in zip(self.layers, self.priors, self.shuffle_layers):
                for s in steps:
                    x, ld = s(x, h=h)
                    logdet = logdet + ld
                x, ld = prior(x, h=h)
                logdet = logdet + ld
                x, ld = shuffle(x)
                logdet = logdet + ld                      # integer 0 for a Shuffle (flow_blocks.py:324)
                outs.append(x[:, prior.z1_channels:])
                x = x[:, :prior.z1_channels]
            outs.append(x)
            return torch.cat(outs[::-1], dim=1), logdet
        pieces = []
        for prior in self.priors:
            pieces.append(x[:, prior.z1_channels:])
            x = x[:, :prior.z1_channels]
        for steps, prior, shuffle in zip(reversed(self.layers), reversed(self.priors), reversed(self.shuffle_layers)):
            x = torch.cat([x, pieces.pop()], dim=1)
            x = shuffle(x, reverse=True)
            x = prior(x, h=h, reverse=True)
            for s in reversed(steps):
                x = s(x, h=h, reverse=True)
        return x


class SupervisedMacowTransformer(nn.Module):
    """Reference models/modules/INN/INN.py:446-481 (keys consumed: :451-467)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        for unsupported in ("attention", "cond_conv"):
            if config.get(unsupported, False):
                raise NotImplementedError(f"oracle covers shipped configs only ({unsupported}=True is not one)")
        assert config["transform"] == "affine" and config["prior_transform"] == "affine"
        assert config["activation"] == "elu" and config["coupling_type"] == "conv"
        self.flow = MultiScaleInternal(config["num_steps"], config["flow_in_channels"], config["flow_mid_channels"],
                                       config["h_channels"], config["factor"], tuple(config["kernel_size"]),
                                       use_1x1=bool(config.get("use1x1", False)),
                                       condition_nice=bool(config.get("condition_nice", False)))

    def forward(self, x, cond, reverse=False):
        if reverse:
            return self.flow(x, cond, reverse=True)
        return self.flow(x, cond)


# --------------------------------------------------------------------------
# loss, LR schedule
# --------------------------------------------------------------------------
def nll(sample):
    """Reference loss.py:75-79 (spatial_mean=False): no log(2*pi) term."""
    return 0.5 * (sample ** 2).sum(dim=[1, 2, 3])


class FlowLoss(nn.Module):
    """Reference loss.py:6-31; draws randn_like(sample) for the logged reference value."""

    def __init__(self, spatial_mean=False, logdet_weight=1.0):
        super().__init__()
        self.spatial_mean = spatial_mean
        self.logdet_weight = logdet_weight

    def forward(self, sample, logdet):
        assert logdet.dim() == 1
        hw = float(sample.shape[-2] * sample.shape[-1]) if self.spatial_mean else 1.0      # loss.py:14-20, 75-77
        nll_loss = nll(sample).mean() / hw
        nlogdet = -logdet.mean() / hw
        loss = nll_loss + self.logdet_weight * nlogdet
        ref = nll(torch.randn_like(sample)).mean() / hw
        return loss, {"flow_loss": loss, "reference_nll_loss": ref, "nlogdet_loss": nlogdet,
                      "nll_loss": nll_loss, "logdet_weight": self.logdet_weight}


def linear_var(act_it, start_it, end_it, start_val, end_val, clip_min, clip_max):
    """Reference utils/general.py:221-228."""
    v = float(end_val - start_val) / (end_it - start_it) * (act_it - start_it) + start_val
    return min(max(v, clip_min), clip_max)


def lr_at(step, lr=1e-3, warm_it=500, end_it=200000):
    """LR rule of PokeMotionModel.on_train_batch_start (second_stage_video.py:238-253)."""
    if step < warm_it:
        return linear_var(step, 0, warm_it, 0.0, lr, 0.0, lr)
    return linear_var(step, warm_it, end_it, lr, 0.0, 0.0, lr)
