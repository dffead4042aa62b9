"""CPU oracle for the first-stage video VAE of iPOKE — TEST INFRASTRUCTURE, NOT PRODUCT.

Plain-PyTorch fp32 restatement of the 3-D ResNet-18 motion encoder, the ConvGRU,
the SPADE-conditioned decoder and the small 2-D poke/image encoders, with the
reference's state-dict keys.  Pinned against golden vectors generated from the
reference's own modules (``oracle/make_goldens.py``).  Imported only by tests,
``__graft_entry__.smoke`` and ``bench.py``'s cpu_baseline leg.

Reference sites are cited per class.  Spectral-normalised convolutions follow
torch.nn.utils.spectral_norm (old-style forward-pre-hook, util.py:52, 252): in
*eval* mode u, v are frozen (weight = weight_orig / sigma), which is how the
second stage uses them (models/second_stage_video.py:269-272); in *train* mode
every forward CALL first runs one power iteration on the buffers
(``spectral_power_iter``) and sigma is taken with clones of the updated u, v --
the decoder is called once per generated frame, so a first-stage training step
sees T - 1 different sigma per weight (pinned by golden ``g13_first_stage_train_mode_128``).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# 3-D ResNet-18 motion encoder  (models/modules/motion_models/motion_encoder.py)
# --------------------------------------------------------------------------
class BasicBlock3d(nn.Module):
    """conv3x3x3-GN16-ReLU-conv3x3x3-GN16 (+1x1x1 strided conv+GN skip) -ReLU.  Reference :45-74, :198-216."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.GroupNorm(16, cout)
        self.conv2 = nn.Conv3d(cout, cout, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.GroupNorm(16, cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv3d(cin, cout, 1, stride=stride, bias=False), nn.GroupNorm(16, cout))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return F.relu(out + res)


class ResNetMotionEncoder(nn.Module):
    """Reference motion_encoder.py:150-241 with layers=[2,2,2,2] (resnet18_alternative, :21-23).

    ``eps`` for the reparameterisation is drawn on the CPU generator like the
    reference's ``torch.FloatTensor(...).normal_()`` (:218-222); pass ``eps=``
    to inject it.
    """

    def __init__(self, dic):
        super().__init__()
        ch = list(dic["ENC_M_channels"])
        self.be_determinstic = bool(dic.get("deterministic", False))   # [sic] reference attribute name
        self.spatial_size = dic["img_size"]
        max_frames = dic["max_frames"]
        self.min_ssize = dic.get("min_spatial_size", 8)
        self.conv1 = nn.Conv3d(3, ch[0], (3, 7, 7), stride=2, padding=(1, 3, 3), bias=False)
        self.bn1 = nn.GroupNorm(16, ch[0])
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
        self.conv_mu = nn.Conv2d(last, dic["z_dim"], 3, 1, 1)
        self.conv_var = nn.Conv2d(last, dic["z_dim"], 3, 1, 1)

    @staticmethod
    def _make(cin, cout, stride):
        return nn.Sequential(BasicBlock3d(cin, cout, stride), BasicBlock3d(cout, cout, 1))

    def features(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        if self.stride4 is not None:
            x = self.layer4(x)
        if self.has5:
            x = self.layer5(x)
        return x.squeeze(2)

    def forward(self, x, eps=None):
        emb = self.features(x)
        mu, logvar = self.conv_mu(emb), self.conv_var(emb)
        if self.be_determinstic:
            return mu, mu, mu
        if eps is None:
            eps = torch.FloatTensor(logvar.size()).normal_().to(mu.device)
        return eps * (0.5 * logvar).exp() + mu, mu, logvar


# --------------------------------------------------------------------------
# ConvGRU  (models/modules/motion_models/rnn.py)
# --------------------------------------------------------------------------
class ConvGRUCell(nn.Module):
    """Reference rnn.py:4-56: u,r = sigmoid(conv[x,h]); o = tanh(conv[x, h*r]); h' = h(1-u) + o*u."""

    def __init__(self, cin, hidden, k=3):
        super().__init__()
        self.reset_gate = nn.Conv2d(cin + hidden, hidden, k, padding=k // 2)
        self.update_gate = nn.Conv2d(cin + hidden, hidden, k, padding=k // 2)
        self.out_gate = nn.Conv2d(cin + hidden, hidden, k, padding=k // 2)

    def forward(self, x, h):
        xh = torch.cat([x, h], dim=1)
        u = torch.sigmoid(self.update_gate(xh))
        r = torch.sigmoid(self.reset_gate(xh))
        o = torch.tanh(self.out_gate(torch.cat([x, h * r], dim=1)))
        return h * (1 - u) + o * u


class ConvGRU(nn.Module):
    """Reference rnn.py:59-133: layer i is fed the *updated* hidden state of layer i-1."""

    def __init__(self, cin, hidden, n_layers, k=3):
        super().__init__()
        self.n_layers = n_layers
        self.cells = nn.Sequential(*[ConvGRUCell(cin if i == 0 else hidden, hidden, k) for i in range(n_layers)])

    def forward(self, x, hidden):
        out = []
        for cell, h in zip(self.cells, hidden):
            x = cell(x, h)
            out.append(x)
        return out


# --------------------------------------------------------------------------
# 2-D conv blocks  (models/modules/autoencoders/util.py)
# --------------------------------------------------------------------------
class _SNConv(nn.Module):
    """Parameter holder with old-style spectral_norm names (bias, weight_orig, weight_u, weight_v).

    ``transposed``: ConvTranspose2d weight [in, out, k, k] with the spectral norm taken over dim=1
    (torch.nn.utils.spectral_norm default for transposed convs).
    """

    def __init__(self, cin, cout, k, transposed=False):
        super().__init__()
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.transposed = transposed
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_orig = nn.Parameter(torch.randn(shape) / math.sqrt(cin * k * k))
        rows = cout
        cols = self.weight_orig.numel() // rows
        self.register_buffer("weight_u", F.normalize(torch.randn(rows), dim=0))
        self.register_buffer("weight_v", F.normalize(torch.randn(cols), dim=0))

    def matrix(self):
        w = self.weight_orig
        if self.transposed:
            w = w.transpose(0, 1)
        return w.reshape(w.shape[0], -1)

    def weight(self):
        u, v = self.weight_u, self.weight_v
        if self.training:
            # torch's hook: iterate in place without grad, then clone so that later calls' in-place updates do not
            # invalidate what this call's backward needs
            self.spectral_power_iter()
            u, v = u.clone(), v.clone()
        sigma = torch.dot(u, torch.mv(self.matrix(), v))
        return self.weight_orig / sigma

    @torch.no_grad()
    def spectral_power_iter(self, eps=1e-12):
        """One train-mode power iteration (torch spectral_norm forward-pre-hook semantics)."""
        m = self.matrix()
        self.weight_v.copy_(F.normalize(torch.mv(m.t(), self.weight_u), dim=0, eps=eps))
        self.weight_u.copy_(F.normalize(torch.mv(m, self.weight_v), dim=0, eps=eps))


class _PlainConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k))
        self.bias = nn.Parameter(torch.zeros(cout))


def _make_norm(kind, ch):
    if kind == "group":
        return nn.GroupNorm(16, ch)
    if kind == "in":
        return nn.InstanceNorm2d(ch)
    assert kind == "none"
    return None


def _act(kind, x, transpose_block=False):
    # util.py:41-42: inside Conv2dTransposeBlock the key "elu" selects nn.ReLU (quirk kept)
    if kind == "elu":
        return F.relu(x) if transpose_block else F.elu(x)
    if kind == "relu":
        return F.relu(x)
    if kind == "tanh":
        return torch.tanh(x)
    assert kind == "none"
    return x


class Conv2dBlock(nn.Module):
    """zero-pad -> conv -> norm -> act.  Reference util.py:195-273."""

    def __init__(self, cin, cout, k, stride, padding, norm="none", activation="elu", snorm=False):
        super().__init__()
        self.stride, self.padding, self.activation = stride, padding, activation
        self.norm = _make_norm(norm, cout)
        self.conv = _SNConv(cin, cout, k) if snorm else _PlainConv(cin, cout, k)

    def forward(self, x):
        w = self.conv.weight() if isinstance(self.conv, _SNConv) else self.conv.weight
        x = F.conv2d(x, w, self.conv.bias, stride=self.stride, padding=self.padding)
        if self.norm is not None:
            x = self.norm(x)
        return _act(self.activation, x)


class Conv2dTransposeBlock(nn.Module):
    """ConvTranspose2d(k, s, padding=p, output_padding=p) -> norm -> act.  Reference util.py:7-73."""

    def __init__(self, cin, cout, k, stride, padding, norm="none", activation="elu", snorm=False):
        super().__init__()
        self.stride, self.padding, self.activation = stride, padding, activation
        self.norm = _make_norm(norm, cout)
        self.snorm = snorm
        if snorm:
            self.conv = _SNConv(cin, cout, k, transposed=True)
        else:
            self.conv = nn.ConvTranspose2d(cin, cout, k, stride, padding=padding, output_padding=padding)

    def forward(self, x):
        if self.snorm:
            x = F.conv_transpose2d(x, self.conv.weight(), self.conv.bias, stride=self.stride,
                                   padding=self.padding, output_padding=self.padding)
        else:
            x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        return _act(self.activation, x, transpose_block=True)


class ResBlock(nn.Module):
    """Reference util.py:106-192.  The skip conv always uses InstanceNorm ("in") + activation."""

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

    def forward(self, x):
        res = self.res_conv(x) if self.convolve_res else x
        return self.conv2(self.conv1(x)) + res


class Spade(nn.Module):
    """GroupNorm(affine=False)(x)*(1+gamma)+beta, gamma/beta from the bilinearly resized start frame.

    Reference util.py:473-500 (align_corners=True).
    """

    def __init__(self, ch, groups=16):
        super().__init__()
        while ch % groups != 0:
            groups -= 1
        self.norm = nn.GroupNorm(groups, ch, affine=False)
        self.conv = nn.Conv2d(3, 128, 3, 1, 1)
        self.conv_gamma = nn.Conv2d(128, ch, 3, 1, 1)
        self.conv_beta = nn.Conv2d(128, ch, 3, 1, 1)

    def modulation(self, y, size):
        y = F.interpolate(y, mode="bilinear", size=size, align_corners=True)
        y = F.leaky_relu(self.conv(y), 0.2)
        return self.conv_gamma(y), self.conv_beta(y)

    def forward(self, x, y):
        gamma, beta = self.modulation(y, x.shape[-2:])
        return self.norm(x) * (1 + gamma) + beta


class SpadeCondConvDecoder(nn.Module):
    """Reference fully_conv_models.py:135-177."""

    def __init__(self, config):
        super().__init__()
        ch = config["dec_channels"]
        sn = config["spectral_norm"]
        self.blocks = nn.ModuleList()
        self.spade_blocks = nn.ModuleList()
        self.in_block = ResBlock(config["z_dim"], ch[0], snorm=sn, norm=config["norm"])
        for i, nf in enumerate(ch[1:]):
            self.blocks.append(ResBlock(ch[i], nf, norm="none", upsampling=True, snorm=sn))
            self.spade_blocks.append(Spade(nf))
        self.out_conv = Conv2dBlock(ch[-1], 3, 3, 1, 1, norm="none", activation="tanh")

    def forward(self, actual_frame, start_frame, del_shape=True):
        x = self.in_block(actual_frame[-1])
        for blk, sp in zip(self.blocks, self.spade_blocks):
            x = sp(blk(x), start_frame)
        return self.out_conv(x)


class ConvEncoder(nn.Module):
    """Deterministic 2-D encoder of FirstStageWrapper.  Reference fully_conv_models.py:28-94."""

    def __init__(self, nf_in, nf_max, n_stages):
        super().__init__()
        nf = 32
        blocks = [Conv2dBlock(nf_in, nf, 3, 2, 1, norm="group", activation="elu", snorm=True)]
        for _ in range(n_stages - 1):
            nxt = min(2 * nf, nf_max)
            blocks.append(ResBlock(nf, nxt, stride=2, norm="group", activation="elu", snorm=True))
            nf = nxt
        self.model = nn.Sequential(*blocks)
        self.bottleneck = nn.Sequential(ResBlock(nf, nf_max, activation="elu", norm="group"))

    def forward(self, x):
        mean = self.model(x)
        out = self.bottleneck(mean)
        return out, mean, None


class FirstStageWrapper(nn.Module):
    """Encoder half of the reference's FirstStageWrapper (fully_conv_models.py:9-26); the
    decoder half is not on the hot path (its checkpoint keys are ignored with strict=False)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        arch = config["architecture"]
        self.be_deterministic = arch["deterministic"]
        assert self.be_deterministic
        n_stages = int(np.log2(config["data"]["spatial_size"][0] // arch["min_spatial_size"]))
        nf_in = arch["nf_in"] + (3 if arch.get("poke_and_image", False) else 0)
        self.encoder = ConvEncoder(nf_in, arch["nf_max"], n_stages)


class SpadeCondMotionModel(nn.Module):
    """Reference models/first_stage_motion_model.py:469-522."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        arch = dict(config["architecture"])
        self.full_sequence = bool(config["training"].get("full_sequence", False))
        arch.update(img_size=config["data"]["spatial_size"][0], max_frames=config["data"]["max_frames"],
                    full_seq=self.full_sequence)
        self.use_motion_bias = bool(arch.get("motion_bias", False))
        self.enc_motion = ResNetMotionEncoder(arch)
        self.n_layers = arch["n_gru_layers"]
        self.rnn = ConvGRU(arch["z_dim"], arch["z_dim"], self.n_layers)
        if self.use_motion_bias:
            s = arch["min_spatial_size"]
            self.motion_bias = nn.Parameter(torch.randn(1, arch["z_dim"], s, s))
        self.gen = SpadeCondConvDecoder(arch)

    def decode(self, motion, start_frame, length):
        hidden = [motion] * self.n_layers
        in_rnn = torch.cat([self.motion_bias] * start_frame.shape[0], dim=0) if self.use_motion_bias else motion
        frames = []
        for _ in range(length):
            hidden = self.rnn(in_rnn, hidden)
            frames.append(self.gen([hidden[-1]], start_frame))
        return torch.stack(frames, dim=1)

    def forward(self, X, eps=None):
        X_in = X if self.full_sequence else X[:, 1:]
        motion, mu, logvar = self.enc_motion(X_in.transpose(1, 2), eps=eps)
        return self.decode(motion, X[:, 0], X.shape[1] - 1), mu, logvar


def kl_loss(mu, logvar):
    """Reference utils/losses.py:47-48."""
    return -0.5 * torch.mean(torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1))


def first_stage_loss(X, X_hat, mu, logvar, w_l1=10.0, w_kl=1e-7):
    """L1 + KL part of MotionModel.training_step (first_stage_motion_model.py:263-272)."""
    return w_l1 * (X[:, 1:] - X_hat).abs().mean() + w_kl * kl_loss(mu, logvar)
