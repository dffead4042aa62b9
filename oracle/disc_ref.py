"""CPU restatement of the first-stage temporal discriminator (TEST INFRASTRUCTURE ONLY -- imported by tests/,
oracle/make_goldens.py and nothing else; the product path never touches it).

Follows reference models/modules/discriminators/patchgan_3d.py:
    conv3x3x3 (:31-40)   spectral-normalised Conv3d 3x3x3, padding 1, stride (stride_t, s, s), no bias
    BasicBlock (:43-63)  conv-GN(16)-ReLU-conv-GN(16), + (downsampled) input, ReLU
    ResNet (:171-258)    stem Conv3d(3,64,(3,7,7),stride (1,2,2),pad (1,3,3)) + GN(16) + ReLU + MaxPool3d(3, (1,2,2), 1);
                         layer1..4 = [2,2,2,2] blocks of 64/128/256/512 planes, spatial strides 1/1/2/2, temporal stride 2 in
                         layers 2-4 unless patch_temp_disc; a layer's first block downsamples through a spectral-normalised
                         3x3x3 conv + GN(16) when stride != 1 or the width changes (:221-232);
                         AvgPool3d((1, ceil(size/16), ceil(size/16))), Linear(512, num_classes, bias=False) per remaining frame
    loss (:263-274)      hinge: mean(relu(1 - pred)) for real, mean(relu(1 + pred)) for fake
    gp2 (:285-294)       mean over the batch of sum((d sum(pred) / d x)^2)  (create_graph=True)
    fmap_loss (:297-304) mean over the four feature maps of mean |f1 - f2|
The 2-D PatchGAN (``PatchDiscriminator`` below) follows models/modules/discriminators/patchgan.py:368-470: spectral-normalised
4x4 convolutions with bias (stride 2, 2, 2, then 1 and the 1-channel output head, padding 1), InstanceNorm2d (no affine) +
LeakyReLU(0.2) after every inner convolution, feature maps after each of them; hinge loss and feature-matching loss as above.
State-dict keys equal the reference's (``conv1.weight_orig / weight_u / weight_v``, ``gn1.weight``, ``layer1.0.conv1...``,
``layer2.0.downsample.0.weight_orig``, ``fc.weight``).  Parity is pinned by oracle/make_goldens.py job g8 (outputs, losses
and gradients of this module asserted against the reference module on the same inputs and weights).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class SNConv3d(nn.Module):
    """Conv3d without bias under torch.nn.utils.spectral_norm semantics (weight_orig, weight_u, weight_v; one power
    iteration per forward call in train mode; sigma = u^T W v with W = weight_orig.reshape(cout, -1))."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.stride, self.pad = tuple(stride), tuple(pad)
        w = torch.empty(cout, cin, *k)
        nn.init.orthogonal_(w)
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", F.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(torch.randn(cin * k[0] * k[1] * k[2]), dim=0, eps=1e-12))

    def effective_weight(self):
        wm = self.weight_orig.reshape(self.weight_orig.shape[0], -1)
        if self.training:
            with torch.no_grad():
                v = F.normalize(torch.mv(wm.t(), self.weight_u), dim=0, eps=1e-12)
                u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
                self.weight_v.copy_(v); self.weight_u.copy_(u)
        u, v = self.weight_u.clone(), self.weight_v.clone()
        return self.weight_orig / torch.dot(u, torch.mv(wm, v))

    def forward(self, x):
        return F.conv3d(x, self.effective_weight(), None, self.stride, self.pad)


class Block(nn.Module):
    def __init__(self, cin, planes, stride=1, stride_t=1, downsample=False):
        super().__init__()
        self.conv1 = SNConv3d(cin, planes, (3, 3, 3), (stride_t, stride, stride), (1, 1, 1))
        self.bn1 = nn.GroupNorm(16, planes)
        self.conv2 = SNConv3d(planes, planes, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        self.bn2 = nn.GroupNorm(16, planes)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(SNConv3d(cin, planes, (3, 3, 3), (stride_t, stride, stride), (1, 1, 1)), nn.GroupNorm(16, planes))

    def forward(self, x):
        out = torch.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return torch.relu(out + res)


class TemporalDiscriminator(nn.Module):
    """``resnet(config=d_t, spatial_size, sequence_length)`` of the reference (ResNet-18 layout)."""

    def __init__(self, spatial_size, config, layers=(2, 2, 2, 2)):
        super().__init__()
        self.gp_weight = config.get("gp_weight", 0.0)
        stride_t = 1 if config.get("patch_temp_disc", False) else 2
        self.conv1 = SNConv3d(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3))
        self.gn1 = nn.GroupNorm(16, 64)
        spec = [(64, 1, 1), (128, 1, stride_t), (256, 2, stride_t), (512, 2, stride_t)]
        inplanes = 64
        for li, ((planes, s, st), n) in enumerate(zip(spec, layers), start=1):
            blocks = [Block(inplanes, planes, s, st, downsample=(s != 1 or inplanes != planes))]
            inplanes = planes
            blocks += [Block(planes, planes) for _ in range(1, n)]
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.last = int(math.ceil(spatial_size / 16))
        self.fc = nn.Linear(512, config.get("num_classes", 1), bias=False)

    def forward(self, x):
        x = torch.relu(self.gn1(self.conv1(x)))
        x = F.max_pool3d(x, (3, 3, 3), (1, 2, 2), 1)
        fmaps = []
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
            fmaps.append(x)
        p = F.avg_pool3d(x, (1, self.last, self.last), 1)
        pred = torch.cat([self.fc(p[:, :, i].reshape(p.shape[0], -1)) for i in range(p.shape[2])], dim=1)
        return pred, fmaps

    @staticmethod
    def loss(pred, real):
        return torch.relu(1.0 - pred).mean() if real else torch.relu(1.0 + pred).mean()

    @staticmethod
    def gp2(pred, x):
        g = torch.autograd.grad(pred.sum(), x, create_graph=True, retain_graph=True)[0]
        return g.pow(2).reshape(x.shape[0], -1).sum(1).mean()

    @staticmethod
    def fmap_loss(f1, f2):
        return sum((a - b).abs().mean() for a, b in zip(f1, f2)) / len(f1)


class SNConv2d(nn.Module):
    """Conv2d with bias under torch.nn.utils.spectral_norm semantics."""

    def __init__(self, cin, cout, k, stride, pad, bias=True):
        super().__init__()
        self.stride, self.pad = stride, pad
        self.weight_orig = nn.Parameter(torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.register_buffer("weight_u", F.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(torch.randn(cin * k * k), dim=0, eps=1e-12))

    def forward(self, x):
        wm = self.weight_orig.reshape(self.weight_orig.shape[0], -1)
        if self.training:
            with torch.no_grad():
                v = F.normalize(torch.mv(wm.t(), self.weight_u), dim=0, eps=1e-12)
                u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
                self.weight_v.copy_(v); self.weight_u.copy_(u)
        u, v = self.weight_u.clone(), self.weight_v.clone()
        return F.conv2d(x, self.weight_orig / torch.dot(u, torch.mv(wm, v)), self.bias, self.stride, self.pad)


class PatchDiscriminator(nn.Module):
    """``PatchDiscriminator(config=d_s)`` of the reference with its default InstanceNorm2d (patchgan.py:368-418)."""

    def __init__(self, config):
        super().__init__()
        n_layers = config.get("n_layers", 3)
        ndf = 64
        self.in_conv = SNConv2d(3, ndf, 4, 2, 1)
        self.layers, self.norms = nn.ModuleList(), nn.ModuleList()
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            self.layers.append(SNConv2d(ndf * prev, ndf * mult, 4, 2, 1))
            self.norms.append(nn.InstanceNorm2d(ndf * mult))
        prev, mult = mult, min(2 ** n_layers, 8)
        self.layers.append(SNConv2d(ndf * prev, ndf * mult, 4, 1, 1))
        self.norms.append(nn.InstanceNorm2d(ndf * mult))
        self.out_conv = SNConv2d(ndf * mult, 1, 4, 1, 1)

    def forward(self, x):
        x = F.leaky_relu(self.in_conv(x), 0.2)
        fmap = []
        for conv, norm in zip(self.layers, self.norms):
            x = F.leaky_relu(norm(conv(x)), 0.2)
            fmap.append(x)
        return self.out_conv(x), fmap

    loss = staticmethod(TemporalDiscriminator.loss)
    fmap_loss = staticmethod(TemporalDiscriminator.fmap_loss)
