"""VGG-19 perceptual loss of the first stage on the HIP kernels (reference utils/losses.py:6-82, call site
models/first_stage_motion_model.py:263: ``w_vgg * vgg_loss(X[:, 1:] frames, X_hat frames)``).

``VGG`` holds the 13 convolutions of ``torchvision.models.vgg19().features[:30]`` under the reference's names
(``slice1.0.weight`` ... ``slice5.28.bias``; ``load_torchvision_features`` maps a torchvision ``features.<idx>`` state dict onto
them -- the pretrained weights are not available offline).  Every convolution is the implicit-GEMM kernel with bias + ReLU in
the epilogue, the 2x2 pools are ``ipoke_maxpool3d_fwd``, the five L1 terms ``ipoke_l1_pair``; the backward pass runs the adjoint
convolutions only -- the weights are frozen, so no weight gradient is formed.  The true frames' maps carry no gradient.
"""
import torch
import torch.nn as nn

from . import _lib, first_stage as FS, first_stage_train as T, nn as K
from .discriminator import _L1MeanFn, max_pool3d

CFG_E = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512)     # features[:30]
SLICE_OF = lambda idx: 1 if idx < 2 else 2 if idx < 7 else 3 if idx < 12 else 4 if idx < 21 else 5
TAPS = (1, 6, 11, 20, 29)                                       # relu1_1, relu2_1, relu3_1, relu4_1, relu5_1


class VGG(nn.Module):
    def __init__(self, requires_grad=False, dtype="bf16"):
        super().__init__()
        self.dtype = dtype
        self.mean, self.std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]          # kept for parity; unused (:32)
        for i in range(1, 6):
            setattr(self, f"slice{i}", nn.ModuleDict())
        self.program = []                                       # ("conv", slice, idx) | ("pool",) | ("tap",)
        idx, cin = 0, 3
        for v in CFG_E:
            if v == "M":
                self.program.append(("pool",)); idx += 1
                continue
            getattr(self, f"slice{SLICE_OF(idx)}")[str(idx)] = FS._Conv(cin, v, 3, 1, 1, bias=True, dims=2)
            self.program.append(("conv", SLICE_OF(idx), str(idx)))
            if idx + 1 in TAPS:
                self.program.append(("tap",))
            idx += 2; cin = v
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision_features(self, sd):
        """``torchvision.models.vgg19().state_dict()`` (or its ``features`` part) -> this module."""
        own = {}
        for k, v in sd.items():
            parts = k.split(".")
            if parts[0] == "features":
                parts = parts[1:]
            if len(parts) == 2 and parts[0].isdigit() and int(parts[0]) < 30:
                own[f"slice{SLICE_OF(int(parts[0]))}.{parts[0]}.{parts[1]}"] = v
        return self.load_state_dict(own, strict=True)

    def forward(self, X):
        """X fp32 [N, 3, H, W] -> the five maps as channels-last records (``nn.CL``), differentiable w.r.t. X."""
        _lib.require_gpu()
        dt = self.dtype
        N, C, H, W = X.shape
        first = self.slice1["0"]
        if X.requires_grad:
            xcl = T._pad_cols(X.permute(0, 2, 3, 1).reshape(-1, C), K.round_up(C, K.e16(dt)), dt)
            h = T.conv(first, K.CL(xcl, N, (1, H, W), C), dt, act=_lib.ACT_RELU)
        else:
            X = X.float()
            st = (X.stride(0), X.stride(1), 0, X.stride(2), X.stride(3))
            h = T.conv(first, None, dt, act=_lib.ACT_RELU, src=(X, N, C, (1, H, W), st))
        out = []
        for op in self.program[1:]:
            if op[0] == "conv":
                h = T.conv(getattr(self, f"slice{op[1]}")[op[2]], h, dt, act=_lib.ACT_RELU)
            elif op[0] == "pool":
                h = max_pool3d(h, (1, 2, 2), (1, 2, 2), (0, 0, 0), dt)
            else:
                out.append(h)
        return out


class VGGLoss(nn.Module):
    def __init__(self, weighted=False, dtype="bf16"):
        super().__init__()
        self.vgg = VGG(dtype=dtype)
        self.weighted = weighted
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        """x: the true frames (no gradient), y: the generated frames; both fp32 [N, 3, H, W]."""
        with torch.no_grad():
            f1 = self.vgg(x.detach())
        f2 = self.vgg(y)
        terms = [_L1MeanFn.apply(b.t, a.t, b.C, self.vgg.dtype) for a, b in zip(f1, f2)]
        if self.weighted:
            return sum(w * t for w, t in zip(self.weights, terms))
        return sum(terms) / len(terms)
