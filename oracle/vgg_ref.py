"""CPU restatement of the first stage's VGG perceptual loss (TEST INFRASTRUCTURE ONLY -- imported by tests/ and
oracle/make_goldens.py; the product path never touches it).

Follows reference utils/losses.py:
    VGG (:6-39)        the first 30 layers of ``torchvision.models.vgg19(pretrained=True).features`` cut into five slices whose
                       outputs are relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (feature indices 1, 6, 11, 20, 29); no input
                       normalisation (:32 is commented out); parameters frozen
    fmap_loss (:58-65) mean over the five maps of mean |f1 - f2|
    VGGLoss (:67-82)   fmap_loss(vgg(x), vgg(y)) (unweighted form, the default)
and its call site models/first_stage_motion_model.py:263: vgg_loss(X[:, 1:] frames, X_hat frames).

torchvision is a third-party dependency that is absent from this image (ipoke.yml:23 lists it unpinned, data_proc.yml:14 pins
0.4.0).  ``vgg19_features`` restates its published architecture: configuration "E" of Simonyan & Zisserman -- 3x3 convolutions
with padding 1 and bias, each followed by ReLU, 2x2/2 max pooling after 2, 2, 4, 4, 4 convolutions of 64, 128, 256, 512, 512
channels -- as an nn.Sequential whose indices equal torchvision's, so that state-dict keys are ``features.<idx>.weight``.
Parity of the reference's own part (slicing, loss) is pinned by oracle/make_goldens.py job g12, which runs the reference's
VGGLoss with ``torchvision.models.vgg19`` resolved to this stack; the stack itself is "parity unpinned" against torchvision.
"""
import torch
import torch.nn as nn

CFG_E = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))        # feature-index ranges of slice1..5


def vgg19_features():
    layers, cin = [], 3
    for v in CFG_E:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class VGG(nn.Module):
    def __init__(self, features=None):
        super().__init__()
        f = vgg19_features() if features is None else features
        for i, (a, b) in enumerate(SLICES):
            seq = nn.Sequential()
            for x in range(a, b):
                seq.add_module(str(x), f[x])
            setattr(self, f"slice{i + 1}", seq)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, X):
        out = []
        for i in range(5):
            X = getattr(self, f"slice{i + 1}")(X)
            out.append(X)
        return out


class VGGLoss(nn.Module):
    def __init__(self, features=None):
        super().__init__()
        self.vgg = VGG(features)

    def forward(self, x, y):
        f1, f2 = self.vgg(x), self.vgg(y)
        return sum(torch.mean(torch.abs(a - b)) for a, b in zip(f1, f2)) / len(f1)
