"""Import helper for the *reference* iPOKE sources (TEST INFRASTRUCTURE ONLY).

This file is only ever used inside the build container, where the read-only
reference checkout lives at ``/root/reference``: ``oracle/make_goldens.py``
imports the reference's own PyTorch modules through it in order to (i) validate
the CPU restatement in ``oracle/`` and (ii) emit the golden vectors committed
under ``tests/golden/``.  Nothing in the product package, ``bench.py`` or the
``-m gpu`` tests imports it, and no reference source travels to the GPU box.

Recipe: the pure-torch sub-modules import once a handful of
absent third-party packages are replaced by auto-mocks; two hard-coded
``.cuda()`` calls on the path are neutralised by making ``Tensor.cuda`` the
identity on this GPU-less host.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("IPOKE_REFERENCE_ROOT", "/root/reference")

_MOCKED_TOPLEVEL = (
    "opt_einsum", "wandb", "lpips", "cv2", "torchvision", "seaborn", "umap",
    "coloredlogs", "kornia", "dotmap", "natsort", "imageio", "tensorflow",
    "tensorflow_hub", "tensorflow_gan", "matplotlib", "skimage", "PIL",
)
_MOCKED_PREFIXES = ("models.pose_estimator", "models.flownet2")


class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        return None


class _MockFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        hit = False
        if top in _MOCKED_TOPLEVEL:
            try:
                # prefer the real package when it exists
                if top not in sys.modules or isinstance(sys.modules[top], mock.MagicMock):
                    real = importlib.machinery.PathFinder.find_spec(top)
                    hit = real is None
            except Exception:
                hit = True
        if any(fullname == p or fullname.startswith(p + ".") for p in _MOCKED_PREFIXES):
            hit = True
        if fullname.startswith("pytorch_lightning.") and fullname not in sys.modules:
            hit = True                      # any lightning sub-module the stub below does not define
        if hit:
            return importlib.machinery.ModuleSpec(fullname, _MockLoader(), is_package=True)
        return None


def _install_lightning_stub():
    import torch.nn as nn
    if "pytorch_lightning" in sys.modules:
        return
    pl = types.ModuleType("pytorch_lightning")
    pl.__path__ = []

    class LightningModule(nn.Module):
        global_step = 0
        current_epoch = 0

        def log(self, *a, **k):
            return None

        def log_dict(self, *a, **k):
            return None

        def optimizers(self):
            return getattr(self, "_optimizers_stub", None)

    class Callback:
        pass

    pl.LightningModule = LightningModule
    pl.Callback = Callback
    pl.Trainer = mock.MagicMock(name="Trainer")
    metrics = types.ModuleType("pytorch_lightning.metrics")
    metrics.__path__ = []

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default)

    metrics.Metric = Metric
    pl.metrics = metrics
    callbacks = types.ModuleType("pytorch_lightning.callbacks")
    callbacks.ModelCheckpoint = mock.MagicMock(name="ModelCheckpoint")
    callbacks.Callback = Callback
    pl.callbacks = callbacks
    loggers = types.ModuleType("pytorch_lightning.loggers")
    loggers.WandbLogger = mock.MagicMock(name="WandbLogger")
    pl.loggers = loggers
    profiler = types.ModuleType("pytorch_lightning.profiler")
    profiler.AdvancedProfiler = mock.MagicMock(name="AdvancedProfiler")
    pl.profiler = profiler
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.metrics"] = metrics
    sys.modules["pytorch_lightning.callbacks"] = callbacks
    sys.modules["pytorch_lightning.loggers"] = loggers
    sys.modules["pytorch_lightning.profiler"] = profiler


_installed = False


def install():
    """Make ``import models.…`` resolve to the reference checkout (idempotent)."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(
            f"reference checkout not found at {REFERENCE_ROOT}; golden vectors can "
            "only be (re)generated inside the build container")
    import torch
    sys.meta_path.insert(0, _MockFinder())
    _install_lightning_stub()
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def ref(modname):
    install()
    return importlib.import_module(modname)
