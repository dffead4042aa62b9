"""Name-keyed deterministic weight fill.

Every tensor of a state dict is filled from ``torch.Generator().manual_seed(crc32(key))``
so that two implementations which share state-dict keys (the reference, the CPU
oracle, this package) obtain identical weights at any model size without
shipping checkpoints.  Used by the golden-vector generator, the parity tests
and ``bench.py`` (random-init weights of the benchmarked architecture).
"""
import zlib

import torch


def _gen(key):
    g = torch.Generator()
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    return g


def _randn(shape, g):
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


def fill_value(key, ref):
    """Deterministic value for state-dict entry ``key`` shaped/typed like ``ref`` (CPU tensor returned)."""
    g = _gen(key)
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(ref.shape)
    if leaf == "initialized":
        return torch.ones(shape, dtype=ref.dtype)
    if leaf == "forward_shuffle_idx":
        return torch.randperm(shape[0], generator=g).to(ref.dtype)
    if leaf == "backward_shuffle_idx":
        fwd = torch.randperm(shape[0], generator=_gen(key[: -len("backward_shuffle_idx")] + "forward_shuffle_idx"))
        return torch.argsort(fwd).to(ref.dtype)
    if not ref.dtype.is_floating_point:
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == "weight_g":
        return (0.1 + 0.02 * _randn(shape, g)).abs() + 0.01
    if leaf == "weight_v" and len(shape) == 4:
        return 0.05 * _randn(shape, g)
    if leaf in ("weight_u", "weight_v"):            # spectral-norm power-iteration vectors
        v = _randn(shape, g)
        return v / (v.norm() + 1e-12)
    if leaf == "log_scale":
        return 0.03 * _randn(shape, g)
    if leaf == "bias" and len(shape) == 3:           # ActNorm bias [C,1,1]
        return 0.03 * _randn(shape, g)
    if leaf == "motion_bias":
        return _randn(shape, g)
    if leaf == "running_var":                        # BatchNorm statistics of the I3D network: positive variances
        return 0.5 + torch.rand(shape, generator=g)
    if leaf in ("gamma",):
        return 1.0 + 0.1 * _randn(shape, g)
    squeezed = [s for s in shape if s != 1]
    if len(squeezed) <= 1:
        if leaf == "weight":                         # norm-layer scale
            return 1.0 + 0.1 * _randn(shape, g)
        return 0.1 * _randn(shape, g)                # biases, beta, actnorm bias
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    gain = 1.4
    if ".conv_var." in key:                          # keep the VAE's log-variance head small
        gain = 0.15
    elif ".conv_mu." in key:
        gain = 0.7
    return _randn(shape, g) * (gain / fan_in ** 0.5)


@torch.no_grad()
def deterministic_fill_(module_or_state_dict, prefix=""):
    """Overwrite every parameter/buffer in place; returns the list of keys touched."""
    sd = module_or_state_dict.state_dict() if hasattr(module_or_state_dict, "state_dict") else module_or_state_dict
    keys = []
    for k, t in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        val = fill_value(prefix + k, t)
        t.copy_(val.to(t.device))
        keys.append(k)
    return keys
