"""Test-side helpers: torch re-statements of the weight "shadow" layouts and small wrappers."""
import torch

from ipoke_amd import ops


def tdt(dtype):
    return ops.torch_dtype(dtype)


def shadow_nt(w, kc_pad, rows_pad=None, ld=None, dtype="f32", row_scale=None):
    """Conv weight [N][Cin][*k] -> [rows_pad][ld], k = tap*kc_pad + c (zero padded)."""
    N, Cin = w.shape[:2]
    taps = w[0, 0].numel()
    rows_pad = N if rows_pad is None else rows_pad
    ld = taps * kc_pad if ld is None else ld
    w3 = w.reshape(N, Cin, taps).permute(0, 2, 1).float()
    if row_scale is not None:
        w3 = w3 * row_scale.view(N, 1, 1)
    buf = torch.zeros(rows_pad, ld, dtype=torch.float32, device=w.device)
    tmp = torch.zeros(N, taps, kc_pad, device=w.device)
    tmp[:, :, :Cin] = w3
    buf[:N, :taps * kc_pad] = tmp.reshape(N, taps * kc_pad)
    return buf.to(tdt(dtype)).contiguous()


def shadow_t(w, n_pad, rows_pad=None, dtype="f32", col_scale=None):
    """Transposed operand: [rows_pad (c_in)][taps*n_pad], entry [c][tap*n_pad + n] = w[n][c][tap]."""
    N, Cin = w.shape[:2]
    taps = w[0, 0].numel()
    rows_pad = Cin if rows_pad is None else rows_pad
    w3 = w.reshape(N, Cin, taps).float()
    if col_scale is not None:
        w3 = w3 * col_scale.view(N, 1, 1)
    tmp = torch.zeros(Cin, taps, n_pad, device=w.device)
    tmp[:, :, :N] = w3.permute(1, 2, 0)
    buf = torch.zeros(rows_pad, taps * n_pad, device=w.device)
    buf[:Cin] = tmp.reshape(Cin, taps * n_pad)
    return buf.to(tdt(dtype)).contiguous()


def wn_scale(g, v):
    return g.flatten() / v.flatten(1).norm(dim=1)


def frag_tile(buf):
    """[rows][K] (rows % 16 == 0, K bytes % 64 == 0) -> the fragment-tiled order of the MCF weight operands (prep.hip:
    tiled_offset): tiles of 16 rows x 64 bytes, 16-byte chunk q of row r at (16 q + r) * 16 bytes inside the tile."""
    rows, K = buf.shape
    e16 = 16 // buf.element_size()
    ks = 4 * e16
    assert rows % 16 == 0 and K % ks == 0
    return buf.view(rows // 16, 16, K // ks, 4, e16).permute(0, 2, 3, 1, 4).contiguous().view(rows, K)


def mcf_shadows(sd, prefix, C, Cc, dtype):
    """Shadows of one MaskedConvFlow from its (reference-keyed) state dict."""
    d = ops.mcf_dims(C, Cc, dtype)
    w1 = sd[prefix + "net.shift_conv.weight"]
    v = sd[prefix + "net.conv1x1.conv.weight_v"]
    g = sd[prefix + "net.conv1x1.conv.weight_g"]
    b = sd[prefix + "net.conv1x1.conv.bias"].float().contiguous()
    H, K2 = 4 * C, 4 * C + Cc
    sc = wn_scale(g, v)
    W1 = shadow_nt(w1, d["Cp"], d["Hr"], d["K1p"], dtype)
    W1T = shadow_t(w1, d["Hq"], d["Cr"], dtype)
    W2 = shadow_nt(v, d["K2p"], d["N2r"], d["K2p"], dtype, row_scale=sc)
    weff = (v.flatten(1) * sc.view(-1, 1))[:, :H]                 # [2C][H]
    W2T = torch.zeros(d["Hr"], d["K3p"], device=v.device)
    W2T[:H, :2 * C] = weff.t()
    return dict(W1=frag_tile(W1), W1T=frag_tile(W1T), W2=frag_tile(W2), W2T=frag_tile(W2T.to(tdt(dtype)).contiguous()), bias=b, dims=d)


def synthetic_batch(B, T, size, seed=1, device="cpu"):
    """Same synthetic batch as oracle/make_goldens.py:synthetic_batch (SURVEY.md §8d inputs)."""
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, T, 3, size, size, generator=g) * 2 - 1
    flow = torch.randn(B, 2, size, size, generator=g)
    mask = (torch.rand(B, 1, size, size, generator=g) < 0.05).float()
    poke = [torch.randn(B, 2, size, size, generator=g) * mask, torch.zeros(B, 5, 2, dtype=torch.int64)]
    batch = {"images": images, "flow": flow, "poke": poke, "sample_ids": torch.zeros(B, T, dtype=torch.int64)}
    return {k: ([p.to(device) for p in v] if isinstance(v, list) else v.to(device)) for k, v in batch.items()}


_FILL_CACHE = {}


@torch.no_grad()
def cached_fill_(module, prefix):
    """``deterministic_fill_`` of a LARGE module (the 1.05 / 1.24 B-parameter flows: ~20 s of CPU random numbers per fill) with the filled
    state dict kept on the device for the rest of the session: the same (class, prefix, shapes) is filled once, later models copy it."""
    from ipoke_amd.utils.detfill import deterministic_fill_
    sd = module.state_dict()
    key = (type(module).__name__, prefix, tuple((k, tuple(v.shape)) for k, v in sd.items()))
    hit = _FILL_CACHE.get(key)
    if hit is None:
        deterministic_fill_(module, prefix=prefix)
        _FILL_CACHE[key] = {k: v.detach().clone() for k, v in module.state_dict().items()}
        return
    for k, v in sd.items():
        v.copy_(hit[k])
