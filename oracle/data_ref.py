"""CPU restatement of the reference's poke simulation and flow resize (TEST INFRASTRUCTURE ONLY -- imported by tests/ and
oracle/make_goldens.py; the product path never touches it).

Follows reference data/base_dataset.py:
    _get_flow (:651-693)   optional division by (raw height / target height) (scale_poke_to_res), bilinear resize with
                           align_corners=True to config["spatial_size"]; an all-zero flow for "zero poke" samples (ids[-1] == -1)
    _get_poke (:507-648)   amplitude = |flow| on the window [poke_size, size - poke_size)^2, shifted to min 0 and scaled to max 1;
                           candidate positions = amplitude > mean + 2 std (fallbacks: > mean + std, > mean); the number of pokes is
                           drawn from [1, min(n_pokes, #candidates)] unless fixed; each poke copies the flow vector at its centre
                           (equal_poke_val) or the flow patch around it into a (2 half + 1)^2 window, later pokes overwriting
                           earlier ones.  Zero-poke samples: centres are drawn among the positions below the 5th amplitude
                           percentile (background) and the values come from positions with amplitude > mean + std (fallback
                           > mean).  poke_centers: int64 [n_pokes, 2] (row, col), -1 padded.
Random draws go through an injected ``randint(low, high, size)`` so that the device path, this restatement and the reference
(whose ``np.random.randint`` is replaced by the same source while the golden vectors are generated) consume identical draws:
value = low + floor(u * (high - low)) with u float32 in [0, 1), in the order (count draw, value-source draws, centre draws).
Parity is pinned by oracle/make_goldens.py job g11.
"""
import numpy as np
import torch
import torch.nn.functional as F


class UniformDraws:
    """randint replacement reading pre-drawn uniforms u[0] (count), u[1 : 1+N] (value sources, zero-poke samples only) and
    u[1+N : 1+2N] (centres), N = n_pokes of the config."""

    def __init__(self, u, n_max, fix_n_pokes, zero):
        self.u, self.n_max, self.calls = np.asarray(u, dtype=np.float32), n_max, 0
        self.plan = ([] if fix_n_pokes else [0]) + ([1, 1 + n_max] if zero else [1 + n_max])

    def __call__(self, low, high=None, size=None):
        if high is None:
            low, high = 0, low
        off = self.plan[self.calls]
        self.calls += 1
        n = 1 if size is None else int(size)
        v = low + np.floor(self.u[off:off + n].astype(np.float64) * (high - low)).astype(np.int64)
        return int(v[0]) if size is None else v


def get_flow(raw, spatial_size, scale_poke_to_res=True):
    """raw: float32 numpy [2, Hs, Ws] -> torch [2, H, W]."""
    if scale_poke_to_res:
        raw = raw / (raw.shape[1] / spatial_size[0])
    return F.interpolate(torch.from_numpy(raw).unsqueeze(0), size=tuple(spatial_size), mode="bilinear", align_corners=True).squeeze(0)


def get_poke(flow, poke_size, n_pokes, randint, zero=False, fix_n_pokes=False, equal_poke_val=True):
    """flow [2, H, W] -> (poke [2, H, W], poke_centers int64 [n_pokes, 2]); raises ValueError when no candidate remains."""
    H, W = flow.shape[1:]
    h0, h1, w0, w1 = poke_size, H - poke_size, poke_size, W - poke_size
    amp = torch.norm(flow[:, h0:h1, w0:w1], 2, dim=0)
    amp = amp - amp.min()
    amp = amp / amp.max()
    std, mean = amp.std(), torch.mean(amp)
    shift = torch.tensor([[h0, w0]])
    if zero:
        bg = torch.lt(amp, np.percentile(amp.numpy(), 5)).nonzero(as_tuple=False)
        src = torch.gt(amp, mean + std).nonzero(as_tuple=False)
        if src.shape[0] == 0:
            src = torch.gt(amp, mean).nonzero(as_tuple=False)
        src = src + shift
        cand = bg
    else:
        cand = torch.gt(amp, mean + std * 2.0).nonzero(as_tuple=False)
        if cand.shape[0] == 0:
            cand = torch.gt(amp, mean + std).nonzero(as_tuple=False)
            if cand.shape[0] == 0:
                cand = torch.gt(amp, mean).nonzero(as_tuple=False)
    cand = cand + shift
    if cand.shape[0] == 0:
        raise ValueError("empty candidate set")
    n = n_pokes if fix_n_pokes else int(randint(1, min(n_pokes, int(cand.shape[0])) + 1))
    if zero:
        src_sel = src[randint(src.shape[0], size=n)]
    sel = cand[randint(cand.shape[0], size=n)]
    half = int(poke_size / 2)
    poke = torch.zeros_like(flow)
    centers = torch.full((n_pokes, 2), -1, dtype=torch.int64)
    for i in range(n):
        r, c = int(sel[i, 0]), int(sel[i, 1])
        sr, sc = (int(src_sel[i, 0]), int(src_sel[i, 1])) if zero else (r, c)
        if equal_poke_val:
            val = flow[:, sr, sc].unsqueeze(-1).unsqueeze(-1)
        else:
            val = flow[:, sr - half:sr + half + 1, sc - half:sc + half + 1]
        poke[:, r - half:r + half + 1, c - half:c + half + 1] = val
    centers[:n] = sel
    return poke, centers
