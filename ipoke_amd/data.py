"""Batched, on-device counterparts of the reference data set's flow / poke preparation (data/base_dataset.py).

``PokeSimulator`` carries the attributes ``BaseDataset.__init__`` derives from the data config (:40-78, :185-186: spatial_size,
poke_size, n_pokes, fix_n_pokes, equal_poke_val, scale_poke_to_res) and offers

    get_flow(raw)                 _get_flow (:651-693)   raw flows [B, 2, Hs, Ws] -> [B, 2, H, W] (scaled to the resolution, bilinear)
    get_poke(flow, zero, u)       _get_poke (:507-648)   -> (poke [B, 2, H, W], poke_centers int64 [B, n_pokes, 2], flow_out, status)

so that a loader only has to deliver raw flows and frames; everything downstream of the file read happens in HBM.  The random
draws are uniforms supplied by the caller (or drawn here from a torch generator): see ``ipoke_poke_simulate`` in
include/ipoke_hip.h for how they map to the reference's ``np.random.randint`` calls.  No CPU path.
"""
import torch

from . import _lib
from ._lib import check, ptr


class FlowError(RuntimeError):
    """No poke candidate in a sample (the reference raises FlowError and draws another sample, base_dataset.py:588-589)."""


class PokeSimulator:
    def __init__(self, config):
        self.config = config
        assert "spatial_size" in config
        self.spatial_size = tuple(config["spatial_size"])
        self.n_pokes = int(config["n_pokes"])
        self.fix_n_pokes = bool(config.get("fix_n_pokes", False)) or self.n_pokes == 1
        self.scale_poke_to_res = bool(config.get("scale_poke_to_res", False))
        self.poke_size = config["poke_size"] if "poke_size" in config else self.spatial_size[0] / 128 * 10
        if int(self.poke_size) != self.poke_size:
            raise ValueError("poke_size must be integral (the reference slices tensors with it)")
        self.poke_size = int(self.poke_size)
        self.equal_poke_val = bool(config.get("equal_poke_val", True))
        self.valid_h = [self.poke_size, self.spatial_size[0] - self.poke_size]
        self.valid_w = [self.poke_size, self.spatial_size[1] - self.poke_size]

    def get_flow(self, raw):
        _lib.require_gpu()
        raw = raw.contiguous().float()
        B, C, Hs, Ws = raw.shape
        H, W = self.spatial_size
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=raw.device)
        div = Hs / H if self.scale_poke_to_res else 1.0
        check(_lib.lib().ipoke_flow_resize(ptr(raw), ptr(out), B, C, Hs, Ws, H, W, div, _lib.current_stream()))
        return out

    def get_poke(self, flow, zero_poke=None, u=None, generator=None, strict=True):
        """flow fp32 [B, 2, H, W] on the device; zero_poke: bool/int [B] (samples with seq_len_idx == -1) or None;
        u: fp32 [B, 1 + 2 n_pokes] uniforms in [0, 1) or None (drawn on the device from ``generator``)."""
        _lib.require_gpu()
        flow = flow.contiguous().float()
        B, C, H, W = flow.shape
        assert C == 2 and (H, W) == self.spatial_size
        dev = flow.device
        if u is None:
            u = torch.rand(B, 1 + 2 * self.n_pokes, device=dev, generator=generator)
        u = u.to(dev, torch.float32).contiguous()
        assert u.shape == (B, 1 + 2 * self.n_pokes)
        zero = None if zero_poke is None else zero_poke.to(dev).to(torch.int32).contiguous()
        poke = torch.empty_like(flow)
        flow_out = torch.empty_like(flow)
        centers = torch.empty(B, self.n_pokes, 2, dtype=torch.int64, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        ws = torch.empty(_lib.lib().ipoke_poke_workspace_bytes(B, H, W, self.poke_size, self.n_pokes), dtype=torch.uint8, device=dev)
        check(_lib.lib().ipoke_poke_simulate(ptr(flow), B, H, W, self.poke_size, self.n_pokes, int(self.fix_n_pokes), int(self.equal_poke_val),
                                             ptr(zero), ptr(u), ptr(poke), ptr(centers), ptr(flow_out), ptr(status), ptr(ws), _lib.current_stream()))
        if strict and bool(status.any()):
            raise FlowError(f"Empty indices array for samples {status.nonzero().flatten().tolist()}")
        return poke, centers, flow_out, status

    def make_batch(self, images, raw_flow, zero_poke=None, u=None, generator=None):
        """The ``batch`` dict the second stage consumes (images, flow, poke = [poke, poke_centers]) from frames already on the device
        and raw flows: the part of BaseDataset.__getitem__ that follows the file reads."""
        flow = self.get_flow(raw_flow)
        # no host synchronisation on the loader path: samples without a candidate (the reference's FlowError) are flagged in
        # ``poke_status`` (int32 [B], 1 = resample), for the caller to inspect when it chooses to
        poke, centers, flow_out, status = self.get_poke(flow, zero_poke, u, generator, strict=False)
        return {"images": images, "flow": flow_out, "poke": [poke, centers], "poke_status": status}
