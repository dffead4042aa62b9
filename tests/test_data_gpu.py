"""Device data path (ipoke_amd/data.py + csrc/data.hip) against golden G11: the reference's own BaseDataset._get_flow / _get_poke
(data/base_dataset.py:507-693) run on synthetic raw flows with its np.random.randint fed from stored uniforms.  Index work is
bit-exact (poke centres, the set of poked pixels); values are copies of flow entries, so they equal the resized flow bit for bit."""
import numpy as np
import pytest
import torch

from ipoke_amd.data import FlowError, PokeSimulator
from tests.conftest import t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _sim(meta, n_pokes=5):
    size, poke_size, zero, equal, fix = (int(v) for v in meta)
    return PokeSimulator({"spatial_size": (size, size), "n_pokes": n_pokes, "poke_size": poke_size, "scale_poke_to_res": True,
                          "equal_poke_val": bool(equal), "fix_n_pokes": bool(fix)}), bool(zero)


def test_flow_resize_and_poke_cases(golden):
    g = golden("g11_data_path")
    for ci in range(int(g["n_cases"])):
        sim, zero = _sim(g[f"meta{ci}"])
        raw = t(g[f"raw{ci}"], DEV).unsqueeze(0)
        flow = sim.get_flow(raw)
        want_flow = t(g[f"flow{ci}"])
        err = (flow[0].cpu() - want_flow).abs().max().item()
        assert err <= 2e-6 * max(1.0, want_flow.abs().max().item()), (ci, err)
        # the poke simulation consumes the reference's resized flow, so that index decisions are compared on identical inputs
        poke, centers, flow_out, status = sim.get_poke(want_flow.unsqueeze(0).to(DEV), torch.tensor([zero]), t(g[f"u{ci}"], DEV).unsqueeze(0))
        assert int(status[0]) == 0
        assert np.array_equal(centers[0].cpu().numpy(), g[f"centers{ci}"]), (ci, centers[0].tolist(), g[f"centers{ci}"].tolist())
        p = poke[0].cpu()
        assert np.array_equal(p.nonzero().numpy().astype(np.int16), g[f"poke_nz{ci}"]), ci
        assert np.array_equal(p[p != 0].numpy(), g[f"poke_val{ci}"]), ci
        assert abs(flow_out.abs().sum().item() - float(g[f"flow_ret_abs{ci}"])) <= 1e-3 * max(1.0, float(g[f"flow_ret_abs{ci}"]))
        if zero:
            assert flow_out.abs().max().item() == 0.0
        else:
            assert torch.equal(flow_out[0].cpu(), want_flow)


def test_batched_equals_per_sample(golden):
    """One launch over a batch == the per-sample results (mixed ordinary / zero-poke samples in one batch)."""
    g = golden("g11_data_path")
    ids = [ci for ci in range(int(g["n_cases"])) if tuple(g[f"meta{ci}"][[0, 1, 3, 4]]) == (128, 5, 1, 0)]
    sim, _ = _sim(g[f"meta{ids[0]}"])
    flows = torch.stack([t(g[f"flow{ci}"]) for ci in ids]).to(DEV)
    zero = torch.tensor([bool(g[f"meta{ci}"][2]) for ci in ids])
    u = torch.stack([t(g[f"u{ci}"]) for ci in ids]).to(DEV)
    poke, centers, flow_out, status = sim.get_poke(flows, zero, u)
    assert int(status.sum()) == 0
    for j, ci in enumerate(ids):
        assert np.array_equal(centers[j].cpu().numpy(), g[f"centers{ci}"])
        assert abs(poke[j].abs().sum().item() - float(g[f"poke_abs{ci}"])) <= 1e-4 * float(g[f"poke_abs{ci}"])


def test_properties_at_batch_size():
    """B = 64 random smooth flows at 128 px: every centre lies in the candidate window, pokes are (2 half + 1)^2 patches holding flow
    vectors of the same sample, the number of centres is within [1, n_pokes], and the own-RNG path is reproducible from a generator."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    coarse = torch.randn(64, 2, 6, 6, device=DEV, generator=gen)
    raw = torch.nn.functional.interpolate(coarse, size=(256, 256), mode="bicubic", align_corners=False) * 10
    sim = PokeSimulator({"spatial_size": (128, 128), "n_pokes": 5, "poke_size": 5, "scale_poke_to_res": True})
    zero = torch.arange(64) % 12 == 0
    batch = sim.make_batch(torch.zeros(64, 16, 3, 128, 128, device=DEV), raw, zero, generator=torch.Generator(device=DEV).manual_seed(9))
    again = sim.make_batch(torch.zeros(64, 16, 3, 128, 128, device=DEV), raw, zero, generator=torch.Generator(device=DEV).manual_seed(9))
    poke, centers = batch["poke"]
    assert torch.equal(poke, again["poke"][0]) and torch.equal(centers, again["poke"][1])
    flow = sim.get_flow(raw)
    n = (centers[:, :, 0] >= 0).sum(1)
    assert int(n.min()) >= 1 and int(n.max()) <= 5
    valid = centers[centers[:, :, 0] >= 0]
    assert int(valid.min()) >= 5 and int(valid.max()) < 123
    assert bool((batch["flow"][zero] == 0).all()) and torch.equal(batch["flow"][~zero], flow[~zero])
    for b in range(64):
        k = int(n[b])
        r, c = centers[b, k - 1].tolist()                       # the last poke is never overwritten
        patch = poke[b, :, r - 2:r + 3, c - 2:c + 3]
        assert bool((patch == patch[:, :1, :1]).all())
        if not bool(zero[b]):
            assert torch.equal(patch[:, 0, 0], flow[b, :, r, c])
        assert int((poke[b].abs().sum(0) > 0).sum()) <= 25 * k


def test_no_candidate_is_reported():
    sim = PokeSimulator({"spatial_size": (64, 64), "n_pokes": 5, "poke_size": 5})
    flow = torch.zeros(2, 2, 64, 64, device=DEV)                # constant amplitude: 0 / 0 -> NaN everywhere, no candidate passes
    flow[1] = torch.randn(2, 64, 64, device=DEV)
    poke, centers, _, status = sim.get_poke(flow, strict=False)
    assert status.tolist() == [1, 0] and int(centers[0].max()) == -1 and poke[0].abs().max().item() == 0
    with pytest.raises(FlowError):
        sim.get_poke(flow)
