"""CPU: the C-ABI library loads without a GPU, exports every symbol of include/ipoke_hip.h, and the native flow
engine's topology (reference state-dict names, shapes, parameter counts) matches the reference's."""
import ctypes
import os
import re
from ctypes import byref, c_int32, c_int64, c_void_p, create_string_buffer

import pytest
import torch

from ipoke_amd import _lib, configs
from tests.conftest import ROOT


def header_functions(names=("ipoke_hip.h", "ipoke_hip_dev.h")):
    """functions declared in the public header and in the developer / test-hook header"""
    out = set()
    for n in names:
        header = open(os.path.join(ROOT, "include", n)).read()
        out |= set(re.findall(r"^[a-z][a-z0-9_ ]*?\**\s*\b(ipoke_[a-z0-9_]+)\s*\(", header, flags=re.M))
    return out


def test_test_hooks_live_in_the_developer_header():
    """the drop-in boundary (ipoke_hip.h) declares no test hook / probe entry point."""
    public = header_functions(("ipoke_hip.h",))
    dev = header_functions(("ipoke_hip_dev.h",))
    hooks = {"ipoke_spin_delay", "ipoke_timing_start", "ipoke_timing_start_all", "ipoke_timing_stop", "ipoke_timing_stop_ex",
             "ipoke_set_dispatch_override", "ipoke_gru_set_fused", "ipoke_last_conv_kernel", "ipoke_conv_forward_repeat"}
    assert hooks <= dev and not (hooks & public)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = header_functions()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ipoke_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib().ipoke_version() >= 100
    assert _lib.lib().ipoke_dtype_size(_lib.BF16) == 2 and _lib.lib().ipoke_dtype_size(_lib.F32) == 4


def test_error_convention_no_gpu_needed():
    """Negative status + thread-local message, never an exception across the ABI."""
    lib = _lib.lib()
    cfg = _lib.FlowConfig()
    cfg.z_channels, cfg.hidden, cfg.cond_channels, cfg.factor, cfg.n_levels = 30, 64, 128, 4, 3      # 30 % 4 != 0
    cfg.kernel_h, cfg.kernel_w, cfg.dtype, cfg.max_batch = 2, 3, _lib.F32, 4
    h = c_void_p()
    rc = lib.ipoke_flow_create(byref(cfg), byref(h))
    assert rc == -1 and b"multiple of factor" in lib.ipoke_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)
    assert lib.ipoke_conv_forward(None, _lib.F32, None) == -1


def _topology(z):
    lib = _lib.lib()
    arch = configs.flow_arch(z)
    cfg = _lib.FlowConfig()
    cfg.z_channels, cfg.hidden, cfg.cond_channels, cfg.factor = z, arch["flow_mid_channels"], 128, 16
    cfg.n_levels = len(arch["num_steps"])
    for i, s in enumerate(arch["num_steps"]):
        cfg.num_steps[i] = s
    cfg.kernel_h, cfg.kernel_w, cfg.dtype, cfg.max_batch = 2, 3, _lib.BF16, 20
    h = c_void_p()
    _lib.check(lib.ipoke_flow_create(byref(cfg), byref(h)))
    name = create_string_buffer(256); off, nd, kind = c_int64(), c_int32(), c_int32(); shape = (c_int64 * 4)()
    names, n_float = [], 0
    for i in range(lib.ipoke_flow_tensor_count(h)):
        _lib.check(lib.ipoke_flow_tensor_info(h, i, name, 256, byref(off), byref(nd), shape, byref(kind)))
        names.append((name.value.decode(), kind.value))
        if kind.value == 0:
            n = 1
            for k in range(nd.value):
                n *= shape[k]
            n_float += n
    nops = lib.ipoke_flow_op_count(h)
    lib.ipoke_flow_destroy(h)
    return names, n_float, nops


@pytest.mark.parametrize("z,expected_params", [(32, 1054426620), (64, 1237326840)])
def test_full_size_topology_matches_reference_census(z, expected_params):
    """6 995 state-dict entries, 1.05 B / 1.24 B parameters, 50 steps, 215 NICE, 800 MCF."""
    names, n_float, nops = _topology(z)
    assert len(names) == 6995
    assert n_float == expected_params
    assert sum(1 for n, k in names if n.endswith("shift_conv.weight")) == 800
    assert sum(1 for n, k in names if n.endswith("net.conv2.weight")) == 215
    assert sum(1 for n, k in names if n.endswith("log_scale")) == 515
    assert sum(1 for n, k in names if k == 1) == 80                               # Shuffle layers
    assert names[0][0] == "flow.layers.0.0.actnorm1.log_scale"
    assert "flow.priors.14.actnorm.bias" in dict(names) and "flow.shuffle_layers.14.backward_shuffle_idx" in dict(names)


@pytest.mark.parametrize("use1x1", [False, True])
def test_reduced_topology_keys_equal_oracle_state_dict(use1x1):
    from ipoke_amd.flow import SupervisedMacowTransformer
    from oracle import flow_ref
    arch = configs.reduced_flow_arch()
    arch["use1x1"] = use1x1       # LU-parametrised 1x1 convs as shuffle layers: l, u, log_s + five float buffers per level
    m = SupervisedMacowTransformer(arch, dtype="f32", device="cpu")
    o = flow_ref.SupervisedMacowTransformer(arch)
    sa, sb = m.state_dict(), o.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        assert tuple(sa[k].shape) == tuple(sb[k].shape) and sa[k].dtype == sb[k].dtype, k
    m.load_state_dict(sb)                                    # reference-keyed checkpoint loads, buffers are mirrored
    assert torch.equal(m.engine.perm[:16].long(), o.flow.layers[0][0].conv1x1.forward_shuffle_idx)
    if use1x1:                                               # float buffers are views of the engine's table
        lu = o.flow.shuffle_layers[0]
        assert torch.equal(m.state_dict()["flow.shuffle_layers.0.permutated"], lu.permutated)
        off = dict((n, o_) for n, o_, _, _ in m.engine.tensors)["flow.shuffle_layers.0.permutated"]
        assert torch.equal(m.engine.fbuf[off:off + 256].view(16, 16), lu.permutated)
        # the host mirror's own initialiser produces a valid decomposition: P (L*lmask+I) (U*umask+diag(s)) is orthogonal
        m2 = SupervisedMacowTransformer(arch, dtype="f32", device="cpu")
        sd = m2.state_dict(); pre = "flow.shuffle_layers.0."
        wl = sd[pre + "l"] * sd[pre + "lmask"] + sd[pre + "eye"]
        wu = sd[pre + "u"] * sd[pre + "umask"] + torch.diag(sd[pre + "sign_s"] * sd[pre + "log_s"].exp())
        w = sd[pre + "permutated"] @ wl @ wu
        assert (w @ w.t() - torch.eye(16)).abs().max() <= 1e-4
    assert m.flow.reshape == "none"
    with pytest.raises(RuntimeError):                        # no CPU fallback: compute needs the GPU
        m(torch.zeros(1, 16, 8, 8), torch.zeros(1, 128, 8, 8))


def test_first_stage_keys_equal_oracle_state_dict():
    from ipoke_amd.first_stage import FirstStageWrapper, SpadeCondMotionModel
    from oracle import vae_ref
    for size in (64, 128):
        cfg = configs.first_stage_config(size, 32, 16)
        a, b = SpadeCondMotionModel(cfg).state_dict(), vae_ref.SpadeCondMotionModel(cfg).state_dict()
        assert set(a) == set(b)
        assert all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)
    a = FirstStageWrapper(configs.encoder2d_config(128, 3)).state_dict()
    b = vae_ref.FirstStageWrapper(configs.encoder2d_config(128, 3)).state_dict()
    assert set(a) == set(b)


def test_second_stage_host_logic_on_cpu(golden):
    """LR rule, config derivation and optimizer wiring of PokeMotionModel (no compute)."""
    from ipoke_amd.second_stage import PokeMotionModel, linear_var
    arch = configs.flow_arch(32, hidden=64, num_steps=[2, 1, 1], factor=4)
    arch["flow_mid_channels_factor"] = 2
    conf = configs.second_stage_config(64, 32, 16, batch_size=2, arch=arch)
    m = PokeMotionModel(conf, dirs={}, dtype="f32", device="cpu", max_batch=2)
    assert m.config["architecture"]["h_channels"] == 128 and m.config["architecture"]["flow_mid_channels"] == 64
    assert m.poke_key == "flow" and m.loss_func.logdet_weight == 1.0
    g = golden("g6_glue_64")
    for it, lr in zip(g["lr_its"], g["lr_vals"]):
        it = int(it)
        got = m.lr_scaling(it) if it < 500 else m.lr_adaptation(it, end_it=200000)
        assert abs(got - float(lr)) < 1e-12
    assert linear_var(250, 0, 500, 0.0, 1e-3, 0.0, 1e-3) == pytest.approx(5e-4)


def test_deterministic_fill_is_name_keyed():
    from ipoke_amd.utils.detfill import fill_value
    a = fill_value("flow.x.weight", torch.empty(4, 3, 3, 3))
    b = fill_value("flow.x.weight", torch.empty(4, 3, 3, 3))
    c = fill_value("flow.y.weight", torch.empty(4, 3, 3, 3))
    assert torch.equal(a, b) and not torch.equal(a, c)
    f = fill_value("s.forward_shuffle_idx", torch.empty(8, dtype=torch.int64))
    r = fill_value("s.backward_shuffle_idx", torch.empty(8, dtype=torch.int64))
    assert torch.equal(r, torch.argsort(f)) and sorted(f.tolist()) == list(range(8))


def test_descriptor_structs_of_the_binding_match_the_header():
    """ctypes mirrors of the descriptor structs (ipoke_amd/_lib.py) against sizeof in the library: a field added on one side only would
    shift every field behind it without any error."""
    import ctypes
    from ipoke_amd import _lib
    out = (ctypes.c_int32 * 16)()
    n = _lib.lib().ipoke_desc_sizes(out, 16)
    mirrors = [_lib.ConvDesc, _lib.WgradDesc, _lib.AffineDesc, _lib.CouplingEpi, _lib.McfDesc, _lib.UnitPairDesc, _lib.FlowConfig,
               _lib.NormDesc, _lib.NormBwdDesc, _lib.RowScaleBwdDesc, _lib.SnJob, _lib.WgradAdam]
    assert n == len(mirrors)
    for i, m in enumerate(mirrors):
        assert ctypes.sizeof(m) == out[i], (m.__name__, ctypes.sizeof(m), out[i])


def test_package_import_sets_the_hardware_queue_default():
    """the step keeps four streams busy; ``import ipoke_amd`` sets GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises unless
    the variable is given (bench.py used to be the only place that did)."""
    import subprocess
    import sys
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import ipoke_amd; "
            "print(os.environ['GPU_MAX_HW_QUEUES'], ipoke_amd.hw_queue_setting())")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "8 ('8', True)", out.stdout
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '4'; import ipoke_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "4", (out.stdout, out.stderr)
