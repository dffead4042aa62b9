"""Host-side mirror of the reference's conditional flow, backed by the native HIP flow engine.

``SupervisedMacowTransformer(config)`` has the call surface of
``models/modules/INN/INN.py:446-481``:

    out, logdet = flow(x, cond)                 # density direction
    x = flow(z, cond, reverse=True)             # sampling direction
    flow.flow.reshape == 'none'                 # read by PokeMotionModel.make_flow_input

and the state-dict keys of the reference (``flow.layers.{L}.{s}.actnorm1.log_scale`` ...), so that
reference checkpoints load unchanged.  All parameters are views of ONE flat fp32 device buffer whose
layout the native engine defines; gradients are written by the engine into a second flat buffer of
the same layout (``p.grad`` of every named parameter is a view of it), which is what the fused Adam
step and the data-parallel all-reduce operate on.

There is no CPU path: forward/reverse raise if the HIP library or a GPU is missing.
"""
import ctypes
from ctypes import byref, c_char_p, c_int32, c_int64, c_void_p, create_string_buffer

import os
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr

TK_PARAM, TK_IDX_FWD, TK_IDX_BWD, TK_FLAG, TK_FBUF = 0, 1, 2, 3, 4


class FlowEngine:
    """Owner of the native handle and of the flat device buffers."""

    def __init__(self, arch, dtype="bf16", max_batch=64, device=None):
        for key in ("attention", "cond_conv", "multistack"):       # augmented_input only widens flow_in_channels (second_stage.py)
            if arch.get(key, False):
                raise NotImplementedError(f"architecture option {key}=True is outside the shipped iPOKE configs")
        if float(arch.get("p_dropout", 0.0)) > 0.0:
            raise NotImplementedError("p_dropout > 0 is not implemented (0.0 in every shipped config)")
        if arch.get("reshape", "none") != "none":
            raise NotImplementedError("reshape != 'none' is not implemented (every shipped config uses 'none')")
        if arch.get("transform", "affine") != "affine" or arch.get("prior_transform", "affine") != "affine":
            raise NotImplementedError("only the affine transform is implemented (shipped configs)")
        if arch.get("activation", "elu") != "elu" or arch.get("coupling_type", "conv") != "conv":
            raise NotImplementedError("only activation='elu', coupling_type='conv' are implemented (shipped configs)")
        self.lib = _lib.lib()
        self.dtype = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        self.dtype_name = "bf16" if self.dtype == _lib.BF16 else "f32"
        cfg = _lib.FlowConfig()
        cfg.z_channels = int(arch["flow_in_channels"])
        cfg.hidden = int(arch["flow_mid_channels"])
        cfg.cond_channels = int(arch["h_channels"])
        cfg.factor = int(arch["factor"])
        steps = list(arch["num_steps"])
        cfg.n_levels = len(steps)
        for i, s in enumerate(steps):
            cfg.num_steps[i] = int(s)
        cfg.kernel_h, cfg.kernel_w = int(arch["kernel_size"][0]), int(arch["kernel_size"][1])
        cfg.dtype = self.dtype
        cfg.max_batch = int(max_batch)
        cfg.use1x1 = int(bool(arch.get("use1x1", False)))
        cfg.condition_nice = int(bool(arch.get("condition_nice", False)))      # macow2.py:1024-1060: the NICE nets see h as well
        self.cfg = cfg
        self.z, self.cond_channels, self.max_batch = cfg.z_channels, cfg.cond_channels, cfg.max_batch
        h = c_void_p()
        check(self.lib.ipoke_flow_create(byref(cfg), byref(h)))
        self.handle = h
        self.n_params = self.lib.ipoke_flow_param_count(h)
        self.n_perm = self.lib.ipoke_flow_index_count(h)
        self.n_fbuf = self.lib.ipoke_flow_float_buffer_count(h)
        self.n_ops = self.lib.ipoke_flow_op_count(h)
        self.tensors = self._tensor_table()
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = None
        self.perm = torch.zeros(max(self.n_perm, 1), dtype=torch.int32, device=self.device)
        # float buffers of the state dict (use1x1: the LU convs' permutated / sign_s / masks), read in place by the engine
        self.fbuf = torch.zeros(max(self.n_fbuf, 1), dtype=torch.float32, device=self.device)
        if self.n_fbuf > 0 and self.device.type == "cuda":
            check(self.lib.ipoke_flow_set_float_buffers(h, ptr(self.fbuf)))
        self.shadow = None
        self._ws = {}
        self._io = {}
        self.shadow_stale = True
        # data-parallel overlap: (npieces, torch.cuda.Stream, fn(begin, end)) -> the backward is issued in pieces and fn is
        # called as soon as grads[begin:end] is final on that stream (see ipoke_flow_backward_pieces)
        self.grad_ready_hook = None
        # event after which the parameters / weight shadows are final (optimizer step issued on another stream)
        self.params_ready_event = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ipoke_flow_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _tensor_table(self):
        out = []
        name = create_string_buffer(256)
        off, ndim, kind = c_int64(), c_int32(), c_int32()
        shape = (c_int64 * 4)()
        for i in range(self.lib.ipoke_flow_tensor_count(self.handle)):
            check(self.lib.ipoke_flow_tensor_info(self.handle, i, name, 256, byref(off), byref(ndim), shape, byref(kind)))
            out.append((name.value.decode(), off.value, tuple(shape[k] for k in range(ndim.value)), kind.value))
        return out

    # ---- buffers -------------------------------------------------------------------------
    def _need_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("the iPOKE flow engine runs on an MI355X only (no CPU path); construct it on a cuda device")

    def ensure_grads(self):
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        return self.grads

    def workspace(self, B, training):
        key = (int(B), bool(training))
        if key not in self._ws:
            n = self.lib.ipoke_flow_workspace_bytes(self.handle, int(B), int(training))
            if n < 0:
                raise RuntimeError("workspace query failed")
            # keep at most one buffer per mode
            for k in [k for k in self._ws if k[1] == key[1]]:
                del self._ws[k]
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
            if os.environ.get("IPOKE_POISON_WS", "0") == "1":       # developer probe: a read before the first write shows up as NaN
                self._ws[key].view(torch.int16)[: n // 2].fill_(0x7FC0)
        return self._ws[key]

    def _staging(self, B):
        """Persistent input/output buffers per batch size: the native engine replays a captured hipGraph whenever it
        is called with the same pointer arguments, so the boundary tensors must not move between steps."""
        st = self._io.get(B)
        if st is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            st = {"x": torch.empty(B, self.z, 8, 8, **f32), "cond": torch.empty(B, self.cond_channels, 8, 8, **f32),
                  "out": torch.empty(B, self.z, 8, 8, **f32), "logdet": torch.empty(B, **f32),
                  "d_out": torch.empty(B, self.z, 8, 8, **f32), "d_logdet": torch.empty(B, **f32),
                  "dx": torch.empty(B, self.z, 8, 8, **f32)}
            if len(self._io) >= 4:
                self._io.pop(next(iter(self._io)))
            self._io[B] = st
        return st

    def prepare_weights(self):
        """Refresh the matrix-core weight shadows from the master weights (after every update)."""
        self._need_gpu()
        if self.shadow is None:
            self.shadow = torch.zeros(self.lib.ipoke_flow_shadow_bytes(self.handle), dtype=torch.uint8, device=self.device)
        check(self.lib.ipoke_flow_prepare_weights(self.handle, ptr(self.params), ptr(self.shadow), _lib.current_stream()))
        self.shadow_stale = False

    def prepare_weights_range(self, begin, end):
        """Refresh the shadows of the parameters in flat[begin:end] (whole levels) on the current stream."""
        self._need_gpu()
        if self.shadow is None:
            self.shadow = torch.zeros(self.lib.ipoke_flow_shadow_bytes(self.handle), dtype=torch.uint8, device=self.device)
        check(self.lib.ipoke_flow_prepare_weights_range(self.handle, ptr(self.params), ptr(self.shadow), int(begin), int(end),
                                                        _lib.current_stream()))

    def adam_range(self, begin, end, m, v, vmax, lr, betas, eps, weight_decay, step, grad_scale=1.0, max_blocks=0):
        """Adam-amsgrad update of flat[begin:end] fused with the refresh of its weight shadows (ipoke_flow_adam_range): the plain
        1x1 weights get their operands written by the optimizer kernel itself.  Current stream."""
        self._need_gpu()
        if self.shadow is None:
            self.shadow = torch.zeros(self.lib.ipoke_flow_shadow_bytes(self.handle), dtype=torch.uint8, device=self.device)
        check(self.lib.ipoke_flow_adam_range(self.handle, ptr(self.params), ptr(self.ensure_grads()), ptr(m), ptr(v), ptr(vmax),
                                             ptr(self.shadow), int(begin), int(end), float(lr), float(betas[0]), float(betas[1]),
                                             float(eps), float(weight_decay), int(step), float(grad_scale), int(max_blocks),
                                             _lib.current_stream()))

    # ---- compute -------------------------------------------------------------------------
    def _prep_inputs(self, x, cond):
        self._need_gpu()
        B = x.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch={self.max_batch} of this flow")
        if tuple(x.shape[1:]) != (self.z, 8, 8):
            raise ValueError(f"flow input must be [B,{self.z},8,8], got {tuple(x.shape)}")
        if self.params_ready_event is not None:
            torch.cuda.current_stream().wait_event(self.params_ready_event)
            self.params_ready_event = None
        st = self._staging(B)
        st["x"].copy_(x.detach())
        if cond is not None:
            if tuple(cond.shape) != (B, self.cond_channels, 8, 8):
                raise ValueError(f"cond must be [B,{self.cond_channels},8,8], got {tuple(cond.shape)}")
            st["cond"].copy_(cond.detach())
        if self.shadow_stale:
            self.prepare_weights()
        return st, B

    def forward(self, x, cond, save_for_backward=False):
        st, B = self._prep_inputs(x, cond)
        ws = self.workspace(B, save_for_backward)
        check(self.lib.ipoke_flow_forward(self.handle, ptr(self.params), ptr(self.perm), ptr(self.shadow), ptr(st["x"]),
                                          ptr(st["cond"]), B, ptr(st["out"]), ptr(st["logdet"]), ptr(ws), int(save_for_backward),
                                          _lib.current_stream()))
        return st["out"].clone(), st["logdet"].clone()

    def init_forward(self, x):
        """Data-dependent initialisation pass (first forward of an uninitialised reference flow)."""
        self._need_gpu()
        if self.params_ready_event is not None:
            torch.cuda.current_stream().wait_event(self.params_ready_event)
            self.params_ready_event = None
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        out = torch.empty_like(x)
        logdet = torch.empty(B, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, False)
        check(self.lib.ipoke_flow_init_forward(self.handle, ptr(self.params), ptr(self.perm), ptr(x), B, ptr(out), ptr(logdet),
                                               ptr(ws), _lib.current_stream()))
        self.shadow_stale = True
        return out, logdet

    def reverse(self, z, cond):
        st, B = self._prep_inputs(z, cond)
        ws = self.workspace(B, False)
        check(self.lib.ipoke_flow_reverse(self.handle, ptr(self.params), ptr(self.perm), ptr(self.shadow), ptr(st["x"]),
                                          ptr(st["cond"]), B, ptr(st["out"]), ptr(ws), _lib.current_stream()))
        return st["out"].clone()

    def side_stream(self):
        """The engine's weight-gradient stream as a torch stream (``ipoke_flow_side_stream``), or None without a GPU."""
        if not torch.cuda.is_available():
            return None
        h = self.lib.ipoke_flow_side_stream(self.handle)
        return torch.cuda.ExternalStream(h) if h else None

    def handoff_timeouts(self):
        """(unit launches, fused conv3 + coupling launches) whose in-launch hand-off gave up since the exchange scratches were last
        initialised (``ipoke_flow_handoff_timeouts``; synchronises the device).  Anything but (0, 0) means a pass finished on garbage;
        the engine's next entry point raises by itself, this is the explicit check of trainers, tests and the benchmark."""
        from ctypes import c_uint32
        out = (c_uint32 * 2)()
        check(self.lib.ipoke_flow_handoff_timeouts(self.handle, out))
        return int(out[0]), int(out[1])

    def assert_handoffs_clean(self):
        tu, tc = self.handoff_timeouts()
        if tu or tc:
            raise RuntimeError(f"in-launch hand-off timed out ({tu} row-split unit launches, {tc} fused conv3 + coupling launches): "
                               "the results of those passes are invalid")

    def backward(self, d_out, d_logdet, need_dx=False):
        self._need_gpu()
        B = d_out.shape[0]
        st = self._staging(B)
        st["d_out"].copy_(d_out)
        st["d_logdet"].copy_(d_logdet)
        grads = self.ensure_grads()
        ws = self.workspace(B, True)
        if self.grad_ready_hook is None:
            check(self.lib.ipoke_flow_backward(self.handle, ptr(self.params), ptr(self.perm), ptr(self.shadow), ptr(st["d_out"]),
                                               ptr(st["d_logdet"]), B, ptr(grads), ptr(st["dx"]) if need_dx else None, ptr(ws),
                                               _lib.current_stream()))
        else:
            npieces, ready_stream, fn = self.grad_ready_hook
            if fn is None:        # the engine applies the optimizer itself (ipoke_flow_set_native_adam): no host callback
                check(self.lib.ipoke_flow_backward_pieces(self.handle, ptr(self.params), ptr(self.perm), ptr(self.shadow),
                                                          ptr(st["d_out"]), ptr(st["d_logdet"]), B, ptr(grads),
                                                          ptr(st["dx"]) if need_dx else None, ptr(ws), int(npieces),
                                                          c_void_p(ready_stream.cuda_stream), _lib.GRAD_READY_FN(), None, _lib.current_stream()))
                return st["dx"].clone() if need_dx else None
            failure = []

            def _ready(user, piece, begin, end):
                # an exception must not cross the C frame (ctypes would print and swallow it): keep the first one,
                # skip the remaining slices, re-raise after the native call has returned
                if failure:
                    return
                try:
                    if getattr(fn, "wants_piece", False):
                        fn(int(begin), int(end), int(piece))
                    else:
                        fn(int(begin), int(end))
                except BaseException as exc:          # noqa: BLE001
                    failure.append(exc)
            cb = _lib.GRAD_READY_FN(_ready)
            check(self.lib.ipoke_flow_backward_pieces(self.handle, ptr(self.params), ptr(self.perm), ptr(self.shadow),
                                                      ptr(st["d_out"]), ptr(st["d_logdet"]), B, ptr(grads),
                                                      ptr(st["dx"]) if need_dx else None, ptr(ws), int(npieces),
                                                      c_void_p(ready_stream.cuda_stream), cb, None, _lib.current_stream()))
            if failure:
                raise failure[0]
        return st["dx"].clone() if need_dx else None


class _FlowFunction(torch.autograd.Function):
    """out, logdet = flow(x, cond).  Parameter gradients are written by the engine straight into the
    flat gradient buffer (every named parameter's ``.grad`` is a view of it); autograd only carries the
    gradient with respect to x."""

    @staticmethod
    def forward(ctx, x, cond, anchor, engine, train):
        out, logdet = engine.forward(x, cond, save_for_backward=train)
        ctx.engine = engine
        ctx.need_dx = x.requires_grad
        return out, logdet

    @staticmethod
    def backward(ctx, d_out, d_logdet):
        dx = ctx.engine.backward(d_out, d_logdet, need_dx=ctx.need_dx)
        return dx, None, None, None, None


class _Node(nn.Module):
    """Generic container reproducing the reference's module nesting (and ModuleList integer indexing)."""

    def __getitem__(self, i):
        return getattr(self, str(i))

    def __len__(self):
        return sum(1 for k in self._modules if k.isdigit())

    def __iter__(self):
        return (self._modules[str(i)] for i in range(len(self)))


class SupervisedMacowTransformer(nn.Module):
    """Drop-in for the reference class of the same name (INN.py:446-481)."""

    def __init__(self, config, dtype="bf16", max_batch=64, device=None, init="reference"):
        super().__init__()
        self.config = config
        self.engine = FlowEngine(config, dtype=dtype, max_batch=max_batch, device=device)
        self.flow = _Node()
        self.flow.reshape = "none"                        # macow2.py:831, read at second_stage_video.py:290
        self.flow.z_channels = None
        self._views = {}          # name -> (kind, offset, shape)
        self._idx_names = []
        self._flag_names = []
        eng = self.engine
        self._anchor = eng.params.requires_grad_(True)
        for name, off, shape, kind in eng.tensors:
            assert name.startswith("flow.")
            parent, leaf = self._descend(name)
            if kind == TK_PARAM:
                n = 1
                for s in shape:
                    n *= s
                view = eng.params.detach()[off:off + n].view(shape)
                parent.register_parameter(leaf, nn.Parameter(view, requires_grad=True))
            elif kind in (TK_IDX_FWD, TK_IDX_BWD):
                parent.register_buffer(leaf, torch.arange(shape[0], dtype=torch.int64, device=eng.device))
                self._idx_names.append((name, off, shape[0]))
            elif kind == TK_FBUF:                  # float buffers are views of the engine's table, like the parameters
                n = 1
                for s_ in shape:
                    n *= s_
                parent.register_buffer(leaf, eng.fbuf[off:off + n].view(shape))
            else:
                parent.register_buffer(leaf, torch.tensor(0, dtype=torch.uint8, device=eng.device))
                self._flag_names.append(name)
            self._views[name] = (kind, off, shape)
        self._initialized = False
        if init == "reference":
            self.reset_parameters()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.sync_buffers())

    # ---- structure ---------------------------------------------------------------------------
    def _descend(self, name):
        parts = name.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        return node, parts[-1]

    def named_tensor(self, name):
        node, leaf = self._descend(name)
        return getattr(node, leaf)

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self.engine.device))
        if probe.device != self.engine.device or probe.dtype != torch.float32:
            raise RuntimeError("the flow's parameters are views of one flat device buffer; construct the module on its "
                               "final device (device=...) instead of calling .to()/.cuda()/.half()")
        return self

    @torch.no_grad()
    def reset_parameters(self):
        """Distribution-equivalent re-statement of the reference constructors' initialisers
        (macow2.py:485-487, macow_utils.py:225-229, nn.Conv2d default, flow_blocks.py:318)."""
        for name, (kind, off, shape) in self._views.items():
            t = self.named_tensor(name)
            leaf = name.rsplit(".", 1)[-1]
            if kind == TK_PARAM:
                if leaf == "log_scale":
                    t.normal_(0.0, 0.05)
                elif leaf == "bias":
                    t.zero_()
                elif leaf == "weight_v":
                    t.normal_(0.0, 0.05)
                elif leaf == "weight_g" or leaf in ("l", "u", "log_s"):
                    pass                                   # set from ||v|| / by _reset_lu below
                else:                                      # plain conv weights: kaiming_uniform(a=sqrt(5))
                    fan_in = shape[1] * shape[2] * shape[3]
                    bound = 1.0 / fan_in ** 0.5
                    t.uniform_(-bound, bound)
            elif kind == TK_IDX_FWD:
                t.copy_(torch.randperm(shape[0]).to(t.device))
            elif kind == TK_FLAG:
                t.zero_()
        self._reset_lu()
        for name, (kind, off, shape) in self._views.items():
            if kind == TK_PARAM and name.endswith("weight_g"):
                v = self.named_tensor(name[:-1] + "v")
                self.named_tensor(name).copy_(v.flatten(1).norm(dim=1).view(shape))
            if kind == TK_IDX_BWD:
                fwd = self.named_tensor(name.replace("backward_shuffle_idx", "forward_shuffle_idx"))
                self.named_tensor(name).copy_(torch.argsort(fwd))
        self.sync_buffers()

    @torch.no_grad()
    def _reset_lu(self):
        """InvertibleConvLU1d.__init__ (macow2.py:597-621): LU decomposition of a random rotation."""
        mods = sorted({n.rsplit(".", 1)[0] for n, (kind, _, _) in self._views.items() if kind == TK_FBUF})
        if not mods:
            return
        import numpy as np
        import scipy.linalg as alg
        for m in mods:
            nf = self._views[m + ".log_s"][2][0]
            w_init = np.linalg.qr(np.random.randn(nf, nf))[0].astype(np.float32)
            p, l, u = alg.lu(w_init)
            s = np.diag(u)
            lmask = np.tril(np.ones_like(w_init), -1)
            vals = {"permutated": p, "sign_s": np.sign(s), "lmask": lmask, "umask": lmask.T, "eye": np.eye(nf, dtype=np.float32),
                    "l": l, "u": np.triu(u, k=1), "log_s": np.log(np.abs(s))}
            for leaf, v in vals.items():
                t_ = self.named_tensor(m + "." + leaf)
                t_.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(t_.device).view(t_.shape))

    def sync_buffers(self):
        """Mirror the int64 shuffle buffers / uint8 flags of the state dict into the engine (after loading)."""
        perm = torch.empty(max(self.engine.n_perm, 1), dtype=torch.int32)
        for name, off, n in self._idx_names:
            perm[off:off + n] = self.named_tensor(name).to("cpu", torch.int32)
        self.engine.perm.copy_(perm)
        flags = [int(self.named_tensor(n)) for n in self._flag_names]
        if all(f == 1 for f in flags):
            self._initialized = True
        elif all(f == 0 for f in flags):
            self._initialized = False
        else:
            raise NotImplementedError("partially initialised flows (mixed `initialized` flags) are not supported")
        self.engine.shadow_stale = True

    def adopt_engine_perm(self):
        """Write the engine's permutation table back into the named int64 shuffle buffers (after a broadcast of it)."""
        perm = self.engine.perm.to("cpu", torch.int64)
        for name, off, n in self._idx_names:
            self.named_tensor(name).copy_(perm[off:off + n])

    def set_graph_mode(self, enable=True):
        """Replay forward / reverse as captured hipGraphs (BASELINE configs[4]: "hipGraph-captured" sampling).  The first call
        with a given batch size runs eagerly, the second captures, later ones replay; results are bit-identical."""
        check(self.engine.lib.ipoke_flow_set_graph(self.engine.handle, int(bool(enable))))

    def mark_weights_updated(self):
        """Call after changing parameters outside of the fused optimizer (e.g. manual edits)."""
        self.engine.shadow_stale = True

    @property
    def flat_params(self):
        return self.engine.params

    @property
    def flat_grads(self):
        return self.engine.ensure_grads()

    def bind_grads(self):
        """Make ``p.grad`` of every named parameter a view of the flat gradient buffer."""
        g = self.engine.ensure_grads()
        for name, (kind, off, shape) in self._views.items():
            if kind == TK_PARAM:
                n = 1
                for s in shape:
                    n *= s
                self.named_tensor(name).grad = g[off:off + n].view(shape)
        return g

    # ---- compute -----------------------------------------------------------------------------
    def _maybe_init(self, x):
        if not self._initialized:
            self.engine.init_forward(x)
            for n in self._flag_names:
                self.named_tensor(n).fill_(1)
            self._initialized = True

    def forward(self, input, cond, reverse=False):
        if reverse:
            return self.reverse(input, cond)
        self._maybe_init(input)
        train = torch.is_grad_enabled() and (self.training or input.requires_grad)
        if train and self.engine.grads is None:
            self.bind_grads()
        anchor = self._anchor if train else self._anchor.detach()
        out, logdet = _FlowFunction.apply(input, cond, anchor, self.engine, train)
        return out, logdet

    def reverse(self, out, cond):
        with torch.no_grad():
            return self.engine.reverse(out, cond)

    def sample(self, shape, cond, device="cpu"):
        z = torch.randn(shape).to(self.engine.device)
        return self.reverse(z, cond)
