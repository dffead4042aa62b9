"""``PokeMotionModel``: the second-stage module of iPOKE with the reference's call surface
(models/second_stage_video.py:31-452, 632-662) on the HIP flow engine and the HIP first-stage VAE.

The reference class is a LightningModule whose ``__init__(config, dirs)`` loads three checkpoints from
``logs/...``; Lightning and those files do not exist here, so this is a plain ``nn.Module`` that keeps the
Lightning method names (``training_step``, ``configure_optimizers``, ``on_train_batch_start`` ...) and takes
the three sub-model configurations from ``config['first_stage']``, ``config['poke_embedder']`` and
``config['conditioner_model']`` (dictionaries with the reference's YAML keys).  Checkpoints are loaded with
``load_first_stage / load_poke_embedder / load_conditioner`` which mirror ``__initialize_*`` (:182-236).
``ipoke_amd.trainer.SecondStageTrainer`` is the loop that drives it (the counterpart of pl.Trainer.fit).
"""
from functools import partial

import numpy as np
import os

import torch
import torch.nn as nn

from .first_stage import FirstStageWrapper, SpadeCondMotionModel
from .flow import SupervisedMacowTransformer
from .loss import FlowLoss
from .optim import FusedAdamAmsgrad


def linear_var(act_it, start_it, end_it, start_val, end_val, clip_min, clip_max):
    """utils/general.py:221-228."""
    act_val = float(end_val - start_val) / (end_it - start_it) * (act_it - start_it) + start_val
    return float(np.clip(act_val, a_min=clip_min, a_max=clip_max))


class PokeMotionModel(nn.Module):
    def __init__(self, config, dirs=None, dtype="bf16", device=None, max_batch=None):
        super().__init__()
        self.config = config
        self.dirs = dirs or {}
        self.dtype = dtype
        self.embed_poke = True
        self.test_mode = config["general"].get("test", "none")
        tr = config["training"]
        self.spatial_mean_for_loss = bool(tr.get("spatial_mean", False))
        logdet_weight = config["logdet_weight"] if "logdet_weight" in config else 1.0      # looked up at top level (:41)
        if tr.get("adabelief", False):
            raise NotImplementedError("AdaBelief can never be selected by shipped configs (key mismatch, SURVEY §2 row 4b)")
        self.n_test_samples = config["testing"]["n_samples_per_data_point"]
        self.global_step = 0
        self.current_epoch = 0
        self._optimizer = None
        self.device_ = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")

        lr = tr["lr"]
        self.apply_lr_scaling = bool(tr.get("lr_scaling", False))
        if self.apply_lr_scaling:
            self.lr_scaling = partial(linear_var, start_it=0, end_it=tr["lr_scaling_max_it"], start_val=0.0, end_val=lr,
                                      clip_min=0.0, clip_max=lr)
        self.custom_lr_decrease = tr["custom_lr_decrease"]
        if self.custom_lr_decrease:
            self.lr_adaptation = partial(linear_var, start_it=tr["lr_scaling_max_it"], start_val=lr, end_val=0.0, clip_min=0.0,
                                         clip_max=lr)
            # on_train_epoch_start (:317-323): end_it = n_epochs * num_training_batches (capped at max_batches_per_epoch)
            self._lr_end_it = tr["n_epochs"] * tr.get("max_batches_per_epoch", 2000)

        # frozen first-stage model, poke embedder, conditioner
        self.first_stage_config = config["first_stage"]
        self.first_stage_model = SpadeCondMotionModel(self.first_stage_config, dirs=self.dirs, train=False, dtype=dtype)
        self.full_seq = bool(tr.get("full_seq", False))
        self.use_cond = config["conditioner"].get("use", True)
        self.poke_emb_config = config["poke_embedder"]
        self.poke_embedder = FirstStageWrapper(self.poke_emb_config, dtype=dtype)
        if self.use_cond:
            self.conditioner_config = config["conditioner_model"]
            self.conditioner = FirstStageWrapper(self.conditioner_config, dtype=dtype)
        arch = config["architecture"]
        self.augment_input = bool(arch.get("augmented_input", False))
        arch["flow_in_channels"] = self.first_stage_config["architecture"]["z_dim"]
        if self.augment_input:
            # noise channels appended to the latent (:66-79, :304-308).  The flow input is detached (:348), so the "trainable"
            # scale / shift never receive a gradient in the reference either: they are kept as parameters for the state dict only
            n_aug = int(arch["augment_channels"])
            arch["flow_in_channels"] += n_aug
            if arch.get("scale_augmentation", False):
                self.scale_augment = torch.nn.Parameter(torch.ones(n_aug), requires_grad=False)
            else:
                self.register_buffer("scale_augment", torch.ones(n_aug))
            if arch.get("shift_augmentation", False):
                self.shift_augment = torch.nn.Parameter(torch.zeros(n_aug), requires_grad=False)
            else:
                self.register_buffer("shift_augment", torch.zeros(n_aug))
        pe_arch = self.poke_emb_config["architecture"]
        self.embed_poke_and_image = bool(pe_arch.get("poke_and_image", False))
        self.poke_key = "flow" if pe_arch.get("flow_ae", False) else "poke"
        assert not (self.poke_key == "flow" and self.embed_poke_and_image)
        arch["h_channels"] = pe_arch["nf_max"] + (self.conditioner_config["architecture"]["nf_max"] if self.use_cond else 0)
        arch["flow_mid_channels"] = int(arch["flow_mid_channels_factor"] * arch["flow_in_channels"])
        arch["ssize"] = pe_arch["min_spatial_size"]
        fs_min = self.first_stage_config["architecture"]["min_spatial_size"]
        self.adapt_poke_emb_ssize = pe_arch["min_spatial_size"] != fs_min
        self.adapt_cond_ssize = self.use_cond and self.conditioner_config["architecture"]["min_spatial_size"] != fs_min
        if self.adapt_poke_emb_ssize:
            # second_stage_video.py:114-118 builds a stride-`factor` Conv2d when the poke latent is SMALLER than the first stage's
            # (factor > 1: 4x4 -> 2x2) and a stride-1/factor transposed block when it is LARGER (16x16 -> 32x32): both move the map AWAY
            # from the first-stage size, and the torch.cat of :311 fails in the reference itself.  Nothing to be compatible with.
            raise NotImplementedError("adapt_poke_emb_ssize: the reference's own adapter resizes in the wrong direction for either ratio "
                                      "(second_stage_video.py:114-118) and cannot run; use a poke embedder with the first stage's latent size")
        if self.adapt_cond_ssize:
            from .first_stage import Conv2dTransposeBlock
            factor = float(fs_min) / self.conditioner_config["architecture"]["min_spatial_size"]
            if factor < 1 or factor != int(factor):
                # :123-126: `nn.Conv2d(..., stride=int(factor))` with factor < 1 is a stride-0 convolution ("non-positive stride")
                raise NotImplementedError("adapt_cond_ssize with a conditioner latent larger than the first stage's builds a stride-0 "
                                          "Conv2d in the reference (second_stage_video.py:123-126): no behaviour to reproduce")
            nf = self.conditioner_config["architecture"]["nf_max"]
            # Conv2dTransposeBlock(nf, nf, st=factor, ks=3, padding=1): ConvTranspose2d + ("elu" -> ReLU); trainable in name only --
            # it is applied under torch.no_grad (:268-290) and so never receives a gradient in the reference either
            self.conv_adapt_cond = Conv2dTransposeBlock(nf, nf, 3, int(factor), 1, norm="none", activation="elu", snorm=False)
        if arch.get("multistack", False):
            raise NotImplementedError("multistack flows are outside the shipped configs")
        mb = max_batch if max_batch is not None else max(config["data"]["batch_size"], 1)
        self.first_stage_model.to(self.device_); self.poke_embedder.to(self.device_)
        if self.use_cond:
            self.conditioner.to(self.device_)
        if self.adapt_cond_ssize:
            self.conv_adapt_cond.to(self.device_)
        self.flow = SupervisedMacowTransformer(arch, dtype=dtype, max_batch=mb, device=self.device_)
        self.loss_func = FlowLoss(spatial_mean=self.spatial_mean_for_loss, logdet_weight=logdet_weight)
        self.logged = {}

    # ---- Lightning-compatible no-ops ---------------------------------------------------------------
    def log(self, name, value, **kw):
        self.logged[name] = value

    def log_dict(self, d, **kw):
        self.logged.update(d)

    def optimizers(self):
        return self._optimizer

    # ---- checkpoints (second_stage_video.py:182-236) -----------------------------------------------
    @staticmethod
    def _sd(path_or_sd):
        sd = torch.load(path_or_sd, map_location="cpu") if isinstance(path_or_sd, str) else path_or_sd
        return sd["state_dict"] if "state_dict" in sd else sd

    def load_checkpoint(self, ckpt, strict=False):
        """Counterpart of ``SecondStageVideoModel``'s ``load_from_checkpoint(..., strict=False)`` (experiments/
        second_stage_video.py:22, experiments/experiment.py:107-143): a Lightning ``.ckpt`` (path or loaded dict) whose
        ``state_dict`` holds ``flow.flow.*`` (weight_g / weight_v, int64 shuffle indices, uint8 ``initialized`` flags),
        ``first_stage_model.*`` (spectral-norm ``weight_orig / weight_u / weight_v``), ``poke_embedder.*``, ``conditioner.*``."""
        res = self.load_state_dict(self._sd(ckpt), strict=strict)
        self.flow.sync_buffers()
        self.first_stage_model.enc_motion.conv1.invalidate()
        self._drop_graphs()
        return res

    def load_state_dict(self, *args, **kwargs):
        res = super().load_state_dict(*args, **kwargs)
        self._drop_graphs()
        return res

    def _drop_graphs(self):
        """Captured graphs hold the ADDRESSES of the cached weight operands: a state-dict load drops those caches (``_Cached``), so the
        graphs of ``set_sample_graph`` / ``set_encoder_graph`` are recaptured on their next use."""
        if getattr(self, "_sample_graphs", None):
            self._sample_graphs = {}
        if getattr(self, "_enc_graphs", None):
            self._enc_graphs = {}

    def load_first_stage(self, ckpt):
        self.first_stage_model.load_state_dict(self._sd(ckpt), strict=False)
        self.first_stage_model.enc_motion.conv1.invalidate()
        self._drop_graphs()

    def load_poke_embedder(self, ckpt):
        sd = self._sd(ckpt)
        sd = {".".join(k.split(".")[1:]): v for k, v in sd.items() if "encoder" in k or "decoder" in k}
        self.poke_embedder.load_state_dict(sd, strict=False)
        self._drop_graphs()

    def load_conditioner(self, ckpt):
        sd = {k: v for k, v in self._sd(ckpt).items() if "encoder" in k or "decoder" in k}
        self.conditioner.load_state_dict(sd, strict=False)
        self._drop_graphs()

    # ---- LR rule (:238-253) ---------------------------------------------------------------------------
    def on_train_epoch_start(self, num_training_batches=None):
        if self.custom_lr_decrease and num_training_batches is not None:
            self._lr_end_it = self.config["training"]["n_epochs"] * num_training_batches

    def on_train_batch_start(self, batch, batch_idx, dataloader_idx=0):
        max_it = self.config["training"]["lr_scaling_max_it"]
        opt = self.optimizers()
        if self.apply_lr_scaling and self.global_step < max_it:
            lr = self.lr_scaling(self.global_step)
            self.log("learning_rate", lr)
            for pg in opt.param_groups:
                pg["lr"] = lr
        if self.custom_lr_decrease and self.global_step >= max_it:
            lr = self.lr_adaptation(self.global_step, end_it=self._lr_end_it)
            for pg in opt.param_groups:
                pg["lr"] = lr

    # ---- hot path ---------------------------------------------------------------------------------------
    def make_flow_input(self, batch, reverse=False, use_kp_poke=False):
        X = batch["images"]
        if use_kp_poke:
            poke, *_ = batch["keypoint_poke"]
            poke = poke.to(torch.float)
        else:
            poke = batch[self.poke_key]
            poke = poke[0] if isinstance(poke, list) else poke
        if self.embed_poke_and_image:
            poke = torch.cat([poke, X[:, 0]], dim=1)
        self.first_stage_model.eval(); self.poke_embedder.eval()
        if self.use_cond:
            self.conditioner.eval()                   # :271-272
        with torch.no_grad():
            poke_emb, *_ = self.poke_embedder.encoder(poke)
            if self.use_cond:
                cond, *_ = self.conditioner.encoder(X[:, 0])
                if self.adapt_cond_ssize:
                    cond = self._adapt_cond(cond)
        if reverse:
            spatial = self.first_stage_config["architecture"]["min_spatial_size"]
            # CPU generator, then moved: torch.randn(...).type_as(X) (:296-300)
            flow_input = torch.randn((X.size(0), self.config["architecture"]["flow_in_channels"], spatial, spatial)).type_as(X).detach()
        else:
            with torch.no_grad():
                flow_input, *_ = self.encode_first_stage(X)
                if self.augment_input:
                    n_aug = self.config["architecture"]["augment_channels"]
                    noise = torch.randn((flow_input.size(0), n_aug, *flow_input.shape[-2:])).type_as(X)
                    noise = self.scale_augment.to(X.device)[None, :, None, None] * noise + self.shift_augment.to(X.device)[None, :, None, None]
                    flow_input = torch.cat([flow_input, noise], dim=1)
        cond = torch.cat([cond, poke_emb], dim=1) if self.use_cond else poke_emb
        return flow_input, cond

    def _adapt_cond(self, cond):
        """conv_adapt_cond (second_stage_video.py:120-129, 286-287): the conditioner's 4x4 (...) latent brought to the first stage's size."""
        from . import nn as K
        y = self.conv_adapt_cond.run(K.from_nchw(cond.float(), self.dtype), self.dtype)
        return K.to_nchw(y, self.dtype)

    def encode_first_stage(self, X):
        with torch.no_grad():
            if self.full_seq:
                X_in = X if self.first_stage_model.full_sequence or self.config["data"]["max_frames"] < 16 else X[:, :-1]
            else:
                X_in = X if self.first_stage_model.full_sequence else X[:, 1:]
            motion, mu, cov = self.first_stage_model.enc_motion(X_in.transpose(1, 2))
        return motion, mu

    def decode_first_stage(self, motion, X, length=None):
        if length is None:
            length = X.size(1) - 1
        return self.first_stage_model.decode(motion, X[:, 0], length)

    def prefetch_flow_input(self, batch, stream, after=None):
        """Run the frozen encoders for ``batch`` on ``stream`` now; the next ``forward_density(batch)`` (same object) picks the
        result up.  The encoders do not depend on the flow's parameters, so this overlaps with the current step's
        backward pass (what a data-loader worker does for the reference's frozen first stage)."""
        cur = torch.cuda.current_stream()
        # the side stream starts after everything already queued on the caller's stream: the batch may have been produced
        # there (non-blocking H2D copies, GPU-side augmentation), and so are the lazily built weight operands of the first
        # call.  It is issued right after the forward pass, so the overlap with the backward pass is kept.
        # ``after``: an event recorded on the caller's stream behind which the side stream may start (the trainer records
        # it right after the forward pass and calls this once the backward pass is queued, so that the host does not hold
        # the backward chain back by the ~7 ms it takes to queue the encoders)
        if after is not None:
            stream.wait_event(after)
        else:
            stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            if getattr(self, "_graph_encoders", False) and not self.augment_input:
                flow_input, cond = self._flow_input_graphed(batch)
            else:
                flow_input, cond = self.make_flow_input(batch)
            ev = torch.cuda.Event(); ev.record(stream)
        for v in batch.values():                       # tensors the side stream reads must not be recycled under it
            for t_ in (v if isinstance(v, (list, tuple)) else [v]):
                if torch.is_tensor(t_) and t_.is_cuda:
                    t_.record_stream(stream)
        flow_input.record_stream(cur); cond.record_stream(cur)
        if not isinstance(getattr(self, "_prefetched", None), dict):
            self._prefetched = {}
        # keyed FIFO: the next batch may be prefetched before this one is consumed (and a loop may feed the same object twice)
        self._prefetched.setdefault(id(batch), []).append((batch, flow_input, cond, ev))

    # ---- the frozen encoders as ONE captured hipGraph -----------------------------------------------------------------------
    def set_encoder_graph(self, enable=True):
        """Replay the device side of ``make_flow_input`` (2-D poke / image encoders + the 3-D motion encoder: ~200 launches that Python
        needs ~4 ms to queue for 3.6 ms of kernels) from one captured hipGraph per input shape when it is PREFETCHED for the next step
        (``prefetch_flow_input``).  The replay is one host call, and inside the graph the two small 2-D encoders run BESIDE the 3-D encoder
        (a second captured stream: no allocator traffic at replay).  MEASURED AND NOT ADOPTED AS DEFAULT (round 4, c2): the hole behind the
        backward pass shrinks from 5.5 to 4.7 ms, but the step is 60.2-61.2 ms against 59.2-59.4 eager -- the forward pass of the next step
        picks up gaps it did not have (IPOKE_ENC_GRAPH=1 turns it on in SecondStageTrainer; bit-identical results either way).  The reparameterisation noise is still drawn on the CPU generator per call, at the same point of the host program as the
        eager path (motion_encoder.py:220), and copied into the graph's input buffer."""
        self._graph_encoders = bool(enable)
        self._enc_graphs = {}

    def _flow_input_device(self, X, poke, eps, side=None):
        """Device side of ``make_flow_input`` (reverse = False) with the noise given; ``side``: a stream for the 2-D encoders."""
        if self.embed_poke_and_image:
            poke = torch.cat([poke, X[:, 0]], dim=1)
        cur = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(cur)
        with torch.cuda.stream(side if side is not None else cur):
            poke_emb, *_ = self.poke_embedder.encoder(poke)
            cond = None
            if self.use_cond:
                cond, *_ = self.conditioner.encoder(X[:, 0])
                if self.adapt_cond_ssize:
                    cond = self._adapt_cond(cond)
        if self.full_seq:
            X_in = X if self.first_stage_model.full_sequence or self.config["data"]["max_frames"] < 16 else X[:, :-1]
        else:
            X_in = X if self.first_stage_model.full_sequence else X[:, 1:]
        flow_input, _, _ = self.first_stage_model.enc_motion(X_in.transpose(1, 2), eps=eps)
        if side is not None:
            cur.wait_stream(side)
        cond = torch.cat([cond, poke_emb], dim=1) if self.use_cond else poke_emb
        return flow_input, cond

    def _flow_input_graphed(self, batch):
        X = batch["images"]
        poke = batch[self.poke_key]
        poke = poke[0] if isinstance(poke, list) else poke
        key = (tuple(X.shape), tuple(poke.shape), X.dtype, poke.dtype)
        ent = self._enc_graphs.get(key)
        if ent is None:                                  # first call of a shape: eager (builds every lazily cached operand)
            flow_input, cond = self.make_flow_input(batch)
            self._enc_graphs[key] = {"state": "warm", "eps_shape": tuple(flow_input.shape)}
            return flow_input, cond
        self.first_stage_model.eval(); self.poke_embedder.eval()
        if self.use_cond:
            self.conditioner.eval()
        enc = self.first_stage_model.enc_motion
        eps = None if enc.be_determinstic else torch.FloatTensor(*ent["eps_shape"]).normal_()      # CPU generator, motion_encoder.py:220
        if ent["state"] == "warm":
            sX, sP = X.clone(), poke.clone()
            sE = None if eps is None else eps.to(X.device)
            torch.cuda.synchronize()
            side = torch.cuda.Stream() if os.environ.get("IPOKE_ENC_GRAPH_SIDE", "1") != "0" else None
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                out = self._flow_input_device(sX, sP, sE, side=side)
            ent.update(state="ready", graph=g, X=sX, poke=sP, eps=sE, out=out, side=side)
        ent["X"].copy_(X, non_blocking=True); ent["poke"].copy_(poke, non_blocking=True)
        if eps is not None:
            ent["eps"].copy_(eps, non_blocking=True)
        ent["graph"].replay()
        # the graph writes into ITS static output buffers: a second replay (a prefetched batch k + 1 while batch k is still queued or in its
        # backward pass) would overwrite them -- hand out copies (two small tensors, ordered behind the replay on this stream)
        return tuple(o.clone() for o in ent["out"])

    def forward_density(self, batch):
        queue = (getattr(self, "_prefetched", None) or {}).get(id(batch))
        pf = queue.pop(0) if queue else None
        if queue is not None and not queue:
            del self._prefetched[id(batch)]
        if pf is not None and pf[0] is batch:
            torch.cuda.current_stream().wait_event(pf[3])
            flow_input, cond = pf[1], pf[2]
        else:
            flow_input, cond = self.make_flow_input(batch)
        out, logdet = self.flow(flow_input.detach(), cond, reverse=False)
        return out, logdet

    # ---- sampling -------------------------------------------------------------------------------------
    def _poke_of(self, batch, use_kp_poke=False):
        if use_kp_poke:
            poke, *_ = batch["keypoint_poke"]
            return poke.to(torch.float)
        poke = batch[self.poke_key]
        return poke[0] if isinstance(poke, list) else poke

    def _sample_device(self, X, poke, z):
        """The device side of one sample of ``forward_sample``: conditioning encoders -> reverse flow -> ConvGRU + SPADE decode.  No host
        synchronisation, no host RNG: this is what ``set_sample_graph`` captures."""
        return self.decode_first_stage(self._sample_motion(X, poke, z), X)

    def _sample_motion(self, X, poke, z):
        """First half of ``_sample_device``: conditioning encoders -> reverse flow (the motion latent of the first stage)."""
        if self.embed_poke_and_image:
            poke = torch.cat([poke, X[:, 0]], dim=1)
        poke_emb, *_ = self.poke_embedder.encoder(poke)
        if self.use_cond:
            cond, *_ = self.conditioner.encoder(X[:, 0])
            if self.adapt_cond_ssize:
                cond = self._adapt_cond(cond)
            cond = torch.cat([cond, poke_emb], dim=1)
        else:
            cond = poke_emb
        out_motion = self.flow(z, cond, reverse=True)
        if self.augment_input:
            out_motion = out_motion[:, :-self.config["architecture"]["augment_channels"]].contiguous()
        return out_motion

    def sample_stream(self, batches, n_logged_vids=1, add_first_frame=False, use_keypoint_pokes=False):
        """``forward_sample(batch)[0]`` for every batch of an iterable (the validation / test loops, `second_stage_video.py:490-584`,
        call it batch after batch), two batches in flight: the reverse flow of batch k+1 runs on the caller's stream while the ConvGRU +
        SPADE decode of batch k runs on a second stream.  The reverse flow is a serial chain of small launches (its inverse units occupy one
        workgroup per sample), the decoder is wide convolutions -- they fill different parts of the chip.  Per batch the same kernels run
        on the same data in the same order as in ``forward_sample`` (bit-identical results, tested); the latent is drawn from the CPU
        generator per batch in the same order.  Yields one CPU tensor per batch, in order, up to two batches late."""
        self.first_stage_model.eval(); self.poke_embedder.eval()
        if self.use_cond:
            self.conditioner.eval()
        if getattr(self, "_decode_stream", None) is None:
            from .utils.streams import overlapping_stream
            self._decode_stream = overlapping_stream()      # not any new stream: see utils/streams.py
        side, spatial = self._decode_stream, self.first_stage_config["architecture"]["min_spatial_size"]

        # Schedule.  The decode stream never WAITS on the device for the flow: a stream parked in an event wait slows the other queue's
        # dispatch (utils/streams.py: a chain of small launches runs 1.9x slower beside a waiting stream, 3-15x on an unlucky one).
        # Instead the host queues the reverse flow of batch k+1 first and only then waits for the flow of batch k to finish -- it has
        # been running while k+1 was queued -- before it queues the decode of k on the second stream.
        @torch.no_grad()                     # per stage, not around the yields: a generator must not hold the caller's grad mode
        def issue_flow(batch):
            X = batch["images"]
            poke = self._poke_of(batch, use_keypoint_pokes)
            # same CPU generator draw as forward_sample, but through pinned memory: a pageable copy would make the host wait for the
            # stream to drain and give up its lead over the device
            z = torch.randn((X.size(0), self.config["architecture"]["flow_in_channels"], spatial, spatial), pin_memory=True)
            z = z.to(device=X.device, dtype=X.dtype, non_blocking=True)
            motion = self._sample_motion(X, poke, z)
            flow_done = torch.cuda.Event(); flow_done.record()
            return X, motion, flow_done

        @torch.no_grad()
        def issue_decode(item):
            X, motion, flow_done = item
            flow_done.synchronize()                      # host-side: nothing waits on the device
            for t_ in (motion, X):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                video = self.decode_first_stage(motion, X)
                if add_first_frame:
                    video = torch.cat([X[:, 0].unsqueeze(1), video], dim=1)
                video = video[:n_logged_vids]
                # the copy to the host is queued right behind this batch's decode
                host = torch.empty(video.shape, dtype=video.dtype, pin_memory=True)
                host.copy_(video, non_blocking=True)
                done = torch.cuda.Event(); done.record(side)
                return host, done

        def fetch(item):
            item[1].synchronize()
            return item[0].clone()            # out of the pinned staging buffer, like ``.cpu()``

        flowing, decoding = None, None       # batch whose reverse flow is queued / batch whose decode is queued
        for batch in batches:
            nxt = issue_flow(batch)
            if flowing is not None:
                out = issue_decode(flowing)
                if decoding is not None:
                    yield fetch(decoding)
                decoding = out
            flowing = nxt
        if flowing is not None:
            out = issue_decode(flowing)
            if decoding is not None:
                yield fetch(decoding)
            decoding = out
        if decoding is not None:
            yield fetch(decoding)

    def set_sample_graph(self, enable=True):
        """BASELINE configs[4] ("flow inverse + VAE decode, hipGraph-captured"): replay the whole device side of ``forward_sample`` --
        the two conditioning encoders, the ~3 000-launch reverse flow, the 15-step ConvGRU and the frame-batched SPADE decoder
        (models/second_stage_video.py:326-382) -- as ONE captured hipGraph per input shape.  The latent is still drawn on the CPU
        generator per call, like the reference (:296-300), and copied into the graph's input buffer; results are bit-identical to the
        eager path (tests/test_bench_configs_gpu.py)."""
        self._graph_sampling = bool(enable)
        self._sample_graphs = {}

    def _sample_graphed(self, X, poke, z):
        key = (tuple(X.shape), tuple(poke.shape), X.dtype, poke.dtype)
        ent = self._sample_graphs.get(key)
        if ent is None:                                  # first call of a shape: eager (builds every lazily cached operand)
            self._sample_graphs[key] = {"state": "warm"}
            return self._sample_device(X, poke, z)
        if ent["state"] == "warm":
            sX, sP, sZ = X.clone(), poke.clone(), z.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._sample_device(sX, sP, sZ)
            ent.update(state="ready", graph=g, X=sX, poke=sP, z=sZ, out=out)
        if self.flow.engine.shadow_stale:            # a replay skips the eager path's refresh of the flow's weight operands (same addresses)
            self.flow.engine.prepare_weights()
        ent["X"].copy_(X); ent["poke"].copy_(poke); ent["z"].copy_(z)
        ent["graph"].replay()
        return ent["out"]

    def forward_sample(self, batch, n_samples=1, n_logged_vids=1, show_progress=False, add_first_frame=False,
                       use_keypoint_pokes=False):
        video_samples = []
        self.first_stage_model.eval(); self.poke_embedder.eval()
        if self.use_cond:
            self.conditioner.eval()
        with torch.no_grad():
            X = batch["images"]
            poke = self._poke_of(batch, use_keypoint_pokes)
            spatial = self.first_stage_config["architecture"]["min_spatial_size"]
            for _ in range(n_samples):
                # CPU generator, then moved: torch.randn(...).type_as(X) (:296-300); the conditioning is recomputed per sample as
                # the reference's make_flow_input(batch, reverse=True) does (:333)
                z = torch.randn((X.size(0), self.config["architecture"]["flow_in_channels"], spatial, spatial)).type_as(X).detach()
                if getattr(self, "_graph_sampling", False):
                    out_video = self._sample_graphed(X, poke, z)
                else:
                    out_video = self._sample_device(X, poke, z)
                if add_first_frame:
                    out_video = torch.cat([X[:, 0].unsqueeze(1), out_video], dim=1)
                video_samples.append(out_video[:n_logged_vids].cpu())
        return video_samples

    def training_step(self, batch, batch_idx):
        out, logdet = self.forward_density(batch)
        loss, loss_dict = self.loss_func(out, logdet)
        self.log_dict(loss_dict)
        self.log("global_step", self.global_step)
        if self._optimizer is not None:
            self.log("learning_rate", self._optimizer.param_groups[0]["lr"])
        return loss

    # ---- validation loop (second_stage_video.py:490-584) --------------------------------------------------
    def attach_fvd(self, i3d=None, dtype="f32"):
        """The reference builds ``self.FVD = FVD(n_samples=...)`` with the Kinetics I3D weights in __init__ (:66); here the metric
        is attached explicitly (the I3D checkpoint is not part of the second-stage checkpoint)."""
        from .fvd import FVD
        self.FVD = FVD(n_samples=self.config["logging"]["n_fvd_samples"], i3d=i3d, dtype=dtype)
        self._fvd_fake, self._fvd_true, self._fvd_fake_x0, self._fvd_true_x0 = [], [], [], []
        return self.FVD

    def validation_step(self, batch, batch_id):
        """NLL terms of the batch, and (for the first n_fvd_samples clips) one sampled video per clip kept for the FVD of the
        epoch; generated and true clips stay on the device.  ssim-val / psnr-val (:511-512) are computed on the device
        (ipoke_amd/metrics.py); lpips-val (:513) needs the lpips package's pretrained network and is not part of this path."""
        with torch.no_grad():
            out, logdet = self.forward_density(batch)
            loss, loss_dict = self.loss_func(out, logdet)
        self.log_dict({"val/" + key: loss_dict[key] for key in loss_dict})
        X = batch["images"]
        # second_stage_video.py:498-513: the sampled clips and ssim / psnr of every batch below the FVD sample budget, whether or not an
        # I3D is attached; only the clip lists kept for the epoch's FVD need the FVD object
        if batch_id <= int(self.config["logging"]["n_fvd_samples"] / X.size(0)):
            X_hat = self.forward_sample(batch, n_logged_vids=X.size(0))[0].to(X.device)
            if getattr(self, "FVD", None) is not None:
                self._fvd_fake.append(X_hat)
                self._fvd_true.append(X[:, 1:])
                self._fvd_fake_x0.append(torch.cat([X[:, 0].unsqueeze(1), X_hat], dim=1))
                self._fvd_true_x0.append(X)
            from . import metrics
            X_hat_log = X_hat.reshape(-1, *X_hat.shape[2:]).type_as(X)
            X_log = X[:, 1:].reshape(-1, *X_hat.shape[2:])
            both = metrics.psnr_ssim(X_hat_log, X_log)
            self.log("ssim-val", both[1])
            self.log("psnr-val", both[0])
        self.log("d_ref_nll-val", torch.abs(loss_dict["reference_nll_loss"] - loss_dict["nll_loss"]))
        self.log("loss-val", loss)
        return {"loss": loss, "batch_idx": batch_id, "loss_dict": loss_dict}

    def validation_epoch_end(self, outputs=None):
        from .fvd import calculate_FVD
        bs = self.first_stage_config["logging"]["bs_i3d"]
        fvd_score = calculate_FVD(self.FVD.i3d, torch.cat(self._fvd_fake), torch.cat(self._fvd_true), batch_size=bs)
        self.log("FVD-val", fvd_score)
        fvd_x0 = calculate_FVD(self.FVD.i3d, torch.cat(self._fvd_fake_x0), torch.cat(self._fvd_true_x0), batch_size=bs)
        self.log("FVD-val-x0", fvd_x0)
        for lst in (self._fvd_fake, self._fvd_true, self._fvd_fake_x0, self._fvd_true_x0):
            lst.clear()
        return fvd_score, fvd_x0

    def configure_optimizers(self):
        tr = self.config["training"]
        self._optimizer = FusedAdamAmsgrad(self.flow, lr=tr["lr"], betas=(0.9, 0.999), weight_decay=tr["weight_decay"], amsgrad=True)
        if not self.custom_lr_decrease:
            raise NotImplementedError("the shipped configs use custom_lr_decrease=True (no scheduler object)")
        return [self._optimizer]
