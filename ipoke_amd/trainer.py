"""Training loop counterpart of ``experiments/second_stage_video.py:46-65`` (pl.Trainer.fit) for one process per GPU:
LR rule -> training_step -> backward -> gradient all-reduce (RCCL) -> fused Adam-amsgrad."""
import os

import torch

from . import dist as D


class SecondStageTrainer:
    """``overlap``: issue the backward in ``n_grad_buckets`` groups of levels; as soon as a group's slice of the flat gradient
    buffer is final it is all-reduced (data parallel; the role of DDP's bucket hooks in the reference's Lightning run) and
    the fused Adam-amsgrad update of that slice is applied, on a separate stream, while the remaining levels are still
    differentiating.  Without overlap the flat buffer is all-reduced in slices and updated after the backward pass."""

    def __init__(self, model, n_grad_buckets=24, overlap=None):
        self.model = model
        from . import warn_if_queues_late
        warn_if_queues_late("SecondStageTrainer")
        tr, bs = model.config["training"], model.config["data"]["batch_size"]
        # experiments/experiment.py:81-88: accumulate_grad_batches = ceil(min_acc_batch_size / batch_size) when that is larger than the
        # batch (1 for every shipped config: 3 < 20).  Lightning 1.1.7 divides each micro-batch loss by the count and steps the
        # optimizer on every k-th batch; global_step counts optimizer steps.  The engine WRITES the flat gradient buffer on every
        # backward pass, so the first k - 1 micro-batches are summed into a second flat buffer and the k-th pass adds it slice by slice
        # right before the slice's exchange / update; the 1 / k is folded into the fused update's grad_scale.
        mab = int(tr.get("min_acc_batch_size", 0) or 0)
        self.accumulate_grad_batches = -(-mab // bs) if mab > bs else 1
        self._acc, self._acc_count = None, 0
        self.opt = model.configure_optimizers()[0]
        self.world = D.world_size()
        self.n_grad_buckets = n_grad_buckets = int(os.environ.get("IPOKE_PIECES", n_grad_buckets))
        if overlap is None:
            overlap = os.environ.get("IPOKE_NO_OVERLAP", "0") != "1"
        # Also on one GPU (no exchange): the per-group Adam updates run underneath the backward chain with a one-workgroup-
        # per-CU grid (81.3 vs 82.3 ms; with the stand-alone 4096-workgroup grid they starve the chain: 85.2 ms).
        self.overlap = bool(overlap) and torch.cuda.is_available()
        # data parallel: reduce-scatter + sharded update + all-gather per slice (ZeRO-1 over the flat buffer) instead of
        # all-reduce + replicated update; IPOKE_GRAD_EXCHANGE=bf16 halves the bytes of the reduce-scatter
        self.zero1 = self.world > 1 and self.overlap and os.environ.get("IPOKE_NO_ZERO1", "0") != "1"
        if self.zero1:
            gd = torch.bfloat16 if os.environ.get("IPOKE_GRAD_EXCHANGE", "f32") == "bf16" else torch.float32
            self.opt.enable_sharding(self.world, D.rank(), gd)
        # one process, no accumulation: the engine queues every piece's update itself (no host callback inside the backward pass)
        self.native_opt = self.overlap and self.world == 1 and os.environ.get("IPOKE_NO_NATIVE_ADAM", "0") != "1"
        want_prefetch = os.environ.get("IPOKE_NO_PREFETCH", "0") != "1" and torch.cuda.is_available()
        own_prefetch = want_prefetch and os.environ.get("IPOKE_PREFETCH_STREAM", "own") != "chain"
        picked = []
        if self.overlap:
            # the streams this trainer keeps busy beside the chain and the engine's weight-gradient stream must each sit on a hardware
            # queue of its own (utils/streams.py: a shared queue serialises them -- 2x on the step when it is the chain's)
            from .utils.streams import distinct_streams
            side = model.flow.engine.side_stream()
            picked = distinct_streams(1 + int(own_prefetch), against=[torch.cuda.current_stream()] + ([side] if side is not None else []))
            self.ready_stream = picked[0]                # the hook itself is installed around the backward of train_step only
        # train_step(batch, next_batch=...): the frozen encoders (first stage, poke, image) of the NEXT batch do not depend on the
        # flow's parameters; they are issued on their own stream right after this step's forward and run underneath its
        # latency-bound backward chain (86.2 -> 83.2 ms at c2).  Every step still runs one encoder pass.
        self.prefetch_stream = None
        if os.environ.get("IPOKE_NO_PREFETCH", "0") != "1" and torch.cuda.is_available():
            # Stream budget: the step is tuned for FOUR busy streams -- chain, weight gradients, ready / optimizer, encoder
            # prefetch.  A fifth busy stream costs +18 ms per step whatever it carries (measured with a second weight-gradient and a
            # second optimizer stream, any GPU_MAX_HW_QUEUES).  At world > 1 the collectives are issued as synchronous ops, which RCCL
            # runs ON the ready stream (ipoke_amd/dist.py): the budget stays at four for every world size.  IPOKE_PREFETCH_STREAM=chain
            # puts the next batch's encoders behind the backward chain on the caller's stream instead (three busy streams; +1.5 ms on
            # one GPU, 52.3 vs 50.8) -- for PyTorch builds whose collectives still run on an internal stream.
            self.prefetch_stream = (picked[1] if len(picked) > 1 else torch.cuda.Stream()) if own_prefetch else torch.cuda.current_stream()
            # IPOKE_ENC_GRAPH=1 (developer A/B, measured slower: PokeMotionModel.set_encoder_graph): the prefetched encoders replayed from
            # one captured hipGraph, gated on the GPU side at the END of the backward pass, instead of ~200 eager launches
            if os.environ.get("IPOKE_ENC_GRAPH", "0") == "1":
                model.set_encoder_graph(True)
        self.enc_graph_at = os.environ.get("IPOKE_ENC_GRAPH_AT", "bwd")        # developer A/B: "fwd" = may start right behind the forward pass
        self.prefetch_at_start = os.environ.get("IPOKE_PREFETCH_AT", "after_bwd") == "start"
        # IPOKE_PREFETCH_AT=piece<k> (developer A/B, measured round 3, NOT adopted): the next batch's encoders are queued when the
        # backward pass has issued its k-th piece (of IPOKE_PIECES), ordered after that point of the chain, so that they overlap the rest
        # of the backward pass instead of following it: 60.7 (k = 11) / 61.4 (k = 6 .. 10) against 59.1 ms -- beside the backward chain and
        # its side streams the full-chip encoder kernels cost more than the 4 ms hole they fill (the same verdict as =start)
        at = os.environ.get("IPOKE_PREFETCH_AT", "")
        self.prefetch_piece = int(at[5:]) if at.startswith("piece") else None
        # IPOKE_PREFETCH_THREAD=1: the next batch's encoders are queued by a second host thread while this thread is inside the engine's
        # backward call (a foreign call: the interpreter lock is free), ordered behind the forward pass on the GPU
        self.prefetch_thread = os.environ.get("IPOKE_PREFETCH_THREAD", "0") == "1"
        model.flow.train()

    def _prefetch_in_thread(self, next_batch, after):
        """Start ``prefetch_flow_input(next_batch)`` on a second host thread; returns the function that joins it (and re-raises)."""
        import threading
        dev, box = torch.cuda.current_device(), {}

        def run():
            try:
                torch.cuda.set_device(dev)              # the current device is per thread
                self.model.prefetch_flow_input(next_batch, self.prefetch_stream, after=after)
            except BaseException as e:                  # noqa: BLE001 -- handed to the joining thread
                box["err"] = e
        th = threading.Thread(target=run, name="ipoke-prefetch")
        th.start()

        def join():
            th.join()
            if "err" in box:
                raise box["err"]
        return join

    def _optimizer_step(self, fn):
        # (Issuing the update on its own stream so that the next step's frozen encoders run underneath it was measured:
        # 88.0 vs 87.1 ms -- both sides stream HBM, and Adam's persistent grid starves the concurrent kernels.  Not done.)
        return fn()

    def _grads_ready(self, begin, end):
        """grads[begin:end] is final at the current point of ``ready_stream``: all-reduce it there (data parallel) and
        apply the optimizer update to that slice, all without blocking the backward chain."""
        flat = self.model.flow.flat_grads
        scale = 1.0 / (self.world * self.accumulate_grad_batches)
        with torch.cuda.stream(self.ready_stream):
            if self._acc_count:                                  # gradients of the earlier micro-batches of this optimizer step
                flat[begin:end].add_(self._acc[begin:end])
            if self.zero1:
                self.opt.step_range_sharded(begin, end, grad_scale=scale)
                return
            if self.world > 1:
                D.allreduce_async(flat[begin:end]).wait()        # orders ready_stream after the collective, host does not block
            self.opt.step_range(begin, end, grad_scale=scale)

    def sync_initial_state(self, batch):
        """Data-dependent ActNorm init happens on the first forward (macow2.py:503-505).  Under DDP the reference lets
        every rank initialise from its own micro-batch and never re-syncs; here every rank adopts rank 0's initialised
        parameters (deliberate, documented deviation: SURVEY.md §8e)."""
        with torch.no_grad():
            self.model.forward_density(batch)
        D.broadcast_(self.model.flow.flat_params, src=0)
        D.broadcast_(self.model.flow.engine.perm, src=0)
        # the named int64 idx buffers of the state dict follow the engine's (now rank 0's) permutation: a later
        # sync_buffers() / state_dict() on any rank sees the same shuffle as the weights
        self.model.flow.adopt_engine_perm()
        self.model.flow.mark_weights_updated()

    def on_train_epoch_start(self, num_training_batches):
        """Call at every epoch start with this rank's number of batches (min(len(loader), max_batches_per_epoch): what
        Lightning's ``trainer.num_training_batches`` holds): the reference's linear LR decay ends at
        n_epochs * num_training_batches (second_stage_video.py:317-323).  Without it the horizon stays at
        n_epochs * max_batches_per_epoch, the value for loaders at least that long."""
        self.model.on_train_epoch_start(num_training_batches)

    def fit(self, batches_per_epoch, n_epochs=None, get_batch=None):
        """Minimal counterpart of pl.Trainer.fit: ``get_batch(epoch, i)`` supplies batches already resident on the GPU."""
        n_epochs = self.model.config["training"]["n_epochs"] if n_epochs is None else n_epochs
        n = min(int(batches_per_epoch), int(self.model.config["training"].get("max_batches_per_epoch", batches_per_epoch)))
        loss = None
        for epoch in range(n_epochs):
            self.model.current_epoch = epoch
            self.on_train_epoch_start(n)
            nxt = get_batch(epoch, 0)
            for i in range(n):
                batch, nxt = nxt, (get_batch(epoch, i + 1) if i + 1 < n else None)
                loss = self.train_step(batch, i, next_batch=nxt)
            # once per epoch (it synchronises): no in-launch hand-off of the flow's kernels gave up (the engine also checks at every
            # entry point as soon as the previous pass's poll has landed)
            self.model.flow.engine.assert_handoffs_clean()
        return loss

    def train_step(self, batch, batch_idx=0, next_batch=None):
        m = self.model
        m.on_train_batch_start(batch, batch_idx, 0)
        prefetch = next_batch is not None and self.prefetch_stream is not None
        if prefetch and self.prefetch_at_start:
            # developer switch IPOKE_PREFETCH_AT=start (measured, NOT adopted: 69.7 vs 63.7 ms): the next batch's frozen encoders
            # queued BEFORE this step's forward, to run underneath it (no weight gradients / optimizer there) instead of behind
            # the backward pass.  The forward chain is latency-bound; full-chip encoder kernels beside it slow it by more
            # than the hole they leave (scripts/probe_host.py: the host needs 29 ms per step and is throttled by the queue depth,
            # so the ~7 ms of queueing are not what costs).
            step_start = torch.cuda.Event()
            step_start.record()
            m.prefetch_flow_input(next_batch, self.prefetch_stream, after=step_start)
            prefetch = False
        loss = m.training_step(batch, batch_idx)
        if prefetch:
            fwd_done = torch.cuda.Event()
            fwd_done.record()                 # the encoders of the next batch may start behind the forward pass ...
        k = self.accumulate_grad_batches
        if k > 1 and self._acc_count < k - 1:
            # a micro-batch that does not end the optimizer step: plain backward, gradients summed aside
            loss.backward()
            if prefetch:
                m.prefetch_flow_input(next_batch, self.prefetch_stream, after=fwd_done)
            flat = m.flow.flat_grads
            if self._acc is None:
                self._acc = torch.empty_like(flat)
            if self._acc_count == 0:
                self._acc.copy_(flat)
            else:
                self._acc.add_(flat)
            self._acc_count += 1
            return loss
        if self.overlap:
            self.opt.begin_step()
            eng = m.flow.engine
            native = self.native_opt and self._acc_count == 0 and not eng.shadow_stale
            if native:
                self.opt.arm_native(grad_scale=1.0 / (self.world * self.accumulate_grad_batches))
            hook_fn = None if native else self._grads_ready
            if native and prefetch and self.prefetch_piece is not None:
                state = {"done": False}

                def hook_fn(begin, end, piece, _state=state):
                    if not _state["done"] and piece >= self.prefetch_piece:
                        _state["done"] = True
                        here = torch.cuda.Event()
                        here.record()
                        m.prefetch_flow_input(next_batch, self.prefetch_stream, after=here)
                hook_fn.wants_piece = True
            eng.grad_ready_hook = (self.n_grad_buckets, self.ready_stream, hook_fn)
            worker = None
            if prefetch and self.prefetch_thread and self.prefetch_piece is None:
                worker = self._prefetch_in_thread(next_batch, fwd_done)
                prefetch = False
            ok = False
            try:
                loss.backward()               # exchanges and updates every slice from the engine's callbacks; on return the
                ok = True                     # current stream is ordered after the ready stream
            finally:
                eng.grad_ready_hook = None    # a backward outside train_step must not apply optimizer updates
                if native and not ok:
                    self.opt.disarm_native()
                if worker is not None:
                    worker()
            if prefetch and not (native and self.prefetch_piece is not None and state["done"]):
                after = fwd_done
                if getattr(m, "_graph_encoders", False) and self.enc_graph_at == "bwd":
                    # a graph replay is ONE host call: the host, several ms ahead of the GPU here, would start it in the middle of the
                    # backward pass; the event holds it until the chain (and the ready stream's last slice) is done
                    after = torch.cuda.Event()
                    after.record()
                m.prefetch_flow_input(next_batch, self.prefetch_stream, after=after)      # queued after the backward pass (host order = GPU start order)
            self._optimizer_step(self.opt.finish_native if native else self.opt.finish_step)
        else:
            loss.backward()
            if prefetch:
                m.prefetch_flow_input(next_batch, self.prefetch_stream, after=fwd_done)
            if self._acc_count:
                m.flow.flat_grads.add_(self._acc)
            if self.world > 1:
                D.allreduce_flat_(m.flow.flat_grads, self.n_grad_buckets)
            self._optimizer_step(lambda: self.opt.step(grad_scale=1.0 / (self.world * k)))
        self._acc_count = 0
        m.global_step += 1
        return loss
