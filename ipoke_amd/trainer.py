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

    def __init__(self, model, n_grad_buckets=12, overlap=None):
        self.model = model
        self.opt = model.configure_optimizers()[0]
        self.world = D.world_size()
        self.n_grad_buckets = n_grad_buckets = int(os.environ.get("IPOKE_PIECES", n_grad_buckets))
        if overlap is None:
            overlap = os.environ.get("IPOKE_NO_OVERLAP", "0") != "1"
        # Also on one GPU (no exchange): the per-group Adam updates run underneath the backward chain with a one-workgroup-
        # per-CU grid (81.3 vs 82.3 ms; with the stand-alone 4096-workgroup grid they starve the chain: 85.2 ms).
        self.overlap = bool(overlap) and torch.cuda.is_available()
        if self.overlap:
            self.ready_stream = torch.cuda.Stream()
            model.flow.engine.grad_ready_hook = (n_grad_buckets, self.ready_stream, self._grads_ready)
        # train_step(batch, next_batch=...): the frozen encoders (first stage, poke, image) of the NEXT batch do not depend on the
        # flow's parameters; they are issued on their own stream right after this step's forward and run underneath its
        # latency-bound backward chain (86.2 -> 83.2 ms at c2).  Every step still runs one encoder pass.
        self.prefetch_stream = None
        if os.environ.get("IPOKE_NO_PREFETCH", "0") != "1" and torch.cuda.is_available():
            self.prefetch_stream = torch.cuda.Stream()
        model.flow.train()

    def _optimizer_step(self, fn):
        # (Issuing the update on its own stream so that the next step's frozen encoders run underneath it was measured:
        # 88.0 vs 87.1 ms -- both sides stream HBM, and Adam's persistent grid starves the concurrent kernels.  Not done.)
        return fn()

    def _grads_ready(self, begin, end):
        """grads[begin:end] is final at the current point of ``ready_stream``: all-reduce it there (data parallel) and
        apply the optimizer update to that slice, all without blocking the backward chain."""
        flat = self.model.flow.flat_grads
        with torch.cuda.stream(self.ready_stream):
            if self.world > 1:
                D.allreduce_async(flat[begin:end]).wait()        # orders ready_stream after the collective, host does not block
            self.opt.step_range(begin, end, grad_scale=1.0 / self.world)

    def sync_initial_state(self, batch):
        """Data-dependent ActNorm init happens on the first forward (macow2.py:503-505).  Under DDP the reference lets
        every rank initialise from its own micro-batch and never re-syncs; here every rank adopts rank 0's initialised
        parameters (deliberate, documented deviation: SURVEY.md §8e)."""
        with torch.no_grad():
            self.model.forward_density(batch)
        D.broadcast_(self.model.flow.flat_params, src=0)
        D.broadcast_(self.model.flow.engine.perm, src=0)
        self.model.flow.mark_weights_updated()

    def train_step(self, batch, batch_idx=0, next_batch=None):
        m = self.model
        m.on_train_batch_start(batch, batch_idx, 0)
        loss = m.training_step(batch, batch_idx)
        if next_batch is not None and self.prefetch_stream is not None:
            m.prefetch_flow_input(next_batch, self.prefetch_stream)
        if self.overlap:
            self.opt.begin_step()
            loss.backward()                   # exchanges and updates every slice from the engine's callbacks; on return the
            self._optimizer_step(self.opt.finish_step)     # current stream is ordered after the ready stream
        else:
            loss.backward()
            if self.world > 1:
                D.allreduce_flat_(m.flow.flat_grads, self.n_grad_buckets)
            self._optimizer_step(lambda: self.opt.step(grad_scale=1.0 / self.world))
        m.global_step += 1
        return loss
