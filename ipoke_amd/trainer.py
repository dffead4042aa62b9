"""Training loop counterpart of ``experiments/second_stage_video.py:46-65`` (pl.Trainer.fit) for one process per GPU:
LR rule -> training_step -> backward -> gradient all-reduce (RCCL) -> fused Adam-amsgrad."""
import torch

from . import dist as D


class SecondStageTrainer:
    def __init__(self, model, n_grad_buckets=8):
        self.model = model
        self.opt = model.configure_optimizers()[0]
        self.world = D.world_size()
        self.n_grad_buckets = n_grad_buckets
        model.flow.train()

    def sync_initial_state(self, batch):
        """Data-dependent ActNorm init happens on the first forward (macow2.py:503-505).  Under DDP the reference lets
        every rank initialise from its own micro-batch and never re-syncs; here every rank adopts rank 0's initialised
        parameters (deliberate, documented deviation: SURVEY.md §8e)."""
        with torch.no_grad():
            self.model.forward_density(batch)
        D.broadcast_(self.model.flow.flat_params, src=0)
        D.broadcast_(self.model.flow.engine.perm, src=0)
        self.model.flow.mark_weights_updated()

    def train_step(self, batch, batch_idx=0):
        m = self.model
        m.on_train_batch_start(batch, batch_idx, 0)
        loss = m.training_step(batch, batch_idx)
        loss.backward()
        if self.world > 1:
            D.allreduce_flat_(m.flow.flat_grads, self.n_grad_buckets)
        self.opt.step(grad_scale=1.0 / self.world)
        m.global_step += 1
        return loss
