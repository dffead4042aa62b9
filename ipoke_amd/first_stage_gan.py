"""First-stage adversarial training step on the HIP kernels: the manual-optimisation step of reference
``models/first_stage_motion_model.py:160-277`` --

    1. X_hat, mu, logvar = model(X)
    2. temporal discriminator (d_t): clips X_true / X_fake = [X_0, X_hat] cut at a random offset to ``mf_dt`` frames;
       hinge loss on disc_t(X_fake.detach()), disc_t(X_true) plus gp_weight * gp2(X_true); Adam step            (:171-201)
    3. spatial discriminator (d_s): ``n_examples`` random true / reconstructed frames; hinge loss; Adam step      (:203-224)
    4. generator: -mean disc_s(x_fake) * 1, gen_weight * (-mean disc_t(X_fake)) + fmap_weight * fmap_loss,
       w_l1 * L1 + w_kl * KL (+ w_vgg * VGG);  Adam step                                                          (:226-277)

with three ``Adam(lr, betas=(0.5, 0.9), weight_decay)`` optimisers (:376-386).  Everything runs through the differentiable
HIP ops of ``first_stage_train`` / ``discriminator`` and ``ipoke_adam_multi``; the gradient penalty is evaluated
forward-over-reverse (``TemporalDiscriminator.gp2``).

The VGG perceptual term (``w_vgg``, utils/losses.py:67-82) is ``ipoke_amd.vgg.VGGLoss``, passed in by the caller with the
torchvision VGG-19 weights loaded (they are not available offline; bench and tests use random-init weights of that
architecture).  One deliberate deviation: the random choices of the reference's step (clip offset, frame examples; numpy's
global RNG) are arguments, so that the caller owns the random stream.
"""
import numpy as np
import torch

from . import first_stage_train as T


class FirstStageGANTrainer:
    def __init__(self, model, disc_t, disc_s, cfg, vgg_loss=None):
        """``cfg``: the reference's config sections -- cfg["training"] (lr, weight_decay, w_l1, w_kl, w_vgg), cfg["d_t"]
        (gp_weight, fmap_weight, gen_weight, max_frames), cfg["d_s"] (n_examples, gen_weight, fmap_weight),
        cfg["data"]["max_frames"]."""
        tr = cfg["training"]
        self.w_vgg, self.vgg_loss = float(tr.get("w_vgg", 0.0)), vgg_loss
        if self.w_vgg != 0.0 and vgg_loss is None:
            raise ValueError("w_vgg != 0 needs vgg_loss=ipoke_amd.vgg.VGGLoss(...) (load the torchvision VGG-19 weights into it)")
        self.model, self.disc_t, self.disc_s, self.cfg = model, disc_t, disc_s, cfg
        self.w_l1, self.w_kl = float(tr["w_l1"]), float(tr["w_kl"])
        self.mf_dt = min(int(cfg["d_t"]["max_frames"]), int(cfg["data"]["max_frames"]))
        kw = dict(lr=tr["lr"], betas=(0.5, 0.9), weight_decay=tr["weight_decay"])
        self.opt_g = T.MultiTensorAdam(model.parameters(), **kw)
        self.opt_dt = T.MultiTensorAdam(disc_t.parameters(), **kw) if disc_t is not None else None
        self.opt_ds = T.MultiTensorAdam(disc_s.parameters(), **kw) if disc_s is not None else None
        self.gamma = float(tr.get("gamma", 0.98))                    # first_stage.yaml:45

    def optimizers(self):
        return [o for o in (self.opt_g, self.opt_ds, self.opt_dt) if o is not None]

    def on_epoch_end(self):
        """The three ExponentialLR(gamma) schedulers of first_stage_motion_model.py:383-388, stepped once per epoch."""
        for o in self.optimizers():
            o.exponential_lr_step(self.gamma)

    def state_dict(self):
        return {"opt_g": self.opt_g.state_dict(), "opt_ds": None if self.opt_ds is None else self.opt_ds.state_dict(),
                "opt_dt": None if self.opt_dt is None else self.opt_dt.state_dict()}

    def load_state_dict(self, sd):
        self.opt_g.load_state_dict(sd["opt_g"])
        if self.opt_ds is not None and sd.get("opt_ds") is not None:
            self.opt_ds.load_state_dict(sd["opt_ds"])
        if self.opt_dt is not None and sd.get("opt_dt") is not None:
            self.opt_dt.load_state_dict(sd["opt_dt"])

    def draw(self, X, rng=np.random):
        """The random choices of one step, as the reference draws them (:175, :204-205)."""
        B, Tn = X.shape[0], X.shape[1]
        offset = int(rng.choice(np.arange(max(1, Tn - self.mf_dt)), 1))
        n_ex = int(self.cfg["d_s"]["n_examples"])
        return offset, rng.choice(np.arange(B * Tn), n_ex), rng.choice(np.arange(B * (Tn - 1)), n_ex)

    def step(self, X, eps, offset, true_ids, fake_ids, power_iteration=True):
        """One training step on X [B, T, 3, H, W] (fp32, GPU).  Returns a dict of detached loss values (the reference's log).
        ``power_iteration=False`` freezes the spectral-norm buffers (eval-mode semantics; used by the parity test)."""
        m, dt_, ds_ = self.model, self.disc_t, self.disc_s
        cdt = self.cfg["d_t"]
        pit = bool(power_iteration)
        log = {}
        loss_rec, X_hat, mu, lv = T.first_stage_forward_loss(m, X, eps, w_l1=self.w_l1, w_kl=self.w_kl, power_iteration=pit)
        X = X.float()
        true_ids = torch.as_tensor(np.asarray(true_ids), device=X.device, dtype=torch.long)
        fake_ids = torch.as_tensor(np.asarray(fake_ids), device=X.device, dtype=torch.long)
        if dt_ is not None:
            X_fake = torch.cat([X[:, 0].unsqueeze(1), X_hat], dim=1)[:, offset:offset + self.mf_dt].permute(0, 2, 1, 3, 4)
            X_true = X[:, offset:offset + self.mf_dt].permute(0, 2, 1, 3, 4).contiguous()
            pf, _ = dt_(X_fake.detach().contiguous(), pit)
            pt, _ = dt_(X_true, pit)
            loss_dt = (dt_.loss(pf, real=False) + dt_.loss(pt, real=True)) / 2.0
            gp = dt_.gp2(X_true) if dt_.gp_weight > 0 else None
            self.opt_dt.zero_grad()
            (loss_dt if gp is None else loss_dt + dt_.gp_weight * gp).backward()
            self.opt_dt.step()
            log.update(loss_d_dt=loss_dt.detach(), loss_gp_dt=torch.zeros(()) if gp is None else gp.detach())
        if ds_ is not None:
            x_true = X.reshape(-1, *X.shape[2:])[true_ids].contiguous()
            x_fake = X_hat.reshape(-1, *X_hat.shape[2:])[fake_ids]
            pf, _ = ds_(x_fake.detach().contiguous(), pit)
            pt, _ = ds_(x_true, pit)
            loss_ds = (ds_.loss(pf, real=False) + ds_.loss(pt, real=True)) / 2.0
            self.opt_ds.zero_grad()
            loss_ds.backward()
            self.opt_ds.step()
            log.update(loss_d_ds=loss_ds.detach())
        self.opt_g.zero_grad()
        # The generator step differentiates the discriminators w.r.t. their INPUT only: their parameter gradients would be thrown
        # away (the next discriminator step starts with zero_grad), and the true clip's feature maps carry no gradient at all.  The
        # reference lets autograd compute both; here the discriminator parameters are frozen for the duration (no weight-gradient
        # GEMMs, no spectral-norm backward) and the true clip runs without a graph.  Losses and generator gradients are unchanged.
        frozen = [p for d in (dt_, ds_) if d is not None for p in d.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        try:
            total = self._generator_loss(loss_rec, X, X_hat, X_fake if dt_ is not None else None, X_true if dt_ is not None else None,
                                         x_fake if ds_ is not None else None, pit, log)
        finally:
            for p in frozen:
                p.requires_grad_(True)
        total.backward()                                             # one backward = the reference's three accumulating ones
        self.opt_g.step()
        m.invalidate_operands()
        log.update(loss=loss_rec.detach(), X_hat=X_hat.detach())
        return log

    def _generator_loss(self, loss_rec, X, X_hat, X_fake, X_true, x_fake, pit, log):
        dt_, ds_, cdt = self.disc_t, self.disc_s, self.cfg["d_t"]
        total = loss_rec
        if self.w_vgg != 0.0:                                        # (:263, :271)
            loss_vgg = self.vgg_loss(X[:, 1:].reshape(-1, *X.shape[2:]), X_hat.reshape(-1, *X_hat.shape[2:]))
            total = total + self.w_vgg * loss_vgg
            log.update(loss_vgg=loss_vgg.detach())
        if ds_ is not None:
            pg, _ = ds_(x_fake, pit)
            loss_gen_ds = -pg.mean()
            total = total + loss_gen_ds                              # (:234 backs it with weight 1)
            log.update(loss_g_s=loss_gen_ds.detach())
        if dt_ is not None:
            pg, fmap_fake = dt_(X_fake, pit)
            with torch.no_grad():
                _, fmap_true = dt_(X_true, pit)
            loss_gen_dt = -pg.mean()
            loss_fmap = dt_.fmap_loss(fmap_fake, fmap_true)
            total = total + float(cdt["gen_weight"]) * loss_gen_dt + float(cdt["fmap_weight"]) * loss_fmap
            log.update(loss_g_t=loss_gen_dt.detach(), loss_fmap_t=loss_fmap.detach())
        return total
