"""Lightning-style wrapper of the one-class STSE training (mirror of the reference's
models/euclidean_encoder_staticCenter.py, models/euclidean_encoder_dynamicCenter.py and
models/hyperbolic_encoder.py `LitEncoder`, which cannot be imported from the snapshot -- SURVEY fact 1 --
and a minimal `Trainer` replacing pytorch_lightning, which is not installed).

Same hooks and config keys (flat yaml -> argparse.Namespace): `forward(x: list[4])`, `setup(stage)`,
`training_step`, `on_train_epoch_end`, `validation_step`, `validation_epoch_end`, `configure_optimizers`,
`post_processing`.  Differences, all deliberate:
  * the optimisation step is the fused HIP sequence of coskad_amd.trainer.STSETrainStep (forward, loss,
    backward, L2 reg, Adam in one call), so `training_step` has already stepped when it returns the loss;
  * centre statistics are device-side running sums (and all-reduced over ranks) instead of python lists /
    torch.cat of every latent (hyperbolic_encoder.py:148-153);
  * post_processing is the vectorised scoring of coskad_amd.utils.eval_utils.
Parity of these wrappers against the reference is unpinned (no importable reference, no recorded numbers).
"""
from __future__ import annotations

import os
from argparse import Namespace
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, parallel
from .models.sts.ae import STSE
from .trainer import STSETrainStep, make_train_step
from .utils import eval_utils


def _joints(args) -> int:
    if getattr(args, "dataset_headless", False):
        return 14
    if getattr(args, "dataset_kp18_format", False):
        return 18
    return 17


class LitEncoder(nn.Module):
    _reg_last = None      # the regulariser's value at the last logging step (training_step reuses it in between)

    def __init__(self, args: Namespace, hyperbolic: Optional[bool] = None) -> None:
        super().__init__()
        self.args = args
        self.hparams = Namespace(args=args)                  # save_hyperparameters()
        self.hyperbolic = bool(getattr(args, "hyperbolic", False) if hyperbolic is None else hyperbolic)
        self.static_center = bool(getattr(args, "static_center", False))
        self.distance = str(getattr(args, "distance", "euclidean")).lower()   # staticCenter.py:66,84: 'mahalanobis' option
        channels = list(getattr(args, "channels", [32, 16, 32]))
        self.eps = float(getattr(args, "center_tolerance", 1e-3))
        self.model = STSE(c_in=args.num_coords, h_dim=args.h_dim, latent_dim=args.latent_dim,
                          n_frames=args.dataset_seg_len, dropout=args.dropout, n_joints=_joints(args),
                          channels=channels, projector=getattr(args, "projector", "linear"),
                          encoder_type=getattr(args, "encoder_type", "STS_GCN"), distance=self.distance)
        self.learning_rate = args.opt_lr
        self.batch_size = getattr(args, "dataset_batch_size", 2048)
        self.logged: Dict[str, float] = {}
        self._engine: Optional[STSETrainStep] = None
        self._epoch = 0
        self._temp = None
        self.gts: Optional[Dict[Tuple[int, int], np.ndarray]] = None   # frame masks; else read from args.gt_path

    # ---- inference ---------------------------------------------------------------------
    def forward(self, x):
        tensor_data, transformation_idx, metadata, actual_frames = x[0], x[1], x[2], x[3]
        hidden_out = self.model(tensor_data)
        return hidden_out, transformation_idx, metadata, actual_frames

    def log(self, name: str, value) -> None:
        self.logged[name] = float(value)

    # ---- centre initialisation (staticCenter.py:95-130, dynamicCenter.py:76-102, hyperbolic_encoder.py:85-135)
    def setup(self, stage: str = None, train_loader: Optional[Callable[[], Iterable]] = None) -> None:
        if stage != "fit":
            return
        dev = next(self.model.parameters()).device
        acc = torch.zeros(ops.HEAD_SLOTS, device=dev)
        L = self.model.latent_dim
        maha = self.distance == 'mahalanobis' and not self.hyperbolic
        gram = torch.zeros(L, L, device=dev) if maha else None
        eye = torch.eye(L, device=dev) if maha else None
        self.model.eval()
        with torch.no_grad():
            for batch in train_loader():
                z = self.model(batch[0].to(dev).contiguous())
                if self.hyperbolic:
                    ops.poincare_head(z, None, need_grad=False, acc=acc)
                elif maha:                                  # centre sums + second moments in one pass
                    ops.mahalanobis_head(z, self.model.c, eye, need_grad=False, acc=acc, gram=gram)
                else:
                    ops.mse_head(z, self.model.c, need_grad=False, acc=acc)
        parallel.allreduce_sum_(acc)
        if maha:
            parallel.allreduce_sum_(gram)
        self.n_samples = float(acc[17]) if not self.hyperbolic else None
        if self.hyperbolic:
            c = ops.midpoint_finalize(acc, L)
            assert bool((c < 1).all()), f"center is out of the ball\nc = {c}"        # hyperbolic_encoder.py:123
            self.model.c.copy_(c)
        elif self.static_center:
            self.model.c.copy_(ops.center_finalize(acc, self.eps, L))                 # staticCenter.py:118-123
        else:
            # dynamicCenter.py:96-100: the initial centre is stored but model.c stays 0 during the first epoch
            self._temp = ops.center_finalize(acc, 0.0, L)
        if maha:                                            # staticCenter.py:124-127,133-142
            from .trainer import inv_cov_from_moments
            mu = self.model.c if self.static_center else self._temp
            self.model.inv_cov_matrix.copy_(inv_cov_from_moments(gram, acc, mu, L))
        self.model.train()
        # `sync_batchnorm` (not a key of the reference's yamls, whose DDP keeps per-rank BatchNorm statistics): optional SyncBN
        extra = {"sync_bn": True} if bool(getattr(self.args, "sync_batchnorm", False)) else {}
        self._engine = make_train_step(self.model, lr=self.learning_rate, alpha=float(getattr(self.args, "alpha", 0.0)),
                                     head="poincare" if self.hyperbolic else ("mahalanobis" if maha else "euclidean"), **extra)
        self._epoch = 0

    # ---- one optimisation step -----------------------------------------------------------
    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        dev = self.model.c.device
        stats = self._engine.step(batch[0].to(dev, non_blocking=True).contiguous())
        # the reference returns head + alpha * reg (staticCenter.py:180-189); the regulariser is evaluated when it is logged
        # (two launches) and that value is reused for the steps in between: at alpha ~ 1e-6 it moves in the 7th digit per step
        if batch_idx % 20 == 0 or self._reg_last is None: # log_every_n_steps=20 (train_COSKAD.py:76)
            self._reg_last = self._engine.reg_loss().reshape(())
        loss = stats[0] + self._engine.alpha * self._reg_last
        if batch_idx % 20 == 0:
            name = "poincare_loss" if self.hyperbolic else "hypersphere_loss"
            self.log(name, stats[0]); self.log("regularization", self._reg_last)
            self.log("loss", float(loss))
        return loss

    def on_train_epoch_end(self) -> None:
        eng = self._engine
        if eng.head == 'mahalanobis':                      # staticCenter.py:146-147 (before the centre update)
            eng.refresh_inv_cov(self.model.c)
        if self.hyperbolic:
            if not self.static_center:                     # hyperbolic_encoder.py:175-183
                c = eng.refresh_center()
                self.log("center/eucl", torch.linalg.norm(c))
            else:
                eng.center_acc.zero_()
        elif self.static_center:
            eng.center_acc.zero_()
        else:                                              # dynamicCenter.py:125-142
            if self._epoch == 0:
                parallel.allreduce_sum_(eng.center_acc)
                self.model.c.copy_(self._temp)
                eng.center_acc.zero_()
            else:
                eng.refresh_center(eps=self.eps)
        self._epoch += 1

    training_epoch_end = on_train_epoch_end

    def configure_optimizers(self) -> Dict:
        """Adam(lr=opt_lr) (+ ReduceLROnPlateau(mode='max', factor=0.2, min_lr=1e-6) on validation_auc when
        args.validation); the HIP engine owns the Adam state, the Trainer applies the plateau rule."""
        return {"optimizer": "adam", "lr": self.learning_rate,
                "lr_scheduler": {"name": "ReduceLROnPlateau", "mode": "max", "factor": 0.2,
                                 "patience": 2 if not self.static_center and not self.hyperbolic else 100,
                                 "min_lr": 1e-6},
                "monitor": "validation_auc"}

    # ---- validation / scoring --------------------------------------------------------------
    def validation_step(self, batch, batch_idx: int = 0):
        dev = next(self.model.parameters()).device
        with torch.no_grad():
            return self.forward([batch[0].to(dev).contiguous(), batch[1], batch[2], batch[3]])

    predict_step = validation_step

    def validation_epoch_end(self, outputs: List) -> float:
        hidden = torch.cat([o[0] for o in outputs], 0)
        trans = torch.cat([o[1] for o in outputs], 0)
        meta = torch.cat([o[2] for o in outputs], 0)
        frames = torch.cat([o[3] for o in outputs], 0)
        return self.post_processing(hidden, trans, meta, frames)

    def window_scores(self, hidden: torch.Tensor) -> torch.Tensor:
        """per-window anomaly score on the device (eval_utils.py:63-67)."""
        if self.hyperbolic:
            _, _, _, score = ops.poincare_head(hidden.contiguous(), self.model.c, need_grad=False, need_score=True)
        elif self.distance == 'mahalanobis':               # eval_utils.py:41-47
            _, _, score = ops.mahalanobis_head(hidden.contiguous(), self.model.c, self.model.inv_cov_matrix,
                                               need_grad=False, need_score=True)
        else:
            _, _, score = ops.mse_head(hidden.contiguous(), self.model.c, need_grad=False, need_score=True)
        return score

    def _load_gts(self) -> Dict[Tuple[int, int], np.ndarray]:
        if self.gts is not None:
            return self.gts
        out = {}
        for fn in sorted(os.listdir(self.args.gt_path)):
            if fn.endswith(".npy"):
                sc, cl = int(fn.split("_")[0]), int(fn.split("_")[1].split(".")[0])
                out[(sc, cl)] = np.load(os.path.join(self.args.gt_path, fn))
        return out

    def _load_hr_masks(self) -> Optional[Dict[Tuple[int, int], np.ndarray]]:
        """Human-related frame subsets (eval_COSKAD.py:86-97, utils/model_utils.py:149-161): `use_hr: True` +
        `hr_mask_path` (a glob of `<scene>_<clip>.npy` boolean masks; the reference hard-codes its own disk path), or
        masks assigned to `self.hr_masks`."""
        if getattr(self, "hr_masks", None) is not None:
            return self.hr_masks
        if getattr(self.args, "use_hr", False) and getattr(self.args, "hr_mask_path", None):
            return eval_utils.hr_masks_from_dir(self.args.hr_mask_path)
        return None

    def post_processing(self, hidden_out, trans, meta, frames) -> float:
        return self._score_windows(self.window_scores(hidden_out), trans, meta, frames)

    def _score_windows(self, scores, trans, meta, frames) -> float:
        """window scores -> frames -> persons -> smoothing -> transformations -> AUC (eval_COSKAD.py:140-253)."""
        num_transform = max(1, int(getattr(self.args, "dataset_num_transform", 1)))
        auc, per_t, gt = eval_utils.score_dataset(scores, trans, meta, frames, self._load_gts(), num_transform,
                                                  smoothing=int(getattr(self.args, "smoothing", 50)),
                                                  dataname=getattr(self.args, "dataset_choice", "UBnormal"),
                                                  pad_size=int(getattr(self.args, "pad_size", -1)),
                                                  hr_masks=self._load_hr_masks())
        self.log("validation_auc", auc)
        self.last_scores = per_t
        return auc


class _AutogradLit(LitEncoder):
    """Shared machinery of the wrappers whose loss involves the decoder (autoencoder, spherical VAE): the model runs
    through its module surface (autograd nodes around the HIP kernels), torch's Adam optimises, gradients are averaged
    over ranks.  Validation outputs are per-window scores computed on the device."""

    model_cls = None

    def _build(self, args: Namespace, **extra) -> None:
        nn.Module.__init__(self)
        self.args, self.hparams = args, Namespace(args=args)
        self.hyperbolic, self.static_center, self.distance = False, True, 'euclidean'
        self.eps = float(getattr(args, "center_tolerance", 1e-3))
        self.model = self.model_cls(c_in=args.num_coords, h_dim=args.h_dim, latent_dim=args.latent_dim,
                                    n_frames=args.dataset_seg_len, dropout=args.dropout, n_joints=_joints(args),
                                    channels=list(getattr(args, "channels", [32, 16, 32])), **extra)
        self.learning_rate = args.opt_lr
        self.batch_size = getattr(args, "dataset_batch_size", 2048)
        self.logged: Dict[str, float] = {}
        self.gts = None
        object.__setattr__(self, "_engine", self)   # the Trainer's plateau rule calls _engine.set_lr (not a submodule)
        self._opt = None
        self._flat = None

    def set_lr(self, lr: float) -> None:
        if self._flat is not None:
            self._flat.set_lr(lr)
            return
        for g in self._opt.param_groups:
            g['lr'] = lr

    def _make_optimiser(self, mode: str, **weights) -> None:
        """The flat-buffer step with the fused Adam (trainer.STSAETrainStep: no autograd around the encoder / decoder chains)
        where the model is within its kernels; torch autograd + torch.optim.Adam over the module surface otherwise."""
        from .trainer import STSAETrainStep
        from .models.common.components import Encoder
        self._flat = None
        on_gpu = next(self.model.parameters()).is_cuda
        if on_gpu and isinstance(self.model.encoder, Encoder) and STSAETrainStep.supports(self.model) \
                and not self.model.encoder.model[-1].is_wide and not self.model.decoder.model[-1].is_wide:
            self._flat = STSAETrainStep(self.model, mode=mode, lr=self.learning_rate,
                                        alpha=float(getattr(self.args, "alpha", 0.0)), **weights)
        else:
            self._opt = torch.optim.Adam(self.model.parameters(), lr=self.learning_rate)

    def _reg_loss(self) -> torch.Tensor:
        """utils/model_utils.py:90-105 (differentiable: it is part of these wrappers' loss)."""
        ps = [p for n, p in self.model.named_parameters() if 'bias' not in n]
        return 0.5 * sum((p ** 2).sum() for p in ps) / len(ps)

    def _optimise(self, loss: torch.Tensor) -> None:
        self._opt.zero_grad(set_to_none=True)
        loss.backward()
        parallel.allreduce_grads_mean_(list(self.model.parameters()))     # one flat bucket per step
        self._opt.step()

    def validation_step(self, batch, batch_idx: int = 0):
        dev = self.model.c.device
        with torch.no_grad():
            return self.window_scores_from_batch(batch[0].to(dev)), batch[1], batch[2], batch[3]

    predict_step = validation_step

    def validation_epoch_end(self, outputs: List) -> float:
        scores, trans, meta, frames = (torch.cat([o[i] for o in outputs], 0) for i in range(4))
        return self._score_windows(scores, trans, meta, frames)

    def on_train_epoch_end(self) -> None:
        pass

    training_epoch_end = on_train_epoch_end

    def configure_optimizers(self) -> Dict:
        """euclidean_autoencoder.py:137-150 / spherical_vae.py:143-156: ReduceLROnPlateau with patience 2."""
        cfg = super().configure_optimizers()
        cfg["lr_scheduler"]["patience"] = 2
        return cfg


class LitAutoEncoder(_AutogradLit):
    """models/euclidean_autoencoder.py: STSAE with loss lambda_ * MSE(x_rec, x) + MSE(z, c) + alpha * reg (:106-118),
    static centre from the initial latents (:76-100), window score = reconstruction error (eval_utils.py:77-105 with its
    default loss_type 'rec').  The reference unpacks the model's outputs in the order of its missing old module
    (SURVEY 8a row a9); the intent -- first the reconstruction, then the latent -- is what is implemented."""

    from .models.sts.ae import STSAE as model_cls

    def __init__(self, args: Namespace) -> None:
        self._build(args)
        self.lambda_ = float(getattr(args, "lambda_", 0.01))

    def forward(self, x):
        z, x_rec = self.model(x[0])
        return x_rec, z, x[0], x[1], x[2], x[3]

    def setup(self, stage: str = None, train_loader=None) -> None:
        if stage != "fit":
            return
        dev = self.model.c.device
        acc = torch.zeros(ops.HEAD_SLOTS, device=dev)
        self.model.eval()
        with torch.no_grad():
            for batch in train_loader():
                z, _ = self.model(batch[0].to(dev))
                ops.mse_head(z.contiguous(), self.model.c, need_grad=False, acc=acc)
        parallel.allreduce_sum_(acc)
        self.model.c.copy_(ops.center_finalize(acc, self.eps, self.model.latent_dim))
        self.model.train()
        self._make_optimiser('ae', lambda_=self.lambda_)

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        x = batch[0].to(self.model.c.device, non_blocking=True)
        if self._flat is not None:
            out = self._flat.step(x)
            if batch_idx % 20 == 0 or self._reg_last is None:   # (the regulariser: evaluated when logged, reused in between)
                self._reg_last = self._flat.reg_loss().reshape(())
            loss = (self.lambda_ * out['rec'] + out['head']).reshape(()) + float(getattr(self.args, "alpha", 0.0)) * self._reg_last
            if batch_idx % 20 == 0:
                self.log("loss", loss)
                self.log("reconstruction_loss", out['rec']); self.log("hypersphere_loss", out['head'])
                self.log("regularization", self._reg_last)
            return loss
        z, x_rec = self.model(x)
        loss_reco = F.mse_loss(x_rec, x)
        loss_h = F.mse_loss(z, self.model.c.expand_as(z))
        loss_reg = self._reg_loss()
        loss = self.lambda_ * loss_reco + loss_h + float(getattr(self.args, "alpha", 0.0)) * loss_reg
        self._optimise(loss)
        if batch_idx % 20 == 0:
            self.log("loss", loss.detach()); self.log("reconstruction_loss", loss_reco.detach())
            self.log("hypersphere_loss", loss_h.detach()); self.log("regularization", loss_reg.detach())
        return loss.detach()

    # score type: the wrapper's own validation calls windows_based_loss_rec_and_hy with its default 'rec'
    # (euclidean_autoencoder.py:197); eval_COSKAD.py derives 'hyp' from its rec_loss_weight = 0 (:58-66) and sets it here
    score_type = 'rec'
    rec_loss_weight = 0.2

    def window_scores_from_batch(self, x: torch.Tensor) -> torch.Tensor:
        z, x_rec = self.model(x)
        return eval_utils.rec_and_hy_window_scores(x, x_rec, z, self.model.c, self.rec_loss_weight, self.score_type)


class LitVAE(_AutogradLit):
    """models/spherical_vae.py: STSVAE with loss phi * MSE(x_rec, x) + alpha * reg + beta * KL(q || p) + gamma * mean(1/kappa)
    (:81-107), `mean_vector` = mean of the epoch's sampled latents (:110-116), window score = 1 - cos(z, mean_vector)
    of the SAMPLED latent (:76-78, 200) -- stochastic, as in the reference."""

    from .models.sts.vae import STSVAE as model_cls

    def __init__(self, args: Namespace) -> None:
        self.distribution = str(getattr(args, "distribution", "ps")).lower()
        self._build(args, distribution=self.distribution, projector=getattr(args, "projector", "linear"))
        self.phi, self.beta, self.gamma = float(args.phi), float(args.beta), float(args.gamma)
        self.warmup_counter = int(getattr(args, "warmup_epochs", 0))
        if not hasattr(self.model, "mean_vector"):
            self.model.register_buffer("mean_vector", torch.zeros(1, args.latent_dim))
        self._zsum, self._zn = None, 0

    def forward(self, x):
        z, x_rec, _ = self.model(x[0])
        return z, x_rec, x[1], x[2], x[3]

    def setup(self, stage: str = None, train_loader=None) -> None:
        if stage == "fit":
            self.model.train()
            self._make_optimiser('vae', phi=self.phi, beta=self.beta, gamma=self.gamma)

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        from .models.sts.vae import kl_ps_uniform
        x = batch[0].to(self.model.c.device, non_blocking=True)
        if self._flat is not None:
            out = self._flat.step(x)
            s = out['z'].sum(0, keepdim=True)
            self._zsum = s if self._zsum is None else self._zsum + s
            self._zn += out['z'].shape[0]
            if batch_idx % 20 == 0 or self._reg_last is None:   # (the regulariser: evaluated when logged, reused in between)
                self._reg_last = self._flat.reg_loss().reshape(())
            loss = ((self.phi * out['rec'] + self.beta * out['head'] + self.gamma * out['exp']).reshape(())
                    + float(getattr(self.args, "alpha", 0.0)) * self._reg_last)
            if batch_idx % 20 == 0:
                self.log("loss", loss)
                self.log("reconstruction_loss", out['rec']); self.log("kl_loss", out['head'])
                self.log("exp_dist_loss", out['exp']); self.log("regularization", self._reg_last)
            return loss
        z, x_rec, (q, p, kappa) = self.model(x)
        with torch.no_grad():
            s = z.sum(0, keepdim=True)
            self._zsum = s if self._zsum is None else self._zsum + s
            self._zn += z.shape[0]
        if self.distribution == 'normal':
            loss_kl = torch.distributions.kl.kl_divergence(q, p).sum(-1).mean()
        else:
            loss_kl = kl_ps_uniform(q, p).mean()
        loss_rec = F.mse_loss(x_rec, x)
        loss_reg = self._reg_loss()
        loss_exp = (1 / kappa).mean()
        loss = self.phi * loss_rec + float(getattr(self.args, "alpha", 0.0)) * loss_reg + self.beta * loss_kl + self.gamma * loss_exp
        self._optimise(loss)
        if batch_idx % 20 == 0:
            self.log("loss", loss.detach()); self.log("reconstruction_loss", loss_rec.detach())
            self.log("kl_loss", loss_kl.detach()); self.log("exp_dist_loss", loss_exp.detach())
            self.log("regularization", loss_reg.detach())
        return loss.detach()

    def on_train_epoch_end(self) -> None:                    # update_state (:110-116)
        if self._zn:
            zs = torch.cat([self._zsum.reshape(-1), self._zsum.new_tensor([float(self._zn)])])
            parallel.allreduce_sum_(zs)
            self.model.mean_vector.copy_((zs[:-1] / zs[-1]).reshape(1, -1))
        if self.warmup_counter > 0:
            self.warmup_counter -= 1
        self._zsum, self._zn = None, 0

    training_epoch_end = on_train_epoch_end

    def window_scores_from_batch(self, x: torch.Tensor) -> torch.Tensor:
        z, _, _ = self.model(x)
        return 1 - F.cosine_similarity(self.model.mean_vector.expand_as(z), z)


class Trainer:
    """The few pytorch_lightning.Trainer behaviours the reference relies on (train_COSKAD.py:70-85,
    eval_COSKAD.py:110-116): fit with per-epoch validation, top-k checkpoints on the monitored value,
    ReduceLROnPlateau, predict from a checkpoint.  Loaders are zero-argument callables returning an iterable of
    [x, trans, meta, frames] batches (each rank builds its own shard)."""

    def __init__(self, max_epochs: int = 1, ckpt_dir: Optional[str] = None, save_top_k: int = 2,
                 monitor: str = "validation_auc") -> None:
        self.max_epochs, self.ckpt_dir, self.save_top_k, self.monitor = max_epochs, ckpt_dir, save_top_k, monitor
        self.history: List[Dict[str, float]] = []
        self._best: List[Tuple[float, str]] = []

    def fit(self, model: LitEncoder, train_loader, val_loader=None) -> None:
        # DDP wrap semantics (train_COSKAD.py:75-78): every rank starts from rank 0's parameters and buffers
        parallel.broadcast_module_(model.model)
        model.setup("fit", train_loader)
        sched = model.configure_optimizers()["lr_scheduler"]
        best, bad, lr = -float("inf"), 0, model.learning_rate
        for epoch in range(self.max_epochs):
            model.model.train()
            for i, batch in enumerate(train_loader()):
                model.training_step(batch, i)
            model.on_train_epoch_end()
            rec = dict(model.logged, epoch=epoch)
            if val_loader is not None:
                auc = self.validate(model, val_loader)
                rec["validation_auc"] = auc
                if auc > best:
                    best, bad = auc, 0
                else:
                    bad += 1
                    if bad > sched["patience"]:
                        lr = max(lr * sched["factor"], sched["min_lr"])
                        model._engine.set_lr(lr)
                        bad = 0
                self._checkpoint(model, epoch, auc)
            elif "loss" in rec:
                # no validation: ModelCheckpoint(monitor='loss', mode='min') (train_COSKAD.py:70-73)
                self._checkpoint(model, epoch, rec["loss"], monitor="loss", mode="min")
            self.history.append(rec)

    def validate(self, model: LitEncoder, loader) -> float:
        # Lightning's DDP wrap broadcasts rank 0's buffers (BatchNorm running statistics, the centre) before every
        # forward (broadcast_buffers=True): validation scores every shard with the model that gets checkpointed
        parallel.broadcast_buffers_(model.model)
        model.model.eval()
        outs = [model.validation_step(b, i) for i, b in enumerate(loader())]
        if parallel.world_size() > 1:
            # shards are wrap-padded to equal length (same number of collectives on every rank): gather, then drop the
            # duplicated windows -- a window is identified by (transformation, scene, clip, person, start)
            cat = [torch.cat([o[i].to(outs[0][0].device) for o in outs], 0) for i in range(4)]
            cat = [parallel.gather_rows(t) for t in cat]
            keep = parallel.dedupe_rows(torch.cat([cat[1].long().reshape(-1, 1), cat[2].long()], 1))
            outs = [tuple(t[keep] for t in cat)]
        return model.validation_epoch_end(outs)

    def predict(self, model: LitEncoder, loader, ckpt_path: Optional[str] = None):
        if ckpt_path:
            load_checkpoint(model, ckpt_path)
        model.model.eval()
        return [model.predict_step(b, i) for i, b in enumerate(loader())]

    def _checkpoint(self, model: LitEncoder, epoch: int, value: float, monitor: Optional[str] = None,
                    mode: str = "max") -> None:
        if not self.ckpt_dir or parallel.rank() != 0:
            return
        os.makedirs(self.ckpt_dir, exist_ok=True)
        path = os.path.join(self.ckpt_dir, f"epoch={epoch}-{monitor or self.monitor}={value:.4f}.ckpt")
        save_checkpoint(model, path, epoch)
        self._best.append((value, path))
        self._best.sort(key=lambda t: -t[0] if mode == "max" else t[0])
        for _, p in self._best[self.save_top_k:]:
            if os.path.exists(p):
                os.remove(p)
        self._best = self._best[:self.save_top_k]


def save_checkpoint(model: LitEncoder, path: str, epoch: int = 0) -> None:
    """Lightning layout: state_dict keys prefixed `model.`, hyper_parameters = the args Namespace."""
    sd = {"model." + k: v.detach().cpu().clone() for k, v in model.model.state_dict().items()}
    torch.save({"state_dict": sd, "hyper_parameters": {"args": vars(model.args)}, "epoch": epoch}, path)


def load_checkpoint(model: LitEncoder, path: str) -> None:
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = {k[len("model."):]: v for k, v in ck["state_dict"].items() if k.startswith("model.")}
    model.model.load_state_dict(sd, strict=True)
