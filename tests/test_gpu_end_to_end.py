"""End to end on the GPU: LitEncoder + Trainer on synthetic windows -> per-frame anomaly scores and AUC,
against the CPU oracle evaluating the SAME trained weights (north_star: scores within 1e-4, AUC within 0.1)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle import ref_scoring as RS

pytestmark = pytest.mark.gpu


def make_args(**kw):
    a = dict(num_coords=2, h_dim=16, latent_dim=8, dataset_seg_len=12, dropout=0, channels=[16, 8, 16],
             projector="linear", encoder_type="STS_GCN", hyperbolic=False, static_center=False,
             center_tolerance=1e-3, opt_lr=2e-3, alpha=1e-6, dataset_batch_size=256, dataset_num_transform=2,
             dataset_headless=False, dataset_kp18_format=False, smoothing=50, dataset_choice="UBnormal", validation=True)
    a.update(kw)
    return Namespace(**a)


# `default_*`: the reference yamls' widths (channels 32-16-32, h_dim 64, latent 16): the eval forward is then the ONE-kernel fused
# encoder (csrc/fused_fwd.hip) + the split-K bottleneck (+ csrc/mlp_head.hip for `mlp`), the training step the flat engine
@pytest.mark.parametrize("mode", ["euclid_dynamic", "euclid_static", "hyperbolic", "mahalanobis_static", "euclid_mlp",
                                  "default_linear", "default_mlp", "default_hyperbolic"])
def test_train_score_auc_parity(mode, tmp_path):
    from coskad_amd.lit import LitEncoder, Trainer, load_checkpoint
    from coskad_amd.utils.synthetic import batches, make_dataset
    torch.manual_seed(0)
    train, _ = make_dataset(n_scenes=2, n_clips=3, n_persons=2, clip_len=100, num_transform=2, anomaly=False, seed=1)
    test, gts = make_dataset(n_scenes=1, n_clips=3, n_persons=2, clip_len=100, num_transform=2, anomaly=True, seed=2)
    wide = dict(channels=[32, 16, 32], h_dim=64, latent_dim=16) if mode.startswith("default_") else {}
    args = make_args(hyperbolic=mode.endswith("hyperbolic"), static_center=mode in ("euclid_static", "mahalanobis_static"),
                     distance="mahalanobis" if mode == "mahalanobis_static" else "euclidean",
                     projector="mlp" if mode.endswith("_mlp") else "linear", **wide)   # 'mlp': what 5 of the 7 reference yamls select
    lit = LitEncoder(args).cuda()
    lit.gts = gts
    tr = Trainer(max_epochs=3, ckpt_dir=str(tmp_path))
    tr.fit(lit, lambda: batches(train, 256, shuffle=True, seed=0), lambda: batches(test, 512))
    auc = tr.history[-1]["validation_auc"]
    assert 0.0 <= auc <= 1.0 and len(tr.history) == 3
    # ---- oracle on the same weights / centre
    st = {k: v.detach().cpu().clone() for k, v in lit.model.state_dict().items()}
    x, trans, meta, frames = test
    with torch.no_grad():
        z = R.stse_encode(x, st, training=False)
        if mode.endswith("hyperbolic"):
            s_ref = R.dist(st["c"][None], R.project(R.expmap0(z)))
        elif mode == "mahalanobis_static":
            s_ref = R.mahalanobis(z, st["c"][None], st["inv_cov_matrix"])           # eval_utils.py:41-47
        else:
            s_ref = R.euclid_window_score(z, st["c"])
    auc_ref, per_t_ref, _ = RS.score_dataset(s_ref.double().numpy(), trans.numpy(), meta.numpy(), frames.numpy(), gts, 2)
    # HIP window scores vs oracle
    lit.model.eval()
    with torch.no_grad():
        z_hip = lit.model(x.cuda())
        s_hip = lit.window_scores(z_hip).cpu()
    if mode.startswith("default_"):
        from coskad_amd import engine
        from coskad_amd.models.graph_layers.stsgcn import layer_tensors
        assert engine.fused_encoder_supported([layer_tensors(l) for l in lit.model.encoder.model], 12, 17)   # the fused eval kernel ran
    np.testing.assert_allclose(z_hip.cpu().numpy(), z.numpy(), rtol=1e-4, atol=1e-4)           # latents: 1e-4
    if mode.endswith("hyperbolic"):
        # the Poincare distance amplifies latent rounding by ~1/(1-|zh|^2) near the ball boundary: check the head
        # on the HIP latents tightly, and the full chain at the amplified tolerance
        # (fp32 conditioning of the reference formula itself: a centre close to the boundary makes the Moebius
        # addition lose digits).  The HIP head must be as close to the fp64 truth as the fp32 oracle is.
        zc = z_hip.cpu()
        s64 = R.dist(st["c"][None].double(), R.project(R.expmap0(zc.double())))
        s32 = R.dist(st["c"][None], R.project(R.expmap0(zc)))
        err_oracle = float(((s32.double() - s64) / s64).abs().max())
        err_hip = float(((s_hip.double() - s64) / s64).abs().max())
        assert err_hip <= 2 * err_oracle + 1e-4, (err_hip, err_oracle)
        tol = dict(rtol=1e-2, atol=1e-2)
    elif mode == "mahalanobis_static":
        # sqrt(d^T VI d) with VI = inverse covariance (condition number ~1e3 here) amplifies the 1e-4 latent tolerance
        tol = dict(rtol=5e-3, atol=1e-3)
        s_same = R.mahalanobis(z_hip.cpu(), st["c"][None], st["inv_cov_matrix"])       # head alone, same latents
        np.testing.assert_allclose(s_hip.numpy(), s_same.numpy(), rtol=2e-4, atol=1e-4)
    else:
        tol = dict(rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(s_hip.numpy(), s_ref.numpy(), **tol)
    for t in per_t_ref:
        np.testing.assert_allclose(lit.last_scores[t], per_t_ref[t], **tol)                    # per-frame scores
    assert abs(auc - auc_ref) < 1e-2                                                            # north_star: +-0.1
    # checkpoint round trip in the Lightning layout
    import glob
    ck = sorted(glob.glob(str(tmp_path / "*.ckpt")))
    assert 1 <= len(ck) <= 2
    lit2 = LitEncoder(args).cuda()
    load_checkpoint(lit2, ck[-1])
    assert all(k.startswith("model.") for k in torch.load(ck[-1], weights_only=False)["state_dict"])


@pytest.mark.parametrize("kind", ["autoencoder", "spherical_vae"])
def test_decoder_wrappers_fit_and_score(kind, tmp_path):
    """LitAutoEncoder (euclidean_autoencoder.py) and LitVAE (spherical_vae.py): fit on synthetic windows, per-window
    scores against the oracle on the same trained weights (autoencoder; the VAE score uses a sampled latent)."""
    from coskad_amd.lit import LitAutoEncoder, LitVAE, Trainer
    from coskad_amd.utils.synthetic import batches, make_dataset
    torch.manual_seed(0)
    train, _ = make_dataset(n_scenes=2, n_clips=2, n_persons=2, clip_len=80, num_transform=2, anomaly=False, seed=1)
    test, gts = make_dataset(n_scenes=1, n_clips=2, n_persons=2, clip_len=80, num_transform=2, anomaly=True, seed=2)
    args = make_args(latent_dim=8, lambda_=0.01, phi=1.0, beta=1e-3, gamma=1e-2, distribution="ps", warmup_epochs=1,
                     decoder_channels=[8, 8], use_decoder=(kind == "autoencoder"), use_vae=(kind == "spherical_vae"))
    lit = (LitAutoEncoder if kind == "autoencoder" else LitVAE)(args).cuda()
    lit.gts = gts
    tr = Trainer(max_epochs=2, ckpt_dir=str(tmp_path))
    tr.fit(lit, lambda: batches(train, 256, shuffle=True, seed=0), lambda: batches(test, 512))
    assert len(tr.history) == 2 and all(0.0 <= h["validation_auc"] <= 1.0 for h in tr.history)
    assert np.isfinite(tr.history[-1]["loss"]) and tr.history[-1]["reconstruction_loss"] > 0
    x = test[0]
    lit.model.eval()
    with torch.no_grad():
        s_hip = lit.window_scores_from_batch(x.cuda()).cpu()
    assert s_hip.shape == (x.shape[0],) and torch.isfinite(s_hip).all()
    if kind == "autoencoder":
        st = {k: v.detach().cpu().clone() for k, v in lit.model.state_dict().items()}
        with torch.no_grad():
            z = R.stse_encode(x, st, training=False)
            xr = R.stsae_decode(z, st, 16, 12, 17, training=False)
        s_ref = ((xr - x) ** 2).reshape(x.shape[0], -1).mean(-1)
        np.testing.assert_allclose(s_hip.numpy(), s_ref.numpy(), rtol=2e-4, atol=1e-5)
        assert float(lit.model.c.abs().min()) >= 1e-3 - 1e-9            # clamped centre (:97-98)
    else:
        assert float(lit.model.mean_vector.abs().sum()) > 0             # update_state ran (:110-116)
        assert (s_hip >= -1e-6).all() and (s_hip <= 2 + 1e-6).all()     # 1 - cos in [0, 2]


def test_checkpoints_without_validation_monitor_loss(tmp_path):
    """No validation loader: top-2 checkpoints on the training loss, lower is better (train_COSKAD.py:70-73)."""
    import glob
    from coskad_amd.lit import LitEncoder, Trainer
    from coskad_amd.utils.synthetic import batches, make_dataset
    train, _ = make_dataset(n_scenes=1, n_clips=2, n_persons=2, clip_len=60, num_transform=1, anomaly=False, seed=1)
    lit = LitEncoder(make_args(validation=False)).cuda()
    tr = Trainer(max_epochs=3, ckpt_dir=str(tmp_path))
    tr.fit(lit, lambda: batches(train, 128, shuffle=True, seed=0), None)
    ck = sorted(glob.glob(str(tmp_path / "*.ckpt")))
    assert len(ck) == 2 and all("loss=" in c for c in ck)
    kept = sorted(float(c.split("loss=")[1][:-5]) for c in ck)
    losses = sorted(h["loss"] for h in tr.history)
    np.testing.assert_allclose(kept, losses[:2], atol=1e-4)
