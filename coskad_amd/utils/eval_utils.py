"""Window -> frame -> clip anomaly scoring (mirror of the reference's utils/eval_utils.py + the scoring loops of
eval_COSKAD.py:140-253 and the wrappers' post_processing), vectorised.

The reference walks transformations x clips x persons in Python and scatters one window at a time
(eval_utils.py:69-74).  Here the per-window scores come from the HIP head kernels (coskad_amd.ops.mse_head /
poincare_head: MSELoss(reduction='none')(c,z).mean(-1) resp. dist(c, zh)), and the aggregation is ONE segmented
reduction keyed on (transformation, scene, clip, person, frame) with torch index ops on whatever device the
scores live on.  Smoothing and AUC use the same scipy / sklearn calls as the reference (eval_utils.py:206,218).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
from scipy.ndimage import gaussian_filter1d
from sklearn.metrics import roc_auc_score


def score_process(score: np.ndarray, win_size: int = 50, dataname: str = 'STC', use_scaler: bool = False) -> np.ndarray:
    """Shift by 11 frames and Gaussian-smooth with sigma 30 (reference eval_utils.py:200-207)."""
    scores_shifted = np.zeros_like(score)
    shift = 8 + (8 // 2) - 1
    scores_shifted[shift:] = score[:-shift]
    return gaussian_filter1d(scores_shifted, 30)


def filter_vectors_by_cond(vecs, cond):
    """reference eval_utils.py:168-172."""
    return [v[cond] for v in vecs]


def frame_scores(window_scores: torch.Tensor, trans: torch.Tensor, meta: torch.Tensor, frames: torch.Tensor,
                 clip_lengths: Dict[Tuple[int, int], int], num_transform: int):
    """-> dict {(t, scene, clip): np.ndarray[n_frames]} of per-frame clip scores BEFORE smoothing:
    mean over the windows covering a frame (windows whose score is exactly 0 count as missing, as the
    reference's `== 0.0 -> NaN`), then max over persons (eval_COSKAD.py:201-211)."""
    dev = window_scores.device
    s = window_scores.reshape(-1).to(torch.float64)
    N, T = frames.shape
    trans, meta, frames = trans.to(dev).long(), meta.to(dev).long(), frames.to(dev).long()
    # compact ids of (t, scene, clip, person) and of (t, scene, clip)
    key_p = torch.stack([trans, meta[:, 0], meta[:, 1], meta[:, 2]], 1)
    uniq_p, pid = torch.unique(key_p, dim=0, return_inverse=True)
    uniq_c, cid_of_p = torch.unique(uniq_p[:, :3], dim=0, return_inverse=True)
    fmax = int(max(clip_lengths.values()))
    idx = (pid[:, None] * fmax + (frames - 1)).reshape(-1)               # `frames - 1`: eval_utils.py:72
    w = s[:, None].expand(N, T).reshape(-1)
    valid = (w != 0).to(torch.float64)
    ssum = torch.zeros(uniq_p.shape[0] * fmax, dtype=torch.float64, device=dev).index_add_(0, idx, w * valid)
    cnt = torch.zeros_like(ssum).index_add_(0, idx, valid)
    mean = torch.where(cnt > 0, ssum / cnt.clamp_min(1), torch.zeros_like(ssum)).reshape(-1, fmax)
    clip = torch.full((uniq_c.shape[0], fmax), -float("inf"), dtype=torch.float64, device=dev)
    clip = clip.scatter_reduce(0, cid_of_p[:, None].expand(-1, fmax), mean, reduce="amax", include_self=True)
    clip = clip.cpu().numpy()
    out = {}
    for i, (t, sc, cl) in enumerate(uniq_c.cpu().tolist()):
        n = clip_lengths.get((sc, cl))
        if n is not None and t < num_transform:
            out[(t, sc, cl)] = clip[i, :n]
    return out


def score_dataset(window_scores, trans, meta, frames, gts: Dict[Tuple[int, int], np.ndarray], num_transform: int,
                  smoothing: int = 50, dataname: str = 'UBnormal'):
    """Full scoring: -> (final AUC, {t: smoothed score vector}, concatenated gt).
    gts: {(scene, clip): frame mask}, clips are concatenated in sorted key order (sorted(os.listdir(gt_path)))."""
    lengths = {k: int(v.shape[0]) for k, v in gts.items()}
    fs = frame_scores(torch.as_tensor(window_scores), torch.as_tensor(trans), torch.as_tensor(meta),
                      torch.as_tensor(frames), lengths, num_transform)
    keys = sorted(gts.keys())
    per_t = {}
    for t in range(num_transform):
        parts = []
        for k in keys:
            raw = fs.get((t, k[0], k[1]))
            if raw is None:
                raw = np.zeros(lengths[k])          # a clip without detections scores 0 everywhere
            parts.append(score_process(raw, win_size=smoothing, dataname=dataname, use_scaler=False))
        per_t[t] = np.concatenate(parts)
    gt = np.concatenate([gts[k] for k in keys])
    pds = np.mean(np.stack(list(per_t.values()), 0), 0)
    return roc_auc_score(gt, pds), per_t, gt


def ROC(y_test, y_pred):
    """AUC (reference eval_utils.py:216-230 without the matplotlib plot)."""
    return roc_auc_score(y_test, y_pred)
