"""Window -> frame -> clip anomaly scoring (mirror of the reference's utils/eval_utils.py + the scoring loops of
eval_COSKAD.py:140-253 and the wrappers' post_processing), vectorised.

The reference walks transformations x clips x persons in Python and scatters one window at a time
(eval_utils.py:69-74).  Here the per-window scores come from the HIP head kernels (coskad_amd.ops.mse_head /
poincare_head: MSELoss(reduction='none')(c,z).mean(-1) resp. dist(c, zh)), and the aggregation is ONE segmented
reduction keyed on (transformation, scene, clip, person, frame) with torch index ops on whatever device the
scores live on.  Smoothing and AUC use the same scipy / sklearn calls as the reference (eval_utils.py:206,218).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

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


def zero_runs(score: np.ndarray, n: int):
    """Inclusive (first, last) index pairs of the maximal runs of exactly-zero entries among frames 0..n-1
    (the `ranges` helper of the reference, eval_utils.py:210-214, applied to the set of zero frames)."""
    z = np.zeros(n + 2, dtype=bool)
    z[1:n + 1] = score[:n] == 0
    starts = np.nonzero(z[1:n + 1] & ~z[0:n])[0]
    ends = np.nonzero(z[1:n + 1] & ~z[2:n + 2])[0]
    return list(zip(starts.tolist(), ends.tolist()))


def pad_scores(score: np.ndarray, gt: np.ndarray, pad_size: int) -> np.ndarray:
    """Widen every interval in which a person is absent (score exactly 0) by `pad_size` frames on the sides that
    touch a presence interval, zeroing the scores there (reference eval_utils.py:232-248; in place, like the
    reference).  Quirks kept: absence is looked for among frames 0..len(gt)-2 only, an all-absent person is left
    alone, and the zeroed range excludes its upper end."""
    n = len(gt)
    last = n - 2
    for a, b in zero_runs(score, n - 1):
        if a == 0 and b == last:
            continue
        lo = a if a == 0 else max(a - pad_size, 0)
        hi = b if (b == last and a != 0) else min(b + pad_size, n)
        score[lo:hi] = 0
    return score


def person_frame_scores(window_scores: torch.Tensor, trans: torch.Tensor, meta: torch.Tensor, frames: torch.Tensor,
                        fmax: int):
    """Segmented mean of the window scores covering each frame, keyed on (transformation, scene, clip, person):
    -> (keys [P,4] int64 (cpu), mean [P,fmax] float64 on the scores' device).  A window whose score is exactly 0
    counts as missing (the reference's `== 0.0 -> NaN` + nanmean, eval_COSKAD.py:201-203); frames nobody covers are 0."""
    dev = window_scores.device
    s = window_scores.reshape(-1).to(torch.float64)
    N, T = frames.shape
    trans, meta, frames = trans.to(dev).long(), meta.to(dev).long(), frames.to(dev).long()
    key_p = torch.stack([trans, meta[:, 0], meta[:, 1], meta[:, 2]], 1)
    uniq_p, pid = torch.unique(key_p, dim=0, return_inverse=True)
    idx = (pid[:, None] * fmax + (frames - 1)).reshape(-1)               # `frames - 1`: eval_utils.py:72
    w = s[:, None].expand(N, T).reshape(-1)
    valid = (w != 0).to(torch.float64)
    ssum = torch.zeros(uniq_p.shape[0] * fmax, dtype=torch.float64, device=dev).index_add_(0, idx, w * valid)
    cnt = torch.zeros_like(ssum).index_add_(0, idx, valid)
    mean = torch.where(cnt > 0, ssum / cnt.clamp_min(1), torch.zeros_like(ssum)).reshape(-1, fmax)
    return uniq_p.cpu(), mean


def frame_scores(window_scores: torch.Tensor, trans: torch.Tensor, meta: torch.Tensor, frames: torch.Tensor,
                 clip_lengths: Dict[Tuple[int, int], int], num_transform: int, pad_size: int = -1,
                 gts: Optional[Dict[Tuple[int, int], np.ndarray]] = None):
    """-> dict {(t, scene, clip): np.ndarray[n_frames]} of per-frame clip scores BEFORE smoothing:
    mean over the windows covering a frame, optional absence padding per person (`pad_size != -1`,
    eval_COSKAD.py:205-206), then max over persons (eval_COSKAD.py:201-211)."""
    fmax = int(max(clip_lengths.values()))
    uniq_p, mean = person_frame_scores(window_scores, trans, meta, frames, fmax)
    dev = mean.device
    uniq_c, cid_of_p = torch.unique(uniq_p[:, :3], dim=0, return_inverse=True)
    if pad_size != -1:
        # per-person interval surgery on the host (a few thousand short vectors), then the max over persons
        m = mean.cpu().numpy()
        for i, (t, sc, cl, _) in enumerate(uniq_p.tolist()):
            n = clip_lengths.get((sc, cl))
            if n is not None:
                ref_gt = gts[(sc, cl)] if gts is not None else np.zeros(n)
                m[i, :n] = pad_scores(m[i, :n].copy(), ref_gt, pad_size)
        mean = torch.from_numpy(m).to(dev)
    clip = torch.full((uniq_c.shape[0], fmax), -float("inf"), dtype=torch.float64, device=dev)
    clip = clip.scatter_reduce(0, cid_of_p.to(dev)[:, None].expand(-1, fmax), mean, reduce="amax", include_self=True)
    clip = clip.cpu().numpy()
    out = {}
    for i, (t, sc, cl) in enumerate(uniq_c.tolist()):
        n = clip_lengths.get((sc, cl))
        if n is not None and t < num_transform:
            out[(t, sc, cl)] = clip[i, :n]
    return out


def score_dataset(window_scores, trans, meta, frames, gts: Dict[Tuple[int, int], np.ndarray], num_transform: int,
                  smoothing: int = 50, dataname: str = 'UBnormal', pad_size: int = -1,
                  hr_masks: Optional[Dict[Tuple[int, int], np.ndarray]] = None):
    """Full scoring: -> (final AUC, {t: smoothed score vector}, concatenated gt).
    gts: {(scene, clip): frame mask}, clips are concatenated in sorted key order (sorted(os.listdir(gt_path))).
    pad_size: eval_COSKAD.py:205-206 (`-1` = off).  hr_masks: {(scene, clip): boolean frame selector} of the
    human-related subsets (utils/model_utils.py:149-161): applied to the clip score and its ground truth before
    smoothing (eval_COSKAD.py:213-215)."""
    lengths = {k: int(v.shape[0]) for k, v in gts.items()}
    fs = frame_scores(torch.as_tensor(window_scores), torch.as_tensor(trans), torch.as_tensor(meta),
                      torch.as_tensor(frames), lengths, num_transform, pad_size=pad_size, gts=gts)
    keys = sorted(gts.keys())
    hr_masks = hr_masks or {}
    per_t = {}
    for t in range(num_transform):
        parts = []
        for k in keys:
            raw = fs.get((t, k[0], k[1]))
            if raw is None:
                raw = np.zeros(lengths[k])          # a clip without detections scores 0 everywhere
            if k in hr_masks:
                raw = raw[hr_masks[k]]
            parts.append(score_process(raw, win_size=smoothing, dataname=dataname, use_scaler=False))
        per_t[t] = np.concatenate(parts)
    gt = np.concatenate([gts[k][hr_masks[k]] if k in hr_masks else gts[k] for k in keys])
    pds = np.mean(np.stack(list(per_t.values()), 0), 0)
    return roc_auc_score(gt, pds), per_t, gt


def hr_masks_from_dir(pattern: str) -> Dict[Tuple[int, int], np.ndarray]:
    """{(scene, clip): boolean mask} from `<scene>_<clip>.npy` files matching a glob (utils/model_utils.py:149-161)."""
    import glob
    import os
    out = {}
    for path in glob.glob(pattern):
        sc, cl = (int(v) for v in os.path.basename(path).split('.')[0].split('_')[:2])
        out[(sc, cl)] = np.load(path)
    return out


def rec_and_hy_window_scores(x: torch.Tensor, x_rec: torch.Tensor, z: torch.Tensor, c: torch.Tensor,
                             rec_loss_weight: float = 0.2, loss_type: str = 'rec') -> torch.Tensor:
    """Per-window score of the autoencoder wrapper (eval_utils.py:77-106): 'rec' = mean squared reconstruction
    error over (T, V, C), 'hyp' = mean squared distance of the latent to the centre, 'rec+hyp' = rec / weight + hyp.
    The 'hyp' part runs on the HIP head kernel when the latents live on the GPU."""
    B = x.shape[0]
    rec = ((x_rec - x) ** 2).reshape(B, -1).mean(-1)
    if loss_type == 'rec':
        return rec
    if z.is_cuda and z.shape[1] <= 16:           # the head kernel keeps a latent in 16 lanes; wider latents: torch expression
        from .. import ops
        _, _, hyp = ops.mse_head(z.contiguous().float(), c.float().contiguous(), need_grad=False, need_score=True)
    else:
        hyp = ((z - c) ** 2).mean(-1)
    if loss_type == 'hyp':
        return hyp
    if loss_type == 'rec+hyp':
        return rec / rec_loss_weight + hyp
    raise ValueError(f"unknown loss_type {loss_type}")


def eval_loss_type(rec_loss_weight: float) -> str:
    """eval_COSKAD.py:58-66: the evaluation script derives the score type from its rec_loss_weight constant."""
    if rec_loss_weight == 0:
        return 'hyp'
    return 'rec' if rec_loss_weight > 100 else 'rec+hyp'


def ROC(y_test, y_pred):
    """AUC (reference eval_utils.py:216-230 without the matplotlib plot)."""
    return roc_auc_score(y_test, y_pred)
