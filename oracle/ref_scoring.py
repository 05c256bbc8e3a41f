"""CPU oracle of the reference's window -> frame -> clip anomaly scoring -- TEST INFRASTRUCTURE ONLY.

A direct (loop-for-loop) restatement of
  utils/eval_utils.py:57-74   windows_based_loss_hy  (scatter window scores onto frames, `frames - 1`)
  utils/eval_utils.py:200-207 score_process          (shift by 8 + 8//2 - 1 = 11, gaussian_filter1d sigma 30)
  eval_COSKAD.py:140-253 / models/euclidean_encoder_dynamicCenter.py:160-243  (transformations x clips x persons:
      0 -> NaN, nanmean over windows, NaN -> 0, max over persons, concat clips, AUC per transformation,
      mean of the smoothed scores over transformations, final roc_auc_score)
The reference functions themselves cannot be imported here (geoopt missing, .cuda() hard-coded: SURVEY 8c), so
this restatement is "parity unpinned" against the reference's code but uses the same scipy / sklearn calls.
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d
from sklearn.metrics import roc_auc_score


def windows_to_frames(score_w, frames_w, n_frames):
    """eval_utils.py:69-74: pose[n, frames[n]-1] = loss[n]."""
    pose = np.zeros((score_w.shape[0], n_frames))
    for n in range(pose.shape[0]):
        pose[n, frames_w[n] - 1] = score_w[n]
    return pose


def score_process(score):
    """eval_utils.py:200-207."""
    shifted = np.zeros_like(score)
    shift = 8 + (8 // 2) - 1
    shifted[shift:] = score[:-shift]
    return gaussian_filter1d(shifted, 30)


def score_dataset(window_scores, trans, meta, frames, gts, num_transform):
    """gts: dict {(scene, clip): 0/1 array per frame}, iterated in sorted order like sorted(os.listdir).
    Returns (final_auc, per-transformation smoothed score vectors, concatenated gt)."""
    keys = sorted(gts.keys())
    per_t, gt_t = {}, {}
    for t in range(num_transform):
        sel_t = trans == t
        s_t, m_t, f_t = window_scores[sel_t], meta[sel_t], frames[sel_t]
        scores, gcat = [], []
        for (scene, clip) in keys:
            gt = gts[(scene, clip)]
            n_frames = gt.shape[0]
            sel_c = (m_t[:, 0] == scene) & (m_t[:, 1] == clip)
            s_c, m_c, f_c = s_t[sel_c], m_t[sel_c], f_t[sel_c]
            per_person = []
            for fig in sorted(set(m_c[:, 2])):
                sel_p = m_c[:, 2] == fig
                mat = windows_to_frames(s_c[sel_p], f_c[sel_p], n_frames)
                mat = np.where(mat == 0.0, np.nan, mat)
                with np.errstate(all="ignore"):
                    import warnings
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        v = np.nanmean(mat, 0)
                per_person.append(np.where(np.isnan(v), 0, v))
            clip_score = np.amax(np.stack(per_person, 0), 0) if per_person else np.zeros(n_frames)
            scores.append(score_process(clip_score))
            gcat.append(gt)
        per_t[t] = np.concatenate(scores)
        gt_t[t] = np.concatenate(gcat)
    pds = np.mean(np.stack(list(per_t.values()), 0), 0)
    return roc_auc_score(gt_t[0], pds), per_t, gt_t[0]
