"""CPU oracle of the reference's window -> frame -> clip anomaly scoring -- TEST INFRASTRUCTURE ONLY.

A direct (loop-for-loop) restatement of
  utils/eval_utils.py:57-74   windows_based_loss_hy  (scatter window scores onto frames, `frames - 1`)
  utils/eval_utils.py:200-207 score_process          (shift by 8 + 8//2 - 1 = 11, gaussian_filter1d sigma 30)
  eval_COSKAD.py:140-253 / models/euclidean_encoder_dynamicCenter.py:160-243  (transformations x clips x persons:
      0 -> NaN, nanmean over windows, NaN -> 0, max over persons, concat clips, AUC per transformation,
      mean of the smoothed scores over transformations, final roc_auc_score)
  utils/eval_utils.py:77-106  windows_based_loss_rec_and_hy (autoencoder score types 'rec' / 'hyp' / 'rec+hyp')
  utils/eval_utils.py:210-248 ranges + pad_scores    (absence intervals widened by pad_size; eval_COSKAD.py:205-206)
  eval_COSKAD.py:213-215      human-related boolean masks applied to clip score and ground truth before smoothing
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


def ranges(nums):
    """eval_utils.py:210-214: inclusive (first, last) pairs of the runs of consecutive integers in `nums`."""
    nums = sorted(set(nums))
    out, i = [], 0
    while i < len(nums):
        j = i
        while j + 1 < len(nums) and nums[j + 1] == nums[j] + 1:
            j += 1
        out.append((nums[i], nums[j]))
        i = j + 1
    return out


def pad_scores(score, gt, pad_size):
    """eval_utils.py:232-248, branch for branch."""
    n = len(gt)
    absent = set(range(n - 1)) - set(np.nonzero(score)[0].tolist())
    todo = []
    for start, end in ranges(absent):
        if start == 0 and end == n - 2:
            continue
        elif start == 0 and end != n - 2:
            todo.append((start, min(end + pad_size, n)))
        elif start != 0 and end == n - 2:
            todo.append((max(start - pad_size, 0), end))
        else:
            todo.append((max(start - pad_size, 0), min(end + pad_size, n)))
    for a, b in todo:
        for i in range(a, b):
            score[i] = 0
    return score


def rec_and_hy_window_scores(x, x_rec, z, c, rec_loss_weight=0.2, loss_type='rec'):
    """eval_utils.py:77-106 up to the scatter: per-window 'rec' / 'hyp' / 'rec+hyp' score (numpy, float64)."""
    w = x.shape[0]
    g = x.transpose(0, 2, 3, 1).reshape(w, -1).astype(np.float64)
    o = x_rec.transpose(0, 2, 3, 1).reshape(w, -1).astype(np.float64)
    rec = ((g - o) ** 2).mean(-1)
    hyp = ((c[None].astype(np.float64) - z.astype(np.float64)) ** 2).mean(-1)
    if loss_type == 'rec+hyp':
        rec = rec / rec_loss_weight
    out = np.zeros(w)
    for n in range(w):
        if loss_type == 'rec':
            out[n] = rec[n]
        if loss_type == 'hyp':
            out[n] = hyp[n]
        if loss_type == 'rec+hyp':
            out[n] = rec[n] + hyp[n]
    return out


def score_dataset(window_scores, trans, meta, frames, gts, num_transform, pad_size=-1, hr_masks=None):
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
                v = np.where(np.isnan(v), 0, v)
                if pad_size != -1:                                   # eval_COSKAD.py:205-206
                    v = pad_scores(v, gt, pad_size)
                per_person.append(v)
            clip_score = np.amax(np.stack(per_person, 0), 0) if per_person else np.zeros(n_frames)
            if hr_masks and (scene, clip) in hr_masks:               # eval_COSKAD.py:213-215
                clip_score = clip_score[hr_masks[(scene, clip)]]
                gt = gt[hr_masks[(scene, clip)]]
            scores.append(score_process(clip_score))
            gcat.append(gt)
        per_t[t] = np.concatenate(scores)
        gt_t[t] = np.concatenate(gcat)
    pds = np.mean(np.stack(list(per_t.values()), 0), 0)
    return roc_auc_score(gt_t[0], pds), per_t, gt_t[0]
