"""Vectorised scoring (coskad_amd/utils/eval_utils.py) against the loop-for-loop oracle of the reference's
scoring (oracle/ref_scoring.py).  Pure host logic: runs without a GPU."""
import numpy as np
import torch

from coskad_amd.utils import eval_utils as E
from coskad_amd.utils.synthetic import make_dataset
from oracle import ref_scoring as RS


def test_frame_and_dataset_scores_match_reference_loops():
    (x, trans, meta, frames), gts = make_dataset(n_scenes=2, n_clips=2, n_persons=3, clip_len=90, num_transform=2, seed=3)
    g = torch.Generator().manual_seed(0)
    scores = torch.rand(x.shape[0], generator=g, dtype=torch.float64)
    scores[::37] = 0.0                               # exact zeros count as 'missing' in the reference
    auc, per_t, gt = E.score_dataset(scores, trans, meta, frames, gts, num_transform=2)
    auc_r, per_t_r, gt_r = RS.score_dataset(scores.numpy(), trans.numpy(), meta.numpy(), frames.numpy(), gts, 2)
    assert np.array_equal(gt, gt_r)
    for t in per_t:
        np.testing.assert_allclose(per_t[t], per_t_r[t], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(auc, auc_r, rtol=1e-12)


def test_score_process_matches_reference_formula():
    v = np.random.default_rng(0).random(300)
    np.testing.assert_allclose(E.score_process(v), RS.score_process(v))


def test_ragged_persons_and_missing_clip():
    (x, trans, meta, frames), gts = make_dataset(n_scenes=1, n_clips=2, n_persons=2, clip_len=80, seed=5)
    keep = ~((meta[:, 1] == 2) & (meta[:, 2] == 1))  # clip 2 keeps a single person
    keep &= ~((meta[:, 1] == 1) & (meta[:, 3] > 30))  # clip 1: no window covers the late frames
    x, trans, meta, frames = x[keep], trans[keep], meta[keep], frames[keep]
    s = torch.rand(x.shape[0], dtype=torch.float64, generator=torch.Generator().manual_seed(1)) + 0.1
    auc, per_t, _ = E.score_dataset(s, trans, meta, frames, gts, 1)
    auc_r, per_t_r, _ = RS.score_dataset(s.numpy(), trans.numpy(), meta.numpy(), frames.numpy(), gts, 1)
    np.testing.assert_allclose(per_t[0], per_t_r[0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(auc, auc_r)
