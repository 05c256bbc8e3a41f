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


def _ragged_case(seed=7):
    (x, trans, meta, frames), gts = make_dataset(n_scenes=2, n_clips=2, n_persons=3, clip_len=100, num_transform=2, seed=seed)
    # ragged presence: person 1 leaves clip (1,1) in the middle, person 2 enters clip (2,2) late, person 0 of (1,2) is absent
    keep = ~((meta[:, 0] == 1) & (meta[:, 1] == 1) & (meta[:, 2] == 1) & (meta[:, 3] > 25) & (meta[:, 3] < 60))
    keep &= ~((meta[:, 0] == 2) & (meta[:, 1] == 2) & (meta[:, 2] == 2) & (meta[:, 3] < 40))
    keep &= ~((meta[:, 0] == 1) & (meta[:, 1] == 2) & (meta[:, 2] == 0))
    x, trans, meta, frames = x[keep], trans[keep], meta[keep], frames[keep]
    s = torch.rand(x.shape[0], dtype=torch.float64, generator=torch.Generator().manual_seed(seed)) + 0.05
    return s, trans, meta, frames, gts


def test_pad_scores_matches_reference_branches():
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(5, 60))
        v = rng.random(n)
        for _ in range(int(rng.integers(0, 4))):           # random absence intervals, incl. at both ends
            a = int(rng.integers(0, n)); b = int(rng.integers(a, min(n, a + 15)))
            v[a:b + 1] = 0
        pad = int(rng.integers(0, 8))
        gt = np.zeros(n)
        np.testing.assert_array_equal(E.pad_scores(v.copy(), gt, pad), RS.pad_scores(v.copy(), gt, pad), err_msg=str((trial, v, pad)))
    allzero = np.zeros(12)
    np.testing.assert_array_equal(E.pad_scores(allzero.copy(), allzero, 3), allzero)


def test_padded_and_masked_scoring_matches_reference_loops():
    s, trans, meta, frames, gts = _ragged_case()
    rng = np.random.default_rng(1)
    hr = {(1, 1): rng.random(100) > 0.3, (2, 2): rng.random(100) > 0.5}    # human-related subsets of two clips
    for pad, masks in ((10, None), (-1, hr), (4, hr)):
        auc, per_t, gt = E.score_dataset(s, trans, meta, frames, gts, 2, pad_size=pad, hr_masks=masks)
        auc_r, per_t_r, gt_r = RS.score_dataset(s.numpy(), trans.numpy(), meta.numpy(), frames.numpy(), gts, 2,
                                                pad_size=pad, hr_masks=masks)
        assert np.array_equal(gt, gt_r)
        for t in per_t:
            np.testing.assert_allclose(per_t[t], per_t_r[t], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(auc, auc_r, rtol=1e-12)


def test_rec_and_hy_score_types():
    g = torch.Generator().manual_seed(2)
    x, xr = torch.randn(9, 2, 12, 17, generator=g), torch.randn(9, 2, 12, 17, generator=g)
    z, c = torch.randn(9, 16, generator=g), torch.randn(16, generator=g)
    for lt in ('rec', 'hyp', 'rec+hyp'):
        got = E.rec_and_hy_window_scores(x, xr, z, c, rec_loss_weight=0.2, loss_type=lt)
        want = RS.rec_and_hy_window_scores(x.numpy(), xr.numpy(), z.numpy(), c.numpy(), 0.2, lt)
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-5)
    z32, c32 = torch.randn(9, 32, generator=g), torch.randn(32, generator=g)           # a latent wider than the head kernel's 16
    for lt in ('hyp', 'rec+hyp'):
        got = E.rec_and_hy_window_scores(x, xr, z32, c32, rec_loss_weight=0.2, loss_type=lt)
        want = RS.rec_and_hy_window_scores(x.numpy(), xr.numpy(), z32.numpy(), c32.numpy(), 0.2, lt)
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-5)
    assert [E.eval_loss_type(w) for w in (0, 0.2, 1000)] == ['hyp', 'rec+hyp', 'rec']     # eval_COSKAD.py:58-66
