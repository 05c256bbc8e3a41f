"""Data pipeline against tests/golden/data_pipeline.npz, which oracle/make_golden_data.py generated
by running the reference's utils/dataset.py::PoseDatasetRobust on the synthetic Morais-format CSV tree stored in the
fixture.  CPU tests: CSV -> windows -> scaler -> items.  GPU test: the HBM-resident loader's gather + affine kernel."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    g = np.load(GOLD)
    root = tmp_path_factory.mktemp("morais")
    for k in g.files:
        if not k.startswith("csv."):
            continue
        _, split, folder, person = k.split(".")
        d = root / ("training" if split == "train" else "testing") / "trajectories" / folder
        d.mkdir(parents=True, exist_ok=True)
        np.savetxt(d / f"{person}.csv", g[k], delimiter=",", fmt="%.6f")
    return g, str(root), str(tmp_path_factory.mktemp("exp"))


def _order(meta):
    return np.lexsort((meta[:, 3], meta[:, 2], meta[:, 1], meta[:, 0]))


def _datasets(root, exp):
    from coskad_amd.utils.dataset import PoseDatasetRobust
    kw = dict(exp_dir=exp, num_transform=5, seg_len=12, vid_res=[1080, 720], num_coords=2)
    tr = PoseDatasetRobust(root, split="train", seg_stride=2, **kw)
    te = PoseDatasetRobust(root, split="test", seg_stride=1, **kw)
    return tr, te


def test_windows_scaler_and_meta_match_reference(tree):
    g, root, exp = tree
    tr, te = _datasets(root, exp)
    np.testing.assert_allclose(tr.scaler.center_, g["scaler.center"], rtol=0, atol=0)
    np.testing.assert_allclose(tr.scaler.scale_, g["scaler.scale"], rtol=0, atol=0)
    for name, ds in (("train", tr), ("test", te)):
        ref_meta, ref_ids, ref_data = g[f"{name}.meta"], g[f"{name}.ids"], g[f"{name}.data"]
        assert len(ds) == int(g[f"{name}.len"]) and ds.segs_data_np.shape == ref_data.shape
        a, b = _order(ds.segs_meta), _order(ref_meta)            # the reference visits folders in os.listdir order
        np.testing.assert_array_equal(ds.segs_meta[a], ref_meta[b])
        np.testing.assert_array_equal(ds.segs_ids[a], ref_ids[b])
        np.testing.assert_array_equal(ds.segs_data_np[a], ref_data[b])      # bit-exact float32 windows
        assert ds.segs_data_np.dtype == np.float32


def test_items_and_transforms_match_reference(tree):
    g, root, exp = tree
    from coskad_amd.utils.dataset import AE_TRANS_MATS
    np.testing.assert_array_equal(AE_TRANS_MATS, g["trans_mats"])
    tr, te = _datasets(root, exp)
    for name, ds in (("train", tr), ("test", te)):
        n = ds.num_samples
        for k, ref_index in enumerate(g[f"{name}.item_index"]):
            t = int(ref_index) // n
            assert t == int(g[f"{name}.item_trans"][k])
            # same window, located by its metadata (window order may differ, see above)
            m = g[f"{name}.item_meta"][k]
            s = int(np.flatnonzero((ds.segs_meta == m).all(1))[0])
            data, tt, meta, ids = ds[t * n + s]
            assert tt == t
            np.testing.assert_array_equal(meta, m)
            np.testing.assert_array_equal(ids, g[f"{name}.item_ids"][k])
            np.testing.assert_allclose(data, g[f"{name}.item_data"][k], rtol=0, atol=1e-7)
            assert data.shape == (2, 12, 17)


def test_edge_cases():
    from coskad_amd.utils.dataset import bbox_centre_coordinates, build_windows, keypoints17_to_coco18
    from collections import OrderedDict
    c = np.zeros((3, 34), np.float32)
    c[1, 0::2] = np.linspace(100, 200, 17)           # a frame with x but no y at all: the reference's (0,0,0,0) box
    c[2, :] = np.tile([300.0, 400.0], 17)            # degenerate box (all joints on one point)
    out = bbox_centre_coordinates(c, [1080, 720])
    assert not out[0].any()                          # empty frame stays empty
    assert not out[1].any() and np.isfinite(out).all()
    assert np.abs(out[2]).max() < 1.0
    # ragged: exactly one window / too short
    tr = OrderedDict([("01-0001_0001", (np.arange(5, 17, dtype=np.int32), np.ones((12, 34), np.float32))),
                      ("01-0001_0002", (np.arange(3, dtype=np.int32), np.ones((3, 34), np.float32)))])
    X, meta, ids = build_windows(tr, 12, 0)
    assert X.shape == (1, 12, 34) and meta.tolist() == [[1, 1, 1, 5]] and ids[0].tolist() == list(range(5, 17))
    with pytest.raises(ValueError):
        build_windows(OrderedDict([("01-0001_0002", tr["01-0001_0002"])]), 12, 0)
    k = keypoints17_to_coco18(np.arange(17 * 2, dtype=np.float64).reshape(17, 2))
    assert k.shape == (18, 2) and k[1].tolist() == [11.0, 12.0]          # neck = mean of joints 5 and 6


@pytest.mark.gpu
def test_device_loader_matches_items(tree):
    g, root, exp = tree
    from coskad_amd.utils.dataset import DeviceLoader
    tr, _ = _datasets(root, exp)
    w = tr.to_device("cuda")
    index = torch.arange(len(tr))
    x = w.gather(index).cpu().numpy()
    ref = np.stack([tr[i][0] for i in range(len(tr))])
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-7)
    # loader: every item covered over the ranks, metadata aligned with the data
    seen = []
    for rank in range(2):
        for xb, tb, mb, fb in DeviceLoader(w, 16, shuffle=True, seed=3, rank=rank, world=2):
            assert xb.is_cuda and xb.shape[1:] == (2, 12, 17)
            for k in range(xb.shape[0]):
                s = int(np.flatnonzero((tr.segs_meta == mb[k].numpy()).all(1))[0])
                seen.append(int(tb[k]) * tr.num_samples + s)
                np.testing.assert_allclose(xb[k].cpu().numpy(), tr[seen[-1]][0], rtol=0, atol=1e-7)
                np.testing.assert_array_equal(fb[k].numpy(), tr.segs_ids[s])
    # DistributedSampler semantics: equal shards, the tail wrap-padded (an odd item count repeats one item)
    assert sorted(set(seen)) == list(range(len(tr)))
    assert len(seen) == 2 * ((len(tr) + 1) // 2)
    # out-of-range indices give zero clips instead of reading out of bounds
    bad = w.gather(torch.tensor([len(tr) + 5, -1]))
    assert float(bad.abs().max()) == 0.0


@pytest.mark.gpu
def test_fit_and_score_from_morais_tree(tree, tmp_path):
    """CSV tree -> HBM-resident loaders -> LitEncoder/Trainer -> per-frame scores and AUC (the train_COSKAD.py flow)."""
    import shutil
    from argparse import Namespace
    from coskad_amd.lit import LitEncoder, Trainer
    from coskad_amd.utils.dataset import get_dataset_and_loader
    g, root, _ = tree
    if not os.path.isdir(os.path.join(root, "validating")):
        shutil.copytree(os.path.join(root, "testing"), os.path.join(root, "validating"))
    args = Namespace(num_coords=2, h_dim=16, latent_dim=8, dataset_seg_len=12, dropout=0, channels=[16, 8, 16],
                     projector="linear", encoder_type="STS_GCN", hyperbolic=False, static_center=False,
                     center_tolerance=1e-3, opt_lr=2e-3, alpha=1e-6, dataset_batch_size=32, dataset_num_transform=5,
                     dataset_headless=False, dataset_kp18_format=False, smoothing=5, dataset_choice="UBnormal",
                     validation=True, dataset_path_to_robust=root, dataset_seg_stride=2, dataset_vid_res=[1080, 720],
                     dataset_normalize_pose=True, dataset_exp_dir=str(tmp_path), seed=0, debug=False)
    _, train_loader, vds, val_loader = get_dataset_and_loader(args, split="train", validation=True)
    assert os.path.exists(tmp_path / "local_robust.pickle")
    lit = LitEncoder(args).cuda()
    # frame masks per (scene, clip) of the validation split, long enough for every frame id; a few frames anomalous
    gts = {}
    for sc, cl in {(int(m[0]), int(m[1])) for m in vds.segs_meta}:
        n = int(vds.segs_ids[(vds.segs_meta[:, 0] == sc) & (vds.segs_meta[:, 1] == cl)].max()) + 1
        gt = np.zeros(n, dtype=np.int64)
        gt[n // 2: n // 2 + 5] = 1
        gts[(sc, cl)] = gt
    lit.gts = gts
    tr = Trainer(max_epochs=2, ckpt_dir=str(tmp_path / "ck"))
    tr.fit(lit, lambda: train_loader, lambda: val_loader)
    assert len(tr.history) == 2 and 0.0 <= tr.history[-1]["validation_auc"] <= 1.0
    assert np.isfinite(tr.history[-1].get("loss", 0.0))
