"""Data pipeline of the reference's `normalization_strategy: 'robust'` path, restated array-wise.

Reference flow (utils/dataset.py:204-327 -> utils/get_robust_data.py:24-185 -> utils/data.py, utils/preprocessing.py,
utils/dataset_utils.py):  Morais-format CSV trajectories `<root>/{training,validating,testing}/trajectories/<SS-CCCC>/
<PPPP>.csv` (rows `frame, x1, y1, ..., x17, y17`, zeros = missing joint) -> drop trajectories shorter than a window ->
bounding-box-centred coordinates per frame -> sliding windows -> RobustScaler (quantiles 10/90, zeros ignored, fitted on
the train split and pickled as `<exp_dir>/local_robust.pickle`) -> [N, 3, T, V] (x, y, 1) -> optional 18-keypoint /
headless layouts -> one of 5 affine `PoseTransform`s per item.

MI355X-first design: the window table is small next to 288 GB of HBM (one million windows = 1.6 GB), so `DeviceLoader`
keeps it RESIDENT on the GPU and forms each batch with one gather + affine kernel (`coskad_gather_transform_f32`):
no per-step host->device traffic, no DataLoader workers.  The numpy `__getitem__` of the reference surface is kept for
parity tests and for CPU-side consumers.

Pinned by tests/golden/data_pipeline.npz (generated from the reference by oracle/make_golden_data.py).
"""
from __future__ import annotations

import math
import os
import pickle
from collections import OrderedDict
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

N_KP = 17


# ---- trajectories -------------------------------------------------------------------------------------------
def load_trajectories(trajectories_path: str, debug: bool = False, split: str = 'train') -> "OrderedDict[str, Tuple[np.ndarray, np.ndarray]]":
    """utils/data.py:218-235.  id `<SS-CCCC>_<PPPP>` -> (frames int32 [n], coordinates float32 [n, 34]).
    Folders and files are visited in sorted order (the reference uses os.listdir order, which is unspecified)."""
    out: "OrderedDict[str, Tuple[np.ndarray, np.ndarray]]" = OrderedDict()
    folders = sorted(os.listdir(trajectories_path))
    if debug and split == 'train':
        folders = folders[:5]
    for folder in folders:
        for name in sorted(os.listdir(os.path.join(trajectories_path, folder))):
            arr = np.loadtxt(os.path.join(trajectories_path, folder, name), dtype=np.float32, delimiter=',', ndmin=2)
            out[folder + '_' + name.split('.')[0]] = (arr[:, 0].astype(np.int32), arr[:, 1:])
    return out


def bbox_centre_coordinates(coords: np.ndarray, video_resolution: Sequence[float]) -> np.ndarray:
    """All frames of a trajectory at once: utils/data.py:9-43 (`compute_bounding_box`: min/max over the non-zero x
    resp. y, 10 % margin, clipped to the frame, rounded half-to-even) and :159-181 (`_from_image_to_centre_bounding_box`:
    missing joints -> centre, subtract the centre, divide by the box size; frames without any joint stay zero)."""
    c = np.asarray(coords, dtype=np.float32)
    n = c.shape[0]
    width, height = np.asarray(video_resolution, dtype=np.float32)
    x, y = c[:, 0::2], c[:, 1::2]
    xm, ym = x != 0, y != 0
    some = xm.any(1) | ym.any(1)                       # `any(kps)`
    boxed = xm.any(1) & ym.any(1)                      # else np.min of an empty array -> the (0, 0, 0, 0) box
    inf = np.float32(np.inf)
    left = np.where(xm, x, inf).min(1)
    right = np.where(xm, x, -inf).max(1)
    top = np.where(ym, y, inf).min(1)
    bottom = np.where(ym, y, -inf).max(1)
    with np.errstate(invalid='ignore', over='ignore'):
        ew = np.float32(0.1) * (right - left + np.float32(1))
        eh = np.float32(0.1) * (bottom - top + np.float32(1))
        l_ = np.rint(np.clip(left - ew, np.float32(0), width - np.float32(1)))
        r_ = np.rint(np.clip(right + ew, np.float32(0), width - np.float32(1)))
        t_ = np.rint(np.clip(top - eh, np.float32(0), height - np.float32(1)))
        b_ = np.rint(np.clip(bottom + eh, np.float32(0), height - np.float32(1)))
    l_, r_, t_, b_ = (np.where(boxed, v, 0).astype(np.float32) for v in (l_, r_, t_, b_))
    cx, cy = (l_ + r_) / np.float32(2), (t_ + b_) / np.float32(2)      # integers / 2: exact in float32
    w, h = r_ - l_, b_ - t_
    xs = np.where(xm, x, cx[:, None]) - cx[:, None]
    ys = np.where(ym, y, cy[:, None]) - cy[:, None]
    with np.errstate(divide='ignore', invalid='ignore'):
        xs = np.where(w[:, None] != 0, xs / w[:, None], np.float32(0))
        ys = np.where(h[:, None] != 0, ys / h[:, None], np.float32(0))
    out = np.empty((n, 2 * x.shape[1]), dtype=np.float32)
    out[:, 0::2], out[:, 1::2] = xs, ys
    out[~some] = c[~some]
    return out


def build_windows(trajs: "OrderedDict[str, Tuple[np.ndarray, np.ndarray]]", seg_len: int, gap: int):
    """utils/preprocessing.py:244-313: every start, `seg_len` frames `gap + 1` apart.
    -> X [N, seg_len, D] f32, meta [N, 4] (scene, clip, person, first frame id), ids [N, seg_len] (frame ids)."""
    total = seg_len + gap * (seg_len - 1)
    Xs, metas, ids = [], [], []
    for tid, (frames, coords) in trajs.items():
        n = len(frames)
        starts = np.arange(0, n - total + 1)
        if starts.size == 0:
            continue
        idx = starts[:, None] + (gap + 1) * np.arange(seg_len)[None, :]
        Xs.append(coords[idx])
        scene, clip = (int(v) for v in tid.split('_')[0].split('-'))
        person = int(tid.split('_')[1])
        metas.append(np.stack([np.full(starts.size, scene), np.full(starts.size, clip), np.full(starts.size, person),
                               frames[starts]], 1).astype(np.int64))
        ids.append(frames[idx].astype(np.int64))
    if not Xs:
        raise ValueError("no trajectory is long enough for one window")
    return np.concatenate(Xs, 0), np.concatenate(metas, 0), np.concatenate(ids, 0)


def scale_robust(X: np.ndarray, scaler=None):
    """utils/data.py:361-370: zeros are missing values; RobustScaler(quantile_range=(10, 90)); missing -> 0."""
    from sklearn.preprocessing import RobustScaler
    shape = X.shape
    Xn = np.where(X == 0.0, np.nan, X).reshape(-1, shape[-1])
    if scaler is None:
        scaler = RobustScaler(quantile_range=(10.0, 90.0))
        scaler.fit(Xn)
    Xs = scaler.transform(Xn)
    return np.where(np.isnan(Xs), 0.0, Xs).reshape(shape), scaler


def keypoints17_to_coco18(kps: np.ndarray) -> np.ndarray:
    """utils/dataset_utils.py:7-19: neck = mean of the shoulders (5, 6), then the OpenPose order."""
    neck = 0.5 * (kps[..., 5, :] + kps[..., 6, :])
    kp = np.concatenate([kps, neck[..., None, :]], axis=-2)
    return kp[..., np.array([0, 17, 6, 8, 10, 5, 7, 9, 12, 14, 16, 11, 13, 15, 2, 1, 4, 3]), :]


# ---- the 5 affine transforms (utils/dataset_utils.py:255-310) ---------------------------------------------------
def aff_trans_mat(sx=1.0, sy=1.0, tx=0.0, ty=0.0, rot=0.0, flip=False) -> np.ndarray:
    """flip . rot . [scale | translate]  as float32 3x3 (each factor rounded to float32 first, like the torch original)."""
    c, s = math.cos(math.radians(rot)), math.sin(math.radians(rot))
    f = np.eye(3, dtype=np.float32)
    if flip:
        f[0, 0] = -1.0
    ts = np.array([[sx, 0, tx], [0, sy, ty], [0, 0, 1]], dtype=np.float32)
    r = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    return (f @ (r @ ts)).astype(np.float32)


AE_TRANS_MATS = np.stack([aff_trans_mat(rot=0, flip=False), aff_trans_mat(rot=0, flip=True),
                          aff_trans_mat(rot=90, flip=False), aff_trans_mat(rot=90, flip=True),
                          aff_trans_mat(rot=45, flip=False)])


def apply_pose_transform(pose: np.ndarray, mat: np.ndarray) -> np.ndarray:
    """pose [3, T, V] = (x, y, conf): transform (x, y, 1), keep conf (utils/dataset_utils.py:272-286)."""
    hom = np.concatenate([pose[:2], np.ones_like(pose[2:3])], 0)
    out = np.einsum('ktv,ck->ctv', hom, mat)
    return np.concatenate([out[:2], pose[2:3]], 0)


# ---- dataset ----------------------------------------------------------------------------------------------------------
class PoseDatasetRobust:
    """Array-wise restatement of utils/dataset.py::PoseDatasetRobust (local features only: `include_global=False`).
    Attributes follow the reference: segs_data_np [N, 3, T, V] f32, segs_meta [N, 4], segs_ids [N, T], num_samples,
    num_transform; item `i` = transform `i // N` of window `i % N` -> [data [num_coords, T, V], trans, meta, ids]."""

    def __init__(self, path_to_robust_data: str, split: str = 'train', exp_dir: str = '', num_transform: int = 5,
                 seg_len: int = 12, seg_stride: int = 1, vid_res: Sequence[int] = (1080, 720), num_coords: int = 2,
                 normalize_pose: bool = True, kp18_format: bool = False, headless: bool = False, debug: bool = False,
                 scaler=None) -> None:
        sub = 'training' if 'train' in split else ('testing' if 'test' in split else 'validating')
        self.split, self.num_coords, self.seg_len = split, num_coords, seg_len
        self.num_transform = max(1, int(num_transform))
        self.trans_mats = AE_TRANS_MATS[:self.num_transform] if num_transform > 0 else AE_TRANS_MATS[:1]
        gap = seg_stride - 1
        trajs = load_trajectories(os.path.join(path_to_robust_data, sub, 'trajectories'), debug=debug, split=split)
        total = seg_len + gap * (seg_len - 1)
        trajs = OrderedDict((k, v) for k, v in trajs.items() if len(v[0]) >= total)          # remove_short_trajectories
        res = np.asarray(vid_res, dtype=np.float32)
        trajs = OrderedDict((k, (f, bbox_centre_coordinates(c, res))) for k, (f, c) in trajs.items())
        X, meta, ids = build_windows(trajs, seg_len, gap)
        self.scaler = scaler
        if normalize_pose:
            path = os.path.join(exp_dir, 'local_robust.pickle')
            if self.scaler is None:
                if split == 'train':
                    _, self.scaler = scale_robust(np.vstack([c for _, c in trajs.values()]))
                    if exp_dir:
                        os.makedirs(exp_dir, exist_ok=True)
                        with open(path, 'wb') as f:
                            pickle.dump(self.scaler, f)
                else:
                    with open(path, 'rb') as f:
                        self.scaler = pickle.load(f)
            X, _ = scale_robust(X, self.scaler)
        kp = X.reshape(*X.shape[:2], N_KP, 2)
        data = np.empty((*kp.shape[:-1], 3))
        data[..., :2], data[..., 2] = kp, 1.0
        if kp18_format:
            data = keypoints17_to_coco18(data)
        if headless:
            data = data[:, :, :14]
        self.segs_data_np = np.transpose(data, (0, 3, 1, 2)).astype(np.float32)
        self.segs_meta, self.segs_ids = meta, ids
        self.metadata = self.segs_meta
        self.num_samples, self.C, self.T, self.V = self.segs_data_np.shape

    def __len__(self) -> int:
        return self.num_transform * self.num_samples

    def __getitem__(self, index: int):
        s, t = index % self.num_samples, index // self.num_samples
        data = apply_pose_transform(self.segs_data_np[s], self.trans_mats[t])[:self.num_coords]
        return [data, t, self.segs_meta[s], self.segs_ids[s]]

    def to_device(self, device="cuda") -> "DeviceWindows":
        return DeviceWindows(self, device)


class DeviceWindows:
    """The window table resident in HBM: xy [N, 2, T*V] f32 (conf is the constant 1 of the robust path) and the
    transform matrices; `gather(index)` = one HIP kernel producing the batch [B, 2, T, V] for dataset indices."""

    def __init__(self, ds: PoseDatasetRobust, device="cuda") -> None:
        if ds.num_coords != 2:
            raise ValueError("the device loader serves the (x, y) layout (num_coords = 2)")
        self.ds = ds
        self.N, self.T, self.V = ds.num_samples, ds.T, ds.V
        self.xy = torch.from_numpy(np.ascontiguousarray(ds.segs_data_np[:, :2])).to(device).reshape(self.N, 2, self.T * self.V).contiguous()
        self.mats = torch.from_numpy(np.ascontiguousarray(ds.trans_mats)).to(device)
        self.meta = torch.from_numpy(ds.segs_meta.astype(np.int64))
        self.frames = torch.from_numpy(ds.segs_ids.astype(np.int32))

    def gather(self, index: torch.Tensor) -> torch.Tensor:
        from .. import ops
        return ops.gather_transform(self.xy, self.mats, index.to(self.xy.device, non_blocking=True), self.T, self.V)


class DeviceLoader:
    """Batches `[x (device), trans_idx, meta, frames]` like the reference's DataLoader over PoseDatasetRobust; rank r of
    W takes items r, r+W, ... of the (optionally shuffled) index list with the tail wrap-padded to ceil(n / W) items per
    rank, as DistributedSampler does under Lightning DDP: every rank runs the same number of batches (and collectives)."""

    def __init__(self, windows: DeviceWindows, batch_size: int, shuffle: bool = False, seed: int = 0, rank: int = 0,
                 world: int = 1) -> None:
        self.w, self.batch_size, self.shuffle, self.seed, self.rank, self.world = windows, batch_size, shuffle, seed, rank, world
        self.epoch = 0

    def __len__(self) -> int:
        n = (len(self.w.ds) + self.world - 1) // self.world
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[List[torch.Tensor]]:
        n = len(self.w.ds)
        if self.shuffle:
            idx = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch))
            self.epoch += 1
        else:
            idx = torch.arange(n)
        from ..parallel import shard_indices
        idx = idx[shard_indices(n, self.rank, self.world)]
        for i in range(0, idx.numel(), self.batch_size):
            j = idx[i:i + self.batch_size]
            s = j % self.w.N
            yield [self.w.gather(j), j // self.w.N, self.w.meta[s], self.w.frames[s]]


def get_dataset_and_loader(args, split: str = 'train', validation: bool = False, rank: int = 0, world: int = 1,
                           device="cuda"):
    """utils/dataset.py:284-327 for `normalization_strategy == 'robust'`; `args` carries the reference's dataset_* keys
    (with or without the prefix).  Returns (dataset, loader[, val_dataset, val_loader])."""
    def g(name, default=None):
        return getattr(args, 'dataset_' + name, getattr(args, name, default))
    common = dict(exp_dir=g('exp_dir', getattr(args, 'ckpt_dir', '')), num_transform=g('num_transform', 5),
                  seg_len=g('seg_len', 12), vid_res=g('vid_res', [1080, 720]), num_coords=getattr(args, 'num_coords', 2),
                  normalize_pose=g('normalize_pose', True), kp18_format=g('kp18_format', False),
                  headless=g('headless', False), debug=getattr(args, 'debug', False))
    root = g('path_to_robust', getattr(args, 'data_dir', ''))
    bs = g('batch_size', 2048)
    ds = PoseDatasetRobust(root, split=split, seg_stride=g('seg_stride', 1) if split == 'train' else 1, **common)
    loader = DeviceLoader(ds.to_device(device), bs, shuffle=(split == 'train'), seed=getattr(args, 'seed', 0), rank=rank, world=world)
    if not validation:
        return ds, loader
    vds = PoseDatasetRobust(root, split='validation', seg_stride=1, scaler=ds.scaler, **common)
    return ds, loader, vds, DeviceLoader(vds.to_device(device), bs, rank=rank, world=world)
