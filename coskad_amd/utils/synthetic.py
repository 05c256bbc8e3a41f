"""Synthetic pose windows in the reference's batch format: x [N,2,T,V] f32, trans [N] i64,
meta [N,4] i64 = (scene, clip, person, start), frames [N,T] i32 (1-based), plus per-clip frame masks.

Normal motion = smooth low-frequency joint trajectories; anomalous segments add high-frequency jitter.
There is no network for the real datasets; this generator stands in for utils/dataset.py's loaders."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


def make_dataset(n_scenes=2, n_clips=3, n_persons=2, clip_len=120, T=12, V=17, num_transform=1, anomaly=True,
                 seed=0):
    rng = np.random.default_rng(seed)
    xs, trans, meta, frames = [], [], [], []
    gts: Dict[Tuple[int, int], np.ndarray] = {}
    t_axis = np.arange(clip_len)[:, None, None]
    for sc in range(1, n_scenes + 1):
        for cl in range(1, n_clips + 1):
            gt = np.zeros(clip_len, dtype=np.int64)
            if anomaly:
                a0 = int(rng.integers(20, clip_len - 50))
                gt[a0:a0 + 30] = 1
            gts[(sc, cl)] = gt
            for pe in range(n_persons):
                phase = rng.uniform(0, 2 * np.pi, size=(1, V, 2))
                amp = rng.uniform(0.2, 0.6, size=(1, V, 2))
                traj = amp * np.sin(2 * np.pi * t_axis / 40.0 + phase)            # [L,V,2]
                if anomaly and pe == 0:
                    traj = traj + gt[:, None, None] * rng.normal(0, 0.8, size=traj.shape)
                for s0 in range(0, clip_len - T + 1):
                    w = traj[s0:s0 + T].transpose(2, 0, 1)                        # [2,T,V]
                    for tr in range(num_transform):
                        ww = w if tr == 0 else w[::-1].copy()                     # tr 1: swap x/y (a cheap 'transform')
                        xs.append(ww)
                        trans.append(tr)
                        meta.append((sc, cl, pe, s0))
                        frames.append(np.arange(s0 + 1, s0 + T + 1))              # 1-based like the reference
    x = torch.from_numpy(np.ascontiguousarray(np.stack(xs), dtype=np.float32))
    return (x, torch.tensor(trans, dtype=torch.int64), torch.tensor(meta, dtype=torch.int64),
            torch.tensor(np.stack(frames), dtype=torch.int32)), gts


def batches(data, batch_size, shuffle=False, seed=0, rank=0, world=1, epoch=0):
    """Minimal DataLoader: yields [x, trans, meta, frames].  Sharding has DistributedSampler semantics: rank r takes
    items r, r+W, ... of the (per-epoch reshuffled: seed + epoch) index list, the tail wrap-padded so that EVERY rank
    gets ceil(n / W) items and therefore issues the same number of collectives per epoch."""
    from ..parallel import shard_indices
    x, trans, meta, frames = data
    n = x.shape[0]
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(seed + epoch)) if shuffle else torch.arange(n)
    idx = idx[shard_indices(n, rank, world)]
    for i in range(0, idx.numel(), batch_size):
        j = idx[i:i + batch_size]
        yield [x[j], trans[j], meta[j], frames[j]]


def synthetic_clips(B: int, C: int = 2, T: int = 12, V: int = 17, seed: int = 0) -> torch.Tensor:
    """SURVEY 8d benchmark input: 0.5 * N(0,1) clipped to +-3 (RobustScaler-ed, bbox-centred poses are O(1)) with 2 %
    exact zeros marking missing joints."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, C, T, V, generator=g) * 0.5).clamp(-3, 3)
    mask = torch.rand(B, 1, T, V, generator=g) < 0.02
    return torch.where(mask, torch.zeros(()), x).contiguous()
