"""Per-layer A/B of the backward chain on one box: ms per train step of the bench workload with the data
kernel of layer i forming the batch reductions of layer i - 1 (the default) against layer i - 1 running its own statistics pass,
one pair at a time, and the bottleneck's chain (engine.FUSE_TOP) the same way; alternating, three repetitions each.
usage: python tools/ab_chain.py [steps]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd import engine
from coskad_amd.models.sts.ae import STSE
from coskad_amd.trainer import make_train_step
import bench
from coskad_amd.utils.synthetic import synthetic_clips

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
torch.manual_seed(0)
m = STSE(bench.C_IN, bench.CHANNELS, bench.HID, bench.LATENT, bench.T, bench.V, 'sts_gcn', 'linear', 'euclidean', 0.0)
m.c.fill_(0.1)
eng = make_train_step(m.cuda().train(), lr=1e-4, alpha=1e-6, head='euclidean')
x = synthetic_clips(4096, bench.C_IN, bench.T, bench.V, seed=1).cuda()
variants = [("all chains (default)", frozenset(), True), ("layer 4 does not carry layer 3", frozenset({3}), True),
            ("layer 3 does not carry layer 2", frozenset({2}), True), ("layer 2 does not carry layer 1", frozenset({1}), True),
            ("bottleneck does not carry layer 4", frozenset(), False)]
res = {v[0]: [] for v in variants}
for rep in range(3):
    for name, skip, top in variants:
        engine.CHAIN_SKIP, engine.FUSE_TOP = skip, top
        for _ in range(15):
            eng.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step(x)
        torch.cuda.synchronize()
        res[name].append(round((time.perf_counter() - t0) / steps * 1e3, 4))
for name, v in res.items():
    print(f"{name:40s} {v}  median {sorted(v)[1]}")
