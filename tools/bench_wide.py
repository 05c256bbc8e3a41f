"""Throughput of the wide stack of the north_star wording (C = 2 -> 256: channels [64, 128, 256], h_dim 256, latent 16)
on one GPU: train step = forward + MSE-to-centre + backward + Adam through the module surface (autograd).  Layer 1 runs on
the fused kernels, the wider layers on the mixing kernels + the strided MFMA GEMM + csrc/wide.hip (ST_GCNN_layer.forward_wide).
Not the judged bench (bench.py keeps BASELINE.json's configs[1]); prints one JSON line.
usage: python tools/bench_wide.py [--batch 1024] [--steps 10]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from coskad_amd.models.sts.ae import STSE  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402  (initialisers / synthetic inputs only)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
T, V, chans, hid, L = 12, 17, [64, 128, 256], 256, 16
st = R.init_stse_state(2, tuple(chans), hid, L, T, V, seed=0)
st["c"] = torch.full((L,), 0.1)
m = STSE(2, chans, hid, L, T, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
m.load_state_dict(st, strict=True)
m.cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
x = R.synthetic_clips(a.batch, 2, T, V, seed=100).cuda()


def step():
    opt.zero_grad(set_to_none=True)
    z = m(x)
    loss = torch.nn.functional.mse_loss(z, m.c.expand_as(z))
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
cs = [2] + chans + [hid]
tv = T * V
fwd = sum(4 * tv * (ci + co) for ci, co in zip(cs[:-1], cs[1:])) + 4 * hid * tv
bwd = 2 * 4 * hid * tv + sum(4 * tv * (co + ci + (ci if i else 0)) for i, (ci, co) in enumerate(zip(cs[:-1], cs[1:])))
print(json.dumps({"workload": f"wide stack 2-64-128-256-256 latent 16, B={a.batch}, T=12, V=17, train step via autograd",
                  "clips_per_s": round(a.batch / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                  "algorithmic_bytes_per_clip": fwd + bwd,
                  "hbm_frac": round(a.batch / dt * (fwd + bwd) / 8e12, 4), "loss": round(float(loss), 6),
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
