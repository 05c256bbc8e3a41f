"""ms per train step of the bench workload in THIS process (for A/B of build- or env-level switches: run it alternately)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd.models.sts.ae import STSE
from coskad_amd.trainer import make_train_step
import bench
from coskad_amd.utils.synthetic import synthetic_clips

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
m = STSE(bench.C_IN, bench.CHANNELS, bench.HID, bench.LATENT, bench.T, bench.V, 'sts_gcn', 'linear', 'euclidean', 0.0)
m.c.fill_(0.1)
eng = make_train_step(m.cuda().train(), lr=1e-4, alpha=1e-6, head='euclidean')
x = synthetic_clips(4096, bench.C_IN, bench.T, bench.V, seed=1).cuda()
out = []
for rep in range(3):
    for _ in range(20):
        eng.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(x)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / steps * 1e3, 4))
print(os.environ.get("AB_TAG", ""), out)
