"""A/B of an engine switch on one box: python tools/ab_engine_flag.py FLAG [steps]  -> ms per train step with FLAG on / off, alternating."""
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

flag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(0)
m = STSE(bench.C_IN, bench.CHANNELS, bench.HID, bench.LATENT, bench.T, bench.V, 'sts_gcn', 'linear', 'euclidean', 0.0)
m.c.fill_(0.1)
eng = make_train_step(m.cuda().train(), lr=1e-4, alpha=1e-6, head='euclidean')
x = synthetic_clips(4096, bench.C_IN, bench.T, bench.V, seed=1).cuda()
res = {True: [], False: []}
for rep in range(6):
    on = rep % 2 == 0
    setattr(engine, flag, on)
    for _ in range(20):
        eng.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(x)
    torch.cuda.synchronize()
    res[on].append((time.perf_counter() - t0) / steps * 1e3)
print(flag, "on:", [round(v, 4) for v in res[True]], "off:", [round(v, 4) for v in res[False]])
