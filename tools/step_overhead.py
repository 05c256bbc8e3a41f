"""Is the train-step loop of bench.py GPU-bound or host-bound?  B = 4096 default stack: per step the wall time of a pipelined loop,
the host's enqueue time alone (clock stopped before the synchronisation), and a hipGraph replay of the same step.
usage: python tools/step_overhead.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                 # noqa: E402
from coskad_amd.models.sts.ae import STSE                     # noqa: E402
from coskad_amd.trainer import STSETrainStep                  # noqa: E402
from coskad_amd.utils.synthetic import synthetic_clips        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sync = torch.cuda.synchronize
med = lambda v: sorted(v)[len(v) // 2]
for graph in (False, True):
    torch.manual_seed(0)
    m = STSE(bench.C_IN, bench.CHANNELS, bench.HID, bench.LATENT, bench.T, bench.V, 'sts_gcn', 'linear', 'euclidean', 0.0)
    m.c.fill_(0.1)
    eng = STSETrainStep(m.cuda().train(), lr=1e-4, alpha=1e-6, head='euclidean', use_graph=graph)
    x = synthetic_clips(B, bench.C_IN, bench.T, bench.V, seed=1).cuda()
    for _ in range(20):
        eng.step(x)
    sync()
    pipe, host = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(50):
            eng.step(x)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        host.append((t1 - t0) / 50)
        pipe.append((t2 - t0) / 50)
    print(f"B={B} graph={graph}: pipelined {med(pipe) * 1e3:.4f} ms/step   host-only enqueue {med(host) * 1e3:.4f} ms/step")
