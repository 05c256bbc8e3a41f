"""Is the eval-mode forward loop of bench.py GPU-bound or host-bound?  Times, at B = 4096 (default stack, linear projector):
  pipelined   : 20 x model(x) between two synchronisations (what bench.py's forward_only measures)
  host only   : the same 20 calls, clock stopped BEFORE the synchronisation (enqueue cost per call)
  graph       : the same forward captured once in a hipGraph and replayed (no Python between the launches)
usage: python tools/eval_overhead.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coskad_amd.models.sts.ae import STSE          # noqa: E402
from coskad_amd.utils.synthetic import synthetic_clips   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
model = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda().eval()
x = synthetic_clips(B, 2, 12, 17, seed=100).cuda()
sync = torch.cuda.synchronize
with torch.no_grad():
    for _ in range(10):
        model(x)
    sync()
    pipe, host = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            model(x)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        host.append((t1 - t0) / 20)
        pipe.append((t2 - t0) / 20)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        z = model(x)
    for _ in range(5):
        g.replay()
    sync()
    gr = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        sync()
        gr.append((time.perf_counter() - t0) / 20)
med = lambda v: sorted(v)[len(v) // 2]
print(f"B={B}  pipelined {med(pipe) * 1e6:.1f} us (min {min(pipe) * 1e6:.1f})   host-only enqueue {med(host) * 1e6:.1f} us (min {min(host) * 1e6:.1f})   "
      f"graph replay {med(gr) * 1e6:.1f} us (min {min(gr) * 1e6:.1f})")
