import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from coskad_amd.models.sts.vae import STSVAE
from coskad_amd.trainer import STSAETrainStep
from coskad_amd.utils.synthetic import synthetic_clips
torch.manual_seed(0)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 25
m = STSVAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'mlp', 'euclidean', 0.0, distribution='ps')
eng = STSAETrainStep(m.cuda().train(), mode='vae', lr=1e-4, alpha=1e-6, phi=1.0, beta=1e-3, gamma=1e-2)
x = synthetic_clips(4096, 2, 12, V, seed=1).cuda()
for _ in range(10): eng.step(x)
torch.cuda.synchronize()
pipe, host = [], []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(20): eng.step(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) / 20); pipe.append((t2 - t0) / 20)
print(f"VAE step: pipelined {sorted(pipe)[2]*1e3:.3f} ms, host-only enqueue {sorted(host)[2]*1e3:.3f} ms")
