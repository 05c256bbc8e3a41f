"""Kernel-by-kernel timeline of the eval-mode encoder forward (STSE) at B = 4096: run under rocprofv3 --kernel-trace; python tools/eval_timeline.py [V]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coskad_amd.models.sts.ae import STSE
from oracle import ref_cpu as R   # synthetic clips only

V = int(sys.argv[1]) if len(sys.argv) > 1 else 25
m = STSE(2, [32, 16, 32], 64, 16, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda().eval()
x = R.synthetic_clips(4096, 2, 12, V, seed=1).cuda()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m(x)
    e1.record()
    torch.cuda.synchronize()
print(f"V={V} eval forward {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
