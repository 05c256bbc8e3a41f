"""Times ops.layer_apply_z (training-mode apply from the stored Z) on the 25-joint layout for the default stack's layers 2-4 at B = 4096."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd import ops
B, T, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 12, 25
torch.manual_seed(0)
out_s = []
for Ci, Co in ((32, 64), (16, 32), (32, 16)):
    x = torch.randn(B, Ci, T, V, device="cuda")
    Z = torch.randn(B, Ci, T, V, device="cuda")
    A = torch.rand(T, V, V, device="cuda"); Tm = torch.rand(V, T, T, device="cuda")
    wfold = torch.randn(2 * Ci, ops.cop(Co), device="cuda") * 0.1
    bias = torch.randn(ops.cop(Co), device="cuda")
    sl = torch.tensor([0.25], device="cuda")
    out = torch.empty(B, Co, T, V, device="cuda")
    f = lambda: ops.layer_apply_z(Z, x, A, Tm, wfold, bias, Co, in_slope=sl, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    out_s.append(f"{Ci}->{Co}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
print(f"B={B} V=25 layer_apply_z: " + ", ".join(out_s))
