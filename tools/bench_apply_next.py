"""Times ops.layer_apply_next (apply + next-layer statistics, csrc/fused_apply_next.hip) for the default stack's layers 1-3 at B = 4096."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd import ops
B, T, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 12, 17
torch.manual_seed(0)
out_s = []
for Ci, Co in ((2, 32), (32, 16), (16, 32)):
    x = torch.randn(B, Ci, T, V, device="cuda")
    Z = torch.randn(B, Ci, T, V, device="cuda")
    A = torch.rand(T, V, V, device="cuda"); Tm = torch.rand(V, T, T, device="cuda")
    wfold = torch.randn(2 * Ci, ops.cop(Co), device="cuda") * 0.1
    bias = torch.randn(ops.cop(Co), device="cuda")
    sl = torch.tensor([0.25], device="cuda")
    ftab = torch.empty(ops.ftab_floats(), device="cuda")
    ops.build_ftabs([A], [Tm], [ftab])
    out = torch.empty(B, Co, T, V, device="cuda"); zn = torch.empty(B, Co, T, V, device="cuda")
    part = torch.empty(ops.layer_apply_next_rows(B, Ci, Co) * 2 * (Co * Co + Co), device="cuda")
    f = lambda: ops.layer_apply_next(Z, x, wfold, bias, Co, sl if Ci > 2 else None, sl, ftab, part, T, V, out=out, Z_next=zn)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    out_s.append(f"{Ci}->{Co}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
print(f"B={B} layer_apply_next: " + ", ".join(out_s))
