"""Scratch timing of layer_apply at BASELINE shapes (not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coskad_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chans = [2, 32, 16, 32, 64]
T, V = 12, 17
torch.manual_seed(0)
tot = 0.0
for i in range(4):
    Ci, Co = chans[i], chans[i + 1]
    x = torch.randn(B, Ci, T, V, device="cuda")
    A = torch.randn(T, V, V, device="cuda") * 0.2
    Tm = torch.randn(V, T, T, device="cuda") * 0.2
    wf = torch.randn(2 * Ci, ops.cop(Co), device="cuda") * 0.1
    b = torch.randn(ops.cop(Co), device="cuda")
    sl = torch.full((1,), 0.25, device="cuda")
    out = torch.empty(B, Co, T, V, device="cuda")
    for _ in range(3):
        ops.layer_apply(x, A, Tm, wf, b, Co, in_slope=sl, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.layer_apply(x, A, Tm, wf, b, Co, in_slope=sl, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = B * (Ci + Co) * T * V * 4
    fl = B * (2 * Ci * V * T * T + 2 * Ci * T * V * V + 4 * Ci * Co * T * V)
    tot += ms
    print(f"L{i+1} {Ci}->{Co}: {ms*1e3:.1f} us  {byts/ms/1e6:.0f} GB/s  {fl/ms/1e9:.1f} TFLOP/s")
print(f"4-layer apply total {tot*1e3:.1f} us -> {B/tot/1e3:.2f} M clips/s")
