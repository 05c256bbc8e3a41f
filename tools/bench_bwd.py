"""Times ops.layer_bwd (stored-Z path) for the four layers of the default stack at B = 4096 on the current GPU."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd import ops

B, T, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 12, 17
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for Ci, Co in ((32, 64), (16, 32), (32, 16)):
    dev = "cuda"
    x = torch.randn(B, Ci, T, V, device=dev) * 0.5
    A = (torch.rand(T, V, V, device=dev) - 0.5) * 0.5
    Tm = (torch.rand(V, T, T, device=dev) - 0.5) * 0.5
    slope = torch.tensor([0.25], device=dev)
    Wt, Wr = torch.randn(Co, Ci, device=dev) * 0.1, torch.randn(Co, Ci, device=dev) * 0.1
    gt, bet, gr, ber = (torch.rand(Co, device=dev) + 0.5 for _ in range(4))
    bt, br = torch.zeros(Co, device=dev), torch.zeros(Co, device=dev)
    rm = [torch.zeros(Co, device=dev) for _ in range(2)]
    rv = [torch.ones(Co, device=dev) for _ in range(2)]
    nbt = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(2)]
    ws = torch.empty(ops.train_stats_ws_bytes(Ci), dtype=torch.uint8, device=dev)
    Z = torch.empty_like(x)
    wfold, bias, stat = ops.layer_train_stats(x, A, Tm, slope, Wt, bt, gt, bet, rm[0], rv[0], nbt[0], Wr, br, gr, ber, rm[1], rv[1], nbt[1], ws, Z=Z)
    dU = torch.randn(B, Co, T, V, device=dev) * 0.1
    g = {"A": torch.empty_like(A), "T": torch.empty_like(Tm), "Wt": torch.empty_like(Wt), "bt": torch.empty_like(bt),
         "gt": torch.empty_like(gt), "bet": torch.empty_like(bet), "Wr": torch.empty_like(Wr), "br": torch.empty_like(br),
         "gr": torch.empty_like(gr), "ber": torch.empty_like(ber), "slope_in": torch.empty(1, device=dev)}
    bws = torch.empty(ops.layer_bwd_ws_bytes(B, Ci, Co, T, V), dtype=torch.uint8, device=dev)
    dIn = torch.empty_like(x)
    t = timeit(lambda: ops.layer_bwd(x, dU, A, Tm, slope, stat, Wt, gt, Wr, gr, g, bws, dIn=dIn, Z=Z))
    out.append(f"{Ci}->{Co}: {t:.0f} us")
print(f"B={B} layer_bwd (all stages): " + ", ".join(out))
