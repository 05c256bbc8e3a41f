"""Per-phase wall-clock of the fused layer backward (needs a library built with -DFB_TIMING: tools/ab_fused.sh build t -DFB_TIMING).
Prints, per layer shape, the mean over blocks of wave 0's per-phase microseconds summed over its clips."""
import sys

import torch

sys.path.insert(0, ".")
from coskad_amd import ops

B, T, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 12, 17
names = ["stage", "temporal", "pass1(dZ)", "dA", "pass2(dXres)", "spatial^T", "dT", "temporal^T", "last-epi", "rowpass", "below-stats", "", "", "", "", "loop-head"]
CHAIN = len(sys.argv) > 2 and sys.argv[2] == "chain"
BELOW = {(32, 64): 16, (16, 32): 32, (32, 16): 2}
torch.manual_seed(0)
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
    below = None
    if CHAIN:
        cb = BELOW[(Ci, Co)]
        rows = ops.layer_bwd_below_rows(B, Ci, Co, cb, T, V)
        below = (torch.randn(B, cb, T, V, device=dev), torch.randn(B, cb, T, V, device=dev), slope if cb > 2 else None,
                 torch.empty(ops.layer_bwd_below_floats(B, Ci, Co, cb, T, V), device=dev))
    for _ in range(3):
        ops.layer_bwd(x, dU, A, Tm, slope, stat, Wt, gt, Wr, gr, g, bws, dIn=dIn, Z=Z, below=below)
    torch.cuda.synchronize()
    t = dIn.flatten()[:256 * 16].view(256, 16).double().mean(0) / 100.0     # 100 MHz ticks -> us
    per = dIn.flatten()[:256 * 16].view(256, 16).double().sum(1) / 100.0
    print(f"   wave-0 totals over blocks: min {float(per.min()):.0f} mean {float(per.mean()):.0f} max {float(per.max()):.0f} us")
    tot = float(t.sum())
    print(f"{Ci}->{Co}: total {tot:.0f} us/wave: " + ", ".join(f"{n} {float(v):.1f}" for n, v in zip(names, t) if n))
