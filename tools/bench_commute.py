"""Times the commuted 32 -> 16 layer (csrc/commute_layer.hip) at B = 4096, 12 x 25: run under rocprofv3 --kernel-trace --stats for the per-kernel
split.  python tools/bench_commute.py [B]"""
import sys
import torch
from coskad_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
T, V = 12, 25
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=dev)
U_prev, dU = r(B, 32, T, V), r(B, 16, T, V)
slope = torch.tensor([0.25], device=dev)
Wt, Wr, A, Tm = r(16, 32) * 0.2, r(16, 32) * 0.2, r(T, V, V) * 0.3, r(V, T, T) * 0.3
gt, bet, gr, ber, bt, br = (r(16) for _ in range(6))
rm, rv = torch.zeros(16, device=dev), torch.ones(16, device=dev)
nbt = torch.zeros((), dtype=torch.int64, device=dev)
into = {"A": torch.empty_like(A), "T": torch.empty_like(Tm), "Wt": torch.empty(16, 32, device=dev), "Wr": torch.empty(16, 32, device=dev),
        "gt": torch.empty(16, device=dev), "bet": torch.empty(16, device=dev), "gr": torch.empty(16, device=dev),
        "ber": torch.empty(16, device=dev), "in_slope": torch.empty(1, device=dev)}


def fwd():
    return ops.commute_fwd(U_prev, slope, Wt, Wr, A, Tm, gt, bet, gr, ber, bt, br, rm, rv, rm.clone(), rv.clone(), nbt, nbt.clone(), 0.1, 1e-5)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


U, saved, _ = fwd()
print(f"B={B}  forward {timeit(fwd):.1f} us   backward {timeit(lambda: ops.commute_bwd(saved, dU, into)):.1f} us")
