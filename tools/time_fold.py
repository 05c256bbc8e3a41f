"""Phase timing of k_bwd_fold (library built with -DCOSKAD_FOLD_TIMING): runs train steps and prints the stamps of the LAST
fold launch of the step (layer 1's) and, with --layer4, of a single layer-4 backward call."""
import ctypes, sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd import _lib, ops
B, T, V = 4096, 12, 17
torch.manual_seed(0)
Ci, Co = 32, 64
x = torch.randn(B, Ci, T, V, device="cuda"); dU = torch.randn(B, Co, T, V, device="cuda")
A = torch.rand(T, V, V, device="cuda") - 0.5; Tm = torch.rand(V, T, T, device="cuda") - 0.5
sl = torch.tensor([0.25], device="cuda")
Wt, Wr = torch.randn(Co, Ci, device="cuda") * 0.2, torch.randn(Co, Ci, device="cuda") * 0.2
one, zero = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
nb = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
ws = torch.empty(ops.train_stats_ws_bytes(Ci), dtype=torch.uint8, device="cuda")
Z = torch.empty_like(x)
for _ in range(3):
    ops.layer_train_stats(x, A, Tm, sl, Wt, zero.clone(), one.clone(), zero.clone(), zero.clone(), one.clone(), nb[0],
                          Wr, zero.clone(), one.clone(), zero.clone(), zero.clone(), one.clone(), nb[1], ws, Z=Z)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 16)()
_lib.lib().coskad_debug_tfold_stamps(out)
st = list(out)[:5]
print("k_train_fold 32->64 phases (us):", [round((b - a) / 100.0, 2) for a, b in zip(st[:-1], st[1:])], "total", round((st[4] - st[0]) / 100.0, 2))
wfold, bias, stat = ops.layer_train_stats(x, A, Tm, sl, Wt, zero.clone(), one.clone(), zero.clone(), zero.clone(), one.clone(), nb[0],
                                          Wr, zero.clone(), one.clone(), zero.clone(), zero.clone(), one.clone(), nb[1], ws, Z=Z)
g = {"A": torch.empty_like(A), "T": torch.empty_like(Tm), "Wt": torch.empty_like(Wt), "bt": torch.empty(Co, device="cuda"),
     "gt": torch.empty(Co, device="cuda"), "bet": torch.empty(Co, device="cuda"), "Wr": torch.empty_like(Wr), "br": torch.empty(Co, device="cuda"),
     "gr": torch.empty(Co, device="cuda"), "ber": torch.empty(Co, device="cuda"), "slope_in": torch.empty(1, device="cuda")}
buf = torch.empty(ops.layer_bwd_ws_bytes(B, Ci, Co, T, V), dtype=torch.uint8, device="cuda")
for _ in range(5):
    ops.layer_bwd(x, dU, A, Tm, sl, stat, Wt, one, Wr, one, g, buf, Z=Z)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 16)()
_lib.lib().coskad_debug_fold_stamps(out)
st = list(out)[:6]
print("k_bwd_fold 32->64 phases (us):", [round((b - a) / 100.0, 2) for a, b in zip(st[:-1], st[1:])], "total", round((st[5] - st[0]) / 100.0, 2))
