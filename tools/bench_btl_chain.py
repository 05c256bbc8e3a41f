"""Times ops.btlnk_bwd_chain (bottleneck backward + the top layer's batch reductions, csrc/btlnk_chain.hip) against the two passes it
replaces (ops.btlnk_bwd + ops.layer_bwd_stats) at B = 4096, 64 x (12 x V) columns, 32 channels below, latent 16."""
import sys
import torch
sys.path.insert(0, ".")
from coskad_amd import engine, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
V = int(sys.argv[2]) if len(sys.argv) > 2 else 17
T, Ci, L = 12, 32, 16
K = 64 * T * V
torch.manual_seed(0)
U = torch.randn(B, 64, T, V, device="cuda"); W = torch.randn(L, K, device="cuda") * 0.01; dz = torch.randn(B, L, device="cuda")
x = torch.randn(B, Ci, T, V, device="cuda"); Z = torch.randn(B, Ci, T, V, device="cuda")
A = torch.randn(T, V, V, device="cuda"); Tm = torch.randn(V, T, T, device="cuda")
sl = torch.tensor([0.25], device="cuda")
dU = torch.empty_like(U); dW = torch.empty_like(W); db = torch.empty(L, device="cuda"); ds = torch.empty(1, device="cuda")
ws = engine.Workspace()
wsb = torch.empty(ops.btlnk_bwd_ws_bytes(B, K, L), dtype=torch.uint8, device="cuda")
wsl = torch.empty(ops.layer_bwd_ws_bytes(B, Ci, 64, T, V), dtype=torch.uint8, device="cuda")


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t_chain = timed(lambda: ops.btlnk_bwd_chain(U, W, dz, sl, dW, db, ds, ws, x, Z, sl, dU=dU))
t_b = timed(lambda: ops.btlnk_bwd(U, W, dz, sl, dW, db, ds, wsb, dU=dU))
t_s = timed(lambda: ops.layer_bwd_stats(x, dU, A, Tm, sl, True, wsl, Z=Z))
print(f"B={B} V={V}: btlnk_bwd_chain {t_chain:.1f} us   vs   btlnk_bwd {t_b:.1f} + layer_bwd_stats {t_s:.1f} = {t_b + t_s:.1f} us")
