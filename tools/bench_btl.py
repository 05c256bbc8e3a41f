"""Times ops.btlnk_bwd (bottleneck backward: dU, dW, db, slope gradient) at B = 4096, K = 64 x 204, L = 16."""
import sys
import torch
sys.path.insert(0, ".")
from coskad_amd import ops
B, K, L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64 * 204, 16
torch.manual_seed(0)
U = torch.randn(B, 64, 12, 17, device="cuda"); W = torch.randn(L, K, device="cuda") * 0.01; dz = torch.randn(B, L, device="cuda")
sl = torch.tensor([0.25], device="cuda")
dU = torch.empty_like(U); dW = torch.empty_like(W); db = torch.empty(L, device="cuda"); ds = torch.empty(1, device="cuda")
ws = torch.empty(ops.btlnk_bwd_ws_bytes(B, K, L), dtype=torch.uint8, device="cuda")
f = lambda: ops.btlnk_bwd(U, W, dz, sl, dW, db, ds, ws, dU=dU)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(f"B={B} btlnk_bwd (both kernels): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
