"""How sensitive are the golden model's gradients (B = 8, batch statistics over 8 x 204 samples) to the rounding of the first
layer's output?  Perturbs U1 = layer-1 pre-activation by (a) random relative noise, (b) a per-channel constant offset of the
same size, and prints the gradient error against the golden file."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_modules as tm
from coskad_amd import ops
import coskad_amd.engine as E
g = dict(np.load("tests/golden/stse_default.npz"))
x = torch.from_numpy(g["x"]).cuda(); c = torch.from_numpy(g["c"]).cuda()
orig = ops.layer_apply_z
def run(mode, eps):
    m, st = tm.build_stse(g)
    m.train()
    def hooked(Z, xx, A, Tm, wfold, bias, Co, in_slope=None, out_slope=None, out=None):
        u = orig(Z, xx, A, Tm, wfold, bias, Co, in_slope=in_slope, out_slope=out_slope, out=out)
        if xx.shape[1] <= 4:
            if mode == "noise": u = u * (1.0 + eps * torch.randn_like(u))
            if mode == "offset": u = u + eps * u.abs().amax(dim=(0, 2, 3), keepdim=True) * torch.sign(torch.randn(1, u.shape[1], 1, 1, device=u.device))
        return u
    ops.layer_apply_z = hooked; E.ops = ops
    z = m(x); loss = ((z - c) ** 2).mean(); loss.backward(); torch.cuda.synchronize()
    worst = 0.0
    for n, p in m.named_parameters():
        ref = g["grad." + n]
        if np.abs(ref).max() > 0 and not n.endswith(("tcn.0.bias", "residual.0.bias")): worst = max(worst, float(np.abs(p.grad.cpu().numpy() - ref).max() / np.abs(ref).max()))
    return worst
torch.manual_seed(0)
for mode in ("none", "noise", "offset"):
    for eps in (1e-7, 1e-6, 1e-5):
        print(mode, eps, "worst gradient error relative to each tensor's max:", f"{run(mode, eps):.2e}")
