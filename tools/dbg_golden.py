import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import state_from  # noqa
import test_gpu_modules as tm
g = dict(np.load("tests/golden/stse_default.npz"))
m, st = tm.build_stse(g)
x = torch.from_numpy(g["x"]).cuda()
print("B", x.shape)
m.train()
c = torch.from_numpy(g["c"]).cuda()
z = m(x)
loss = ((z - c) ** 2).mean()
loss.backward()
for n, p in m.named_parameters():
    ref = g["grad." + n]
    a = p.grad.cpu().numpy()
    err = np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30)
    print(f"{n:45s} rel-to-max err {err:.2e}")
# compare every layer_apply_z call of a second forward against the recompute kernel
from coskad_amd import ops
orig = ops.layer_apply_z
def hooked(Z, x, A, Tm, wfold, bias, Co, in_slope=None, out_slope=None, out=None):
    u = orig(Z, x, A, Tm, wfold, bias, Co, in_slope=in_slope, out_slope=out_slope, out=out)
    ref = ops.layer_apply(x, A, Tm, wfold, bias, Co, in_slope=in_slope)
    print("apply_z", tuple(x.shape), "->", Co, "max err", float((u - ref).abs().max()), "max", float(ref.abs().max()), "contig", x.is_contiguous(), Z.is_contiguous())
    return u
ops.layer_apply_z = hooked
import coskad_amd.engine as E
E.ops = ops
z2 = m(x)
