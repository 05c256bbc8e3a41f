import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_backward as tb
from coskad_amd import ops
from oracle import ref_cpu as R
case = [x for x in tb.LAYER_CASES if x[:3] == (32, 64, 17)][0]
tb.test_layer_backward(*case); print("pre-case ok")
g = torch.Generator().manual_seed(0)
V = 25
x = torch.randn(6, 8, 12, V, generator=g); dZ = torch.randn(6, 8, 12, V, generator=g)
A = torch.randn(12, V, V, generator=g) * 0.2; Tm = torch.randn(V, 12, 12, generator=g) * 0.2
Ar = A.clone().requires_grad_(True); Tr = Tm.clone().requires_grad_(True)
(R.gcn(x, Ar, Tr) * dZ).sum().backward()
dA, dT = ops.gcn_bwd_params(x.cuda(), dZ.cuda(), A.cuda(), Tm.cuda())
ea = (dA.cpu() - Ar.grad).abs(); et = (dT.cpu() - Tr.grad).abs()
print("gcn_bwd_params V=25: max err dA", float(ea.max()), "dT", float(et.max()), "bad t:", sorted(set(np.argwhere(ea.numpy() > 1e-3)[:, 0].tolist())))
# forward+adjoint gcn check too
z = ops.gcn(x.cuda(), A.cuda(), Tm.cuda()).cpu()
print("gcn fwd err", float((z - R.gcn(x, A, Tm)).abs().max()))
