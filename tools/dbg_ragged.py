import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_backward import make_layer_state, dev
from coskad_amd import ops
Ci, Co, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T, V = 12, 17
st = make_layer_state(Ci, Co, V, seed=Ci + Co)
g = torch.Generator().manual_seed(B)
x_pre = dev(torch.randn(B, Ci, T, V, generator=g))
probe = dev(torch.randn(B, Co, T, V, generator=g) * 0.1)
d = {k[2:]: dev(v) for k, v in st.items()}
sl = dev(torch.tensor([0.2]))
Wt, Wr = d["tcn.0.weight"].reshape(Co, Ci), d["residual.0.weight"].reshape(Co, Ci)
ws = torch.empty(max(ops.train_stats_ws_bytes(Ci), ops.layer_bwd_ws_bytes(B, Ci, Co, T, V)), dtype=torch.uint8, device="cuda")
Z = torch.empty_like(x_pre)
_, _, stat = ops.layer_train_stats(
    x_pre, d["gcn.A"], d["gcn.T"], sl, Wt, d["tcn.0.bias"], d["tcn.1.weight"], d["tcn.1.bias"],
    d["tcn.1.running_mean"], d["tcn.1.running_var"], d["tcn.1.num_batches_tracked"],
    Wr, d["residual.0.bias"], d["residual.1.weight"], d["residual.1.bias"],
    d["residual.1.running_mean"], d["residual.1.running_var"], d["residual.1.num_batches_tracked"], ws, Z=Z)
def run(zz):
    z = lambda *s_: torch.full(s_, float("nan"), device="cuda")
    gr = {"A": z(T, V, V), "T": z(V, T, T), "Wt": z(Co, Ci), "bt": z(Co), "gt": z(Co), "bet": z(Co), "Wr": z(Co, Ci),
          "br": z(Co), "gr": z(Co), "ber": z(Co), "slope_in": z(1)}
    dIn = torch.empty_like(x_pre)
    ops.layer_bwd(x_pre, probe, d["gcn.A"], d["gcn.T"], sl, stat, Wt, d["tcn.1.weight"], Wr, d["residual.1.weight"], gr, ws, dIn=dIn, Z=zz)
    torch.cuda.synchronize()
    return dIn, gr
dF, gF = run(Z)
dS, gS = run(None)
for k in gS:
    a, b = gF[k].cpu().numpy(), gS[k].cpu().numpy()
    bad = np.abs(a - b) > 2e-3 * np.abs(b) + 2e-4 * np.abs(b).max()
    print(k, "bad", int(bad.sum()), "of", bad.size, "max|b|", float(np.abs(b).max()))
    if bad.sum() and bad.ndim == 3:
        idx = np.argwhere(bad)
        print("  idx sample", idx[:40].tolist())
# oracle for dA
sys.path.insert(0, "oracle")
import importlib
from oracle import ref_cpu as R
stc = {k: v.clone() for k, v in st.items()}
for k in stc:
    if R.is_param_key(k) and stc[k].is_floating_point():
        stc[k].requires_grad_(True)
xo = x_pre.cpu().clone().requires_grad_(True)
so = torch.tensor([0.2], requires_grad=True)
U = R.st_gcnn_layer(R.prelu(xo, so), stc, "L", training=True, return_preact=True)
(U * probe.cpu()).sum().backward()
ref = stc["L.gcn.A"].grad.numpy()
for name, gg in (("fused", gF), ("split", gS)):
    a = gg["A"].cpu().numpy()
    bad = np.abs(a - ref) > 2e-3 * np.abs(ref) + 2e-4 * np.abs(ref).max()
    print(name, "vs oracle: bad", int(bad.sum()))
a = gF["A"].cpu().numpy()
print("fused", a[0, 3, 10:17], "\nref  ", ref[0, 3, 10:17], "\nsplit", gS["A"].cpu().numpy()[0, 3, 10:17])
print("diff/|ref| t=0 v=3:", (a[0, 3, 12:16] - ref[0, 3, 12:16]))
