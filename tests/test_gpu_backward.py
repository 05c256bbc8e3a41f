"""GPU parity of the backward kernels, heads and optimiser against the CPU oracle's autograd."""
import numpy as np
import pytest
import torch

from conftest import state_from
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def dev(t):
    return None if t is None else t.detach().cuda().contiguous()


def close(a, b, rtol=2e-4, atol_rel=2e-5, msg=""):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    scale = max(float(np.abs(b).max()), 1e-8)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * scale, err_msg=msg)


def make_layer_state(Ci, Co, V, seed, identity=False):
    g = torch.Generator().manual_seed(seed)
    T = 12
    st = {}
    p = "L"
    st[p + ".gcn.A"] = (torch.rand(T, V, V, generator=g) * 2 - 1) / V ** 0.5
    st[p + ".gcn.T"] = (torch.rand(V, T, T, generator=g) * 2 - 1) / T ** 0.5
    for br, bn in (("tcn.0", "tcn.1"), ("residual.0", "residual.1")):
        if br.startswith("residual") and identity:
            continue
        st[f"{p}.{br}.weight"] = (torch.rand(Co, Ci, 1, 1, generator=g) * 2 - 1) / Ci ** 0.5
        st[f"{p}.{br}.bias"] = (torch.rand(Co, generator=g) * 2 - 1) / Ci ** 0.5
        st[f"{p}.{bn}.weight"] = 1 + 0.2 * torch.randn(Co, generator=g)
        st[f"{p}.{bn}.bias"] = 0.2 * torch.randn(Co, generator=g)
        st[f"{p}.{bn}.running_mean"] = torch.zeros(Co)
        st[f"{p}.{bn}.running_var"] = torch.ones(Co)
        st[f"{p}.{bn}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    st[p + ".prelu.weight"] = torch.full((1,), 0.25)
    return st


LAYER_CASES = [
    # Ci, Co, V, B, first(no input act / no dx), identity
    (2, 32, 17, 37, True, False),
    (32, 16, 17, 9, False, False),
    (16, 32, 17, 11, False, False),
    (32, 64, 17, 5, False, False),
    (64, 32, 17, 3, False, False),
    (8, 8, 25, 6, False, True),
    (4, 8, 14, 7, False, False),
    (8, 2, 18, 5, False, False),
]


@pytest.mark.parametrize("Ci,Co,V,B,first,identity", LAYER_CASES)
def test_layer_backward(Ci, Co, V, B, first, identity):
    from coskad_amd import ops
    T = 12
    st = make_layer_state(Ci, Co, V, seed=Ci * 100 + Co, identity=identity)
    g = torch.Generator().manual_seed(7)
    x_pre = torch.randn(B, Ci, T, V, generator=g)
    probe = torch.randn(B, Co, T, V, generator=g)
    slope = torch.tensor([0.2])
    # oracle
    pk = [k for k in st if R.is_param_key(k) and st[k].is_floating_point()]
    stc = {k: v.clone() for k, v in st.items()}
    for k in pk:
        stc[k].requires_grad_(True)
    xo = x_pre.clone().requires_grad_(not first)
    so = slope.clone().requires_grad_(not first)
    X = xo if first else R.prelu(xo, so)
    U = R.st_gcnn_layer(X, stc, "L", training=True, return_preact=True)
    (U * probe).sum().backward()
    # HIP
    d = {k[2:]: dev(v) for k, v in st.items()}
    ws = torch.empty(max(ops.train_stats_ws_bytes(Ci), ops.layer_bwd_ws_bytes(B, Ci, Co, T, V)), dtype=torch.uint8, device="cuda")
    sl = None if first else dev(slope)
    Wt = d["tcn.0.weight"].reshape(Co, Ci)
    Wr = None if identity else d["residual.0.weight"].reshape(Co, Ci)
    wfold, bias, stat = ops.layer_train_stats(
        dev(x_pre), d["gcn.A"], d["gcn.T"], sl, Wt, d["tcn.0.bias"], d["tcn.1.weight"], d["tcn.1.bias"],
        d["tcn.1.running_mean"], d["tcn.1.running_var"], d["tcn.1.num_batches_tracked"],
        Wr, d.get("residual.0.bias"), d.get("residual.1.weight"), d.get("residual.1.bias"),
        d.get("residual.1.running_mean"), d.get("residual.1.running_var"), d.get("residual.1.num_batches_tracked"), ws)
    u_hip = ops.layer_apply(dev(x_pre), d["gcn.A"], d["gcn.T"], wfold, bias, Co, in_slope=sl)
    close(u_hip, U, rtol=1e-4, atol_rel=1e-5, msg="forward preact")
    z = lambda *s: torch.full(s, float("nan"), device="cuda")  # poison: kernels must overwrite
    grads = {"A": z(T, V, V), "T": z(V, T, T), "Wt": z(Co, Ci), "bt": z(Co), "gt": z(Co), "bet": z(Co)}
    if not identity:
        grads.update({"Wr": z(Co, Ci), "br": z(Co), "gr": z(Co), "ber": z(Co)})
    if not first:
        grads["slope_in"] = z(1)
    dIn = ops.layer_bwd(dev(x_pre), dev(probe), d["gcn.A"], d["gcn.T"], sl, stat, Wt, d["tcn.1.weight"], Wr,
                        d.get("residual.1.weight"), grads, ws, need_dx=not first)
    gmax = max(float(stc[k].grad.abs().max()) for k in pk if stc[k].grad is not None)
    ref = {"A": stc["L.gcn.A"].grad, "T": stc["L.gcn.T"].grad, "Wt": stc["L.tcn.0.weight"].grad.reshape(Co, Ci),
           "bt": stc["L.tcn.0.bias"].grad, "gt": stc["L.tcn.1.weight"].grad, "bet": stc["L.tcn.1.bias"].grad}
    if not identity:
        ref.update({"Wr": stc["L.residual.0.weight"].grad.reshape(Co, Ci), "br": stc["L.residual.0.bias"].grad,
                    "gr": stc["L.residual.1.weight"].grad, "ber": stc["L.residual.1.bias"].grad})
    for k, r in ref.items():
        a, b = grads[k].cpu().numpy(), r.numpy()
        assert np.isfinite(a).all(), k
        np.testing.assert_allclose(a, b, rtol=5e-4, atol=5e-5 * max(np.abs(b).max(), 1e-9) + 2e-5 * gmax, err_msg=k)
    if not first:
        close(dIn, xo.grad, rtol=5e-4, atol_rel=5e-5, msg="dIn")
        close(grads["slope_in"], so.grad, rtol=5e-4, atol_rel=1e-4, msg="dslope_in")


@pytest.mark.parametrize("Ci,Co,V,B", [(32, 16, 17, 1031), (16, 32, 17, 1026), (32, 64, 17, 1033), (32, 32, 17, 515),
                                       (32, 16, 25, 1029), (16, 32, 25, 1025), (32, 64, 25, 1027), (16, 16, 25, 771)])
def test_layer_stored_z_ragged_batches_vs_oracle(Ci, Co, V, B):
    """The stored-Z training path of one layer -- statistics pass (k_fwd_moments / k_fwd_moments_bpc) writing Z, apply from the
    stored Z (k_layer_apply_bpc / _ring / _flat), backward (k_bwd_stats_bpc / _flat / _ring, fold, k_layer_bwd_bpc or
    k_bwd_data_bpc + k_gcn_params_bpc) -- against the CPU oracle's autograd of stsgcn.py:94-116 at B ~ 1000 clips, not a
    multiple of anything the kernels tile by: several rounds of every persistent workgroup and a partial last one."""
    from coskad_amd import ops
    T = 12
    st = make_layer_state(Ci, Co, V, seed=Ci * 100 + Co + V)
    g = torch.Generator().manual_seed(11)
    x_pre = torch.randn(B, Ci, T, V, generator=g)
    probe = torch.randn(B, Co, T, V, generator=g) / (B * T * V) ** 0.5
    slope = torch.tensor([0.2])
    pk = [k for k in st if R.is_param_key(k) and st[k].is_floating_point()]
    stc = {k: v.clone() for k, v in st.items()}
    for k in pk:
        stc[k].requires_grad_(True)
    xo = x_pre.clone().requires_grad_(True)
    so = slope.clone().requires_grad_(True)
    U = R.st_gcnn_layer(R.prelu(xo, so), stc, "L", training=True, return_preact=True)
    (U * probe).sum().backward()
    d = {k[2:]: dev(v) for k, v in st.items()}
    ws = torch.empty(max(ops.train_stats_ws_bytes(Ci), ops.layer_bwd_ws_bytes(B, Ci, Co, T, V)), dtype=torch.uint8, device="cuda")
    sl, xd = dev(slope), dev(x_pre)
    Wt, Wr = d["tcn.0.weight"].reshape(Co, Ci), d["residual.0.weight"].reshape(Co, Ci)
    guard = torch.full((B + 2, Ci, T, V), 7.0, device="cuda")
    Z = guard[1:B + 1]
    wfold, bias, stat = ops.layer_train_stats(
        xd, d["gcn.A"], d["gcn.T"], sl, Wt, d["tcn.0.bias"], d["tcn.1.weight"], d["tcn.1.bias"],
        d["tcn.1.running_mean"], d["tcn.1.running_var"], d["tcn.1.num_batches_tracked"],
        Wr, d["residual.0.bias"], d["residual.1.weight"], d["residual.1.bias"],
        d["residual.1.running_mean"], d["residual.1.running_var"], d["residual.1.num_batches_tracked"], ws, Z=Z)
    assert bool((guard[0] == 7.0).all()) and bool((guard[B + 1] == 7.0).all()), "Z written outside its rows"
    with torch.no_grad():
        close(Z, R.gcn(R.prelu(x_pre, slope), st["L.gcn.A"], st["L.gcn.T"]), rtol=1e-4, atol_rel=1e-5, msg="stored Z")
    u_hip = ops.layer_apply_z(Z, xd, d["gcn.A"], d["gcn.T"], wfold, bias, Co, in_slope=sl)
    close(u_hip, U, rtol=1e-4, atol_rel=1e-5, msg="forward preact")
    z = lambda *s: torch.full(s, float("nan"), device="cuda")  # poison: kernels must overwrite
    grads = {"A": z(T, V, V), "T": z(V, T, T), "Wt": z(Co, Ci), "bt": z(Co), "gt": z(Co), "bet": z(Co),
             "Wr": z(Co, Ci), "br": z(Co), "gr": z(Co), "ber": z(Co), "slope_in": z(1)}
    gd = torch.full((B + 2, Ci, T, V), 7.0, device="cuda")
    dIn = ops.layer_bwd(xd, dev(probe), d["gcn.A"], d["gcn.T"], sl, stat, Wt, d["tcn.1.weight"], Wr, d["residual.1.weight"], grads, ws,
                        need_dx=True, dIn=gd[1:B + 1], Z=Z)
    assert bool((gd[0] == 7.0).all()) and bool((gd[B + 1] == 7.0).all()), "dIn written outside its rows"
    gmax = max(float(stc[k].grad.abs().max()) for k in pk if stc[k].grad is not None)
    ref = {"A": stc["L.gcn.A"].grad, "T": stc["L.gcn.T"].grad, "Wt": stc["L.tcn.0.weight"].grad.reshape(Co, Ci),
           "bt": stc["L.tcn.0.bias"].grad, "gt": stc["L.tcn.1.weight"].grad, "bet": stc["L.tcn.1.bias"].grad,
           "Wr": stc["L.residual.0.weight"].grad.reshape(Co, Ci), "br": stc["L.residual.0.bias"].grad,
           "gr": stc["L.residual.1.weight"].grad, "ber": stc["L.residual.1.bias"].grad}
    for k, r in ref.items():
        a, b = grads[k].cpu().numpy(), r.numpy()
        assert np.isfinite(a).all(), k
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * max(np.abs(b).max(), 1e-9) + 5e-5 * gmax, err_msg=k)
    close(dIn, xo.grad, rtol=1e-3, atol_rel=1e-4, msg="dIn")
    close(grads["slope_in"], so.grad, rtol=1e-3, atol_rel=2e-4, msg="dslope_in")


@pytest.mark.parametrize("Ci,Co,V,B", [(2, 32, 17, 37), (2, 32, 17, 1500), (3, 32, 25, 21), (2, 64, 14, 9), (4, 8, 18, 5)])
def test_first_layer_backward_stored_z_vs_oracle(Ci, Co, V, B):
    """The few-channel layer (no dIn, raw input) on the stored-Z path = csrc/first_layer.hip (k_first_stats, k_first_bwd)
    against the torch-autograd oracle of stsgcn.py:94-116 on the CPU."""
    from coskad_amd import ops
    T = 12
    st = make_layer_state(Ci, Co, V, seed=Ci * 100 + Co + V)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, T, V, generator=g)
    probe = torch.randn(B, Co, T, V, generator=g)
    pk = [k for k in st if R.is_param_key(k) and st[k].is_floating_point()]
    stc = {k: v.clone() for k, v in st.items()}
    for k in pk:
        stc[k].requires_grad_(True)
    U = R.st_gcnn_layer(x, stc, "L", training=True, return_preact=True)
    (U * probe).sum().backward()
    d = {k[2:]: dev(v) for k, v in st.items()}
    ws = torch.empty(max(ops.train_stats_ws_bytes(Ci), ops.layer_bwd_ws_bytes(B, Ci, Co, T, V)), dtype=torch.uint8, device="cuda")
    Wt, Wr = d["tcn.0.weight"].reshape(Co, Ci), d["residual.0.weight"].reshape(Co, Ci)
    Z = torch.empty(B, Ci, T, V, device="cuda")
    _, _, stat = ops.layer_train_stats(
        dev(x), d["gcn.A"], d["gcn.T"], None, Wt, d["tcn.0.bias"], d["tcn.1.weight"], d["tcn.1.bias"],
        d["tcn.1.running_mean"], d["tcn.1.running_var"], d["tcn.1.num_batches_tracked"],
        Wr, d["residual.0.bias"], d["residual.1.weight"], d["residual.1.bias"],
        d["residual.1.running_mean"], d["residual.1.running_var"], d["residual.1.num_batches_tracked"], ws, Z=Z)
    z = lambda *s_: torch.full(s_, float("nan"), device="cuda")
    grads = {"A": z(T, V, V), "T": z(V, T, T), "Wt": z(Co, Ci), "bt": z(Co), "gt": z(Co), "bet": z(Co), "Wr": z(Co, Ci),
             "br": z(Co), "gr": z(Co), "ber": z(Co)}
    ops.layer_bwd(dev(x), dev(probe), d["gcn.A"], d["gcn.T"], None, stat, Wt, d["tcn.1.weight"], Wr, d["residual.1.weight"],
                  grads, ws, need_dx=False, Z=Z)
    gmax = max(float(stc[k].grad.abs().max()) for k in pk if stc[k].grad is not None)
    ref = {"A": stc["L.gcn.A"].grad, "T": stc["L.gcn.T"].grad, "Wt": stc["L.tcn.0.weight"].grad.reshape(Co, Ci),
           "bt": stc["L.tcn.0.bias"].grad, "gt": stc["L.tcn.1.weight"].grad, "bet": stc["L.tcn.1.bias"].grad,
           "Wr": stc["L.residual.0.weight"].grad.reshape(Co, Ci), "br": stc["L.residual.0.bias"].grad,
           "gr": stc["L.residual.1.weight"].grad, "ber": stc["L.residual.1.bias"].grad}
    for k, r in ref.items():
        a, b = grads[k].cpu().numpy(), r.numpy()
        assert np.isfinite(a).all(), k
        np.testing.assert_allclose(a, b, rtol=5e-4, atol=5e-5 * max(np.abs(b).max(), 1e-9) + 2e-5 * gmax, err_msg=k)


@pytest.mark.parametrize("Ci,Co,B,V", [(32, 64, 1031, 17), (16, 32, 2053, 17), (32, 16, 1026, 17), (16, 16, 771, 17), (16, 64, 1537, 17),
                                       (32, 32, 800, 17), (32, 64, 515, 25), (16, 32, 1027, 25), (32, 16, 600, 25), (16, 16, 5, 25),
                                       (16, 64, 40, 25), (32, 32, 513, 25)])
def test_fused_backward_ragged_batch_vs_split_kernels(Ci, Co, B, V):
    """Stored-Z path (17 joints: csrc/fused_bwd.hip, one kernel, one clip per workgroup; 25 joints: csrc/fused_stats.hip flat
    reductions, bwd_data_bpc.hip, gcn_params_bpc.hip; several clips per workgroup, ragged last round) against the recompute path
    of the same library (k_bwd_reduce / k_bwd_data / k_bwd_gcn_params: block-per-tile kernels, no stored Z): two independent
    implementations of stsgcn.py:94-116's autograd.  dIn sits inside a guarded buffer: the clip-per-workgroup kernels must not
    write a byte outside the tensor."""
    from coskad_amd import ops
    T = 12
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
        n, guard = x_pre.numel(), 4096
        buf = torch.full((n + 2 * guard,), 12345.0, device="cuda")
        dIn = buf[guard:guard + n].view_as(x_pre)
        ops.layer_bwd(x_pre, probe, d["gcn.A"], d["gcn.T"], sl, stat, Wt, d["tcn.1.weight"], Wr, d["residual.1.weight"], gr, ws,
                      dIn=dIn, Z=zz)
        torch.cuda.synchronize()
        assert bool((buf[:guard] == 12345.0).all()) and bool((buf[guard + n:] == 12345.0).all()), "wrote outside dIn"
        return dIn.clone(), gr

    dIn_f, g_f = run(Z)
    dIn_s, g_s = run(None)
    close(dIn_f, dIn_s.cpu(), rtol=5e-4, atol_rel=5e-5, msg="dIn")
    for k in g_s:
        a, b = g_f[k].cpu().numpy(), g_s[k].cpu().numpy()
        assert np.isfinite(a).all(), k
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-4 * max(np.abs(b).max(), 1e-9), err_msg=k)


@pytest.mark.parametrize("C0,C1,C2,B", [(2, 32, 16, 37), (2, 32, 16, 1026), (2, 32, 16, 4099), (32, 16, 32, 29), (32, 16, 32, 1031),
                                        (32, 16, 32, 4100), (16, 32, 64, 41), (16, 32, 64, 1027), (16, 32, 64, 4097)])
def test_backward_chain_below_stats_vs_own_pass(C0, C1, C2, B):
    """coskad_layer_bwd_chain_f32: the (C1 -> C2) data kernel also forms the batch reductions of the (C0 -> C1) layer below it
    (csrc/fused_bwd.hip, NS = 1 / 2) -- against the same two layers with the lower layer's own statistics pass (k_first_stats /
    k_bwd_stats_ring).  The three pairs of the default stack; ragged batches (several clips per wave, partial last round)."""
    from coskad_amd import ops
    T, V = 12, 17
    st1, st2 = make_layer_state(C0, C1, V, seed=5), make_layer_state(C1, C2, V, seed=6)
    g = torch.Generator().manual_seed(B)
    x = dev(torch.randn(B, C0, T, V, generator=g))
    probe = dev(torch.randn(B, C2, T, V, generator=g) * 0.1)
    d1, d2 = {k[2:]: dev(v) for k, v in st1.items()}, {k[2:]: dev(v) for k, v in st2.items()}
    sl0 = dev(torch.tensor([0.3])) if C0 > 2 else None      # x is a pre-activation unless it is the raw input
    sl = dev(torch.tensor([0.2]))
    ws = torch.empty(max(ops.train_stats_ws_bytes(C1), ops.train_stats_ws_bytes(C0), ops.layer_bwd_ws_bytes(B, C1, C2, T, V),
                         ops.layer_bwd_ws_bytes(B, C0, C1, T, V)), dtype=torch.uint8, device="cuda")

    def stats(xin, dd, Ci, Co, slope):
        Z = torch.empty(B, Ci, T, V, device="cuda")
        wf, bi, stat = ops.layer_train_stats(
            xin, dd["gcn.A"], dd["gcn.T"], slope, dd["tcn.0.weight"].reshape(Co, Ci), dd["tcn.0.bias"], dd["tcn.1.weight"], dd["tcn.1.bias"],
            dd["tcn.1.running_mean"], dd["tcn.1.running_var"], dd["tcn.1.num_batches_tracked"],
            dd["residual.0.weight"].reshape(Co, Ci), dd["residual.0.bias"], dd["residual.1.weight"], dd["residual.1.bias"],
            dd["residual.1.running_mean"], dd["residual.1.running_var"], dd["residual.1.num_batches_tracked"], ws, Z=Z)
        return Z, wf, bi, stat
    Z1, wf1, b1, stat1 = stats(x, d1, C0, C1, sl0)
    U1 = ops.layer_apply_z(Z1, x, d1["gcn.A"], d1["gcn.T"], wf1, b1, C1, in_slope=sl0)
    Z2, _, _, stat2 = stats(U1, d2, C1, C2, sl)

    def grads(Ci, Co, slope):
        z = lambda *s_: torch.full(s_, float("nan"), device="cuda")
        gr = {"A": z(T, V, V), "T": z(V, T, T), "Wt": z(Co, Ci), "bt": z(Co), "gt": z(Co), "bet": z(Co), "Wr": z(Co, Ci),
              "br": z(Co), "gr": z(Co), "ber": z(Co)}
        if slope:
            gr["slope_in"] = z(1)
        return gr

    def run(chain):
        g2, g1 = grads(C1, C2, True), grads(C0, C1, sl0 is not None)
        below, buf, rows = None, None, 0
        if chain:
            rows = ops.layer_bwd_below_rows(B, C1, C2, C0, T, V)
            assert rows > 0
            n, guard = ops.layer_bwd_below_floats(B, C1, C2, C0, T, V), 1024
            buf = torch.full((n + 2 * guard,), 12345.0, device="cuda")
            below = (x, Z1, sl0, buf[guard:guard + n])
        dU1 = ops.layer_bwd(U1, probe, d2["gcn.A"], d2["gcn.T"], sl, stat2, d2["tcn.0.weight"].reshape(C2, C1), d2["tcn.1.weight"],
                            d2["residual.0.weight"].reshape(C2, C1), d2["residual.1.weight"], g2, ws, Z=Z2, below=below)
        dX = ops.layer_bwd(x, dU1, d1["gcn.A"], d1["gcn.T"], sl0, stat1, d1["tcn.0.weight"].reshape(C1, C0), d1["tcn.1.weight"],
                           d1["residual.0.weight"].reshape(C1, C0), d1["residual.1.weight"], g1, ws, need_dx=sl0 is not None, Z=Z1,
                           stats_in=(below[3], rows) if chain else None)
        torch.cuda.synchronize()
        if chain:
            assert bool((buf[:1024] == 12345.0).all()) and bool((buf[-1024:] == 12345.0).all()), "wrote outside below_stats"
        return dU1, dX, g2, g1

    dU_c, dX_c, g2c, g1c = run(True)
    dU_o, dX_o, g2o, g1o = run(False)
    assert torch.equal(dU_c, dU_o)
    # the upper layer's own results do not depend on the chain: dU bit for bit; its dA / dT / slope sums are summed over a
    # different number of workgroups (the chain variant runs two waves per SIMD, the plain 16-channel one three): fp32 rounding only
    for k in g2o:
        a, b_ = g2c[k].cpu().numpy(), g2o[k].cpu().numpy()
        np.testing.assert_allclose(a, b_, rtol=2e-5, atol=2e-6 * max(np.abs(b_).max(), 1e-9), err_msg="upper layer " + k)
    if dX_o is not None:
        close(dX_c, dX_o.cpu(), rtol=1e-3, atol_rel=1e-4, msg="dX of the lower layer")
    gmax = max(float(v.abs().max()) for v in g1o.values())
    for k in g1o:
        a, b_ = g1c[k].cpu().numpy(), g1o[k].cpu().numpy()
        assert np.isfinite(a).all(), k
        np.testing.assert_allclose(a, b_, rtol=1e-3, atol=1e-4 * max(np.abs(b_).max(), 1e-9) + 2e-5 * gmax, err_msg="lower layer " + k)


@pytest.mark.parametrize("B,hid,V,L,with_slope", [(37, 64, 17, 16, True), (5, 8, 25, 8, True), (16, 4, 17, 4, False), (300, 16, 17, 16, True)])
def test_bottleneck(B, hid, V, L, with_slope):
    from coskad_amd import ops
    g = torch.Generator().manual_seed(B + hid)
    T = 12
    K = hid * T * V
    U = torch.randn(B, hid, T, V, generator=g, requires_grad=True)
    W = (torch.randn(L, K, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(L, generator=g).requires_grad_(True)
    a = torch.tensor([0.3], requires_grad=True)
    X = R.prelu(U, a) if with_slope else U
    z = R.linear(X.reshape(B, -1), W, b)
    dz = torch.randn(B, L, generator=g)
    (z * dz).sum().backward()
    sl = dev(a) if with_slope else None
    zh = ops.btlnk_fwd(dev(U), dev(W), dev(b), sl)
    close(zh, z, rtol=1e-4, atol_rel=1e-5, msg="z")
    ws = torch.empty(ops.btlnk_bwd_ws_bytes(B, K, L), dtype=torch.uint8, device="cuda")
    dW = torch.full((L, K), float("nan"), device="cuda")
    db = torch.full((L,), float("nan"), device="cuda")
    da = torch.full((1,), float("nan"), device="cuda")
    dU = ops.btlnk_bwd(dev(U), dev(W), dev(dz), sl, dW, db, da if with_slope else None, ws)
    close(dU, U.grad, msg="dU")
    close(dW, W.grad, msg="dW")
    close(db, b.grad, msg="db")
    if with_slope:
        close(da, a.grad, rtol=5e-4, atol_rel=1e-4, msg="dslope")


def test_mse_head():
    from coskad_amd import ops
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1000, 16, generator=g, requires_grad=True)
    c = torch.randn(16, generator=g)
    loss = R.mse_to_center(z, c)
    loss.backward()
    acc = torch.zeros(ops.HEAD_SLOTS, device="cuda")
    stats, dz, score = ops.mse_head(dev(z), dev(c), need_score=True, acc=acc)
    stats, dz2, _ = ops.mse_head(dev(z), dev(c), acc=acc)  # second call accumulates again
    close(stats[0], loss, rtol=1e-5)
    close(dz, z.grad)
    close(score, R.euclid_window_score(z.detach(), c), rtol=1e-5)
    close(stats[1:17], z.detach().sum(0), rtol=1e-4, atol_rel=1e-5)
    close(acc[1:17], 2 * z.detach().sum(0), rtol=1e-4, atol_rel=1e-5)
    assert float(acc[17]) == 2000.0
    cc = ops.center_finalize(acc, 0.5, 16).cpu()
    np.testing.assert_allclose(cc.numpy(), R.clamp_center(z.detach().mean(0), 0.5).numpy(), rtol=1e-4, atol=1e-6)


def test_poincare_head_matches_oracle_and_golden(golden):
    from coskad_amd import ops
    g = golden("hyper_math.npz")
    u = torch.from_numpy(g["u"]).clone().requires_grad_(True)
    c = torch.from_numpy(g["a"][3])
    loss, zh = R.poincare_loss(u, c)
    loss.backward()
    stats, dz, zh_hip, score = ops.poincare_head(dev(u), dev(c), need_zh=True, need_score=True)
    np.testing.assert_allclose(zh_hip.cpu().numpy(), g["project_expmap0"], rtol=1e-5, atol=2e-6)  # the reference itself
    np.testing.assert_allclose(score.cpu().numpy(), g["dist_bcast"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(stats[0]), float(g["loss"]), rtol=1e-4)
    np.testing.assert_allclose(dz.cpu().numpy(), g["dloss_du"], rtol=2e-3, atol=2e-6)
    close(dz, u.grad, rtol=2e-3, atol_rel=2e-5)
    # on-ball distance
    sc2 = ops.poincare_dist(dev(torch.from_numpy(g["project_expmap0"])), dev(c)).cpu()
    np.testing.assert_allclose(sc2.numpy(), g["dist_bcast"], rtol=2e-4, atol=2e-5)


def test_poincare_logmap0_vs_reference_golden(golden):
    """logmap0 (hyper_math.py:367-370) on the reference's own points and values; expmap0 -> logmap0 returns the tangent vector
    wherever neither the tanh clamp nor the ball projection has cut it."""
    from coskad_amd import ops
    g = golden("hyper_math.npz")
    p = torch.from_numpy(g["project_expmap0"])
    out = ops.poincare_logmap0(dev(p)).cpu()
    np.testing.assert_allclose(out.numpy(), g["logmap0"], rtol=2e-5, atol=1e-7)
    close(dev(out), R.logmap0(p), rtol=2e-5, atol_rel=1e-6)
    u = torch.from_numpy(g["u"])
    inner = (u.norm(dim=-1) > 1e-4) & (u.norm(dim=-1) < 2.0)
    _, _, zh, _ = ops.poincare_head(dev(u), None, need_zh=True, need_grad=False)
    back = ops.poincare_logmap0(zh).cpu()
    np.testing.assert_allclose(back[inner].numpy(), u[inner].numpy(), rtol=2e-4, atol=1e-6)


def test_poincare_center(golden):
    from coskad_amd import ops
    g = golden("stse_default.npz")
    z = torch.from_numpy(g["train.z"])
    c = torch.from_numpy(g["hyp.c"])
    acc = torch.zeros(ops.HEAD_SLOTS, device="cuda")
    stats, dz, zh, _ = ops.poincare_head(dev(z), dev(c), need_zh=True, acc=acc)
    np.testing.assert_allclose(zh.cpu().numpy(), g["hyp.zh"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(float(stats[0]), float(g["hyp.loss"]), rtol=1e-5)
    np.testing.assert_allclose(dz.cpu().numpy(), g["hyp.dz"], rtol=5e-4, atol=1e-7)
    mid = ops.midpoint_finalize(acc, 16).cpu()
    np.testing.assert_allclose(mid.numpy(), g["hyp.center"], rtol=1e-3, atol=2e-5)  # Klein-model mean == gyromidpoint


def test_adam_and_reg():
    from coskad_amd import ops
    g = torch.Generator().manual_seed(3)
    n = 10007
    p0 = torch.randn(n, generator=g)
    mask = (torch.rand(n, generator=g) > 0.3).float()
    reg_scale = 0.5 / 7
    np.testing.assert_allclose(float(ops.sqnorm(dev(p0), dev(mask), reg_scale)), float(reg_scale * (mask * p0 * p0).sum()), rtol=1e-5)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-2)
    ph, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    alpha = 0.1
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        opt.zero_grad()
        pt.grad = gr + alpha * 2 * reg_scale * mask * pt.detach()
        opt.step()
        ops.adam(ph, dev(gr), m, v, dev(mask), 1e-2, 0.9, 0.999, 1e-8, step, reg_coef=alpha * 2 * reg_scale)
    np.testing.assert_allclose(ph.cpu().numpy(), pt.detach().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,Ci,V,L,with_slope,in_act", [(37, 32, 17, 16, True, True), (1031, 32, 17, 16, True, True), (5, 16, 17, 8, True, False),
                                                        (300, 32, 25, 16, True, True), (33, 16, 25, 9, False, True), (4099, 32, 17, 16, True, True)])
def test_bottleneck_backward_with_top_layer_reductions(B, Ci, V, L, with_slope, in_act):
    """coskad_btlnk_bwd_chain_f32 (csrc/btlnk_chain.hip): the bottleneck backward (ae.py:97-101 under autograd) AND the top
    layer's BatchNorm-backward batch reductions (stsgcn.py:94-116) from one position-major pass, against torch autograd for
    dU / dW / db / dslope and fp64 einsums for P = sum dU Z^T, Q = sum dU PReLU(x)^T, s = sum dU.  Ragged chunks (B not a multiple
    of 16), the masked last position tile (204 = 12 x 16 + 12; 300 = 18 x 16 + 12), latent sizes below 16, guard bands around dU."""
    from coskad_amd import engine, ops
    g = torch.Generator().manual_seed(B + Ci + V)
    T, hid = 12, 64
    K = hid * T * V
    U = torch.randn(B, hid, T, V, generator=g, requires_grad=True)
    W = (torch.randn(L, K, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(L, generator=g).requires_grad_(True)
    a = torch.tensor([0.3], requires_grad=True)
    X = R.prelu(U, a) if with_slope else U
    z = R.linear(X.reshape(B, -1), W, b)
    dz = torch.randn(B, L, generator=g)
    (z * dz).sum().backward()
    x_in = torch.randn(B, Ci, T, V, generator=g)
    Zs = torch.randn(B, Ci, T, V, generator=g)
    sl = dev(a) if with_slope else None
    sl_in = dev(torch.tensor([0.2])) if in_act else None
    dW = torch.full((L, K), float("nan"), device="cuda")
    db = torch.full((L,), float("nan"), device="cuda")
    da = torch.full((1,), float("nan"), device="cuda")
    guard = torch.full((B + 2, hid, T, V), 7.0, device="cuda")
    ws = engine.Workspace()
    dU, (buf, rows) = ops.btlnk_bwd_chain(dev(U), dev(W), dev(dz), sl, dW, db, da if with_slope else None, ws, dev(x_in), dev(Zs), sl_in,
                                          dU=guard[1:B + 1])
    assert rows > 0
    assert bool((guard[0] == 7.0).all()) and bool((guard[B + 1] == 7.0).all()), "dU written outside its rows"
    close(dU, U.grad, msg="dU")
    close(dW, W.grad, msg="dW")
    close(db, b.grad, msg="db")
    if with_slope:
        close(da, a.grad, rtol=5e-4, atol_rel=1e-4, msg="dslope")
    got = ops.chain_sums(buf, rows, Ci, hid).cpu().numpy()
    d = U.grad.double()
    Xb = (torch.where(x_in > 0, x_in, 0.2 * x_in) if in_act else x_in).double()
    P = torch.einsum("botv,bctv->oc", d, Zs.double()).reshape(-1)
    Q = torch.einsum("botv,bctv->oc", d, Xb).reshape(-1)
    s = d.sum(dim=(0, 2, 3))
    want = torch.cat([P, Q, s]).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    # the same call again: bit-identical (fixed-order sums)
    dW2 = torch.empty_like(dW)
    dU2, (buf2, rows2) = ops.btlnk_bwd_chain(dev(U), dev(W), dev(dz), sl, dW2, None, None, ws, dev(x_in), dev(Zs), sl_in)
    assert rows2 == rows and torch.equal(dU2, dU) and torch.equal(dW2, dW)
    assert torch.equal(ops.chain_sums(buf2, rows, Ci, hid), ops.chain_sums(buf, rows, Ci, hid))


@pytest.mark.parametrize("Ci,Co,V,B", [(32, 64, 25, 1027), (16, 32, 25, 515), (32, 16, 25, 1025), (16, 16, 25, 9), (32, 32, 25, 3),
                                       (16, 64, 25, 30), (32, 64, 17, 1031), (32, 32, 17, 40), (16, 32, 17, 700), (32, 16, 17, 600)])
def test_bwd_stats_kernels_vs_torch(Ci, Co, V, B):
    """Stage 1 of the layer backward alone (coskad_layer_bwd_stats_f32: the batch reductions behind both BatchNorm backward folds,
    stsgcn.py:94-116 under autograd) on the stored-Z path -- csrc/fused_stats.hip: one clip per workgroup at 12 x 17 with 32 input
    channels and a wide output, flat positions on the 25-joint layout, wave-per-clip otherwise -- against fp64 einsums:
    P = sum dU Z^T, Q = sum dU PReLU(x)^T, s = sum dU.  Ragged batches (several clips per workgroup, partial last round)."""
    from coskad_amd import ops
    T = 12
    g = torch.Generator().manual_seed(Ci + 3 * Co + V + B)
    x = dev(torch.randn(B, Ci, T, V, generator=g))
    Z = dev(torch.randn(B, Ci, T, V, generator=g))
    dU = dev(torch.randn(B, Co, T, V, generator=g) * 0.3)
    A = dev((torch.rand(T, V, V, generator=g) * 2 - 1) / V ** 0.5)
    Tm = dev((torch.rand(V, T, T, generator=g) * 2 - 1) / T ** 0.5)
    sl = dev(torch.tensor([0.2]))
    ws = torch.empty(ops.layer_bwd_ws_bytes(B, Ci, Co, T, V), dtype=torch.uint8, device="cuda")
    buf, rows = ops.layer_bwd_stats(x, dU, A, Tm, sl, True, ws, Z=Z)
    assert rows > 0
    got = ops.chain_sums(buf, rows, Ci, Co).cpu().numpy()
    X = torch.where(x > 0, x, 0.2 * x).double()
    d = dU.double()
    P = torch.einsum("botv,bctv->oc", d, Z.double()).reshape(-1)
    Q = torch.einsum("botv,bctv->oc", d, X).reshape(-1)
    s = d.sum(dim=(0, 2, 3))
    want = torch.cat([P, Q, s]).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("N,C,V,with_add", [(37, 64, 17, True), (5, 256, 17, False), (9, 64, 25, True), (3, 8, 18, False)])
def test_gcn_bwd_params_dx_vs_separate_kernels(N, C, V, with_add):
    """coskad_gcn_bwd_params_dx_f32 (dA, dT and the adjoint mix from one pass over dZ; the wide layers' backward) against
    coskad_gcn_bwd_params_f32 + coskad_gcn_f32(adjoint) (+ the addend), ConvTemporalGraphical under autograd (stsgcn.py:143-156)."""
    from coskad_amd import ops
    T = 12
    g = torch.Generator().manual_seed(N + C + V)
    x = dev(torch.randn(N, C, T, V, generator=g))
    dZ = dev(torch.randn(N, C, T, V, generator=g) * 0.3)
    add = dev(torch.randn(N, C, T, V, generator=g)) if with_add else None
    A = dev((torch.rand(T, V, V, generator=g) * 2 - 1) / V ** 0.5)
    Tm = dev((torch.rand(V, T, T, generator=g) * 2 - 1) / T ** 0.5)
    dA, dT, dX = ops.gcn_bwd_params_dx(x, dZ, A, Tm, add=add)
    dA0, dT0 = ops.gcn_bwd_params(x, dZ, A, Tm)
    dX0 = ops.gcn(dZ, A, Tm, adjoint=True)
    if add is not None:
        dX0 = dX0 + add
    np.testing.assert_array_equal(dA.cpu().numpy(), dA0.cpu().numpy())
    np.testing.assert_array_equal(dT.cpu().numpy(), dT0.cpu().numpy())
    np.testing.assert_allclose(dX.cpu().numpy(), dX0.cpu().numpy(), rtol=1e-5, atol=1e-5)
