"""GPU parity: HIP forward kernels vs the CPU oracle / golden vectors (through the C ABI)."""
import numpy as np
import pytest
import torch

from conftest import state_from
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize("V,C,N", [(17, 32, 5), (17, 2, 70), (25, 8, 9), (14, 4, 3), (18, 16, 7)])
@pytest.mark.parametrize("adjoint", [False, True])
def test_gcn_matches_oracle(V, C, N, adjoint):
    from coskad_amd import ops
    g = torch.Generator().manual_seed(V * 100 + C)
    x = torch.randn(N, C, 12, V, generator=g)
    A = torch.randn(12, V, V, generator=g) * 0.3
    Tm = torch.randn(V, 12, 12, generator=g) * 0.3
    if not adjoint:
        ref = R.gcn(x, A, Tm)
    else:
        xx = x.clone().requires_grad_(True)  # adjoint = vector-Jacobian product
        probe = torch.randn(N, C, 12, V, generator=g)
        (R.gcn(xx, A, Tm) * probe).sum().backward()
        ref, x = xx.grad, probe
    out = ops.gcn(dev(x), dev(A), dev(Tm), adjoint=adjoint).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def _eval_chain(st, x):
    """eval-mode encoder through bn_fold + layer_apply, exchanging pre-activations."""
    from coskad_amd import ops
    n = R.n_layers(st, "encoder.model")
    h, slope, acts = dev(x), None, []
    for i in range(n):
        p = f"encoder.model.{i}"
        d = {k[len(p) + 1:]: dev(v) for k, v in st.items() if k.startswith(p + ".") and v.is_floating_point()}
        Co, Ci = d["tcn.0.weight"].shape[:2]
        has_res = "residual.0.weight" in d
        wfold, bias = ops.bn_fold(
            d["tcn.0.weight"].reshape(Co, Ci), d["tcn.0.bias"], d["tcn.1.weight"], d["tcn.1.bias"],
            d["tcn.1.running_mean"], d["tcn.1.running_var"],
            d["residual.0.weight"].reshape(Co, Ci) if has_res else None,
            d.get("residual.0.bias"), d.get("residual.1.weight"), d.get("residual.1.bias"),
            d.get("residual.1.running_mean"), d.get("residual.1.running_var"))
        u = ops.layer_apply(h, d["gcn.A"], d["gcn.T"], wfold, bias, Co, in_slope=slope)
        acts.append(ops.layer_apply(h, d["gcn.A"], d["gcn.T"], wfold, bias, Co, in_slope=slope,
                                    out_slope=d["prelu.weight"]).cpu())
        h, slope = u, d["prelu.weight"]
    return acts


@pytest.mark.parametrize("name", ["stse_default.npz", "stse_v25.npz", "stse_b1.npz"])
def test_eval_layers_match_golden(golden, name):
    g = golden(name)
    st = state_from(g)
    acts = _eval_chain(st, torch.from_numpy(g["x"]))
    for i, a in enumerate(acts):
        np.testing.assert_allclose(a.numpy(), g[f"eval.act{i}"], rtol=1e-4, atol=1e-4, err_msg=f"layer {i}")


def test_eval_layers_ragged_batch():
    """batch not a multiple of the clip tile, against the oracle."""
    st = R.init_stse_state(seed=3)
    x = R.synthetic_clips(37, seed=4)
    ref = []
    with torch.no_grad():
        R.stse_encode(x, st, training=False, collect=ref)
    acts = _eval_chain(st, x)
    for i, (a, r) in enumerate(zip(acts, ref)):
        np.testing.assert_allclose(a.numpy(), r.numpy(), rtol=1e-4, atol=1e-4, err_msg=f"layer {i}")


def _train_chain(st, x):
    """train-mode encoder: batch stats (moments kernels) + apply; returns per-layer activations
    and the updated BN buffers."""
    from coskad_amd import ops
    n = R.n_layers(st, "encoder.model")
    h, slope, acts, bufs = dev(x), None, [], {}
    ws = torch.empty(ops.train_stats_ws_bytes(64), dtype=torch.uint8, device="cuda")
    for i in range(n):
        p = f"encoder.model.{i}"
        d = {k[len(p) + 1:]: dev(v) for k, v in st.items() if k.startswith(p + ".")}
        Co, Ci = d["tcn.0.weight"].shape[:2]
        has_res = "residual.0.weight" in d
        wfold, bias, stat = ops.layer_train_stats(
            h, d["gcn.A"], d["gcn.T"], slope,
            d["tcn.0.weight"].reshape(Co, Ci), d["tcn.0.bias"], d["tcn.1.weight"], d["tcn.1.bias"],
            d["tcn.1.running_mean"], d["tcn.1.running_var"], d["tcn.1.num_batches_tracked"],
            d["residual.0.weight"].reshape(Co, Ci) if has_res else None, d.get("residual.0.bias"),
            d.get("residual.1.weight"), d.get("residual.1.bias"), d.get("residual.1.running_mean"),
            d.get("residual.1.running_var"), d.get("residual.1.num_batches_tracked"), ws)
        u = ops.layer_apply(h, d["gcn.A"], d["gcn.T"], wfold, bias, Co, in_slope=slope)
        acts.append(torch.where(u > 0, u, d["prelu.weight"] * u).cpu())
        for k in d:
            if "running" in k or "num_batches" in k:
                bufs[f"{p}.{k}"] = d[k].cpu()
        h, slope = u, d["prelu.weight"]
    return acts, bufs


@pytest.mark.parametrize("name", ["stse_default.npz", "stse_v25.npz", "stse_b1.npz"])
def test_train_forward_matches_oracle_and_golden(golden, name):
    g = golden(name)
    st = state_from(g)
    x = torch.from_numpy(g["x"])
    st_ref = {k: v.clone() for k, v in st.items()}
    ref = []
    with torch.no_grad():
        zref = R.stse_encode(x, st_ref, training=True, collect=ref)
    np.testing.assert_allclose(zref.numpy(), g["train.z"], rtol=1e-5, atol=2e-6)  # oracle == reference
    acts, bufs = _train_chain(st, x)
    for i, (a, r) in enumerate(zip(acts, ref)):
        np.testing.assert_allclose(a.numpy(), r.numpy(), rtol=1e-4, atol=1e-4, err_msg=f"layer {i}")
    for k, v in bufs.items():
        np.testing.assert_allclose(v.numpy(), g["sd1." + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_train_forward_large_ragged_batch():
    st = R.init_stse_state(seed=5)
    x = R.synthetic_clips(1031, seed=6)  # > kMaxGrid tiles on the Ci=32 layers, ragged
    st_ref = {k: v.clone() for k, v in st.items()}
    ref = []
    with torch.no_grad():
        R.stse_encode(x, st_ref, training=True, collect=ref)
    acts, bufs = _train_chain(st, x)
    for i, (a, r) in enumerate(zip(acts, ref)):
        np.testing.assert_allclose(a.numpy(), r.numpy(), rtol=1e-4, atol=1e-4, err_msg=f"layer {i}")
    for k, v in bufs.items():
        np.testing.assert_allclose(v.numpy(), st_ref[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("Ci,Co,B", [(32, 64, 1031), (16, 32, 2053), (32, 16, 1026), (16, 16, 7), (32, 32, 5), (16, 64, 6)])
def test_apply_ring_ragged_batch_vs_recompute_kernel(Ci, Co, B):
    """Training-mode apply on the stored-Z path (csrc/fused_apply.hip: wave-per-clip K-ring GEMM, several clips per wave,
    ragged last round) against the recompute kernel of the same library (k_layer_apply / _m: mixes X itself) with the same
    folded weights: two independent implementations of stsgcn.py:94-116's forward.  The output sits inside a guarded buffer."""
    from coskad_amd import ops
    T, V = 12, 17
    g = torch.Generator().manual_seed(Ci * 7 + Co + B)
    x = (torch.randn(B, Ci, T, V, generator=g)).cuda()
    A = ((torch.rand(T, V, V, generator=g) * 2 - 1) / V ** 0.5).cuda()
    Tm = ((torch.rand(V, T, T, generator=g) * 2 - 1) / T ** 0.5).cuda()
    sl = torch.tensor([0.2], device="cuda")
    Wt, Wr = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda(), (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda()
    one, zero = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    gt, gr = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.rand(Co, generator=g) + 0.5).cuda()
    bt, br = (torch.randn(Co, generator=g) * 0.1).cuda(), (torch.randn(Co, generator=g) * 0.1).cuda()
    nb = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
    ws = torch.empty(ops.train_stats_ws_bytes(Ci), dtype=torch.uint8, device="cuda")
    Z = torch.empty_like(x)
    wfold, bias, _ = ops.layer_train_stats(x, A, Tm, sl, Wt, zero.clone(), gt, bt, zero.clone(), one.clone(), nb[0],
                                           Wr, zero.clone(), gr, br, zero.clone(), one.clone(), nb[1], ws, Z=Z)
    n, guard = B * Co * T * V, 4096
    buf = torch.full((n + 2 * guard,), 12345.0, device="cuda")
    out = buf[guard:guard + n].view(B, Co, T, V)
    ops.layer_apply_z(Z, x, A, Tm, wfold, bias, Co, in_slope=sl, out=out)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 12345.0).all()) and bool((buf[guard + n:] == 12345.0).all()), "wrote outside the output"
    ref = ops.layer_apply(x, A, Tm, wfold, bias, Co, in_slope=sl)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
