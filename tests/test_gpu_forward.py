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


@pytest.mark.parametrize("Ci,Co,B,V", [(32, 64, 1031, 17), (16, 32, 2053, 17), (32, 16, 1026, 17), (16, 16, 7, 17), (32, 32, 5, 17),
                                       (16, 64, 6, 17), (32, 64, 1027, 25), (16, 32, 1030, 25), (32, 16, 1025, 25), (16, 16, 9, 25),
                                       (32, 32, 3, 25), (16, 64, 515, 25)])
def test_apply_ring_ragged_batch_vs_recompute_kernel(Ci, Co, B, V):
    """Training-mode apply on the stored-Z path (csrc/fused_apply.hip / fused_apply_bpc.hip at 17 joints, fused_apply_flat.hip at
    25: K-ring GEMMs, several clips per wave / workgroup, ragged last round) against the recompute kernel of the same library
    (k_layer_apply / _m: mixes X itself) with the same folded weights: two independent implementations of stsgcn.py:94-116's
    forward.  The output sits inside a guarded buffer."""
    from coskad_amd import ops
    T = 12
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


def _rand_layer(g, Ci, Co, T=12, V=17):
    """random parameters of one ST_GCNN layer (conv weights, BN affine, mixing matrices) on the device"""
    d = {"A": ((torch.rand(T, V, V, generator=g) * 2 - 1) / V ** 0.5).cuda(),
         "T": ((torch.rand(V, T, T, generator=g) * 2 - 1) / T ** 0.5).cuda(),
         "Wt": (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda(), "Wr": (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda(),
         "gt": (torch.rand(Co, generator=g) + 0.5).cuda(), "gr": (torch.rand(Co, generator=g) + 0.5).cuda(),
         "bet": (torch.randn(Co, generator=g) * 0.1).cuda(), "ber": (torch.randn(Co, generator=g) * 0.1).cuda(),
         "slope": torch.tensor([0.1 + 0.3 * float(torch.rand(1, generator=g))], device="cuda")}
    return d


def _bn_bufs(Co):
    return [torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda"),
            torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")]


@pytest.mark.parametrize("Ci,Co,Cn,B", [(2, 32, 16, 1031), (32, 16, 32, 1026), (16, 32, 64, 2053), (16, 16, 16, 7), (32, 32, 16, 5),
                                        (2, 16, 32, 3), (32, 16, 32, 1)])
def test_apply_next_vs_separate_kernels(Ci, Co, Cn, B):
    """csrc/fused_apply_next.hip (layer i's apply + layer i+1's Z and BatchNorm moments in one wave-per-clip kernel, several
    clips per wave, ragged last round) against the separate kernels of the same library: U_i, Z_{i+1}, and everything the
    fold makes of the moment partials (folded weights, stat block, running statistics).  Outputs sit in guarded buffers."""
    from coskad_amd import ops
    T, V = 12, 17
    g = torch.Generator().manual_seed(Ci * 131 + Co * 7 + Cn + B)
    x = torch.randn(B, Ci, T, V, generator=g).cuda()
    L1, L2 = _rand_layer(g, Ci, Co), _rand_layer(g, Co, Cn)
    sl_in = torch.tensor([0.2], device="cuda") if Ci > 2 else None
    zero1 = torch.zeros(Co, device="cuda")
    ws = torch.empty(ops.train_stats_ws_bytes(64), dtype=torch.uint8, device="cuda")
    Z = torch.empty_like(x)
    b1 = _bn_bufs(Co)
    wfold, bias, _ = ops.layer_train_stats(x, L1["A"], L1["T"], sl_in, L1["Wt"], zero1.clone(), L1["gt"], L1["bet"], b1[0], b1[1], b1[2],
                                           L1["Wr"], zero1.clone(), L1["gr"], L1["ber"], b1[3], b1[4], b1[5], ws, Z=Z)
    # separate kernels: apply, then the next layer's statistics pass over U
    U_ref = ops.layer_apply_z(Z, x, L1["A"], L1["T"], wfold, bias, Co, in_slope=sl_in)
    zero2 = torch.zeros(Cn, device="cuda")
    Z2_ref = torch.empty_like(U_ref)
    br = _bn_bufs(Cn)
    ref = ops.layer_train_stats(U_ref, L2["A"], L2["T"], L1["slope"], L2["Wt"], zero2.clone(), L2["gt"], L2["bet"], br[0], br[1], br[2],
                                L2["Wr"], zero2.clone(), L2["gr"], L2["ber"], br[3], br[4], br[5], ws, Z=Z2_ref)
    # fused kernel
    ftab = torch.empty(ops.ftab_floats(), device="cuda")
    ops.build_ftabs([L2["A"]], [L2["T"]], [ftab])
    n, guard = B * Co * T * V, 4096
    bufU = torch.full((n + 2 * guard,), 12345.0, device="cuda")
    bufZ = torch.full((n + 2 * guard,), 12345.0, device="cuda")
    rows_max = ops.layer_apply_next_rows(B, Ci, Co)
    E = 2 * (Co * Co + Co)
    bufP = torch.full((rows_max * E + 2 * guard,), 12345.0, device="cuda")
    U, Z2, partials = bufU[guard:guard + n].view(B, Co, T, V), bufZ[guard:guard + n].view(B, Co, T, V), bufP[guard:guard + rows_max * E]
    _, _, rows = ops.layer_apply_next(Z, x, wfold, bias, Co, sl_in, L1["slope"], ftab, partials, T, V, out=U, Z_next=Z2)
    bf = _bn_bufs(Cn)
    got = ops.layer_train_fold(partials, rows, B, T, V, L2["Wt"], zero2.clone(), L2["gt"], L2["bet"], bf[0], bf[1], bf[2],
                               L2["Wr"], zero2.clone(), L2["gr"], L2["ber"], bf[3], bf[4], bf[5], ws)
    torch.cuda.synchronize()
    for name, b, m in (("U", bufU, n), ("Z", bufZ, n), ("partials", bufP, rows_max * E)):
        assert bool((b[:guard] == 12345.0).all()) and bool((b[guard + m:] == 12345.0).all()), f"wrote outside {name}"
    assert rows == rows_max
    np.testing.assert_allclose(U.cpu().numpy(), U_ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(Z2.cpu().numpy(), Z2_ref.cpu().numpy(), rtol=1e-4, atol=2e-5)
    for name, a, r in zip(("wfold", "bias", "stat"), got, ref):
        np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=name)
    for i, (a, r) in enumerate(zip(bf, br)):
        np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=f"bn buffer {i}")


@pytest.mark.parametrize("name,B", [("stse_default.npz", 0), ("stse_default.npz", 203)])
def test_chain_forward_fused_next_vs_separate(golden, name, B):
    """engine.chain_forward in training mode with the next-layer statistics fused into the apply kernels (default) against the
    same chain with a statistics pass per layer: activations, stat blocks, stored Z and BatchNorm buffers."""
    from coskad_amd import engine
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.models.graph_layers.stsgcn import layer_tensors
    g = golden(name)
    st = state_from(g)
    x = torch.from_numpy(g["x"]) if B == 0 else R.synthetic_clips(B, seed=9)
    res = []
    for fuse in (True, False):
        m = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st, strict=True)
        m.cuda().train()
        layers = [layer_tensors(l) for l in m.encoder.model]
        old = engine.FUSE_NEXT
        engine.FUSE_NEXT = fuse
        try:
            u, ctx = engine.chain_forward(x.cuda(), layers, True, engine.Workspace(), want_ctx=True)
        finally:
            engine.FUSE_NEXT = old
        torch.cuda.synchronize()
        res.append((u, ctx, {k: v.clone() for k, v in m.state_dict().items()}))
    (u1, c1, s1), (u0, c0, s0) = res
    np.testing.assert_allclose(u1.cpu().numpy(), u0.cpu().numpy(), rtol=1e-4, atol=1e-4)
    for i in range(4):
        np.testing.assert_allclose(c1.inputs[i].cpu().numpy(), c0.inputs[i].cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=f"input {i}")
        np.testing.assert_allclose(c1.zs[i].cpu().numpy(), c0.zs[i].cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=f"Z {i}")
        np.testing.assert_allclose(c1.stats[i].cpu().numpy(), c0.stats[i].cpu().numpy(), rtol=1e-3, atol=1e-4, err_msg=f"stat {i}")
    for k in s0:
        np.testing.assert_allclose(s1[k].cpu().numpy(), s0[k].cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("B", [5, 1037])
def test_chain_forward_fused_next_vs_separate_25_joints(B):
    """The same on the 25-joint layout (csrc/fused_apply_flat.hip, NX form: the next layer's statistics pass on the apply kernel of
    layers 2 and 3; the first layer's apply and the 64-channel layer keep their own passes): activations, stat blocks, stored Z, buffers."""
    from coskad_amd import engine
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.models.graph_layers.stsgcn import layer_tensors
    torch.manual_seed(3)
    proto = STSE(2, [32, 16, 32], 64, 16, 12, 25, 'sts_gcn', 'linear', 'euclidean', 0.0)
    st = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    x = R.synthetic_clips(B, 2, 12, 25, seed=B)
    res = []
    for fuse in (True, False):
        m = STSE(2, [32, 16, 32], 64, 16, 12, 25, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st, strict=True)
        m.cuda().train()
        layers = [layer_tensors(l) for l in m.encoder.model]
        old = engine.FUSE_NEXT
        engine.FUSE_NEXT = fuse
        try:
            u, ctx = engine.chain_forward(x.cuda(), layers, True, engine.Workspace(), want_ctx=True)
        finally:
            engine.FUSE_NEXT = old
        torch.cuda.synchronize()
        res.append((u, ctx, {k: v.clone() for k, v in m.state_dict().items()}))
    (u1, c1, s1), (u0, c0, s0) = res
    np.testing.assert_allclose(u1.cpu().numpy(), u0.cpu().numpy(), rtol=1e-4, atol=1e-4)
    for i in range(4):
        np.testing.assert_allclose(c1.inputs[i].cpu().numpy(), c0.inputs[i].cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=f"input {i}")
        np.testing.assert_allclose(c1.zs[i].cpu().numpy(), c0.zs[i].cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=f"Z {i}")
        np.testing.assert_allclose(c1.stats[i].cpu().numpy(), c0.stats[i].cpu().numpy(), rtol=1e-3, atol=1e-4, err_msg=f"stat {i}")
    for k in s0:
        np.testing.assert_allclose(s1[k].cpu().numpy(), s0[k].cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("V,B", [(25, 3), (25, 1030), (17, 37)])
def test_eval_first_layer_inside_the_second_layers_kernel(V, B, monkeypatch):
    """Eval mode: layer 1 (2 -> 32) formed on the VALU inside layer 2's kernel (csrc/eval_layer_bpc.hip, FIRST form) against the same
    stack run layer by layer (17 joints: a non-default stack, the default one runs the fused encoder)."""
    import numpy as np
    from coskad_amd import engine, ops
    from coskad_amd.models.sts.ae import STSE
    from oracle import ref_cpu as R
    torch.manual_seed(7)
    chans = [32, 16, 32] if V == 25 else [32, 32]
    m = STSE(2, chans, 64, 16, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda()
    with torch.no_grad():
        for mod in m.modules():                                # non-trivial running statistics / PReLU weights
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0.0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
            if isinstance(mod, torch.nn.PReLU):
                mod.weight.fill_(0.2)
    m.eval()
    x = R.synthetic_clips(B, 2, 12, V, seed=B).cuda()
    calls = []
    real = ops.layer_first_pair_apply
    monkeypatch.setattr(ops, "layer_first_pair_apply", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(engine, "EVAL_FIRST_PAIR", on)
        with torch.no_grad():
            outs[on] = m(x).cpu().numpy()
    assert len(calls) == 1                                     # the pair kernel ran, in the first pass only
    np.testing.assert_allclose(outs[True], outs[False], rtol=2e-4, atol=2e-5)
