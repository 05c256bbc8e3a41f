"""The CPU oracle (oracle/ref_cpu.py) replayed against the golden vectors the
reference itself produced (oracle/make_golden.py).  Runs anywhere (no GPU)."""
import numpy as np
import pytest
import torch

from conftest import state_from
from oracle import ref_cpu as R

TOL = dict(rtol=1e-5, atol=2e-6)
# conv biases feeding a train-mode BatchNorm have an analytically ZERO gradient; autograd
# returns ~1e-7 rounding noise there, so gradients get an absolute floor of 2e-6.
GRAD_FLOOR = 1e-5  # x the largest gradient entry of the model


def t(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize("name", ["stse_default.npz", "stse_v25.npz", "stse_b1.npz"])
def test_stse_eval(golden, name):
    g = golden(name)
    st = state_from(g)
    acts = []
    with torch.no_grad():
        z = R.stse_encode(t(g["x"]), st, training=False, collect=acts)
    for i, a in enumerate(acts):
        np.testing.assert_allclose(a.numpy(), g[f"eval.act{i}"], **TOL)
    np.testing.assert_allclose(z.numpy(), g["eval.z"], **TOL)


@pytest.mark.parametrize("name", ["stse_default.npz", "stse_v25.npz", "stse_b1.npz"])
def test_stse_train_step(golden, name):
    g = golden(name)
    st = state_from(g)
    params = {k: v.requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    st.update(params)
    z = R.stse_encode(t(g["x"]), st, training=True)
    np.testing.assert_allclose(z.detach().numpy(), g["train.z"], **TOL)
    loss_h = R.mse_to_center(z, t(g["c"]))
    loss_r = R.calc_reg_loss(list(params.items()))
    np.testing.assert_allclose(loss_h.item(), g["train.loss_hypersphere"], rtol=1e-5)
    np.testing.assert_allclose(loss_r.item(), g["train.loss_reg"], rtol=1e-5)
    (loss_h + float(g["alpha"]) * loss_r).backward()
    gmax = max(np.abs(g["grad." + k]).max() for k in params)
    for k, p in params.items():
        ref = g["grad." + k]
        scale = max(np.abs(ref).max(), 1e-6)
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-5 * scale + GRAD_FLOOR * gmax, err_msg=k)
    # BN running stats after one step
    for k, v in g.items():
        if k.startswith("sd1.") and k != "sd1.c":
            np.testing.assert_allclose(st[k[4:]].detach().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)


def test_stse_poincare_head(golden):
    g = golden("stse_default.npz")
    z = t(g["train.z"]).clone().requires_grad_(True)
    loss, zh = R.poincare_loss(z, t(g["hyp.c"]))
    np.testing.assert_allclose(zh.detach().numpy(), g["hyp.zh"], **TOL)
    np.testing.assert_allclose(loss.item(), g["hyp.loss"], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(z.grad.numpy(), g["hyp.dz"], rtol=1e-4, atol=1e-7)
    # the Lorentz factor is ill-conditioned for points at the project() clamp: feed the fixture's zh
    np.testing.assert_allclose(R.poincare_mean(t(g["hyp.zh"])).numpy(), g["hyp.center"], **TOL)
    np.testing.assert_allclose(R.poincare_mean(zh.detach()).numpy(), g["hyp.center"], rtol=1e-3, atol=2e-5)
    # geoopt's gyromidpoint formula equals the Klein-model mean (parity-unpinned vs geoopt itself)
    np.testing.assert_allclose(R.weighted_midpoint(t(g["hyp.zh"])).numpy(), g["hyp.center"], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("name,hid,V", [("stsae_small.npz", 16, 17), ("stsae_v25.npz", 64, 25)])
def test_stsae(golden, name, hid, V):
    g = golden(name)
    st = state_from(g)
    x = t(g["x"])
    with torch.no_grad():
        z = R.stse_encode(x, st, training=False)
        xr = R.stsae_decode(z, st, hid, 12, V, training=False)
    np.testing.assert_allclose(z.numpy(), g["eval.z"], **TOL)
    np.testing.assert_allclose(xr.numpy(), g["eval.xrec"], **TOL)
    params = {k: v.requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    st.update(params)
    z = R.stse_encode(x, st, training=True)
    xr = R.stsae_decode(z, st, hid, 12, V, training=True)
    np.testing.assert_allclose(z.detach().numpy(), g["train.z"], **TOL)
    np.testing.assert_allclose(xr.detach().numpy(), g["train.xrec"], rtol=1e-4, atol=1e-5)
    loss = ((xr - x) ** 2).mean() + R.mse_to_center(z, t(g["c"]))
    np.testing.assert_allclose(loss.item(), g["train.loss"], rtol=1e-5)
    loss.backward()
    gmax = max(np.abs(g["grad." + k]).max() for k in params)
    for k, p in params.items():
        ref = g["grad." + k]
        scale = max(np.abs(ref).max(), 1e-6)
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=5e-4, atol=5e-5 * scale + GRAD_FLOOR * gmax, err_msg=k)
    for k, v in g.items():
        if k.startswith("sd1.") and k != "sd1.c":
            np.testing.assert_allclose(st[k[4:]].detach().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)


def test_hyper_math(golden):
    g = golden("hyper_math.npz")
    u, a = t(g["u"]), t(g["a"])
    e = R.expmap0(u)
    p = R.project(e)
    np.testing.assert_allclose(e.numpy(), g["expmap0"], **TOL)
    np.testing.assert_allclose(p.numpy(), g["project_expmap0"], **TOL)
    np.testing.assert_allclose(R.project(t(g["raw"])).numpy(), g["project_raw"], **TOL)
    np.testing.assert_allclose(R.mobius_add(a, p).numpy(), g["mobius_add"], **TOL)
    np.testing.assert_allclose(R.dist(a, p).numpy(), g["dist"], rtol=1e-4, atol=2e-6)  # artanh amplifies 1 ulp by 1/(1-x^2) ~ 5e2 at the project() clamp
    np.testing.assert_allclose(R.dist(a[3][None], p).numpy(), g["dist_bcast"], rtol=1e-4, atol=2e-6)  # artanh amplifies 1 ulp by 1/(1-x^2) ~ 5e2 at the project() clamp
    np.testing.assert_allclose(R.dist0(p).numpy(), g["dist0"], rtol=1e-4, atol=2e-6)  # artanh amplifies 1 ulp by 1/(1-x^2) ~ 5e2 at the project() clamp
    np.testing.assert_allclose(R.logmap0(p).numpy(), g["logmap0"], rtol=1e-4, atol=2e-6)  # artanh amplifies 1 ulp by 1/(1-x^2) ~ 5e2 at the project() clamp
    # Lorentz factors of points AT the project() clamp are ill-conditioned (1-|k|^2 ~ 1e-6 in fp32):
    # replay on the fixture's own points, same op order -> tight tolerance
    np.testing.assert_allclose(R.poincare_mean(t(g["project_expmap0"])).numpy(), g["poincare_mean"], **TOL)
    uu = u.clone().requires_grad_(True)
    loss = R.dist(a[3][None], R.project(R.expmap0(uu))).mean()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(uu.grad.numpy(), g["dloss_du"], rtol=2e-4, atol=1e-6)


def test_init_state_keys_match_reference(golden):
    g = golden("stse_default.npz")
    ref_keys = {k[4:]: v.shape for k, v in g.items() if k.startswith("sd0.")}
    st = R.init_stse_state()
    assert set(st) == set(ref_keys)
    for k, v in st.items():
        assert tuple(v.shape) == tuple(ref_keys[k]), k
