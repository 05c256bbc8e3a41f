"""A 32 -> 16 ST_GCNN layer by commutation (csrc/commute_layer.hip) against the layer's formula in torch fp64 with autograd
(reference models/graph_layers/stsgcn.py:56-80 the mixing, 94-116 the layer; BatchNorm2d in training mode)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_layer(U_prev, slope, A, Tm, Wt, bt, gt, bet, Wr, br, gr, ber, eps):
    """stsgcn.py:108-116 in training mode, fp64: -> (pre-activation output, batch means / biased variances of both branches)"""
    X = torch.where(U_prev > 0, U_prev, slope * U_prev) if slope is not None else U_prev
    Y = torch.einsum('nctv,vtq->ncqv', X, Tm)
    Z = torch.einsum('nctv,tvw->nctw', Y, A)
    t = torch.einsum('oc,nctv->notv', Wt, Z) + bt.view(1, -1, 1, 1)
    r = torch.einsum('oc,nctv->notv', Wr, X) + br.view(1, -1, 1, 1)

    def bn(x, g, b):
        m = x.mean((0, 2, 3), keepdim=True)
        v = x.var((0, 2, 3), unbiased=False, keepdim=True)
        return (x - m) / torch.sqrt(v + eps) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1), m.flatten(), v.flatten()

    tn, mt, vt = bn(t, gt, bet)
    rn, mr, vr = bn(r, gr, ber)
    return tn + rn, (mt, vt, mr, vr)


def _params(dev, gen, T, V):
    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=gen, device=dev) * scale).contiguous()
    return dict(A=rnd(T, V, V, scale=0.3), Tm=rnd(V, T, T, scale=0.4), Wt=rnd(16, 32, scale=0.25), bt=rnd(16, scale=0.2),
                gt=1.0 + rnd(16, scale=0.2), bet=rnd(16, scale=0.2), Wr=rnd(16, 32, scale=0.25), br=rnd(16, scale=0.2),
                gr=1.0 + rnd(16, scale=0.2), ber=rnd(16, scale=0.2))


@pytest.mark.parametrize("V,B,with_slope", [(25, 3, True), (25, 37, True), (25, 5, False), (25, 1100, True), (17, 5, True), (17, 1037, True)])
def test_commuted_layer_matches_the_layer_formula(V, B, with_slope):
    from coskad_amd import ops
    dev = torch.device("cuda:0")
    T = 12
    assert ops.commute_ok(T, V, 32, 16)
    gen = torch.Generator(device=dev).manual_seed(1234 + B)
    p = _params(dev, gen, T, V)
    U_prev = torch.randn(B, 32, T, V, generator=gen, device=dev)
    slope = torch.tensor([0.25], device=dev) if with_slope else None
    dU = torch.randn(B, 16, T, V, generator=gen, device=dev) / (B * T * V) ** 0.5
    eps, mom = 1e-5, 0.1
    rm_t, rv_t = torch.zeros(16, device=dev), torch.ones(16, device=dev)
    rm_r, rv_r = torch.zeros(16, device=dev), torch.ones(16, device=dev)
    nbt_t = torch.zeros((), dtype=torch.int64, device=dev)
    nbt_r = torch.zeros((), dtype=torch.int64, device=dev)
    U, saved, _ = ops.commute_fwd(U_prev, slope, p["Wt"], p["Wr"], p["A"], p["Tm"], p["gt"], p["bet"], p["gr"], p["ber"], p["bt"], p["br"], rm_t, rv_t,
                               rm_r, rv_r, nbt_t, nbt_r, mom, eps)
    into = {"A": torch.empty_like(p["A"]), "T": torch.empty_like(p["Tm"]), "Wt": torch.empty(16, 32, device=dev),
            "Wr": torch.empty(16, 32, device=dev), "gt": torch.empty(16, device=dev), "bet": torch.empty(16, device=dev),
            "gr": torch.empty(16, device=dev), "ber": torch.empty(16, device=dev)}
    if with_slope:
        into["in_slope"] = torch.empty(1, device=dev)
    d_in, _ = ops.commute_bwd(saved, dU, into)

    # fp64 reference with autograd
    q = {k: v.double().requires_grad_(True) for k, v in p.items()}
    Ud = U_prev.double().requires_grad_(True)
    sd = slope.double().requires_grad_(True) if with_slope else None
    ref, (mt, vt, mr, vr) = _ref_layer(Ud, sd, q["A"], q["Tm"], q["Wt"], q["bt"], q["gt"], q["bet"], q["Wr"], q["br"], q["gr"], q["ber"], eps)
    (ref * dU.double()).sum().backward()

    def close(got, want, what, tol=2e-4):
        want = want.to(torch.float64)
        err = (got.double() - want).abs().max().item()
        scale = want.abs().max().item() + 1e-12
        assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"

    close(U, ref.detach(), "U")
    n = B * T * V
    close(rm_t, mom * mt.detach(), "running_mean tcn")
    close(rv_t, 0.9 + mom * vt.detach() * n / (n - 1), "running_var tcn")
    close(rm_r, mom * mr.detach(), "running_mean residual")
    close(rv_r, 0.9 + mom * vr.detach() * n / (n - 1), "running_var residual")
    assert int(nbt_t) == 1 and int(nbt_r) == 1
    close(d_in, Ud.grad, "dU_prev")
    close(into["A"], q["A"].grad, "dA")
    close(into["T"], q["Tm"].grad, "dT")
    close(into["Wt"], q["Wt"].grad, "dWt")
    close(into["Wr"], q["Wr"].grad, "dWr")
    close(into["gt"], q["gt"].grad, "dgamma tcn")
    close(into["bet"], q["bet"].grad, "dbeta tcn")
    close(into["gr"], q["gr"].grad, "dgamma residual")
    close(into["ber"], q["ber"].grad, "dbeta residual")
    if with_slope:
        close(into["in_slope"], sd.grad, "dslope")


def _state(m, eng):
    import numpy as np  # noqa: F401
    return ({n: v.cpu().numpy().copy() for n, v in eng.fp.gviews.items()},
            {k: v.cpu().numpy().copy() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})


def _compare(a, b):
    import numpy as np
    gmax = max(np.abs(v).max() for v in a[0].values())
    for n, ref in a[0].items():
        np.testing.assert_allclose(b[0][n], ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max() + 2e-5 * gmax, err_msg=n)
    for k, ref in a[1].items():
        np.testing.assert_allclose(b[1][k], ref, rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("V", [25, 17])
def test_autoencoder_step_with_commuted_layers_equals_the_layer_kernels(V, monkeypatch):
    """The default-width autoencoder step with its 32 -> 16 layers on csrc/commute_layer.hip (trainer._FlatStack `commute` segments: at 25
    joints encoder layer 2 and decoder layer 2, at 17 joints the decoder's only -- the encoder's chained kernels stay) against the same
    step on the 32-channel layer kernels: losses, every gradient (incl. the PReLU slopes on both sides of the commuted layers), running
    statistics; ragged batch."""
    import numpy as np
    from coskad_amd import trainer
    from coskad_amd.models.sts.ae import STSAE
    from oracle import ref_cpu as R
    torch.manual_seed(11)
    proto = STSAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
    st = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    x = R.synthetic_clips(37, 2, 12, V, seed=12).cuda()
    res = {}
    for on in (False, True):
        monkeypatch.setattr(trainer, "COMMUTE", on)
        m = STSAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st)
        m.cuda().train()
        eng = trainer.STSAETrainStep(m, mode='ae', lr=0.0, alpha=0.0, lambda_=0.8)
        assert any(s[0] == 'commute' for s in eng.enc.segs) == (on and V == 25)
        assert any(s[0] == 'commute' for s in eng.dec.segs) == on
        out = eng.step(x)
        torch.cuda.synchronize()
        res[on] = (float(out['rec']), float(out['head'])) + _state(m, eng)
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-5)
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=1e-5)
    _compare(res[False][2:], res[True][2:])


@pytest.mark.parametrize("ride", [True, False])
def test_encoder_step_with_a_commuted_layer_equals_the_chain(ride, monkeypatch):
    """STSETrainStep at 25 joints: the encoder as a _FlatStack with layer 2 commuted vs the plain chain; then three optimiser steps
    stay together (loss curve).  ride: layer 3's statistics pass on the commuted layer's last kernel, or as its own launch."""
    import numpy as np
    from coskad_amd import trainer
    from coskad_amd.models.sts.ae import STSE
    from oracle import ref_cpu as R
    torch.manual_seed(21)
    V = 25
    proto = STSE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
    st = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    x = R.synthetic_clips(53, 2, 12, V, seed=22).cuda()
    res = {}
    monkeypatch.setattr(trainer, "COMMUTE_NEXT", ride)
    for on in (False, True):
        monkeypatch.setattr(trainer, "COMMUTE", on)
        m = STSE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st)
        m.cuda().train()
        eng = trainer.STSETrainStep(m, lr=0.0, alpha=0.0)
        assert (eng.stack is not None) == on
        loss = float(eng.step(x)[0])
        torch.cuda.synchronize()
        first = (loss,) + _state(m, eng)
        eng.set_lr(1e-3)
        curve = [float(eng.step(x)[0]) for _ in range(3)]
        res[on] = (first, curve)
    np.testing.assert_allclose(res[True][0][0], res[False][0][0], rtol=1e-5)
    _compare(res[False][0][1:], res[True][0][1:])
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=2e-4)
