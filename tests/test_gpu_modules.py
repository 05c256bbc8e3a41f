"""GPU parity of the module mirror (STSE / STSAE with the reference's names) against the golden vectors:
reference checkpoints load by key, forward and torch-autograd backward match."""
import numpy as np
import pytest
import torch

from conftest import state_from

pytestmark = pytest.mark.gpu


def build_stse(g, cls=None, **kw):
    from coskad_amd.models.sts.ae import STSE
    st = state_from(g)
    chans = []
    i = 0
    while f"encoder.model.{i}.tcn.0.weight" in st:
        chans.append(st[f"encoder.model.{i}.tcn.0.weight"].shape[0])
        i += 1
    V = st["encoder.model.0.gcn.A"].shape[1]
    m = (cls or STSE)(input_dim=2, layer_channels=chans[:-1], hidden_dimension=chans[-1], latent_dim=st["c"].shape[0],
                      n_frames=12, n_joints=V, encoder_type='STS_GCN', projector='linear', distance='euclidean', dropout=0.0, **kw)
    missing, unexpected = m.load_state_dict(st, strict=True)
    return m.cuda(), st


@pytest.mark.parametrize("name", ["stse_default.npz", "stse_v25.npz", "stse_b1.npz"])
def test_stse_eval_and_train_step(golden, name):
    g = golden(name)
    m, st = build_stse(g)
    assert set(m.state_dict().keys()) == set(st.keys())
    x = torch.from_numpy(g["x"]).cuda()
    m.eval()
    with torch.no_grad():
        z = m(x)
    np.testing.assert_allclose(z.cpu().numpy(), g["eval.z"], rtol=1e-4, atol=1e-4)
    # one train step with torch autograd driving the HIP nodes
    m.train()
    c = torch.from_numpy(g["c"]).cuda()
    z = m(x)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["train.z"], rtol=1e-4, atol=1e-4)
    loss = ((z - c) ** 2).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["train.loss_hypersphere"], rtol=1e-4)
    gmax = max(np.abs(g["grad." + n]).max() for n, _ in m.named_parameters())
    for n, p in m.named_parameters():
        ref = g["grad." + n]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max() + 2e-5 * gmax, err_msg=n)
    for k, v in g.items():
        if k.startswith("sd1.") and k != "sd1.c":
            np.testing.assert_allclose(m.state_dict()[k[4:]].cpu().numpy(), v, rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("below,nxt", [(False, True), (True, False), (False, False)])
def test_stse_train_step_without_the_fusions(golden, below, nxt):
    """The same reference gradients with the cross-layer fusions switched off one at a time (engine.FUSE_BELOW: backward chain,
    engine.FUSE_NEXT: apply + next-layer statistics): the per-layer statistics kernels stay covered."""
    from coskad_amd import engine
    g = golden("stse_default.npz")
    m, st = build_stse(g)
    x = torch.from_numpy(g["x"]).cuda()
    c = torch.from_numpy(g["c"]).cuda()
    old = engine.FUSE_BELOW, engine.FUSE_NEXT
    engine.FUSE_BELOW, engine.FUSE_NEXT = below, nxt
    try:
        m.train()
        z = m(x)
        ((z - c) ** 2).mean().backward()
        torch.cuda.synchronize()
    finally:
        engine.FUSE_BELOW, engine.FUSE_NEXT = old
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["train.z"], rtol=1e-4, atol=1e-4)
    gmax = max(np.abs(g["grad." + n]).max() for n, _ in m.named_parameters())
    for n, p in m.named_parameters():
        ref = g["grad." + n]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max() + 2e-5 * gmax, err_msg=n)


def test_legacy_keywords(golden):
    from coskad_amd.models.sts.ae import STSE
    m = STSE(c_in=2, h_dim=64, latent_dim=16, n_frames=12, dropout=0.0, n_joints=17, channels=[32, 16, 32],
             projector='linear', encoder_type='STS_GCN')
    assert sum(p.numel() for p in m.parameters()) == 239716
    with pytest.raises(ValueError):
        STSE(c_in=2, h_dim=64, latent_dim=16, n_frames=12, dropout=0.0, n_joints=17, channels=[32, 16, 32], encoder_type='nope')


@pytest.mark.parametrize("name", ["stsae_small.npz", "stsae_v25.npz"])
def test_stsae(golden, name):
    """stsae_v25: the DEFAULT widths (32-16-32, hidden 64, latent 8) on the 25-joint layout = BASELINE config 4's model
    shape (different LDS budgets and kernel dispatch than V = 17), encoder + decoder, eval + train + every gradient."""
    from coskad_amd.models.sts.ae import STSAE
    g = golden(name)
    m, st = build_stse(g, cls=STSAE)
    x = torch.from_numpy(g["x"]).cuda()
    m.eval()
    with torch.no_grad():
        z, xr = m(x)
    np.testing.assert_allclose(z.cpu().numpy(), g["eval.z"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xr.cpu().numpy(), g["eval.xrec"], rtol=1e-4, atol=1e-4)
    m.train()
    z, xr = m(x)
    np.testing.assert_allclose(xr.detach().cpu().numpy(), g["train.xrec"], rtol=2e-4, atol=2e-4)
    c = torch.from_numpy(g["c"]).cuda()
    loss = ((xr - x) ** 2).mean() + ((z - c) ** 2).mean()
    np.testing.assert_allclose(loss.item(), g["train.loss"], rtol=1e-4)
    loss.backward()
    gmax = max(np.abs(g["grad." + n]).max() for n, _ in m.named_parameters())
    for n, p in m.named_parameters():
        ref = g["grad." + n]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max() + 5e-5 * gmax, err_msg=n)


def test_cpu_tensor_fails_loudly():
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd._lib import CoskadHipError
    m = STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    with pytest.raises(CoskadHipError):
        m(torch.zeros(2, 2, 12, 17))


def test_stsvae_forward_backward():
    """STSVAE: deterministic heads against the oracle, sampled latent on the sphere, gradients flow."""
    from coskad_amd.models.sts.vae import STSVAE, kl_ps_uniform
    from oracle import ref_cpu as R
    torch.manual_seed(0)
    m = STSVAE(2, [16, 8, 16], 16, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0, distribution='ps').cuda()
    x = R.synthetic_clips(6, seed=3).cuda()
    m.eval()
    with torch.no_grad():
        zm, zv = m.encode(x)
    st = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        flat = R.stse_encode(x.cpu(), st, training=False)          # btlnk = Identity -> flattened encoder output
        zm_ref, zv_ref = R.stsvae_heads(flat, st, 'ps')
    np.testing.assert_allclose(zm.cpu().numpy(), zm_ref.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(zv.cpu().numpy(), zv_ref.numpy(), rtol=1e-3, atol=1e-4)
    m.train()
    z, xr, (q, p, kappa) = m(x)
    np.testing.assert_allclose(z.norm(dim=-1).detach().cpu().numpy(), 1.0, atol=1e-4)
    loss = ((xr - x) ** 2).mean() + kl_ps_uniform(q, p).mean() + (1 / kappa).mean()    # spherical_vae.py:81-107
    loss.backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in m.parameters())


def test_mahalanobis_head_vs_oracle():
    """loss, gradient, per-window score, centre sums, second moments and the inverse covariance (a13)."""
    from coskad_amd import ops
    from coskad_amd.trainer import inv_cov_from_moments
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(5)
    B, L = 1000, 16
    z = torch.randn(B, L, generator=g) * torch.linspace(0.2, 2.0, L)
    c = torch.randn(L, generator=g) * 0.1
    A = torch.randn(L, L, generator=g)
    VI = A @ A.T / L + 0.5 * torch.eye(L) + 0.05 * torch.randn(L, L, generator=g)      # not exactly symmetric
    zr = z.clone().requires_grad_(True)
    d_ref = R.mahalanobis(zr, c[None], VI)
    d_ref.mean().backward()
    acc = torch.zeros(ops.HEAD_SLOTS, device="cuda")
    gram = torch.zeros(L, L, device="cuda")
    stats, dz, score = ops.mahalanobis_head(z.cuda(), c.cuda(), VI.cuda(), need_score=True, acc=acc, gram=gram)
    np.testing.assert_allclose(score.cpu().numpy(), d_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(stats[0]), float(d_ref.detach().mean()), rtol=1e-5)
    np.testing.assert_allclose(dz.cpu().numpy(), zr.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(acc[1:1 + L].cpu().numpy(), z.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    assert float(acc[17]) == B
    np.testing.assert_allclose(gram.cpu().numpy(), (z.T.double() @ z.double()).numpy(), rtol=1e-4, atol=1e-2)
    # compute_inv_cov_mat (staticCenter.py:133-142) = inverse(sum (z-mu)(z-mu)^T / (n-1))
    mu = z.mean(0)
    S = ((z - mu)[:, :, None] @ (z - mu)[:, None, :]).sum(0)
    ref_inv = torch.inverse(S / (B - 1))
    got = inv_cov_from_moments(gram, acc, mu.cuda(), L).cpu()
    np.testing.assert_allclose(got.numpy(), ref_inv.numpy(), rtol=2e-3, atol=1e-4)
    # a second batch accumulates
    ops.mahalanobis_head(z.cuda(), c.cuda(), VI.cuda(), need_grad=False, acc=acc, gram=gram)
    np.testing.assert_allclose(gram.cpu().numpy(), 2 * (z.T.double() @ z.double()).numpy(), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("enc", ["learnable_gcn", "static_gcn"])
def test_plain_gcn_encoders_vs_reference(enc):
    """STSE with the plain-GCN encoders (a16): library GEMMs + the HIP bottleneck, against reference outputs/gradients."""
    import os
    from coskad_amd.models.sts.ae import STSE
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "stse_altgcn.npz"))
    m = STSE(2, [8, 4], 8, 8, 12, 17, enc, 'linear', 'euclidean', 0.0)
    sd = {k[len(enc) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(enc + ".sd.")}
    assert set(sd) == set(m.state_dict())                               # same keys as the reference
    m.load_state_dict(sd, strict=True)
    m.cuda()
    x = torch.from_numpy(g[enc + ".x"]).cuda()
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x).cpu().numpy(), g[enc + ".z_eval"], rtol=1e-4, atol=1e-5)
    m.train()
    z = m(x)
    loss = torch.nn.functional.mse_loss(z, torch.full((8,), 0.05, device="cuda"))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g[enc + ".loss"]), rtol=1e-5)
    for k, p in m.named_parameters():
        ref = g[f"{enc}.grad.{k}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)


def test_split_backward_on_side_stream_matches(golden):
    """coskad_layer_bwd_data_f32 + coskad_layer_gcn_params_f32 on a second stream (the split, per-kernel backward) against
    the single-call backward, which runs the fused wave-per-clip kernel (csrc/fused_bwd.hip) for 16 / 32-channel layers:
    two implementations of the same arithmetic, equal up to fp32 summation order (three Adam steps apart)."""
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    from oracle import ref_cpu as R
    x = R.synthetic_clips(64, seed=9).cuda()
    flats, grads = [], []
    for side in (False, True):
        st = R.init_stse_state(2, (32, 16, 32), 64, 16, 12, 17, seed=0)
        st["c"] = torch.full((16,), 0.1)
        m = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st, strict=True)
        eng = STSETrainStep(m.cuda().train(), lr=1e-3, alpha=1e-6, side_stream=side)
        eng.step(x)
        torch.cuda.synchronize()
        grads.append(eng.fp.grad.clone())
        for _ in range(2):
            eng.step(x)
        torch.cuda.synchronize()
        flats.append(eng.fp.flat.clone())
    # the gradients of the first step: equal up to fp32 summation order
    g0, g1 = grads[0].cpu().numpy(), grads[1].cpu().numpy()
    np.testing.assert_allclose(g0, g1, rtol=2e-3, atol=2e-5 * float(np.abs(g1).max()))
    # three Adam steps later: Adam turns the sign noise of near-zero gradient elements into +-lr steps, so the parameters
    # are only required to stay within the three steps' reach of each other
    np.testing.assert_allclose(flats[0].cpu().numpy(), flats[1].cpu().numpy(), rtol=2e-3, atol=3.5e-3)


def test_wide_stack_c256_vs_oracle():
    """The C = 2 -> 256 stack of the north_star wording (channels [64, 128, 256], h_dim 256): layer 1 on the fused
    kernels, the wider layers on the mixing kernels + the strided MFMA GEMM (1x1 convolutions and their gradients) + the
    BatchNorm / residual / PReLU kernels of csrc/wide.hip; outputs, every gradient and the running statistics against the oracle."""
    from coskad_amd.models.sts.ae import STSE
    from oracle import ref_cpu as R
    st = R.init_stse_state(2, (64, 128, 256), 256, 16, 12, 17, seed=2)
    st["c"] = torch.full((16,), 0.05)
    x = R.synthetic_clips(6, seed=8)
    m = STSE(2, [64, 128, 256], 256, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    assert set(m.state_dict()) == set(st)
    m.load_state_dict(st, strict=True)
    m.cuda()
    assert [l.is_wide for l in m.encoder.model] == [False, True, True, True]
    m.eval()
    with torch.no_grad():
        z = m(x.cuda())
        zr = R.stse_encode(x, {k: v.clone() for k, v in st.items()}, training=False)
    np.testing.assert_allclose(z.cpu().numpy(), zr.numpy(), rtol=1e-4, atol=1e-4)
    m.train()
    z = m(x.cuda())
    loss = torch.nn.functional.mse_loss(z, st["c"].cuda().expand_as(z))
    loss.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    sto = dict(st)
    sto.update(params)
    zt = R.stse_encode(x, sto, training=True)
    lref = R.mse_to_center(zt, st["c"])
    lref.backward()
    np.testing.assert_allclose(float(loss.detach()), float(lref.detach()), rtol=1e-4)
    for k, p in m.named_parameters():
        if k.endswith(("tcn.0.bias", "residual.0.bias")):
            continue
        ref = params[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-3, atol=1e-3 * np.abs(ref).max() + 1e-9, err_msg=k)
    for k, v in sto.items():
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(m.state_dict()[k].cpu().numpy(), v.detach().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_autograd_train_step_matches_fast_path_and_handles_mlp():
    """AutogradTrainStep (module surface + torch Adam) == STSETrainStep (flat buffers + HIP Adam) on a model both take;
    make_train_step routes the plain-GCN encoders to it (the `mlp` projector and wide stacks stay on the flat step)."""
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import AutogradTrainStep, STSETrainStep, make_train_step
    from oracle import ref_cpu as R
    x = R.synthetic_clips(48, seed=4).cuda()
    outs = []
    for cls in (STSETrainStep, AutogradTrainStep):
        st = R.init_stse_state(2, (8, 4, 8), 8, 8, 12, 17, seed=1)
        st["c"] = torch.full((8,), 0.1)
        m = STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st, strict=True)
        eng = cls(m.cuda().train(), lr=1e-3, alpha=1e-4, head='euclidean')
        losses = [float(eng.step(x)[0]) for _ in range(3)]
        outs.append((losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, float(eng.reg_loss())))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-5)
    for k in outs[0][1]:
        if k.endswith(("tcn.0.bias", "residual.0.bias")):
            continue      # zero-gradient biases: torch's autograd noise makes torch Adam move them by +-lr
        np.testing.assert_allclose(outs[0][1][k].float().numpy(), outs[1][1][k].float().numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
    for kw in (dict(projector='mlp'), dict(encoder_type='learnable_gcn')):
        args = dict(encoder_type='sts_gcn', projector='linear')
        args.update(kw)
        m = STSE(2, [8, 4, 8], 8, 8, 12, 17, args['encoder_type'], args['projector'], 'euclidean', 0.0).cuda().train()
        eng = make_train_step(m, lr=1e-3, alpha=1e-6, head='euclidean')
        # the `mlp` projector (what 5 of the 7 reference yamls select) stays on the flat-buffer HIP step
        assert isinstance(eng, STSETrainStep if kw.get('projector') == 'mlp' else AutogradTrainStep)
        l0 = float(eng.step(x)[0])
        for _ in range(5):
            l1 = float(eng.step(x)[0])
        assert np.isfinite(l1) and l1 < l0
    assert isinstance(make_train_step(STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda(), lr=1e-3), STSETrainStep)


def test_eval_fold_cache_tracks_weight_changes(golden):
    """Eval-mode folded weights are cached per layer and refreshed when any input of the fold changes in place."""
    g = golden("stse_default.npz")
    m, st = build_stse(g)
    x = torch.from_numpy(g["x"]).cuda()
    m.eval()
    with torch.no_grad():
        z0 = m(x).clone()
        z1 = m(x)                                   # served from the cache
        assert torch.equal(z0, z1)
        layer = m.encoder.model[1]
        cache = layer.__dict__["_fold_cache"]
        assert "fold" in cache or "fused" in cache      # per-layer fold, or the fused encoder's operand streams
        layer.tcn[1].running_var.mul_(1.5)          # in-place change of a BN buffer -> stale cache must not be used
        z2 = m(x)
        assert not torch.allclose(z0, z2)
        layer.tcn[1].running_var.div_(1.5)
        np.testing.assert_allclose(m(x).cpu().numpy(), z0.cpu().numpy(), rtol=1e-5, atol=1e-6)
        layer.tcn[0].weight.data.copy_(layer.tcn[0].weight.data * 1.0)      # version bump, same values
        assert torch.equal(m(x), m(x))


def test_stsvae_v25_default_width_vs_reference_golden(golden):
    """BASELINE config 4's model (spherical VAE, 25 joints, default widths 32-16-32 / 64, latent 8): encoder, mean head
    and decoder pinned by the REFERENCE's STSAE outputs on the same weights -- with the `linear` projector the VAE's
    `fc_mean` is the autoencoder's bottleneck (vae.py:147-150) and its decoder is the autoencoder's (vae.py:93-132) --
    then one stochastic training step through the whole wrapper loss."""
    from coskad_amd.models.sts.vae import STSVAE, kl_ps_uniform
    g = golden("stsae_v25.npz")
    st = state_from(g)
    m = STSVAE(2, [32, 16, 32], 64, 8, 12, 25, 'sts_gcn', 'linear', 'euclidean', 0.0, distribution='ps')
    own = m.state_dict()
    for k, v in st.items():
        if k.startswith(("encoder.", "decoder.", "rev_btlnk.")):
            assert own[k].shape == v.shape, k
            own[k] = v
    own["fc_mean.weight"], own["fc_mean.bias"] = st["btlnk.weight"], st["btlnk.bias"]
    m.load_state_dict(own, strict=True)
    m.cuda().eval()
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        zm, zv = m.encode(x)
        zref = torch.from_numpy(g["eval.z"]).cuda()
        np.testing.assert_allclose(zm.cpu().numpy(), (zref / zref.norm(dim=-1, keepdim=True)).cpu().numpy(), rtol=1e-4, atol=1e-4)
        assert bool((zv >= 1).all())                                                  # softplus + 1 (vae.py:85)
        # the concentration head against the oracle's heads on the oracle's own flattened encoder output (vae.py:79-85)
        from oracle import ref_cpu as R
        sto = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        flat = R.stse_encode(x.cpu(), sto, training=False)
        zm_o, zv_o = R.stsvae_heads(flat, sto, 'ps')
        np.testing.assert_allclose(zv.cpu().numpy(), zv_o.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(zm.cpu().numpy(), zm_o.numpy(), rtol=1e-4, atol=1e-4)
        xr = m.decode(zref, (x.shape[0], 64, 12, 25, 1))
        np.testing.assert_allclose(xr.cpu().numpy(), g["eval.xrec"], rtol=1e-4, atol=1e-4)
    m.train()
    torch.manual_seed(0)
    z, xr, (q, p, kappa) = m(x)
    assert xr.shape == x.shape
    np.testing.assert_allclose(z.norm(dim=-1).detach().cpu().numpy(), 1.0, atol=1e-4)
    loss = ((xr - x) ** 2).mean() + kl_ps_uniform(q, p).mean() + (1 / kappa).mean()    # spherical_vae.py:81-107
    loss.backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in m.parameters())


@pytest.mark.parametrize("hidden", [None, [12, 10]])
def test_mlp_projector_hip_path_vs_oracle(hidden):
    """projector='mlp' (components.py:209-226 intent): wide Linear on the bottleneck kernels + [BatchNorm1d, ReLU, Linear]
    blocks on csrc/mlp_head.hip.  Forward (train + eval), every gradient, running statistics and one optimisation step
    against the CPU oracle (`ref_cpu.mlp`), through both the autograd surface and the flat-buffer STSETrainStep."""
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    from oracle import ref_cpu as R
    torch.manual_seed(3)
    kw = {} if hidden is None else dict(projector_hidden_layers=hidden)
    m = STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'mlp', 'euclidean', 0.0, **kw)
    with torch.no_grad():                         # non-trivial BN statistics / affine
        for mod in m.btlnk.net:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.add_(0.2 * torch.randn_like(mod.weight)); mod.bias.add_(0.1 * torch.randn_like(mod.bias))
                mod.running_mean.add_(0.1 * torch.randn_like(mod.running_mean)); mod.running_var.mul_(1.3)
    m.c.copy_(torch.linspace(-0.1, 0.1, 8))
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = R.synthetic_clips(40, seed=6)
    m.cuda()
    # eval forward
    m.eval()
    with torch.no_grad():
        z = m(x.cuda())
        z_ref = R.stse_encode(x, {k: v.clone() for k, v in st.items()}, training=False)
    np.testing.assert_allclose(z.cpu().numpy(), z_ref.numpy(), rtol=1e-4, atol=1e-4)
    # train forward + backward through the autograd surface
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    so = {k: v.clone() for k, v in st.items()}
    so.update(params)
    zo = R.stse_encode(x, so, training=True)
    lo = R.mse_to_center(zo, st["c"])
    lo.backward()
    m.train()
    zt = m(x.cuda())
    np.testing.assert_allclose(zt.detach().cpu().numpy(), zo.detach().numpy(), rtol=2e-4, atol=2e-4)
    loss = ((zt - m.c) ** 2).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-4)
    gmax = max(float(p.grad.abs().max()) for p in params.values())
    for n, p in m.named_parameters():
        ref = params[n].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max() + 5e-5 * gmax, err_msg=n)
    for k, v in so.items():                       # running statistics after the step (BatchNorm1d: unbiased variance)
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(m.state_dict()[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    # the flat-buffer step: same loss and same parameters as torch Adam on the oracle's gradients
    m2 = STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'mlp', 'euclidean', 0.0, **kw)
    m2.load_state_dict(st, strict=True)
    eng = STSETrainStep(m2.cuda().train(), lr=1e-3, alpha=0.0, head='euclidean')
    stats = eng.step(x.cuda())
    np.testing.assert_allclose(float(stats[0]), lo.item(), rtol=1e-4)
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    opt.step()
    new = m2.state_dict()
    for k, p in params.items():
        if k.endswith(("tcn.0.bias", "residual.0.bias", "btlnk.net.0.bias")) or (k.startswith("btlnk.net.") and k.endswith("bias")
                                                                                 and k[:-4] + "weight" in st and st[k[:-4] + "weight"].dim() == 2
                                                                                 and k != max((q for q in st if q.startswith("btlnk.net.") and q.endswith(".bias") and st[q[:-4] + "weight"].dim() == 2))):
            continue   # Linear biases in front of a train-mode BatchNorm: analytically zero gradient (autograd noise moves torch's Adam)
        np.testing.assert_allclose(new[k].cpu().numpy(), p.detach().numpy(), rtol=2e-3, atol=3e-4, err_msg=k)


@pytest.mark.parametrize("B,H,L", [(203, 16, 16), (4099, 16, 16), (130, 12, 7), (64, 8, 8), (77, 24, 16)])
def test_mlp_head_kernels_vs_torch(B, H, L):
    """csrc/mlp_head.hip through the C ABI -- the 64-rows-per-block kernels of the hidden, out <= 16 shapes (several blocks, a ragged
    last one, padded columns) and the general kernels (24 features) -- against torch's BatchNorm1d -> ReLU -> Linear: train-mode
    forward, every gradient, the running statistics; eval-mode forward and input gradient."""
    from coskad_amd import ops
    g = torch.Generator().manual_seed(B + H)
    y = torch.randn(B, H, generator=g) * 1.3 + 0.4
    bn = torch.nn.BatchNorm1d(H)
    lin = torch.nn.Linear(H, L)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(H, generator=g)); bn.bias.copy_(0.2 * torch.randn(H, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(H, generator=g)); bn.running_var.copy_(1 + 0.2 * torch.rand(H, generator=g))
    probe = torch.randn(B, L, generator=g)
    dev = lambda t: t.detach().clone().cuda().contiguous()
    for training in (True, False):
        bn.train(training)
        rm, rv, nb = dev(bn.running_mean), dev(bn.running_var), bn.num_batches_tracked.detach().clone().cuda()
        yr = y.clone().requires_grad_(True)
        for p in list(bn.parameters()) + list(lin.parameters()):
            p.grad = None
        zr = lin(torch.relu(bn(yr)))
        (zr * probe).sum().backward()
        z, stat = ops.mlp_head_fwd(dev(y), dev(bn.weight), dev(bn.bias), rm, rv, nb, dev(lin.weight), dev(lin.bias), training,
                                   momentum=bn.momentum, eps=bn.eps)
        np.testing.assert_allclose(z.cpu().numpy(), zr.detach().numpy(), rtol=1e-4, atol=1e-4)
        nanf = lambda *s_: torch.full(s_, float("nan"), device="cuda")
        gr = {"gamma": nanf(H), "beta": nanf(H), "W2": nanf(L, H), "b2": nanf(L)}
        dy = ops.mlp_head_bwd(dev(y), stat, dev(bn.weight), dev(bn.bias), dev(lin.weight), dev(probe), gr, training)
        torch.cuda.synchronize()
        scale = float(yr.grad.abs().max())
        np.testing.assert_allclose(dy.cpu().numpy(), yr.grad.numpy(), rtol=1e-3, atol=1e-4 * scale + 1e-6)
        for name, ref in (("gamma", bn.weight.grad), ("beta", bn.bias.grad), ("W2", lin.weight.grad), ("b2", lin.bias.grad)):
            a, b = gr[name].cpu().numpy(), ref.numpy()
            assert np.isfinite(a).all(), name
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * np.abs(b).max() + 1e-6, err_msg=name)
        if training:
            np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=1e-4, atol=1e-6)
            assert int(nb) == int(bn.num_batches_tracked)


def _randomise_bn(m):
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.weight.add_(0.2 * torch.randn_like(mod.weight)); mod.bias.add_(0.1 * torch.randn_like(mod.bias))
                mod.running_mean.add_(0.1 * torch.randn_like(mod.running_mean)); mod.running_var.mul_(1.3)


@pytest.mark.parametrize("B", [5, 261])
def test_fused_eval_forward_with_mlp_projector(B):
    """`projector: 'mlp'` (5 of the reference's 7 yamls; components.py:209-226) on the eval-mode fast path: ONE fused encoder
    kernel, the first Linear on its tile-major output, the [BatchNorm1d, ReLU, Linear] block on csrc/mlp_head.hip --
    against the CPU oracle and against the layer-by-layer path of the same module."""
    from coskad_amd.models.sts.ae import STSE
    from oracle import ref_cpu as R
    torch.manual_seed(11)
    m = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'mlp', 'euclidean', 0.0)
    _randomise_bn(m)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = R.synthetic_clips(B, seed=12)
    m.cuda().eval()
    with torch.no_grad():
        assert m._project_fused(x.cuda()) is not None, "the fused path must take the default geometry with an mlp projector"
        z = m(x.cuda())
        z_ref = R.stse_encode(x, st, training=False)
    with torch.enable_grad():                     # a gradient is needed -> layer-by-layer kernels
        z_layers = m(x.cuda()).detach()
    np.testing.assert_allclose(z.cpu().numpy(), z_ref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(z.cpu().numpy(), z_layers.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("projector", ["linear", "mlp"])
def test_fused_eval_vae_heads(projector):
    """STSVAE.encode in eval mode on the fused encoder kernel: `linear` -> fc_mean | fc_var stacked on ONE bottleneck pass
    (vae.py:147-150), `mlp` -> the projector as in STSE, then both heads as one MFMA GEMM (vae.py:141-146); against the oracle's
    heads and the layer-by-layer path."""
    from coskad_amd.models.sts.vae import STSVAE
    from oracle import ref_cpu as R
    torch.manual_seed(5)
    m = STSVAE(2, [32, 16, 32], 64, 8, 12, 17, 'sts_gcn', projector, 'euclidean', 0.0, distribution='ps')
    _randomise_bn(m)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = R.synthetic_clips(37, seed=8)
    m.cuda().eval()
    with torch.no_grad():
        assert m._heads_fused(x.cuda()) is not None
        zm, zv = m.encode(x.cuda())
        flat = R.stse_encode(x, st, training=False)          # linear: the flattened encoder output; mlp: the projector's output
        zm_ref, zv_ref = R.stsvae_heads(flat, st, 'ps')
    with torch.enable_grad():
        zm_l, zv_l = m.encode(x.cuda())
    np.testing.assert_allclose(zm.cpu().numpy(), zm_ref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(zv.cpu().numpy(), zv_ref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(zm.cpu().numpy(), zm_l.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(zv.cpu().numpy(), zv_l.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("Ci,Co", [(16, 32), (16, 16)])
def test_train_mode_dropout_matches_oracle_with_the_same_mask(Ci, Co):
    """nn.Dropout(p) at the end of `tcn` (stsgcn.py:66) in training mode: the layer's counter-based mask, recomputed in the
    backward, against the oracle with the SAME explicit mask (forward, every gradient, running statistics); eval mode is the
    identity; the mask has the right rate and scale and changes with the seed."""
    from coskad_amd import ops
    from coskad_amd.models.graph_layers.stsgcn import ST_GCNN_layer
    from oracle import ref_cpu as R
    torch.manual_seed(4)
    p_drop, B, T, V = 0.3, 9, 12, 17
    layer = ST_GCNN_layer(Ci, Co, (1, 1), 1, T, V, p_drop)
    st = {"l." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    x = torch.randn(B, Ci, T, V)
    layer.cuda().train()
    seed = 123456789
    layer._dropout_args = lambda: (p_drop, seed)
    mask = ops.dropout_mask((B, Co, T, V), p_drop, seed, "cuda")
    vals = torch.unique(mask).cpu().numpy()
    np.testing.assert_allclose(vals, [0.0, 1 / (1 - p_drop)], rtol=1e-6)
    n = mask.numel()
    assert abs(float((mask == 0).float().mean()) - p_drop) < 4 * (p_drop * (1 - p_drop) / n) ** 0.5
    assert not torch.equal(mask, ops.dropout_mask((B, Co, T, V), p_drop, seed + 1, "cuda"))
    xg = x.cuda().requires_grad_(True)
    out = layer(xg)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).cuda()
    (out * w).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    so = dict(st)
    so.update(params)
    xo = x.clone().requires_grad_(True)
    ref = R.st_gcnn_layer(xo, so, "l", training=True, drop_mask=mask.cpu())
    (ref * w.cpu()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=1e-3, atol=1e-4)
    gmax = max(float(p_.grad.abs().max()) for p_ in params.values())
    for name, prm in layer.named_parameters():
        r = params["l." + name].grad.numpy()       # (conv biases in front of a train-mode BatchNorm: exactly 0 here, rounding noise in autograd)
        np.testing.assert_allclose(prm.grad.cpu().numpy(), r, rtol=2e-3, atol=2e-4 * np.abs(r).max() + 5e-5 * gmax, err_msg=name)
    for k, v in layer.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), so["l." + k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    # two training forwards draw different masks; eval mode: no dropout at all
    del layer._dropout_args
    torch.manual_seed(0)
    a, b = layer(x.cuda()).detach(), layer(x.cuda()).detach()
    assert not torch.equal(a, b)
    layer.eval()
    with torch.no_grad():
        e = layer(x.cuda())
        ref_e = R.st_gcnn_layer(x, {k: v.cpu() for k, v in {"l." + k: v for k, v in layer.state_dict().items()}.items()}, "l", training=False)
    np.testing.assert_allclose(e.cpu().numpy(), ref_e.numpy(), rtol=1e-4, atol=1e-4)


def test_train_step_captured_in_a_hip_graph_equals_the_eager_step(golden):
    """STSETrainStep(use_graph=True): the whole step (apply + next-layer statistics kernels, folds with their table-building
    blocks, fused backward, Adam) captured once in a hipGraph and replayed -- same parameters as the eager step after three steps
    on changing inputs (every kernel argument that changes between steps lives in device memory)."""
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    from oracle import ref_cpu as R
    g = golden("stse_default.npz")
    st = state_from(g)
    xs = [R.synthetic_clips(64, seed=20 + i).cuda() for i in range(3)]
    res = []
    for use_graph in (False, True):
        m = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st, strict=True)
        m.c.fill_(0.05)
        eng = STSETrainStep(m.cuda().train(), lr=1e-3, alpha=1e-4, head='euclidean', use_graph=use_graph)
        # the first graph-mode call takes the step twice on its input (an eager warm-up outside the capture, then the replay)
        seq = xs if use_graph else [xs[0]] + xs
        losses = [float(eng.step(x)[0]) for x in seq]
        torch.cuda.synchronize()
        res.append((losses[-2:], {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
    (l0, s0), (l1, s1) = res
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    for k in s0:
        np.testing.assert_allclose(s1[k].numpy(), s0[k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def _torch_twin(layer):
    """pure-torch CPU twin of an ST_GCNN_layer (reference composition, stsgcn.py:94-116) sharing its BatchNorm configuration"""
    import copy
    from oracle import ref_cpu as R
    tcn, res, prelu = copy.deepcopy(layer.tcn).cpu(), copy.deepcopy(layer.residual).cpu(), copy.deepcopy(layer.prelu).cpu()
    A, T = layer.gcn.A.detach().cpu().clone().requires_grad_(True), layer.gcn.T.detach().cpu().clone().requires_grad_(True)

    def fwd(x):
        return prelu(tcn(R.gcn(x, A, T)) + res(x))
    return fwd, tcn, res, (A, T)


@pytest.mark.gpu
@pytest.mark.parametrize("Ci,Co,V", [(16, 32, 17), (32, 16, 25), (64, 128, 17)])
@pytest.mark.parametrize("mode", ["cumulative", "untracked"])
def test_layer_batchnorm_non_default_configurations(Ci, Co, V, mode):
    """nn.BatchNorm2d(momentum=None) (cumulative moving average: factor 1 / num_batches_tracked) and track_running_stats=False
    (batch statistics in eval mode too) on the tile kernels (16 -> 32, 32 -> 16) and the wide path (64 -> 128), against torch's
    own modules composed as the reference composes them (stsgcn.py:56-80,94-116; torch/nn/modules/batchnorm.py)."""
    from coskad_amd.models.graph_layers.stsgcn import ST_GCNN_layer
    torch.manual_seed(3)
    layer = ST_GCNN_layer(Ci, Co, [1, 1], 1, 12, V, 0.0)
    kw = dict(momentum=None) if mode == "cumulative" else dict(track_running_stats=False)
    for seq in (layer.tcn, layer.residual):
        bn = torch.nn.BatchNorm2d(Co, **kw)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        seq[1] = bn
    fwd, tcn, res, _ = _torch_twin(layer)
    layer.cuda().train()
    for m in (tcn, res):
        m.train()
    g = torch.Generator().manual_seed(5)
    for step in range(3):                                     # the averaging factor changes every step: 1, 1/2, 1/3
        x = torch.randn(6 + step, Ci, 12, V, generator=g) * (1.0 + step)
        out = layer(x.cuda())
        ref = fwd(x)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
        for mine, theirs in ((layer.tcn[1], tcn[1]), (layer.residual[1], res[1])):
            if mode == "cumulative":
                assert int(mine.num_batches_tracked) == int(theirs.num_batches_tracked) == step + 1
                np.testing.assert_allclose(mine.running_mean.cpu().numpy(), theirs.running_mean.numpy(), rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(mine.running_var.cpu().numpy(), theirs.running_var.numpy(), rtol=1e-4, atol=1e-5)
            else:
                assert mine.running_mean is None and mine.num_batches_tracked is None
    # gradients of the last training forward
    xg = torch.randn(5, Ci, 12, V, generator=g)
    w = torch.randn(5, Co, 12, V, generator=g)
    xa = xg.clone().cuda().requires_grad_(True)
    (layer(xa) * w.cuda()).sum().backward()
    xb = xg.clone().requires_grad_(True)
    (fwd(xb) * w).sum().backward()
    scale = float(xb.grad.abs().max())
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.numpy(), rtol=2e-3, atol=2e-4 * scale)
    np.testing.assert_allclose(layer.tcn[0].weight.grad.cpu().numpy(), tcn[0].weight.grad.numpy(), rtol=2e-3,
                               atol=2e-4 * float(tcn[0].weight.grad.abs().max()))
    # eval mode: running statistics (cumulative) / batch statistics (untracked)
    layer.eval(); tcn.eval(); res.eval()
    xe = torch.randn(7, Ci, 12, V, generator=g)
    with torch.no_grad():
        np.testing.assert_allclose(layer(xe.cuda()).cpu().numpy(), fwd(xe).numpy(), rtol=1e-4, atol=1e-4)
    # load_state_dict (torch writes num_batches_tracked: its version moves) restarts the host's mirror of the counter
    if mode == "cumulative":
        sd = {k: v.clone() for k, v in layer.state_dict().items()}
        sd["tcn.1.num_batches_tracked"].fill_(9); sd["residual.1.num_batches_tracked"].fill_(9)
        layer.load_state_dict(sd)
        tcn[1].num_batches_tracked.fill_(9); res[1].num_batches_tracked.fill_(9)
        layer.train(); tcn.train(); res.train()
        out, ref = layer(xe.cuda()), fwd(xe)
        np.testing.assert_allclose(layer.tcn[1].running_var.cpu().numpy(), tcn[1].running_var.numpy(), rtol=1e-4, atol=1e-5)
        assert int(layer.tcn[1].num_batches_tracked) == 10


@pytest.mark.gpu
def test_mlp_projector_batchnorm_momentum_none_stays_on_the_kernels():
    """BatchNorm1d(momentum=None) inside the `mlp` projector: HIP path (hip_ok), values and running statistics as torch's"""
    from coskad_amd.models.common.components import MLP
    torch.manual_seed(1)
    mlp = MLP(64 * 12 * 17, 16, [16])
    mlp.net[1] = torch.nn.BatchNorm1d(16, momentum=None)
    assert mlp.hip_ok
    import copy
    twin = copy.deepcopy(mlp.net).train()
    mlp.cuda().train()
    from coskad_amd.models.sts.ae import _BottleneckFn
    from coskad_amd import engine
    ws = engine.Workspace()
    g = torch.Generator().manual_seed(2)
    for step in range(2):
        U = torch.randn(9, 64, 12, 17, generator=g)
        z = mlp.forward_preact(U.cuda(), None, ws, _BottleneckFn.apply)
        ref = twin(U.reshape(9, -1))
        np.testing.assert_allclose(z.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(mlp.net[1].running_var.cpu().numpy(), twin[1].running_var.numpy(), rtol=1e-4, atol=1e-6)
        assert int(mlp.net[1].num_batches_tracked) == step + 1


@pytest.mark.gpu
@pytest.mark.parametrize("chans,hid", [((64, 128, 256), 256), ((32, 128, 16), 64)])
def test_flat_train_step_on_wide_stacks(chans, hid):
    """STSETrainStep (flat buffers, fused Adam, explicit wide_forward / wide_backward) on stacks with layers beyond the tile
    kernels -- the C = 2 -> 256 stack (wide layers last) and a stack with a wide layer BETWEEN tile runs: the gradients it leaves in
    the flat buffer equal the oracle's autograd (reference components.py:70-105, stsgcn.py:94-116), and three steps equal
    AutogradTrainStep's (module surface + torch Adam) parameters and running statistics; make_train_step selects it."""
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import AutogradTrainStep, STSETrainStep, make_train_step
    from oracle import ref_cpu as R
    st = R.init_stse_state(2, chans, hid, 16, 12, 17, seed=2)
    st["c"] = torch.full((16,), 0.05)
    x = R.synthetic_clips(6, seed=8)

    def build():
        m = STSE(2, list(chans), hid, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict({k: v.clone() for k, v in st.items()}, strict=True)
        return m.cuda().train()

    m = build()
    assert any(l.is_wide for l in m.encoder.model)
    eng = make_train_step(m, lr=0.0, alpha=0.0, head='euclidean')           # lr 0: the step leaves the gradients, not an update
    assert isinstance(eng, STSETrainStep) and eng.stack is not None
    stats = eng.step(x.cuda())
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    sto = dict(st)
    sto.update(params)
    lref = R.mse_to_center(R.stse_encode(x, sto, training=True), st["c"])
    lref.backward()
    np.testing.assert_allclose(float(stats[0]), float(lref.detach()), rtol=1e-4)
    for k in eng.fp.names:
        if k.endswith(("tcn.0.bias", "residual.0.bias")):
            continue
        ref = params[k].grad.numpy()
        np.testing.assert_allclose(eng.fp.gviews[k].cpu().numpy(), ref, rtol=5e-3, atol=1e-3 * np.abs(ref).max() + 1e-9, err_msg=k)
    outs = []
    for cls in (STSETrainStep, AutogradTrainStep):
        m = build()
        e = cls(m, lr=1e-3, alpha=1e-4, head='euclidean')
        losses = [float(e.step(x.cuda())[0]) for _ in range(3)]
        outs.append((losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-4)
    for k in outs[0][1]:
        if k.endswith(("tcn.0.bias", "residual.0.bias")):
            continue
        a, b = outs[0][1][k].float().numpy(), outs[1][1][k].float().numpy()
        np.testing.assert_allclose(a, b, rtol=5e-3, atol=5e-3 * 1e-3 + 1e-4 * np.abs(b).max(), err_msg=k)
