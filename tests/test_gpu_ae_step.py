"""GPU parity of the flat-buffer train step of the decoder models (coskad_amd.trainer.STSAETrainStep: encoder, bottleneck,
rev_btlnk, decoder, reconstruction head, everything backwards, fused Adam -- no autograd around the chains) against the
REFERENCE's gradients (tests/golden/stsae_*.npz, generated from models/sts/ae.py::STSAE) and against the module-surface
autograd path of the same library for the spherical VAE (whose sampler is not in the reference snapshot)."""
import numpy as np
import pytest
import torch

from conftest import state_from

pytestmark = pytest.mark.gpu


def _build(g, cls, **kw):
    st = state_from(g)
    chans, i = [], 0
    while f"encoder.model.{i}.tcn.0.weight" in st:
        chans.append(st[f"encoder.model.{i}.tcn.0.weight"].shape[0])
        i += 1
    V = st["encoder.model.0.gcn.A"].shape[1]
    m = cls(input_dim=2, layer_channels=chans[:-1], hidden_dimension=chans[-1], latent_dim=st["c"].shape[0], n_frames=12,
            n_joints=V, encoder_type='STS_GCN', projector='linear', distance='euclidean', dropout=0.0, **kw)
    return m, st


@pytest.mark.parametrize("name", ["stsae_small.npz", "stsae_v25.npz"])
def test_flat_autoencoder_step_matches_reference_gradients(golden, name):
    """loss = MSE(x_rec, x) + MSE(z, c) (euclidean_autoencoder.py:106-118 with lambda_ = 1, alpha = 0): losses, every parameter
    gradient and the BatchNorm running statistics after the step against the reference's own autograd run.  stsae_v25 is
    BASELINE config 4's geometry: its first decoder layer (64 channels on 25 joints) is beyond the tile kernels and runs the
    composed path inside the flat step."""
    from coskad_amd.models.sts.ae import STSAE
    from coskad_amd.trainer import STSAETrainStep
    g = golden(name)
    m, st = _build(g, STSAE)
    m.load_state_dict(st, strict=True)
    m.c.copy_(torch.from_numpy(g["c"]))          # the centre the reference's loss was taken against
    m.cuda().train()
    eng = STSAETrainStep(m, mode='ae', lr=0.0, alpha=0.0, lambda_=1.0)
    x = torch.from_numpy(g["x"]).cuda()
    out = eng.step(x)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(out['rec']) + float(out['head']), g["train.loss"], rtol=1e-4)
    np.testing.assert_allclose(out['z'].cpu().numpy(), g["train.z"], rtol=2e-4, atol=2e-4)
    names = [n for n, _ in m.named_parameters()]
    gmax = max(np.abs(g["grad." + n]).max() for n in names)
    for n in names:
        ref = g["grad." + n]
        np.testing.assert_allclose(eng.fp.gviews[n].cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max() + 5e-5 * gmax, err_msg=n)
    sd = m.state_dict()
    for k, v in g.items():
        if k.startswith("sd1.") and ("running" in k or "num_batches" in k) and k[4:] in sd:
            np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), v, rtol=1e-4, atol=1e-5, err_msg=k)
    for n in names:                                  # lr = 0: the fused Adam left every parameter where it was
        assert torch.equal(sd[n].cpu(), st[n]), n



def test_flat_autoencoder_step_equals_torch_adam_on_the_module_surface(golden):
    """two steps with lr > 0 and alpha > 0: the flat step (fused Adam, regulariser folded into it) and the module-surface path
    (autograd + calc_reg_loss in the loss + torch.optim.Adam) end with the same parameters."""
    from coskad_amd.models.sts.ae import STSAE
    from coskad_amd.trainer import STSAETrainStep
    g = golden("stsae_small.npz")
    x = torch.from_numpy(g["x"]).cuda()
    c = torch.from_numpy(g["c"])
    res = []
    for flat in (True, False):
        m, st = _build(g, STSAE)
        m.load_state_dict(st, strict=True)
        m.c.copy_(c)
        m.cuda().train()
        if flat:
            eng = STSAETrainStep(m, mode='ae', lr=1e-3, alpha=1e-3, lambda_=0.5)
            for _ in range(2):
                eng.step(x)
        else:
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                z, xr = m(x)
                ps = [p for n, p in m.named_parameters() if 'bias' not in n]
                reg = 0.5 * sum((p ** 2).sum() for p in ps) / len(ps)
                (0.5 * ((xr - x) ** 2).mean() + ((z - m.c) ** 2).mean() + 1e-3 * reg).backward()
                opt.step()
        torch.cuda.synchronize()
        res.append({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    for k in res[0]:
        if k.endswith(("tcn.0.bias", "residual.0.bias")):
            continue        # analytically-zero gradients: autograd's rounding noise random-walks them under Adam
        np.testing.assert_allclose(res[0][k].numpy(), res[1][k].numpy(), rtol=2e-3, atol=3e-4, err_msg=k)


@pytest.mark.parametrize("V,projector", [(17, 'linear'), (25, 'linear'), (25, 'mlp')])
def test_flat_vae_step_matches_module_autograd(V, projector):
    """spherical VAE (spherical_vae.py:81-107): phi * MSE + beta * KL + gamma * mean(1 / kappa).  The flat step and the
    module-surface autograd path draw the same PowerSpherical sample under the same torch seed; losses and every gradient
    must agree (the encoder / decoder themselves are pinned by the reference goldens in the test above)."""
    from coskad_amd.models.sts.vae import STSVAE, kl_ps_uniform
    from coskad_amd.trainer import STSAETrainStep
    from oracle import ref_cpu as R
    torch.manual_seed(3)
    proto = STSVAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', projector, 'euclidean', 0.0, distribution='ps')
    st = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    x = R.synthetic_clips(24, 2, 12, V, seed=4).cuda()
    phi, beta, gamma = 0.7, 0.3, 0.2
    m1 = STSVAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', projector, 'euclidean', 0.0, distribution='ps')
    m1.load_state_dict(st)
    m1.cuda().train()
    torch.manual_seed(11)
    z, xr, (q, p, kappa) = m1(x)
    l_rec, l_kl, l_exp = ((xr - x) ** 2).mean(), kl_ps_uniform(q, p).mean(), (1 / kappa).mean()
    (phi * l_rec + beta * l_kl + gamma * l_exp).backward()
    m2 = STSVAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', projector, 'euclidean', 0.0, distribution='ps')
    m2.load_state_dict(st)
    m2.cuda().train()
    eng = STSAETrainStep(m2, mode='vae', lr=0.0, alpha=0.0, phi=phi, beta=beta, gamma=gamma)
    torch.manual_seed(11)
    out = eng.step(x)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out['z'].cpu().numpy(), z.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(out['rec']), float(l_rec), rtol=1e-4)
    np.testing.assert_allclose(float(out['head']), float(l_kl), rtol=1e-4)
    np.testing.assert_allclose(float(out['exp']), float(l_exp), rtol=1e-4)
    if projector == 'mlp':                           # (spherical_vae.yaml:37 selects the mlp projector) running statistics of its BatchNorm1d
        for (k1, b1), (k2, b2) in zip(m1.named_buffers(), m2.named_buffers()):
            if "btlnk" in k1:
                np.testing.assert_allclose(b2.cpu().numpy(), b1.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k1)
    grads = {n: p.grad for n, p in m1.named_parameters()}
    gmax = max(float(v.abs().max()) for v in grads.values() if v is not None)
    for n, ref in grads.items():
        if ref is None:
            continue
        r = ref.cpu().numpy()
        np.testing.assert_allclose(eng.fp.gviews[n].cpu().numpy(), r, rtol=2e-3, atol=2e-4 * np.abs(r).max() + 5e-5 * gmax, err_msg=n)


def test_flat_autoencoder_step_default_widths_vs_oracle():
    """The reference's DEFAULT widths (32-16-32, hidden 64, latent 16) on 17 joints: the decoder 64 -> 32 -> 16 -> 32 -> 2 runs on
    the tile kernels with the apply + next-layer-statistics fusion for its middle layers; losses, every gradient and the
    BatchNorm running statistics of one flat step against the CPU oracle's autograd (oracle/ref_cpu.py, pinned by the goldens)."""
    from coskad_amd.models.sts.ae import STSAE
    from coskad_amd.trainer import STSAETrainStep
    from oracle import ref_cpu as R
    st = R.init_stse_state(seed=7, decoder=True)
    st["c"] = torch.linspace(-0.1, 0.1, 16)
    x = R.synthetic_clips(10, seed=8)
    m = STSAE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    m.load_state_dict(st, strict=True)
    eng = STSAETrainStep(m.cuda().train(), mode='ae', lr=0.0, alpha=0.0, lambda_=0.3)
    out = eng.step(x.cuda())
    torch.cuda.synchronize()
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    so = {k: v.clone() for k, v in st.items()}
    so.update(params)
    z = R.stse_encode(x, so, training=True)
    xr = R.stsae_decode(z, so, 64, 12, 17, training=True)
    l_rec, l_h = ((xr - x) ** 2).mean(), R.mse_to_center(z, st["c"])
    (0.3 * l_rec + l_h).backward()
    np.testing.assert_allclose(float(out['rec']), float(l_rec), rtol=1e-4)
    np.testing.assert_allclose(float(out['head']), float(l_h), rtol=1e-4)
    gmax = max(float(p.grad.abs().max()) for p in params.values())
    for n, p in params.items():
        r = p.grad.numpy()
        np.testing.assert_allclose(eng.fp.gviews[n].cpu().numpy(), r, rtol=2e-3, atol=2e-4 * np.abs(r).max() + 5e-5 * gmax, err_msg=n)
    sd = m.state_dict()
    for k, v in so.items():
        if "running" in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


class _GivenDirichlet(torch.autograd.Function):
    """torch.distributions.dirichlet._Dirichlet with the draw handed in: forward returns x, backward is _Dirichlet_backward."""

    @staticmethod
    def forward(ctx, conc, x):
        ctx.save_for_backward(x, conc)
        return x.clone()

    @staticmethod
    def backward(ctx, go):
        x, conc = ctx.saved_tensors
        total = conc.sum(-1, True).expand_as(conc)
        grad = torch._dirichlet_grad(x, conc, total)
        return grad * (go - (x * go).sum(-1, True)), None


@pytest.mark.parametrize("L", [2, 8, 16])
def test_power_spherical_head_kernels_vs_torch_autograd(L):
    """csrc/vae_head.hip against the torch restatement it replaces in the flat VAE step (coskad_amd/models/sts/vae.py: _finish_heads,
    PowerSpherical.rsample / entropy, kl_ps_uniform; reference vae.py:79-91,104-118, spherical_vae.py:86-94) on the SAME noise: sample,
    per-clip KL and 1 / kappa, and the gradients of  <w, z> + w_kl sum kl + w_exp sum 1/kappa  w.r.t. the raw head outputs (columns
    of one [B, L + 1] tensor), incl. rows near the poles and large / small concentrations."""
    import torch.nn.functional as F
    from coskad_amd import ops
    from coskad_amd.models.sts.vae import HypersphericalUniform, PowerSpherical, kl_ps_uniform
    g = torch.Generator().manual_seed(5 + L)
    B = 700
    H2 = torch.randn(B, L + 1, generator=g)
    H2[:, L] = torch.linspace(-6, 25, B)                 # softplus region, identity region (> 20)
    H2[0, :L] = 0.0; H2[0, 0] = 3.0                       # mu = e1: the Householder vector degenerates (F.normalize's eps)
    H2[1, :L] = 0.0; H2[1, 0] = -2.0                      # mu = -e1
    w = torch.randn(B, L, generator=g)
    w_kl, w_exp = 0.3 / B, 0.2 / B
    dev = torch.device("cuda")
    H2d = H2.to(dev)
    z, kl, ik, saved = ops.ps_head_forward(H2d[:, :L], H2d[:, L:L + 1])
    dH2 = torch.empty_like(H2d)
    ops.ps_head_backward(saved, w.to(dev), w_kl, w_exp, dH2[:, :L], dH2[:, L:L + 1])
    x, eps = saved[6], saved[7]
    # ---- the torch restatement on the same draws -----------------------------------------------------------------------------------
    Hr = H2d.clone().requires_grad_(True)
    Z_mean = Hr[:, :L] / torch.norm(Hr[:, :L], dim=-1, keepdim=True)
    Z_var = F.softplus(Hr[:, L:]) + 1
    q = PowerSpherical(loc=Z_mean, scale=Z_var.squeeze(-1))
    zb = _GivenDirichlet.apply(torch.stack([q.alpha, q.beta], -1), x)[..., 0]
    t = (2 * zb - 1).unsqueeze(-1)
    v = F.normalize(eps, dim=-1)
    y = torch.cat([t, torch.sqrt(torch.clamp(1 - t * t, min=0)) * v], -1)
    e1 = torch.zeros_like(Z_mean)
    e1[..., 0] = 1
    u = F.normalize(e1 - Z_mean, dim=-1)
    z_ref = y - 2 * (y * u).sum(-1, keepdim=True) * u
    kl_ref = kl_ps_uniform(q, HypersphericalUniform(L - 1, device=dev))
    ik_ref = (1 / Z_var).squeeze(-1)
    ((z_ref * w.to(dev)).sum() + w_kl * kl_ref.sum() + w_exp * ik_ref.sum()).backward()
    np.testing.assert_allclose(z.cpu().numpy(), z_ref.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(kl.cpu().numpy(), kl_ref.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(ik.cpu().numpy(), ik_ref.detach().cpu().numpy(), rtol=1e-6)
    ref = Hr.grad.cpu().numpy()
    got = dH2.cpu().numpy()
    ok = np.isfinite(ref).all(axis=1)                     # (mu = e1 exactly: torch's own gradient is 0 / 0 there)
    assert ok.sum() >= B - 2
    np.testing.assert_allclose(got[ok], ref[ok], rtol=2e-3, atol=2e-6 + 2e-4 * np.abs(ref[ok]).max())


@pytest.mark.parametrize("variant", ["kernel", "explicit", "autograd", "graph"])
@pytest.mark.parametrize("name", ["stsae_small.npz", "stsae_v25.npz"])
def test_folded_first_decoder_layer_equals_the_layer_by_layer_path(golden, name, variant, monkeypatch):
    """coskad_amd/lowrank.py (rev_btlnk + the decoder's first layer as ONE streaming pass over a rank-(latent + 1) input) against
    the layer-by-layer flat step on the same model: loss, latents, every gradient, the running statistics -- at 17 joints (where
    the layer would run on the tile kernels) and at 25 (where it would take the composed wide path; the mode the flat step picks
    by itself there: the reference-gradient test above runs through it)."""
    from coskad_amd import lowrank
    from coskad_amd.models.sts.ae import STSAE
    from coskad_amd.trainer import STSAETrainStep
    g = golden(name)
    res = {}
    graph = variant == "graph"
    monkeypatch.setattr(lowrank, "GRAPH_FOLD", graph)      # the fold replayed as two hipGraphs / launched eagerly ...
    monkeypatch.setattr(lowrank, "EXPLICIT", variant in ("explicit", "kernel"))   # ... its backward written out / torch autograd
    monkeypatch.setattr(lowrank, "FOLD_KERNEL", variant == "kernel")              # ... the statistics algebra on csrc/lowrank_fold.hip
    for mode in ("never", "always"):
        monkeypatch.setattr(lowrank, "MODE", mode)
        m, st = _build(g, STSAE)
        m.load_state_dict(st, strict=True)
        m.c.copy_(torch.from_numpy(g["c"]))
        m.cuda().train()
        eng = STSAETrainStep(m, mode='ae', lr=0.0, alpha=0.0, lambda_=0.7)
        assert (eng.lowrank is not None) == (mode == "always")
        out = eng.step(torch.from_numpy(g["x"]).cuda())
        if mode == "always":
            assert (getattr(eng.lowrank, "_graphs", {}).get(float(g["x"].shape[0] * 12 * m.n_joints)) is not None) == graph
        out = eng.step(torch.from_numpy(g["x"]).cuda())       # lr = 0: the second step (a graph REPLAY) repeats the first
        torch.cuda.synchronize()
        res[mode] = (float(out['rec']), float(out['head']), {n: v.cpu().numpy().copy() for n, v in eng.fp.gviews.items()},
                     {k: v.cpu().numpy().copy() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    monkeypatch.setattr(lowrank, "MODE", "wide")
    m, st = _build(g, STSAE)
    assert (STSAETrainStep(m.cuda().train(), mode='ae').lowrank is not None) == (name == "stsae_v25.npz")
    a, b = res["never"], res["always"]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-5)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-5)
    gmax = max(np.abs(v).max() for v in a[2].values())
    for n, ref in a[2].items():
        np.testing.assert_allclose(b[2][n], ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max() + 2e-5 * gmax, err_msg=n)
    for k, ref in a[3].items():
        np.testing.assert_allclose(b[3][k], ref, rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("name,mode", [("stsae_v25.npz", "wide"), ("stsae_small.npz", "always")])
def test_eval_decode_through_the_folded_first_layer(golden, name, mode, monkeypatch):
    """Eval mode: STSAE.decode with rev_btlnk + the first decoder layer as one streaming pass (lowrank.fold_eval, BatchNorm from the
    running statistics) reconstructs what the layer-by-layer decode does -- and what the reference does (golden eval outputs)."""
    from coskad_amd import lowrank
    from coskad_amd.models.sts.ae import STSAE
    g = golden(name)
    m, st = _build(g, STSAE)
    m.load_state_dict(st, strict=True)
    m.cuda().eval()
    for bn in [b for b in m.decoder.modules() if isinstance(b, torch.nn.BatchNorm2d)]:     # non-trivial running statistics
        bn.running_mean.uniform_(-0.3, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        monkeypatch.setattr(lowrank, "MODE", "never")
        z0, r0 = m(x)
        monkeypatch.setattr(lowrank, "MODE", mode)
        assert lowrank.eval_supported(m.rev_btlnk, m.decoder.model[0])
        z1, r1 = m(x)
    assert torch.equal(z0, z1)
    scale = float(r0.abs().max())
    np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
    # and with the reference's own weights and statistics: its eval-mode reconstruction
    m.load_state_dict(st, strict=True)
    with torch.no_grad():
        _, r2 = m(x)
    key = "eval.xrec" if "eval.xrec" in g else None
    if key:
        np.testing.assert_allclose(r2.cpu().numpy(), g[key], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("V,projector", [(17, 'linear'), (25, 'mlp')])
def test_vae_eval_forward_head_on_hip_equals_the_torch_head(V, projector):
    """STSVAE.forward in eval mode without autograd (the scoring forward, spherical_vae.py:76-78) takes csrc/vae_head.hip for the
    normalisation, softplus + 1 and the PowerSpherical sample; with autograd enabled the same call runs the torch restatement: same
    seed -> same sample, reconstruction and concentration."""
    from coskad_amd.models.sts.vae import STSVAE
    from oracle import ref_cpu as R
    torch.manual_seed(7)
    m = STSVAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', projector, 'euclidean', 0.0, distribution='ps').cuda().eval()
    x = R.synthetic_clips(40, 2, 12, V, seed=9).cuda()
    torch.manual_seed(21)
    with torch.no_grad():
        z1, r1, (q1, _, k1) = m(x)
    torch.manual_seed(21)
    z2, r2, (q2, _, k2) = m(x)
    np.testing.assert_allclose(k1.cpu().numpy(), k2.detach().cpu().numpy(), rtol=1e-4)
    np.testing.assert_allclose(q1.loc.cpu().numpy(), q2.loc.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(z1.cpu().numpy(), z2.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(r1.cpu().numpy(), r2.detach().cpu().numpy(), rtol=2e-4, atol=2e-5 * float(r2.abs().max()) + 1e-5)


def test_eval_decode_cache_follows_training(golden):
    """the cached folded images of the eval decode are rebuilt after a flat training step (whose kernels write parameters and
    running statistics through raw pointers: no torch version counter moves) and after a torch-side parameter write"""
    from coskad_amd.models.sts.ae import STSAE
    from coskad_amd.trainer import STSAETrainStep
    g = golden("stsae_v25.npz")
    m, st = _build(g, STSAE)
    m.load_state_dict(st, strict=True)
    m.cuda()
    x = torch.from_numpy(g["x"]).cuda()
    eng = STSAETrainStep(m.train(), mode='ae', lr=1e-2, alpha=0.0, lambda_=1.0)

    def recon():
        m.eval()
        with torch.no_grad():
            return m(x)[1].clone()
    r0 = recon()
    assert torch.equal(recon(), r0) and "_lowrank_eval" in m.decoder.model[0].__dict__
    m.train()
    eng.step(x)
    r1 = recon()
    assert float((r1 - r0).abs().max()) > 1e-4                # the step moved the decoder
    from coskad_amd import lowrank
    keep = lowrank.MODE
    try:
        lowrank.MODE = "never"
        ref = recon()
    finally:
        lowrank.MODE = keep
    np.testing.assert_allclose(r1.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
    with torch.no_grad():
        m.rev_btlnk.bias.add_(0.05)                           # torch-side write: the version counter moves
    assert float((recon() - r1).abs().max()) > 1e-5


@pytest.mark.parametrize("V", [17, 25])
def test_narrow_output_layer_by_commutation_equals_the_layer_kernels(V, monkeypatch):
    """The decoder's last layer (32 -> 2) with its convolutions commuted in front of the mixing (csrc/last_layer.hip + the few-channel
    kernels on a virtual 4 -> 2 layer: trainer._FlatStack `narrow` segments) against the same layer on the tile kernels: losses,
    every gradient (incl. the producing layer's PReLU slope), the running statistics; default widths, ragged batch."""
    from coskad_amd import trainer
    from coskad_amd.models.sts.ae import STSAE
    from oracle import ref_cpu as R
    torch.manual_seed(5)
    proto = STSAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
    st = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    x = R.synthetic_clips(37, 2, 12, V, seed=6).cuda()
    res = {}
    for on in (False, True):
        monkeypatch.setattr(trainer, "NARROW_OUT", on)
        m = STSAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.load_state_dict(st)
        m.cuda().train()
        eng = trainer.STSAETrainStep(m, mode='ae', lr=0.0, alpha=0.0, lambda_=0.8)
        assert any(s[0] == 'narrow' for s in eng.dec.segs) == on
        out = eng.step(x)
        torch.cuda.synchronize()
        res[on] = (float(out['rec']), float(out['head']), {n: v.cpu().numpy().copy() for n, v in eng.fp.gviews.items()},
                   {k: v.cpu().numpy().copy() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    a, b = res[False], res[True]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-5)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-5)
    gmax = max(np.abs(v).max() for v in a[2].values())
    for n, ref in a[2].items():
        np.testing.assert_allclose(b[2][n], ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max() + 2e-5 * gmax, err_msg=n)
    for k, ref in a[3].items():
        np.testing.assert_allclose(b[3][k], ref, rtol=1e-4, atol=1e-6, err_msg=k)


def test_eval_after_training_through_the_narrow_and_folded_paths_sees_the_new_weights():
    """eval -> train steps -> eval on one model: the second evaluation must use the trained weights although the narrow / folded
    training paths never touch the real layers' eval-mode fold caches (regression: stale BatchNorm-folded weights of the decoder's
    last layer after the first validation)."""
    from coskad_amd import lowrank
    from coskad_amd.models.sts.ae import STSAE
    from coskad_amd.trainer import STSAETrainStep
    from oracle import ref_cpu as R
    torch.manual_seed(2)
    keep = lowrank.MODE
    lowrank.MODE = 'always'
    try:
        m = STSAE(2, [16, 8, 16], 16, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda()
        eng = STSAETrainStep(m.train(), mode='ae', lr=5e-3, alpha=0.0, lambda_=1.0)
        assert eng.lowrank is not None and any(s[0] == 'narrow' for s in eng.dec.segs)
        x = R.synthetic_clips(64, 2, 12, 17, seed=3).cuda()

        def recon():
            m.eval()
            with torch.no_grad():
                return m(x)[1].cpu()
        r0 = recon()
        m.train()
        for _ in range(5):
            eng.step(x)
        r1 = recon()
        st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            ref = R.stsae_decode(R.stse_encode(x.cpu(), st, training=False), st, 16, 12, 17, training=False)
        assert float((r1 - r0).abs().max()) > 1e-3
        np.testing.assert_allclose(r1.numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)
    finally:
        lowrank.MODE = keep
