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
