"""The fused eval-mode encoder kernel (csrc/fused_fwd.hip) through the C ABI against the CPU oracle / the reference's
golden vectors: tile-major activations element for element, latents within 1e-4 (north_star), ragged batch sizes, the
persistent clip loop, bit-exact chunking, plan invalidation after training."""
import numpy as np
import pytest
import torch

from conftest import state_from
from coskad_amd import fused_plan as FP
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _model(st):
    from coskad_amd.models.sts.ae import STSE
    m = STSE(2, [32, 16, 32], 64, st["c"].shape[0], 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    m.load_state_dict(st, strict=True)
    return m.cuda().eval()


def _perturbed_state(seed, latent=16):
    st = R.init_stse_state(2, (32, 16, 32), 64, latent, 12, 17, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in st.items():
        if k.endswith("running_mean"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        if k.endswith("running_var"):
            v.mul_(1 + 0.3 * torch.rand(v.shape, generator=g))
        if (".tcn.1." in k or ".residual.1." in k) and k.endswith(("weight", "bias")):
            v.add_(0.2 * torch.randn(v.shape, generator=g))
        if k.endswith("prelu.weight"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
    st["c"] = torch.zeros(latent)
    return st


@pytest.mark.parametrize("B", [1, 9, 64])
def test_tile_major_activations_and_latents(B):
    from coskad_amd import engine, ops
    from coskad_amd.models.graph_layers.stsgcn import layer_tensors
    st = _perturbed_state(3)
    m = _model(st)
    x = R.synthetic_clips(B, seed=5)
    acts = []
    with torch.no_grad():
        z_ref = R.stse_encode(x, st, training=False, collect=acts)
    layers = [layer_tensors(l) for l in m.encoder.model]
    assert engine.fused_encoder_supported(layers, 12, 17)
    plan = engine.FusedEncoderPlan().get(layers, m.btlnk.weight)
    H = ops.fused_encoder(x.cuda(), plan.tab, plan.wreg, plan.slopes)
    torch.cuda.synchronize()
    assert H.shape == (B, FP.KP)
    h = H.cpu().numpy().reshape(B, FP.NTILE, 4, 64, 4)
    ref = acts[-1].numpy().reshape(B, 64, 12 * 17)                 # activated last layer of the oracle
    j, q = np.arange(64) & 15, np.arange(64) >> 4
    for tile in range(FP.NTILE):
        p = FP.out_position(tile, j)
        ok = p >= 0
        for ot in range(4):
            for r in range(4):
                o = 16 * ot + 4 * q + r
                np.testing.assert_allclose(h[:, tile, ot, ok, r], ref[:, o[ok], p[ok]], rtol=1e-4, atol=1e-4,
                                           err_msg=f"tile {tile} ot {ot} r {r}")
                assert np.all(h[:, tile, ot, ~ok, r] == 0)
    with torch.no_grad():
        z = m(x.cuda())
    np.testing.assert_allclose(z.cpu().numpy(), z_ref.numpy(), rtol=1e-4, atol=1e-4)


def test_reference_golden_latents(golden):
    g = golden("stse_default.npz")
    st = state_from(g)
    m = _model(st)
    with torch.no_grad():
        z = m(torch.from_numpy(g["x"]).cuda())
    assert "_fused_plan" in m.__dict__                       # the fused kernel is what ran
    np.testing.assert_allclose(z.cpu().numpy(), g["eval.z"], rtol=1e-4, atol=1e-4)


def test_persistent_loop_chunking_and_latent8():
    """B > 4 waves x 256 CUs: every wave walks several clips; chunked batches give bit-identical latents; latent 8."""
    st = _perturbed_state(7, latent=8)
    m = _model(st)
    B = 2600
    x = R.synthetic_clips(B, seed=11)
    with torch.no_grad():
        z_ref = R.stse_encode(x, st, training=False)
        xg = x.cuda()
        z = m(xg)
        zc = torch.cat([m(xg[:1000]), m(xg[1000:1003]), m(xg[1003:])])
    np.testing.assert_allclose(z.cpu().numpy(), z_ref.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(z, zc)


def test_plan_follows_the_weights():
    """A training step changes weights and running statistics through raw-pointer kernels: the next eval forward must
    rebuild the operand streams."""
    from coskad_amd.trainer import STSETrainStep
    st = _perturbed_state(9)
    st["c"] = torch.full((16,), 0.05)
    m = _model(st)
    x = R.synthetic_clips(32, seed=2).cuda()
    with torch.no_grad():
        z0 = m(x).clone()
    m.train()
    eng = STSETrainStep(m, lr=1e-2, alpha=0.0, head='euclidean')
    eng.step(x)
    m.eval()
    with torch.no_grad():
        z1 = m(x)
    assert not torch.allclose(z0, z1)
    st1 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        z_ref = R.stse_encode(x.cpu(), st1, training=False)
    np.testing.assert_allclose(z1.cpu().numpy(), z_ref.numpy(), rtol=1e-4, atol=1e-4)
