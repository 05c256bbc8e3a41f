"""Host-side logic that needs no GPU: module surface, state_dict contract, flat-parameter views,
regulariser mask, loud failure on CPU tensors."""
import os

import numpy as np
import pytest
import torch

from conftest import state_from
from oracle import ref_cpu as R


def test_state_dict_contract(golden):
    from coskad_amd.models.sts.ae import STSE, STSAE
    g = golden("stse_default.npz")
    st = state_from(g)
    m = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'STS_GCN', 'linear', 'euclidean', 0.0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(st.keys())          # same names, same order as the reference
    for k in sd:
        assert tuple(sd[k].shape) == tuple(st[k].shape), k
    m.load_state_dict(st, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 239716
    ga = golden("stsae_small.npz")
    ma = STSAE(2, [16, 8, 16], 16, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    ma.load_state_dict(state_from(ga), strict=True)
    mm = STSE(2, [8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'mahalanobis', 0.0)
    assert tuple(mm.inv_cov_matrix.shape) == (8, 8)


def test_reference_initialisers():
    from coskad_amd.models.graph_layers.stsgcn import ConvTemporalGraphical, ST_GCNN_layer
    torch.manual_seed(0)
    g = ConvTemporalGraphical(12, 17)
    assert g.A.shape == (12, 17, 17) and g.T.shape == (17, 12, 12)
    assert g.A.abs().max() <= 1 / 17 ** 0.5 and g.T.abs().max() <= 1 / 12 ** 0.5   # stsgcn.py:134-140
    layer = ST_GCNN_layer(4, 4, (1, 1), 1, 12, 17, 0.0)
    assert isinstance(layer.residual, torch.nn.Identity)                            # stsgcn.py:79-80
    assert float(layer.prelu.weight) == 0.25
    with pytest.raises(AssertionError):
        ST_GCNN_layer(4, 8, (2, 1), 1, 12, 17, 0.0)                                 # stsgcn.py:40-41
    with pytest.raises(NotImplementedError):
        ST_GCNN_layer(4, 8, (3, 3), 1, 12, 17, 0.0)


def test_unknown_types_raise_like_reference():
    from coskad_amd.models.sts.ae import STSE
    with pytest.raises(ValueError):
        STSE(2, [8], 8, 8, 12, 17, 'nope', 'linear', 'euclidean', 0.0)      # ae.py:142
    with pytest.raises(ValueError):
        STSE(2, [8], 8, 8, 12, 17, 'sts_gcn', 'nope', 'euclidean', 0.0)     # ae.py:164
    m = STSE(2, [8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    with pytest.raises(AssertionError):
        m(torch.zeros(2, 2, 12))                                            # ae.py:88


def test_product_path_has_no_cpu_fallback():
    from coskad_amd._lib import CoskadHipError
    from coskad_amd.models.sts.ae import STSE
    m = STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    with pytest.raises(CoskadHipError, match="no CPU fallback"):
        m(torch.zeros(2, 2, 12, 17))


def test_flat_params_and_reg_mask():
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import FlatParams
    m = STSE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    # every tensor starts on a 16-byte boundary of the flat buffer (float4 weight loads): padded to multiples of 4 floats
    assert fp.flat.numel() == sum((p.numel() + 3) // 4 * 4 for p in m.parameters())
    assert all(off % 4 == 0 for off in fp.offsets.values()) and all(v.data_ptr() % 16 == 0 for v in fp.views.values())
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    # parameters are views of the flat buffer: writing the buffer changes the module
    fp.flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())
    fp.flat.copy_(torch.arange(fp.flat.numel(), dtype=torch.float32))
    named = list(m.named_parameters())
    assert float(named[0][1].reshape(-1)[0]) == 0.0
    # reg mask == calc_reg_loss's name filter (utils/model_utils.py:92)
    n_nonbias = sum(1 for n, _ in named if 'bias' not in n)
    assert fp.n_reg_tensors == n_nonbias
    ref = R.calc_reg_loss([(n, p.detach()) for n, p in named])
    mine = 0.5 / fp.n_reg_tensors * float((fp.reg_mask * fp.flat * fp.flat).sum())
    np.testing.assert_allclose(mine, float(ref), rtol=1e-6)


def test_bench_byte_model():
    import bench
    fwd, bwd = bench.algorithmic_bytes_per_clip()
    assert fwd == 236704
    assert abs(bwd - 354208) <= 64  # SURVEY 8d (+ the dz read)


def test_config_contract():
    """flat yaml -> Namespace -> init_sub_args (reference utils/argparser.py:10-45)."""
    import argparse
    import yaml
    from coskad_amd.utils.argparser import init_sub_args
    raw = yaml.safe_load(open("config/synthetic/euclidean_encoder.yaml"))
    raw["create_experiment_dir"] = False
    raw["data_dir"] = "/data/UBnormal"
    args, data_args, ae_args, res_args, opt_args = init_sub_args(argparse.Namespace(**raw))
    assert args.ckpt_dir == "./checkpoints/UBnormal/STSE_euclidean_dynamic_synthetic"
    assert args.gt_path == "/data/UBnormal/validating/test_frame_mask"          # UBnormal + validation
    assert data_args.seg_len == 12 and data_args.batch_size == 2048 and ae_args.epochs == 3 and opt_args.lr == 1e-4
    # the wrapper builds the encoder the yaml describes
    from coskad_amd.lit import LitEncoder
    lit = LitEncoder(args)
    assert lit.model.n_joints == 17 and lit.model.latent_dim == 16 and not lit.hyperbolic


def test_power_spherical_sampler_moments():
    """power_spherical is un-vendored in the reference -> statistical checks of the restated sampler."""
    from coskad_amd.models.sts.vae import HypersphericalUniform, PowerSpherical, kl_ps_uniform
    torch.manual_seed(0)
    d, n = 8, 200000
    mu = torch.nn.functional.normalize(torch.randn(d), dim=0)
    for kappa in (2.0, 20.0):
        q = PowerSpherical(mu.expand(n, d), torch.full((n,), kappa))
        x = q.rsample()
        np.testing.assert_allclose(x.norm(dim=-1).numpy(), 1.0, atol=1e-5)            # on the sphere
        a, b = (d - 1) / 2 + kappa, (d - 1) / 2
        np.testing.assert_allclose(float((x @ mu).mean()), (a - b) / (a + b), atol=5e-3)  # E[mu.x] = 2E[Beta]-1
        resid = x - (x @ mu)[:, None] * mu
        assert float(resid.mean(0).abs().max()) < 5e-3                                 # symmetric around mu
    # KL to the uniform: >= 0, grows with the concentration, ~0 for kappa -> 0
    p = HypersphericalUniform(d - 1)
    kl = kl_ps_uniform(PowerSpherical(mu.expand(3, d), torch.tensor([1e-4, 5.0, 50.0])), p)
    assert abs(float(kl[0])) < 1e-3 and 0 < float(kl[1]) < float(kl[2])


def test_stsvae_surface():
    from coskad_amd.models.sts.vae import STSVAE
    m = STSVAE(2, [8, 4, 8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0, distribution='ps')
    keys = set(m.state_dict().keys())
    assert {"fc_mean.weight", "fc_var.weight", "rev_btlnk.weight", "decoder.model.0.gcn.A", "threshold_dist", "c"} <= keys
    assert m.fc_var.out_features == 1 and isinstance(m.btlnk, torch.nn.Identity)     # vae.py:150,161
    with pytest.raises(ValueError):
        STSVAE(2, [8], 8, 8, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0, distribution='nope')


def test_static_gcn_adjacency_matches_reference_buffer():
    """EncoderStaticPlainGCN's fixed adjacency (alternative_components.py:213-229,243-259), built array-wise."""
    import os
    from coskad_amd.models.common.alternative_components import EncoderStaticPlainGCN
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "stse_altgcn.npz"))
    enc = EncoderStaticPlainGCN(2, [8, 4], 8, 12, 17, 0.0)
    np.testing.assert_array_equal(enc.Adj.numpy(), g["static_gcn.sd.encoder.Adj"])
    np.testing.assert_allclose(enc.Adj.sum(1).numpy(), 1.0, rtol=1e-6)


def test_legacy_wrapper_import_names():
    """The reference's Lightning wrappers import their models under package names that are absent from the snapshot
    (models/euclidean_encoder_staticCenter.py:19, spherical_vae.py:16, euclidean_autoencoder.py:18)."""
    from coskad_amd.models.sts import ae, vae
    from coskad_amd.models.stsae.stsae_hidden_hypersphere import STSAE
    from coskad_amd.models.stse.stse_hidden_hypersphere import STSE
    from coskad_amd.models.stsve.stsve_hidden_hypersphere import STSVE
    assert STSE is ae.STSE and STSAE is ae.STSAE and STSVE is vae.STSVAE


def test_oracle_mlp_equals_torch_sequential_of_the_reference_blocks():
    """`oracle.ref_cpu.mlp` pinned against what the reference's MLP.build_model would assemble if its constructor ran
    (components.py:209-226: [Linear, BatchNorm1d, ReLU(inplace)] per hidden size + a final Linear, as nn.Sequential):
    eval and train mode outputs, gradients and the running statistics after the step."""
    import torch.nn as nn
    from oracle import ref_cpu as R
    torch.manual_seed(0)
    sizes, inp, out = [12, 10], 40, 6
    layers, k = [], inp
    for h in sizes:
        layers += [nn.Linear(k, h), nn.BatchNorm1d(h), nn.ReLU(inplace=True)]
        k = h
    layers.append(nn.Linear(k, out))
    net = nn.Sequential(*layers)
    with torch.no_grad():
        for m in net:
            if isinstance(m, nn.BatchNorm1d):
                m.weight.add_(0.3 * torch.randn_like(m.weight)); m.bias.add_(0.2 * torch.randn_like(m.bias))
                m.running_mean.add_(0.1 * torch.randn_like(m.running_mean)); m.running_var.mul_(1.7)
    x = torch.randn(33, inp)
    st = {"btlnk.net." + k: v.detach().clone() for k, v in net.state_dict().items()}
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(R.mlp(x, st, "btlnk", training=False).numpy(), net(x).numpy(), rtol=1e-6, atol=1e-6)
    net.train()
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if v.is_floating_point() and "running" not in k}
    so = dict(st)
    so.update(params)
    y_o = R.mlp(x, so, "btlnk", training=True)
    y_t = net(x)
    np.testing.assert_allclose(y_o.detach().numpy(), y_t.detach().numpy(), rtol=1e-5, atol=1e-6)
    (y_o ** 2).mean().backward()
    (y_t ** 2).mean().backward()
    for n, p in net.named_parameters():
        np.testing.assert_allclose(params["btlnk.net." + n].grad.numpy(), p.grad.numpy(), rtol=1e-4, atol=1e-6, err_msg=n)
    for n, b in net.named_buffers():
        np.testing.assert_allclose(so["btlnk.net." + n].numpy(), b.numpy(), rtol=1e-6, atol=1e-7, err_msg=n)


REF_CONFIGS = "/root/reference/config"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="the reference tree is only present in the build container")
def test_reference_yaml_files_drive_the_wrappers():
    """Config contract (train_COSKAD.py:15-62): every yaml the reference ships is read by argparser.init_sub_args and builds
    the wrapper train_COSKAD.py would select, with the model family / projector / latent size the file names.
    `UBnormal/euclidean_autoencoder.yaml` is malformed in the reference itself (yaml.load raises there too): expected."""
    import argparse
    import glob
    import yaml
    from coskad_amd import lit
    from coskad_amd.models.common.components import MLP
    from coskad_amd.utils.argparser import init_sub_args
    files = sorted(glob.glob(os.path.join(REF_CONFIGS, "*", "*.yaml")))
    assert len(files) >= 7
    built, broken = 0, []
    for f in files:
        try:
            raw = yaml.load(open(f), Loader=yaml.FullLoader)
        except yaml.YAMLError:
            broken.append(os.path.relpath(f, REF_CONFIGS))
            continue
        args = argparse.Namespace(**raw)
        args.exp_dir = "/tmp/coskad_contract"       # nothing is written: the wrappers only read it
        args, *_ = init_sub_args(args)
        cls = lit.LitAutoEncoder if args.use_decoder else lit.LitVAE if args.use_vae else lit.LitEncoder
        m = cls(args)
        assert m.model.latent_dim == raw["latent_dim"], f
        if not args.use_decoder:
            proj = m.model.btlnk
            want_mlp = raw.get("projector", "linear") == "mlp"
            assert isinstance(proj, MLP) == want_mlp, f
        if str(raw.get("encoder_type", "sts_gcn")).lower() == "sts_gcn":
            chans = [l.tcn[0].weight.shape[0] for l in m.model.encoder.model]
            assert chans == list(raw["channels"]) + [raw["h_dim"]], f
        else:
            assert type(m.model.encoder).__name__.lower().startswith("encoder" + str(raw["encoder_type"]).lower().split("_")[0]), f
        built += 1
    assert built >= 6 and broken == ["UBnormal/euclidean_autoencoder.yaml"], (built, broken)


def test_bn_momentum_host_mirror_of_num_batches_tracked():
    """ops.bn_momentum: nn.BatchNorm's exponential_average_factor per training forward -- the fixed momentum, or (momentum=None)
    1 / num_batches_tracked counted after this batch, from a host mirror that is re-read when torch writes the counter
    (torch/nn/modules/batchnorm.py, _BatchNorm.forward)."""
    import torch
    from coskad_amd import ops
    assert ops.bn_momentum(torch.nn.BatchNorm2d(4, momentum=0.3)) == pytest.approx(0.3)
    bn = torch.nn.BatchNorm2d(4, momentum=None)
    assert [ops.bn_momentum(bn) for _ in range(3)] == [1.0, 0.5, pytest.approx(1 / 3)]   # the kernels advance the device counter
    bn.num_batches_tracked.fill_(9)                       # torch-side write (load_state_dict does the same): version moves
    assert ops.bn_momentum(bn) == pytest.approx(0.1)
    assert ops.bn_momentum(bn) == pytest.approx(1 / 11)
    free = torch.nn.BatchNorm1d(4, track_running_stats=False)
    assert ops.bn_momentum(free) == pytest.approx(0.1) and ops.bn_batch_stats(free, False) and not ops.bn_batch_stats(bn, False)
    free2 = torch.nn.BatchNorm1d(4, momentum=None, track_running_stats=False)
    assert ops.bn_momentum(free2) == 0.0
