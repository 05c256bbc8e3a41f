"""Two / four ranks on ONE GPU (gloo carries the collectives; RCCL refuses two ranks per device): the data-parallel train
step -- shard clips r::W, per-rank BatchNorm statistics, the two-bucket all-reduce of the flat gradient buffer, Adam with the
1/W scale -- must equal the oracle's two-shard computation (mean of shard gradients, torch Adam)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
CFG = dict(input_dim=2, layer_channels=(16, 8, 16), hidden=16, latent=8, T=12, V=17)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from coskad_amd import parallel
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    st = R.init_stse_state(CFG["input_dim"], CFG["layer_channels"], CFG["hidden"], CFG["latent"], seed=0)
    st["c"] = torch.linspace(-0.1, 0.1, CFG["latent"])
    x = R.synthetic_clips(48, seed=9)
    m = STSE(2, list(CFG["layer_channels"]), CFG["hidden"], CFG["latent"], 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    m.load_state_dict(st, strict=True)
    m.cuda().train()
    eng = STSETrainStep(m, lr=1e-3, alpha=0.0, head='euclidean')
    assert eng.world == world
    idx = parallel.shard_indices(48, rank, world)
    stats = eng.step(x[idx].cuda())
    c = eng.refresh_center(eps=1e-3).cpu()
    torch.cuda.synchronize()
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if v.is_floating_point()}, float(stats[0]), c.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_step_matches_oracle(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # oracle: per-shard autograd, mean of gradients, torch Adam
    from coskad_amd import parallel
    st = R.init_stse_state(CFG["input_dim"], CFG["layer_channels"], CFG["hidden"], CFG["latent"], seed=0)
    cvec = torch.linspace(-0.1, 0.1, CFG["latent"])
    x = R.synthetic_clips(48, seed=9)
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    zs, total = [], None
    for r in range(world):
        s2 = {k: v.clone() for k, v in st.items()}
        s2.update(params)
        z = R.stse_encode(x[parallel.shard_indices(48, r, world)], s2, training=True)
        zs.append(z.detach())
        loss = R.mse_to_center(z, cvec) / world
        loss.backward()
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    opt.step()
    for r in range(world):
        got = res[r][1]
        for k, p in params.items():
            if k.endswith(("tcn.0.bias", "residual.0.bias")):
                continue   # analytically zero gradient; autograd noise makes torch's Adam move them
            np.testing.assert_allclose(got[k], p.detach().numpy(), rtol=2e-3, atol=3e-4, err_msg=f"rank {r} {k}")
    # all ranks hold identical parameters and the same all-reduced centre = mean over ALL clips
    for r in range(1, world):
        for k in res[0][1]:
            if "running" not in k:
                np.testing.assert_array_equal(res[0][1][k], res[r][1][k], err_msg=k)
    zall = torch.cat(zs).mean(0)
    np.testing.assert_allclose(res[0][3], R.clamp_center(zall, 1e-3).numpy(), rtol=1e-3, atol=1e-5)
    for r in range(1, world):
        np.testing.assert_array_equal(res[0][3], res[r][3])


DEF = dict(input_dim=2, layer_channels=(32, 16, 32), hidden=64, latent=16)


def _worker_syncbn(rank, world, port, q, bounds):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from coskad_amd import parallel
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    st = R.init_stse_state(DEF["input_dim"], DEF["layer_channels"], DEF["hidden"], DEF["latent"], seed=0)
    st["c"] = torch.linspace(-0.1, 0.1, DEF["latent"])
    x = R.synthetic_clips(bounds[-1], seed=9)
    m = STSE(2, list(DEF["layer_channels"]), DEF["hidden"], DEF["latent"], 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0)
    m.load_state_dict(st, strict=True)
    m.cuda().train()
    eng = STSETrainStep(m, lr=1e-3, alpha=0.0, head='euclidean', sync_bn=True)
    assert eng.world == world and eng.sync_group is not None
    eng.step(x[bounds[rank]:bounds[rank + 1]].cuda())
    torch.cuda.synchronize()
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if v.is_floating_point()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bounds", [(0, 24, 48), (0, 20, 37, 50)])
def test_sync_bn_ranks_match_global_batch_oracle(bounds):
    """Optional SyncBN: with the ranks' fp64 moment sums (forward) and P / Q / sdU sums (backward) added at every
    BatchNorm boundary, W ranks on shards of a batch -- of unequal size for W = 3 -- reproduce the single-process oracle step on
    the WHOLE batch: parameters after Adam and the running statistics.  Default stack: the fused apply + next-statistics kernels,
    the backward chain and the top layer's own reduction all take part."""
    world, nclips = len(bounds) - 1, bounds[-1]
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, world, port, q, bounds)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    st = R.init_stse_state(DEF["input_dim"], DEF["layer_channels"], DEF["hidden"], DEF["latent"], seed=0)
    cvec = torch.linspace(-0.1, 0.1, DEF["latent"])
    x = R.synthetic_clips(nclips, seed=9)
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    s2 = {k: v.clone() for k, v in st.items()}
    s2.update(params)
    # every rank's loss is the mean over ITS clips and the gradients are averaged over the ranks: the global objective is the mean
    # of the per-rank means (equal to the plain mean when the shards have equal sizes)
    z = R.stse_encode(x, s2, training=True)            # ONE forward over the whole batch: global BatchNorm statistics
    loss = sum(R.mse_to_center(z[bounds[r]:bounds[r + 1]], cvec) for r in range(world)) / world
    loss.backward()
    torch.optim.Adam(list(params.values()), lr=1e-3).step()
    for r in range(world):
        got = res[r][1]
        for k, p in params.items():
            if k.endswith(("tcn.0.bias", "residual.0.bias")):
                continue   # analytically zero gradient (see above)
            np.testing.assert_allclose(got[k], p.detach().numpy(), rtol=2e-3, atol=3e-4, err_msg=f"rank {r} {k}")
        for k, v in s2.items():
            if "running" in k:
                np.testing.assert_allclose(got[k], v.numpy(), rtol=1e-4, atol=1e-6, err_msg=f"rank {r} {k}")
    for r in range(1, world):
        for k in res[0][1]:
            np.testing.assert_array_equal(res[0][1][k], res[r][1][k], err_msg=k)


def _worker_unsynced(rank, world, port, q, backend):
    """Every rank builds its model from its OWN random initialisation (no shared load_state_dict): the broadcast that
    Trainer.fit performs (Lightning's DDP wrap, train_COSKAD.py:75-78) must make the replicas identical, and one
    data-parallel step must keep them identical."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from coskad_amd import parallel
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    torch.manual_seed(1000 + rank)
    m = STSE(2, list(CFG["layer_channels"]), CFG["hidden"], CFG["latent"], 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda()
    before = float(m.btlnk.weight.detach().abs().sum())
    parallel.broadcast_module_(m)
    m.train()
    eng = STSETrainStep(m, lr=1e-3, alpha=1e-6, head='euclidean')
    x = R.synthetic_clips(48, seed=9)
    eng.step(x[parallel.shard_indices(48, rank, world)].cuda())
    eng.refresh_center(eps=1e-3)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if v.is_floating_point()}
    q.put((rank, sd, before, dist.get_world_size(), dist.get_backend()))
    dist.barrier(device_ids=[dev]) if backend == "nccl" else dist.barrier()
    dist.destroy_process_group()


def _run_unsynced(world, backend):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unsynced, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[3] == world and r[4] == backend for r in res)
    assert res[0][2] != res[1][2]                       # the ranks really started from different weights
    for r in range(1, world):
        for k, v in res[0][1].items():
            if "running" not in k:                      # BatchNorm running statistics stay per rank (no SyncBN)
                np.testing.assert_array_equal(v, res[r][1][k], err_msg=k)


def test_unsynced_init_is_broadcast_gloo():
    _run_unsynced(2, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (>= 2 GPUs)")
def test_two_rank_step_rccl():
    """The same check over RCCL (backend 'nccl'), one rank per GPU."""
    _run_unsynced(2, "nccl")
