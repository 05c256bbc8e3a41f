"""world_size-2 `gloo` test of the data-parallel path on CPU: shard -> per-rank gradients (oracle autograd
stands in for the kernels) -> flat all-reduce mean -> every rank holds the mean of the shard gradients
(= what the reference's DDP computes); centre statistics all-reduce -> the global centre."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(st, x, c):
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    s2 = {k: v.clone() for k, v in st.items()}
    s2.update(params)
    z = R.stse_encode(x, s2, training=True)
    R.mse_to_center(z, c).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in params.values()])
    stats = torch.zeros(19)
    stats[1:1 + z.shape[1]] = z.detach().sum(0)
    stats[17] = z.shape[0]
    return flat, stats


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coskad_amd import parallel
    torch.set_num_threads(1)
    st = R.init_stse_state(2, (8, 4, 8), 8, 8, 12, 17, seed=0)
    x = R.synthetic_clips(12, seed=5)
    c = torch.full((8,), 0.05)
    idx = parallel.shard_indices(12, rank, world)
    flat, stats = _shard_grads(st, x[idx], c)
    parallel.allreduce_mean_(flat)
    parallel.allreduce_sum_(stats)
    gathered = parallel.gather_rows(idx.float()[:, None])
    ragged = parallel.gather_rows(torch.arange(11)[rank::world].float()[:, None])   # 6 rows on rank 0, 5 on rank 1
    assert sorted(ragged[:, 0].tolist()) == list(range(11)), ragged
    q.put((rank, flat.numpy(), stats.numpy(), gathered.numpy()))
    dist.destroy_process_group()


def test_two_rank_gradient_and_centre_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from coskad_amd import parallel
    st = R.init_stse_state(2, (8, 4, 8), 8, 8, 12, 17, seed=0)
    x = R.synthetic_clips(12, seed=5)
    c = torch.full((8,), 0.05)
    g0, s0 = _shard_grads(st, x[parallel.shard_indices(12, 0, 2)], c)
    g1, s1 = _shard_grads(st, x[parallel.shard_indices(12, 1, 2)], c)
    for r in range(world):
        np.testing.assert_allclose(res[r][1], ((g0 + g1) / 2).numpy(), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(res[r][2], (s0 + s1).numpy(), rtol=1e-6)
        assert sorted(res[r][3][:, 0].tolist()) == list(range(12))        # every clip scored exactly once
    # centre from all-reduced statistics == centre of the un-sharded batch
    z_all = R.stse_encode(x, {k: v.clone() for k, v in st.items()}, training=False)
    assert res[0][2][17] == 12


def test_shard_indices_cover_and_pad():
    from coskad_amd.parallel import shard_indices
    a, b, c = (shard_indices(10, r, 3) for r in range(3))
    assert len(a) == len(b) == len(c) == 4
    assert sorted(set(torch.cat([a, b, c]).tolist())) == list(range(10))
    assert a.tolist() == [0, 3, 6, 9] and b.tolist() == [1, 4, 7, 0]


def _worker_bcast(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coskad_amd import parallel
    from coskad_amd.utils.synthetic import batches, make_dataset
    torch.manual_seed(100 + rank)                      # every rank: its own random initialisation
    m = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.PReLU())
    m[1].num_batches_tracked.fill_(7 + rank)
    parallel.broadcast_module_(m)
    flat = torch.cat([t.detach().reshape(-1).double() for t in list(m.parameters()) + list(m.buffers())])
    # equal-length shards: n % (W * bs) == 1 used to give rank 0 one more batch than rank 1 (mismatched collectives)
    data, _ = make_dataset(n_scenes=1, n_clips=1, n_persons=1, clip_len=12 + 16, anomaly=False, seed=0)   # 17 windows
    n = data[0].shape[0]
    nb = sum(1 for _ in batches(data, 4, rank=rank, world=world))
    outs = list(batches(data, 4, rank=rank, world=world))
    cat = [torch.cat([o[i] for o in outs], 0) for i in range(4)]
    cat = [parallel.gather_rows(t) for t in cat]
    keep = parallel.dedupe_rows(torch.cat([cat[1].long().reshape(-1, 1), cat[2].long()], 1))
    q.put((rank, flat.numpy(), n, nb, cat[2][keep][:, 3].tolist(), int(cat[0].shape[0])))
    dist.destroy_process_group()


def test_module_broadcast_and_equal_shards():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bcast, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])          # rank 1 now holds rank 0's parameters and buffers
    torch.manual_seed(100)
    m0 = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.PReLU())
    np.testing.assert_array_equal(res[0][1][:8], m0[0].weight.detach().reshape(-1).double().numpy())
    n = res[0][2]
    assert n == 17 and res[0][3] == res[1][3] == 3                # ceil(ceil(17/2)/4) batches on BOTH ranks
    for r in res:
        assert r[5] == 18 and sorted(r[4]) == list(range(17))     # 18 gathered rows, 17 distinct windows after the dedupe


def _worker_unused(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coskad_amd import parallel
    torch.manual_seed(0)
    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2))
    frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)
    a.grad = torch.full((3,), float(rank + 1))
    if rank == 0:                       # rank 1 never touched `b` (a branch its shard skipped)
        b.grad = torch.full((2, 2), 4.0)
    parallel.allreduce_grads_mean_([a, frozen, b])
    q.put((rank, a.grad.numpy(), b.grad.numpy(), frozen.grad))
    dist.destroy_process_group()


def test_grad_bucket_layout_is_rank_independent_with_unused_parameters():
    """One rank has no gradient for a parameter: the flat bucket is laid out over every parameter that requires a gradient
    (zeros where missing), so lengths agree, nothing hangs or lands in the wrong tensor, and every rank ends with the mean."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unused, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ga, gb, gf in res:
        np.testing.assert_allclose(ga, np.full(3, 1.5))
        np.testing.assert_allclose(gb, np.full((2, 2), 2.0))
        assert gf is None
