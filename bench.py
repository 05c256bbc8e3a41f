#!/usr/bin/env python
"""bench.py -- pose-clips/sec of the STS-GCN encoder train step (fwd + loss + reg + bwd + Adam).

Workload (BASELINE.json configs[1]): euclidean_encoder_dynamicCenter, synthetic clips
B=4096 per GPU, T=12, V=17, C=2, channels 2->32->16->32->64, latent 16, fp32.
  python bench.py --gpus N --steps K --warmup W
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's
form), or, when WORLD_SIZE is unset, this script starts those N ranks itself as a child process BEFORE touching the
GPU, forwards their output and exits with the child's code.
Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region.

roofline  : the dominant kernel's ALGORITHMIC bytes per launch / its average duration, measured with HIP
            events on the launch stream inside the timed region, against 8 TB/s (MI355X_MICROARCH.md).
cpu_baseline: the CPU oracle (oracle/ref_cpu.py: plain PyTorch CPU ops, pinned to the reference by golden
            vectors) doing the same train step on a bounded sample on this box's host cores (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, V, C_IN = 12, 17, 2
CHANNELS, HID, LATENT = [32, 16, 32], 64, 16
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def algorithmic_bytes_per_clip():
    """every layer reads its input once and writes its output once (fp32)."""
    chans = [C_IN] + CHANNELS + [HID]
    tv = T * V
    fwd = sum(4 * tv * (ci + co) for ci, co in zip(chans[:-1], chans[1:])) + 4 * HID * tv + 4 * LATENT
    # backward per layer: read dOut, read saved input, write dIn (no dIn for layer 1); bottleneck: read U, dz; write dU
    bwd = (2 * 4 * HID * tv + 2 * 4 * LATENT)
    for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:])):
        bwd += 4 * tv * (co + ci + (ci if i > 0 else 0))
    return fwd, bwd


def _median_time(fn, warmup: int, iters: int) -> float:
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_b: int, fwd_b: int):
    """SURVEY 8d protocol: 3 warm-ups, median of 10 iterations; (a) the train step (fwd + mse-to-centre + 1e-6 reg + bwd +
    Adam) on `sample_b` clips, (b) the eval-mode forward under no_grad on `fwd_b` clips (the B = 4096 forward leg)."""
    import torch
    from oracle import ref_cpu as R
    # the GPU box gives a one-GPU job a 16-core share; os.cpu_count() reports the whole host and
    # oversubscribing OpenMP threads makes the CPU path crawl
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
    st = R.init_stse_state(seed=0)
    st["c"] = torch.full((LATENT,), 0.1)
    params = {k: v.requires_grad_(True) for k, v in st.items() if R.is_param_key(k) and v.is_floating_point()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4)
    x = R.synthetic_clips(sample_b, seed=1)

    def step():
        opt.zero_grad(set_to_none=True)
        z = R.stse_encode(x, st, training=True)
        loss = R.mse_to_center(z, st["c"]) + 1e-6 * R.calc_reg_loss(list(params.items()))
        loss.backward()
        opt.step()

    dt = _median_time(step, 3, 10)
    xf = R.synthetic_clips(fwd_b, seed=2)

    def fwd():
        with torch.no_grad():
            R.stse_encode(xf, st, training=False)

    dtf = _median_time(fwd, 3, 10)
    return {"value": round(sample_b / dt, 1), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": _cpu_model(), "host_cores_visible": ncpu,
            "sample": f"median of 10 train steps (fwd+loss+reg+bwd+Adam) after 3 warm-ups, CPU oracle, B={sample_b} "
                      f"synthetic clips [B,2,12,17], default stack 2-32-16-32-64, latent 16",
            "forward": {"value": round(fwd_b / dtf, 1), "unit": "clips/s", "ms": round(dtf * 1e3, 2),
                        "sample": f"median of 10 eval-mode forwards (no_grad) after 3 warm-ups on B={fwd_b} clips"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU (weak scaling)")
    ap.add_argument("--head", default="euclidean", choices=["euclidean", "poincare"])
    ap.add_argument("--graph", type=int, default=int(os.environ.get("COSKAD_GRAPH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to "
                    "rehearse the multi-rank control flow on a single GPU)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="clips per CPU-baseline train step")
    ap.add_argument("--cpu-fwd-sample", type=int, default=4096, help="clips per CPU-baseline forward")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU as a CHILD of this process, before anything here initialises the GPU
        # (never re-exec a process that has touched the device); stdout/stderr pass through, the child's code is ours
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from coskad_amd import _lib
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.trainer import STSETrainStep
    from coskad_amd.utils.synthetic import synthetic_clips

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dev = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":   # "nccl" = RCCL on ROCm; bind the communicator to this rank's GPU up front
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)

    B = args.batch
    torch.manual_seed(0)                     # the modules' own (reference) initialisers; same weights on every rank
    model = STSE(C_IN, CHANNELS, HID, LATENT, T, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
    model.c.fill_(0.1)
    model.cuda().train()
    eng = STSETrainStep(model, lr=1e-4, alpha=1e-6, head=args.head, use_graph=bool(args.graph))
    x = synthetic_clips(B, C_IN, T, V, seed=100 + rank).cuda()   # each rank: its own shard of clips

    def sync():
        if world > 1:
            dist.barrier(device_ids=[dev]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(x)
    sync()
    # dominant kernel (rocprof, profiles/r02_kernel_instances.csv): the fused backward of layer 4 (C_in 32 -> C_out 64).  The library brackets each
    # of its launches with HIP events on the launch stream (coskad_probe_*), inside the timed region.
    import ctypes
    lib = _lib.lib()
    KID_LAYER_APPLY, KID_BWD_DATA = 1, 2
    probing = not args.graph
    if probing:
        lib.coskad_probe_begin(KID_BWD_DATA, CHANNELS[-1], HID)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = eng.step(x)
    sync()
    dt = time.perf_counter() - t0
    probe_ms, probe_n = ctypes.c_float(0), ctypes.c_int(0)
    if probing:
        lib.coskad_probe_end(ctypes.byref(probe_ms), ctypes.byref(probe_n))
    # secondary (outside the timed region): the layer-4 forward kernel, for the forward-roofline target
    fwd_ms, fwd_n = ctypes.c_float(0), ctypes.c_int(0)
    if probing:
        # every rank runs these steps (they contain the gradient all-reduce); only rank 0 reads its probe
        lib.coskad_probe_begin(KID_LAYER_APPLY, CHANNELS[-1], HID)
        for _ in range(5):
            eng.step(x)
        sync()
        lib.coskad_probe_end(ctypes.byref(fwd_ms), ctypes.byref(fwd_n))
    # the WHOLE layer-4 backward (every kernel that shares SURVEY 8d's layer-backward bytes: reductions, folds, data
    # path, dA/dT), so that the dominant kernel's fraction is not read as the layer's
    lbw_ms, lbw_n = ctypes.c_float(0), ctypes.c_int(0)
    if probing:
        lib.coskad_probe_begin(6, CHANNELS[-1], HID)
        for _ in range(5):
            eng.step(x)
        sync()
        lib.coskad_probe_end(ctypes.byref(lbw_ms), ctypes.byref(lbw_n))
    # forward-only (eval-mode encoder + bottleneck; SURVEY 8d's forward roofline target), outside the timed region
    model.eval()
    fz_ms, fz_n = ctypes.c_float(0), ctypes.c_int(0)
    with torch.no_grad():
        for _ in range(5):
            model(x)
        sync()
        lib.coskad_probe_begin(7, C_IN, HID)      # the fused encoder kernel's own launches (HIP events on its stream)
        tf0 = time.perf_counter()
        for _ in range(50):
            model(x)
        sync()
        fwd_dt = (time.perf_counter() - tf0) / 50
        lib.coskad_probe_end(ctypes.byref(fz_ms), ctypes.byref(fz_n))
    model.train()
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = float(stats[0])

    if rank == 0:
        fwd_b, bwd_b = algorithmic_bytes_per_clip()
        roof = roof_fwd = None
        tvb = 4 * T * V
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this script)
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")) as f:
                traffic = json.load(f)
        except OSError:
            pass
        if probe_n.value:
            # SURVEY 8d, backward of one layer: read dOut (C_out), read the saved input (C_in), write dIn (C_in)
            byts = B * tvb * (HID + 2 * CHANNELS[-1])
            ach = byts / (probe_ms.value * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_layer_bwd_fused<2,4> (layer 4 backward, 64 -> 32 channels: data path and dA/dT in one "
                                               "wave-per-clip kernel)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": (traffic.get("bwd_fused layer4", {}).get("hbm_bytes_per_launch") if B == 4096 else None),
                    "traffic_source": "profiles/r02_hbm_traffic.json (PMC FETCH_SIZE/WRITE_SIZE, B=4096)",
                    "algorithmic_bytes_per_launch": byts,
                    "avg_launch_us": round(probe_ms.value * 1e3, 2), "launches": probe_n.value}
            # the same launches against the fp32 MFMA roof: convs Bt.dU, Br.dU (C_in x C_out each),
            # Kt.Z, Kr.X (C_in x C_in each) + forward temporal mix, both adjoint mixes, dA and dT (3 T + 2 V per element)
            ci, co = CHANNELS[-1], HID
            flops = B * (2 * T * V * (2 * ci * co + 2 * ci * ci) + 2 * ci * T * V * (3 * T + 2 * V))
            tf = flops / (probe_ms.value * 1e-3) / 1e12
            roof["mfma_f32"] = {"achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "flops_per_launch": flops}
        roof_lbw = None
        if lbw_n.value:
            byts = B * tvb * (HID + 2 * CHANNELS[-1])
            ach = byts / (lbw_ms.value * 1e-3) / 1e9
            roof_lbw = {"bound": "hbm", "what": "layer 4 backward, ALL its kernels (batch reductions, fp64 folds, data path, "
                        "dA/dT) against the layer's algorithmic bytes (read dOut, read saved input, write dIn)",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes": byts, "avg_us": round(lbw_ms.value * 1e3, 2), "calls": lbw_n.value}
        if fwd_n.value:
            byts = B * tvb * (CHANNELS[-1] + HID)        # layer 4 forward: read 32 channels, write 64
            ach = byts / (fwd_ms.value * 1e-3) / 1e9
            roof_fwd = {"bound": "hbm", "kernel": "k_layer_apply_ring<2,4,1> (layer 4 training forward from the stored Z, 32 -> 64 channels)",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": byts,
                        "traffic": (traffic.get("layer_apply layer4", {}).get("hbm_bytes_per_launch") if B == 4096 else None),
                        "avg_launch_us": round(fwd_ms.value * 1e3, 2), "launches": fwd_n.value}
        out = {
            "metric": "pose_clips_per_sec_fwd_bwd", "value": round(world * B * args.steps / dt, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "euclidean_encoder_dynamicCenter train step (fwd+mse-to-centre+L2 reg+bwd+Adam), "
                                   f"synthetic clips B={B}/GPU T={T} V={V} C={C_IN}, channels 2-32-16-32-64, latent 16",
                       "clips_per_gpu": B, "global_batch": B * world, "head": args.head,
                       "parallelism": f"dp{world}", "hip_graph": bool(args.graph)},
            "algorithmic_bytes_per_clip": {"fwd": fwd_b, "bwd": bwd_b},
            "step_hbm_frac": round(world * B * args.steps / dt * (fwd_b + bwd_b) / world / (HBM_PEAK_GBS * 1e9), 4),
            "final_loss": round(loss, 6),
            "forward_only": {"value": round(B / fwd_dt, 1), "unit": "clips/s per GPU", "ms": round(fwd_dt * 1e3, 4),
                             "hbm_frac": round(B / fwd_dt * fwd_b / (HBM_PEAK_GBS * 1e9), 4),
                             "what": "eval-mode STSE forward: ONE fused encoder kernel (activations resident in LDS / registers) + "
                                     "the split-K bottleneck, BN folded from running stats; hbm_frac prices SURVEY 8d's "
                                     "layer-materialised 236 704 B/clip against 8 TB/s (north_star target: 0.50)",
                             "fused_encoder_kernel_us": round(fz_ms.value * 1e3, 2) if fz_n.value else None,
                             # the fully fused path's own lower bound is input + latent only (1 696 B/clip): it is
                             # FMA-bound, so it is priced against the fp32 matrix peak (3 946 992 FLOP per clip)
                             "fused_inference": {"hbm_bytes_per_clip": 4 * C_IN * T * V + 4 * LATENT,
                                                 "hbm_frac": round(B / fwd_dt * (4 * C_IN * T * V + 4 * LATENT) / (HBM_PEAK_GBS * 1e9), 5),
                                                 "flops_per_clip": 3946992,
                                                 "mfma_f32_frac": round(B / fwd_dt * 3946992 / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)}},
            "roofline": roof,
            "roofline_fwd_layer4": roof_fwd,
            "roofline_layer4_backward": roof_lbw,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_fwd_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
