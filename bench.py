#!/usr/bin/env python
"""bench.py -- pose-clips/sec of the STS-GCN encoder train step (fwd + loss + reg + bwd + Adam).

Workload (BASELINE.json configs[1]): euclidean_encoder_dynamicCenter, synthetic clips
B=4096 per GPU, T=12, V=17, C=2, channels 2->32->16->32->64, latent 16, fp32.
  python bench.py --gpus N --steps K --warmup W
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's
form), or, when WORLD_SIZE is unset, this script starts those N ranks itself as a child process BEFORE touching the
GPU, forwards their output and exits with the child's code.
Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region.

roofline  : the step's dominant kernel -- k_layer_bwd_bpc<2,4,2,1>, layer 4's backward data path + dA / dT -- priced on SURVEY 8d's
            ALGORITHMIC bytes of that layer's backward (read dOut, read the saved input, write dIn: 104 448 B per clip) over its
            average launch time, measured with HIP events on the launch stream inside the timed region, against 8 TB/s
            (MI355X_MICROARCH.md).  Operands the kernel moves beyond 8d (the stored Z, the layer below's rows for the backward
            chain) are listed as `extra_operand_bytes`, never added to the numerator.  `roofline.layer` = every launch of the
            layer's backward call on the same bytes, `roofline.step` = the whole step on 8d's 590 976 B per clip,
            `roofline.kernels` = the per-kernel table (8d bytes, us, fraction) of the step's main kernels; the committed
            rocprofv3 version of that table is profiles/r04_roofline.json (tools/summarize_profiles.py).
legs      : the other shapes BASELINE.json / north_star name, each its own small timed loop (same protocol: warm-up,
            barrier + synchronize, K steps): wide 2-64-128-256-256, V = 25 encoder, V = 25 spherical VAE (config 4's
            model), Poincare head (config 3), `projector: mlp` (what 5 of the reference's 7 yamls select).
cpu_baseline: the CPU oracle (oracle/ref_cpu.py: plain PyTorch CPU ops, pinned to the reference by golden
            vectors) doing the same train step on this box's host cores (rank 0, N=1): B = 4096 (the GPU's shape) and
            SURVEY 8d's 3 + 10 protocol on a B = 1024 sample.
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


def algorithmic_bytes_per_clip(chans=None, tv=T * V, latent=LATENT):
    """every layer reads its input once and writes its output once (fp32)."""
    chans = chans or [C_IN] + CHANNELS + [HID]
    hid = chans[-1]
    fwd = sum(4 * tv * (ci + co) for ci, co in zip(chans[:-1], chans[1:])) + 4 * hid * tv + 4 * latent
    # backward per layer: read dOut, read saved input, write dIn (no dIn for layer 1); bottleneck: read U, dz; write dU
    bwd = (2 * 4 * hid * tv + 2 * 4 * latent)
    for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:])):
        bwd += 4 * tv * (co + ci + (ci if i > 0 else 0))
    return fwd, bwd


def algorithmic_flops_per_clip(chans, t=T, v=V, latent=LATENT):
    """forward = sum_layers [2 Ci V T^2 + 2 Ci T V^2 + 4 Ci Co T V] + 2 hid T V latent; fwd + bwd = 3 x forward."""
    f = sum(2 * ci * v * t * t + 2 * ci * t * v * v + 4 * ci * co * t * v for ci, co in zip(chans[:-1], chans[1:]))
    f += 2 * chans[-1] * t * v * latent
    return f, 3 * f


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


def cpu_baseline(full_b: int, sample_b: int, fwd_b: int):
    """(a) the train step (fwd + mse-to-centre + 1e-6 reg + bwd + Adam) at the GPU's batch `full_b` (1 warm-up, median of 3),
    (b) SURVEY 8d's protocol (3 warm-ups, median of 10) on `sample_b` clips, (c) the eval-mode forward under no_grad on
    `fwd_b` clips (3 + 10)."""
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

    def make_step(x):
        def step():
            opt.zero_grad(set_to_none=True)
            z = R.stse_encode(x, st, training=True)
            loss = R.mse_to_center(z, st["c"]) + 1e-6 * R.calc_reg_loss(list(params.items()))
            loss.backward()
            opt.step()
        return step

    dt_s = _median_time(make_step(R.synthetic_clips(sample_b, seed=1)), 3, 10)
    dt_f = _median_time(make_step(R.synthetic_clips(full_b, seed=3)), 1, 3) if full_b else None
    xf = R.synthetic_clips(fwd_b, seed=2)

    def fwd():
        with torch.no_grad():
            R.stse_encode(xf, st, training=False)

    dtf = _median_time(fwd, 3, 10)
    what = "train step (fwd+loss+reg+bwd+Adam), CPU oracle, synthetic clips [B,2,12,17], default stack 2-32-16-32-64, latent 16"
    out = {"value": round((full_b / dt_f) if dt_f else (sample_b / dt_s), 1), "unit": "clips/s", "cores": torch.get_num_threads(),
           "kind": "port", "cpu_model": _cpu_model(), "host_cores_visible": ncpu,
           "sample": (f"median of 3 {what} at B={full_b} (the GPU's batch) after 1 warm-up" if dt_f else
                      f"median of 10 {what} at B={sample_b} after 3 warm-ups"),
           "protocol_8d": {"value": round(sample_b / dt_s, 1), "unit": "clips/s", "ms": round(dt_s * 1e3, 1),
                           "sample": f"SURVEY 8d protocol: 3 warm-ups, median of 10 train steps at B={sample_b}"},
           "forward": {"value": round(fwd_b / dtf, 1), "unit": "clips/s", "ms": round(dtf * 1e3, 2),
                       "sample": f"median of 10 eval-mode forwards (no_grad) after 3 warm-ups on B={fwd_b} clips"}}
    if dt_f:
        out["ms_per_step"] = round(dt_f * 1e3, 1)
    return out


def run_legs(B, rank, world, sync, steps, warmup, only=None):
    """The other shapes the north_star names, each timed like the main loop (warm-up, sync, `steps` steps, sync)."""
    import torch
    from coskad_amd.models.sts.ae import STSE
    from coskad_amd.models.sts.vae import STSVAE
    from coskad_amd.trainer import STSAETrainStep, STSETrainStep, make_train_step
    from coskad_amd.utils.synthetic import synthetic_clips

    def timed(fn, k=steps, w=warmup):
        for _ in range(w):
            fn()
        sync()
        reps = []                                     # median of three blocks of k calls: one host hiccup does not decide a leg's figure
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            sync()
            reps.append((time.perf_counter() - t0) / k)
        return sorted(reps)[1]

    legs = {}

    def leg(name, build):
        if only is not None and name != only:
            return
        try:
            legs[name] = build()
        except Exception as e:            # a leg must never take the headline number down with it
            legs[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    def hbm(clips_s, bytes_clip):
        return round(clips_s * bytes_clip / (HBM_PEAK_GBS * 1e9), 4)

    def enc_leg(v, head, projector, what):
        def build():
            torch.manual_seed(0)
            m = STSE(C_IN, CHANNELS, HID, LATENT, T, v, 'sts_gcn', projector, 'euclidean', 0.0)
            m.c.fill_(0.1)
            eng = make_train_step(m.cuda().train(), lr=1e-4, alpha=1e-6, head=head)
            x = synthetic_clips(B, C_IN, T, v, seed=300 + rank).cuda()
            dt = timed(lambda: eng.step(x))
            fb, bb = algorithmic_bytes_per_clip(tv=T * v)
            r = {"workload": what, "engine": type(eng).__name__, "ms_per_step": round(dt * 1e3, 4),
                 "clips_per_s": round(world * B / dt, 1),
                 "roofline": {"bound": "hbm", "frac": hbm(B / dt, fb + bb), "algorithmic_bytes_per_clip": fb + bb, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s"}}
            m.eval()
            with torch.no_grad():
                dtf = timed(lambda: m(x), k=max(5, steps), w=3)
            r["forward_only"] = {"ms": round(dtf * 1e3, 4), "clips_per_s": round(B / dtf, 1),
                                 "layerwise_equiv_hbm_frac": hbm(B / dtf, fb)}
            return r
        return build

    leg("poincare_head", enc_leg(V, 'poincare', 'linear', f"hyperbolic_encoder train step (Poincare head: expmap0 / project / dist to the "
                                 f"centre, gyromidpoint sums), B={B}/GPU T={T} V={V}, default stack, latent {LATENT}"))
    leg("mlp_projector", enc_leg(V, 'euclidean', 'mlp', f"euclidean encoder with projector: 'mlp' (Linear-BatchNorm1d-ReLU-Linear, hidden "
                                 f"[{LATENT}]) train step + eval forward, B={B}/GPU T={T} V={V}, default stack"))
    leg("v25_encoder", enc_leg(25, 'euclidean', 'linear', f"euclidean encoder train step on the 25-joint layout, B={B}/GPU T={T} V=25, default "
                               f"stack, latent {LATENT}"))

    def vae_leg(v=25):
        torch.manual_seed(0)
        m = STSVAE(C_IN, CHANNELS, HID, 8, T, v, 'sts_gcn', 'mlp', 'euclidean', 0.0, distribution='ps')   # projector: spherical_vae.yaml:37
        eng = STSAETrainStep(m.cuda().train(), mode='vae', lr=1e-4, alpha=1e-6, phi=1.0, beta=1.0, gamma=1.0)
        x = synthetic_clips(B, C_IN, T, v, seed=400 + rank).cuda()
        dt = timed(lambda: eng.step(x))
        chans = [C_IN] + CHANNELS + [HID]
        fb, bb = algorithmic_bytes_per_clip(tv=T * v, latent=9)
        dchans = chans[::-1]
        fd = sum(4 * T * v * (ci + co) for ci, co in zip(dchans[:-1], dchans[1:]))
        bd = sum(4 * T * v * (co + 2 * ci) for ci, co in zip(dchans[:-1], dchans[1:]))
        total = fb + bb + fd + bd
        m.eval()
        with torch.no_grad():                                  # scoring forward: encoder, heads, sample, decoder (spherical_vae.py:76-78)
            dtf = timed(lambda: m(x), k=max(5, steps), w=3)
        m.train()
        fwd_only = {"ms": round(dtf * 1e3, 4), "clips_per_s": round(B / dtf, 1), "layerwise_equiv_hbm_frac": hbm(B / dtf, fb + fd),
                    "what": "eval-mode forward of the whole VAE (latent sampled as in the reference's scoring)"}
        return {"forward_only": fwd_only, "workload": f"spherical_vae train step (BASELINE config 4's model: STSVAE, projector 'mlp', PowerSpherical latent 8, decoder; phi MSE + "
                            f"beta KL + gamma mean(1/kappa) + alpha reg), B={B}/GPU T={T} V={v}, default widths", "engine": "STSAETrainStep",
                "ms_per_step": round(dt * 1e3, 4), "clips_per_s": round(world * B / dt, 1),
                "roofline": {"bound": "hbm", "frac": hbm(B / dt, total), "algorithmic_bytes_per_clip": total, "peak": HBM_PEAK_GBS, "unit": "GB/s"}}

    leg("v25_spherical_vae", vae_leg)
    leg("v17_spherical_vae", lambda: vae_leg(17))           # the shape of the reference's shipped config/UBnormal/spherical_vae.yaml

    def wide_leg():
        torch.manual_seed(0)
        chans = [C_IN, 64, 128, 256, 256]
        m = STSE(C_IN, chans[1:-1], chans[-1], LATENT, T, V, 'sts_gcn', 'linear', 'euclidean', 0.0)
        m.c.fill_(0.1)
        eng = make_train_step(m.cuda().train(), lr=1e-4, alpha=1e-6, head='euclidean')
        x = synthetic_clips(B, C_IN, T, V, seed=500 + rank).cuda()
        dt = timed(lambda: eng.step(x), k=max(3, steps // 4), w=2)
        _, fl = algorithmic_flops_per_clip(chans)
        tf = B / dt * fl / 1e12
        return {"workload": f"wide stack 2-64-128-256-256 (north_star's C=2->256) train step, B={B}/GPU T={T} V={V}, latent {LATENT}",
                "engine": type(eng).__name__, "ms_per_step": round(dt * 1e3, 3), "clips_per_s": round(world * B / dt, 1),
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "flops_per_clip": fl}}

    leg("wide_c256", wide_leg)
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU (weak scaling)")
    ap.add_argument("--head", default="euclidean", choices=["euclidean", "poincare"])
    ap.add_argument("--graph", type=int, default=int(os.environ.get("COSKAD_GRAPH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra workloads (wide, V=25, VAE, Poincare, mlp)")
    ap.add_argument("--leg-steps", type=int, default=20)
    ap.add_argument("--dry-collectives", action="store_true", help="rehearsal of the N-rank data-parallel step (any backend; `--gpus 2 "
                    "--backend gloo` runs on ONE GPU): after every step all ranks compare the two gradient buckets' element counts, the "
                    "1 / world folded into Adam and the step count, and at the end a checksum of the parameters (ranks train on different "
                    "clips: equal parameters mean the all-reduces did their job); prints backend / world / devices as the N-GPU line does")
    ap.add_argument("--profile-only", action="store_true", help="warm-up + the timed steps only, no probes (rocprofv3 passes of "
                    "tools/collect_profiles.sh: every kernel then runs exactly once per step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to "
                    "rehearse the multi-rank control flow on a single GPU)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="clips per CPU-baseline train step (SURVEY 8d protocol leg)")
    ap.add_argument("--cpu-full", type=int, default=4096, help="clips of the CPU-baseline step at the GPU's batch (0: skip)")
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

    if args.dry_collectives:
        if world < 2:
            raise SystemExit("--dry-collectives needs >= 2 ranks: python bench.py --gpus 2 --backend gloo --dry-collectives")
        n_total, tail = eng.fp.grad.numel(), eng.tail_off
        desc = {"bucket_bottleneck_elems": n_total - tail, "bucket_encoder_elems": tail, "gscale": 1.0 / eng.world, "world": eng.world}
        for _ in range(args.warmup + args.steps):
            eng.step(x)
            got = [None] * world
            dist.all_gather_object(got, dict(desc, steps=eng.steps))
            assert all(g == got[0] for g in got), f"ranks disagree on the step's collectives: {got}"
        sums = [None] * world
        dist.all_gather_object(sums, (float(eng.fp.flat.double().sum()), float(eng.fp.flat.double().abs().sum())))
        assert all(s_ == sums[0] for s_ in sums), f"parameters differ across ranks after {eng.steps} steps: {sums}"
        devs = [None] * world
        dist.all_gather_object(devs, f"rank {rank} pid {os.getpid()} cuda:{dev} {torch.cuda.get_device_name(dev)}")
        if rank == 0:
            print(json.dumps({"dry_collectives": "ok", "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
                              "world": world, "devices": devs, "steps_checked": eng.steps, **desc,
                              "allreduce_bytes_per_step": {"bucket_bottleneck (async, behind the bottleneck backward)": 4 * (n_total - tail),
                                                           "bucket_encoder (at the end of the backward)": 4 * tail},
                              "parameter_checksums_equal_on_all_ranks": True}), flush=True)
        dist.destroy_process_group()
        return
    for _ in range(args.warmup):
        eng.step(x)
    sync()
    # The step's dominant kernel is layer 4's fused backward (k_layer_bwd_bpc<2,4,2,1>: rocprof, profiles/*_kernel_instances.csv).
    # The library brackets every launch of it with HIP events on the launch stream (coskad_probe_*), inside the timed region.
    import ctypes
    lib = _lib.lib()
    KID_LAYER_APPLY, KID_BWD_DATA, KID_LAYER_BWD, KID_FUSED, KID_BTLNK_BWD = 1, 2, 6, 7, 8
    probing = not args.graph and not args.profile_only
    if probing:
        lib.coskad_probe_stride(8)                   # every 8th step's launch: a probed launch costs ~5 us of stream time
        lib.coskad_probe_begin(KID_BWD_DATA, CHANNELS[-1], HID)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = eng.step(x)
    sync()
    dt = time.perf_counter() - t0
    dom_ms_c, dom_n_c = ctypes.c_float(0), ctypes.c_int(0)
    if probing:
        lib.coskad_probe_end(ctypes.byref(dom_ms_c), ctypes.byref(dom_n_c))
        lib.coskad_probe_stride(1)
    dom_ms, dom_n = dom_ms_c.value, dom_n_c.value
    if args.profile_only:
        if rank == 0:
            print(json.dumps({"metric": "pose_clips_per_sec_fwd_bwd", "value": round(world * B * args.steps / dt, 1), "unit": "clips/s",
                              "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps, "profile_only": True}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    def probe_steps(kid, ci, co, n=5):
        # every rank runs these steps (they contain the gradient all-reduce); only rank 0 reads its probe
        ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
        if probing:
            lib.coskad_probe_begin(kid, ci, co)
            for _ in range(n):
                eng.step(x)
            sync()
            lib.coskad_probe_end(ctypes.byref(ms), ctypes.byref(cnt))
        return ms.value, cnt.value

    # secondary (outside the timed region): the per-kernel table -- every main kernel of the step through the same probe
    chans = [C_IN] + CHANNELS + [HID]
    probes = {}
    for li in range(4):
        probes[f"fwd L{li + 1}"] = probe_steps(KID_LAYER_APPLY, chans[li], chans[li + 1])
        probes[f"bwd L{li + 1}"] = probe_steps(KID_BWD_DATA, chans[li], chans[li + 1])
    probes["bwd bottleneck"] = probe_steps(KID_BTLNK_BWD, CHANNELS[-1], LATENT)
    lbw_ms, lbw_n = probe_steps(KID_LAYER_BWD, CHANNELS[-1], HID)
    fwd_ms, fwd_n = probes["fwd L4"]
    # training forward alone (chain + bottleneck: what runs in front of the head), outside the timed region
    from coskad_amd import engine as _engine, ops as _opsf
    def train_fwd():
        U, _ = _engine.chain_forward(x, eng.layers, True, eng.ws, want_ctx=True)
        return _opsf.btlnk_fwd(U, model.btlnk.weight, model.btlnk.bias, eng.layers[-1].slope, ws=eng.ws)
    for _ in range(3):
        train_fwd()
    sync()
    tf0 = time.perf_counter()
    for _ in range(20):
        train_fwd()
    sync()
    train_fwd_dt = (time.perf_counter() - tf0) / 20
    # forward-only (eval-mode encoder + bottleneck; SURVEY 8d's forward roofline target), outside the timed region
    model.eval()
    fz_ms, fz_n = ctypes.c_float(0), ctypes.c_int(0)
    with torch.no_grad():
        for _ in range(5):
            model(x)
        sync()
        reps = []                                         # the forward is ~6 launches: the median of 5 blocks of 20 calls, so that
        for _ in range(5):                                # one host hiccup does not decide the figure
            tf0 = time.perf_counter()
            for _ in range(20):
                model(x)
            sync()
            reps.append((time.perf_counter() - tf0) / 20)
        fwd_dt = sorted(reps)[len(reps) // 2]
        # the fused encoder kernel's own launches (HIP events on its stream), in a loop of their own: the two event records per
        # forward are ~7 % of a 0.23 ms forward
        lib.coskad_probe_begin(KID_FUSED, C_IN, HID)
        for _ in range(20):
            model(x)
        sync()
        lib.coskad_probe_end(ctypes.byref(fz_ms), ctypes.byref(fz_n))
    model.train()
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = float(stats[0])
    devices = [f"cuda:{dev} {torch.cuda.get_device_name(dev)}"]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank {rank} pid {os.getpid()} {devices[0]}")
        devices = gathered
    del eng
    torch.cuda.empty_cache()
    legs = None if (args.no_legs or args.graph) else run_legs(B, rank, world, sync, args.leg_steps, 5)

    if rank == 0:
        fwd_b, bwd_b = algorithmic_bytes_per_clip()
        tvb = 4 * T * V
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this script)
        traffic, traffic_src = {}, None
        for tag in ("r04", "r03", "r02"):
            try:
                with open(os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json")) as f:
                    traffic, traffic_src = json.load(f), f"profiles/{tag}_hbm_traffic.json"
                break
            except OSError:
                pass
        roof = roof_fwd = None
        ci, co = CHANNELS[-1], HID
        layer_bytes = B * tvb * (co + 2 * ci)      # SURVEY 8d, backward of one layer: read dOut (C_out), read the saved input (C_in), write dIn (C_in)
        from coskad_amd import engine as _eng, ops as _ops
        cb = CHANNELS[-2]
        chained = bool(_eng.FUSE_BELOW and _ops.layer_bwd_below_rows(B, ci, co, cb, T, V))
        # operands beyond 8d that the dominant kernel moves: the stored Z of its own layer, and (backward chain) the stored Z and
        # input of the layer below, whose batch reductions it forms from the dU rows it holds
        extra_bytes = B * tvb * ci + (B * tvb * 2 * cb if chained else 0)
        step_bytes = B * (fwd_b + bwd_b)
        step_s = dt / args.steps
        tr = lambda key: (traffic.get(key, {}).get("hbm_bytes_per_launch") if B == 4096 else None)
        if dom_n:
            ach = layer_bytes / (dom_ms * 1e-3) / 1e9
            # the same launches against the fp32 MFMA roof: convs Bt.dU, Br.dU (C_in x C_out each), Kt.Z, Kr.X
            # (C_in x C_in each) + forward temporal mix, both adjoint mixes, dA and dT (3 T + 2 V per element)
            flops = B * (2 * T * V * (2 * ci * co + 2 * ci * ci) + 2 * ci * T * V * (3 * T + 2 * V))
            if chained:
                flops += B * 2 * T * V * 2 * ci * cb          # P and Q of the layer below: C_in x C_in(below) each
            tf = flops / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "hbm", "kernel": "k_layer_bwd_bpc<2,4,2,1>",
                    "what": "the step's dominant kernel (layer 4 backward, 64 -> 32 channels: dZ, dA, dT, adjoint mixing, dXres, PReLU' "
                            "in one pass" + (", + the batch reductions of layer 3: backward chain" if chained else "") + ") on SURVEY 8d's "
                            "bytes of that layer's backward (read dOut 64 ch, read the saved input 32 ch, write dIn 32 ch = 104 448 B per clip)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": tr("bwd_fused layer4"),
                    "traffic_source": f"{traffic_src} (PMC: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, B=4096)",
                    "algorithmic_bytes_per_launch": layer_bytes, "extra_operand_bytes": extra_bytes,
                    "avg_launch_us": round(dom_ms * 1e3, 2), "launches": dom_n,
                    "probe": "HIP events on the launch stream around every 8th step's launch, inside the timed region",
                    "mfma_f32": {"achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "flops_per_launch": flops},
                    "step": {"bytes": step_bytes, "bytes_per_clip": fwd_b + bwd_b, "ms": round(step_s * 1e3, 4),
                             "achieved": round(step_bytes / step_s / 1e9, 1), "frac": round(step_bytes / step_s / (HBM_PEAK_GBS * 1e9), 4),
                             "what": "whole train step (driver-timed) on SURVEY 8d's 590 976 B per clip"}}
            if lbw_n:
                achl = layer_bytes / (lbw_ms * 1e-3) / 1e9
                roof["layer"] = {"what": "every launch of layer 4's backward call (fp64 fold, the dominant kernel, partial-row sums) on the "
                                         "same 8d bytes; the layer's batch reductions ride on the bottleneck backward's kernel "
                                         "(kernels['bwd bottleneck'])" if _eng.FUSE_TOP else
                                         "every launch of layer 4's backward call (batch reductions, fp64 fold, the dominant kernel, "
                                         "partial-row sums) on the same 8d bytes",
                                 "avg_us": round(lbw_ms * 1e3, 2), "launches": lbw_n, "achieved": round(achl, 1),
                                 "frac": round(achl / HBM_PEAK_GBS, 4)}
            # per-kernel table: SURVEY 8d's bytes of each layer's forward / backward against the layer's main kernel
            kb = {"bwd bottleneck": 2 * 4 * HID * T * V + 4 * LATENT}
            for li in range(4):
                kb[f"fwd L{li + 1}"] = tvb * (chans[li] + chans[li + 1])
                kb[f"bwd L{li + 1}"] = tvb * (chans[li + 1] + chans[li] + (chans[li] if li > 0 else 0))
            roof["kernels"] = {k: {"bytes_8d": B * kb[k], "avg_us": round(ms * 1e3, 2),
                                   "frac": round(B * kb[k] / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
                               for k, (ms, n) in probes.items() if n}
        if fwd_n:
            byts = B * tvb * (ci + co)        # layer 4 forward: read 32 channels, write 64
            ach = byts / (fwd_ms * 1e-3) / 1e9
            roof_fwd = {"bound": "hbm", "kernel": "k_layer_apply_bpc<2> (layer 4 training forward from the stored Z, 32 -> 64 channels: one clip per workgroup)",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": byts,
                        "traffic": tr("layer_apply layer4"),
                        "avg_launch_us": round(fwd_ms * 1e3, 2), "launches": fwd_n}
        kp_bytes = 13 * 4 * 64 * 4 * 4        # tile-major activation the fused encoder writes and the bottleneck reads back
        fused_actual = 4 * C_IN * T * V + 2 * kp_bytes + 4 * LATENT
        out = {
            "metric": "pose_clips_per_sec_fwd_bwd", "value": round(world * B * args.steps / dt, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "euclidean_encoder_dynamicCenter train step (fwd+mse-to-centre+L2 reg+bwd+Adam), "
                                   f"synthetic clips B={B}/GPU T={T} V={V} C={C_IN}, channels 2-32-16-32-64, latent 16",
                       "clips_per_gpu": B, "global_batch": B * world, "head": args.head,
                       "parallelism": f"dp{world}", "hip_graph": bool(args.graph)},
            "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if world > 1 else None,
            "world": world, "devices": devices,
            "algorithmic_bytes_per_clip": {"fwd": fwd_b, "bwd": bwd_b},
            "step_hbm_frac": round(world * B * args.steps / dt * (fwd_b + bwd_b) / world / (HBM_PEAK_GBS * 1e9), 4),
            "final_loss": round(loss, 6),
            "forward_only": {"value": round(B / fwd_dt, 1), "unit": "clips/s per GPU", "ms": round(fwd_dt * 1e3, 4), "ms_blocks": [round(r * 1e3, 4) for r in reps],
                             # SURVEY 8d's target prices a LAYER-MATERIALISED forward (236 704 B/clip); the fused kernel never moves
                             # those bytes, so this is an equivalent-throughput figure, not an HBM utilisation
                             "layerwise_equiv_hbm_frac": round(B / fwd_dt * fwd_b / (HBM_PEAK_GBS * 1e9), 4),
                             "hbm_frac_actual": round(B / fwd_dt * fused_actual / (HBM_PEAK_GBS * 1e9), 4),
                             "actual_bytes_per_clip": fused_actual,
                             "what": "eval-mode STSE forward: ONE fused encoder kernel (activations resident in LDS / registers) + "
                                     "the split-K bottleneck, BN folded from running stats.  layerwise_equiv_hbm_frac = clips/s x SURVEY 8d's "
                                     "236 704 B/clip / 8 TB/s (north_star target 0.50: >= 16.9 M clips/s); hbm_frac_actual = what the two "
                                     "kernels really move (clip in, tile-major activation out and back in, latent out)",
                             "fused_encoder_kernel_us": round(fz_ms.value * 1e3, 2) if fz_n.value else None,
                             # the fully fused path's own lower bound is input + latent only (1 696 B/clip): it is
                             # FMA-bound, so it is priced against the fp32 matrix peak (3 946 992 FLOP per clip)
                             "fused_inference": {"hbm_bytes_per_clip": 4 * C_IN * T * V + 4 * LATENT,
                                                 "hbm_frac": round(B / fwd_dt * (4 * C_IN * T * V + 4 * LATENT) / (HBM_PEAK_GBS * 1e9), 5),
                                                 "flops_per_clip": 3946992,
                                                 "mfma_f32_frac": round(B / fwd_dt * 3946992 / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)}},
            "train_forward": {"ms": round(train_fwd_dt * 1e3, 4), "clips_per_s": round(B / train_fwd_dt, 1),
                              "layerwise_hbm_frac": round(B / train_fwd_dt * fwd_b / (HBM_PEAK_GBS * 1e9), 4),
                              "what": "training-mode forward (batch-statistics BatchNorm: statistics, folds, apply kernels, bottleneck) on "
                                      "SURVEY 8d's 236 704 B per clip -- the layer-materialised schedule that bound describes"},
            "roofline": roof,
            "roofline_fwd_layer4": roof_fwd,
            "legs": legs,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_full if B == 4096 else 0, args.cpu_sample, args.cpu_fwd_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
