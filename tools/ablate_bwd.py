"""Scratch: phase ablation of k_bwd_data_f (library built with -DCOSKAD_ABLATE, env COSKAD_ABL = bit mask of skipped phases).
Each mask runs in its own process (the mask is latched at first launch); prints avg us of the layer-4 / layer-2 launches."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from coskad_amd import _lib
    from coskad_amd.trainer import STSETrainStep
    from coskad_amd.models.sts.ae import STSE
    from oracle import ref_cpu as R
    torch.manual_seed(0)
    model = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda().train()
    eng = STSETrainStep(model, lr=1e-4, alpha=1e-6, head='euclidean')
    x = R.synthetic_clips(4096, 2, 12, 17, seed=100).cuda()
    for _ in range(3):
        eng.step(x)
    torch.cuda.synchronize()
    lib = _lib.lib()
    res = []
    kid = int(os.environ.get("ABL_KID", "2"))
    for ci, co in ((32, 64), (32, 16)):
        lib.coskad_probe_begin(kid, ci, co)
        for _ in range(5):
            eng.step(x)
        torch.cuda.synchronize()
        ms, n = ctypes.c_float(0), ctypes.c_int(0)
        lib.coskad_probe_end(ctypes.byref(ms), ctypes.byref(n))
        res.append(f"{ms.value*1e3:7.1f}")
    print(f"abl={int(os.environ.get('COSKAD_ABL','0')):3d}/{int(os.environ.get('COSKAD_ABLG','0')):3d}/{int(os.environ.get('COSKAD_DBG','0')):3d}  L4 {res[0]} us   L2 {res[1]} us", flush=True)
else:
    names = {0: "full", 1: "-stage", 2: "-phase0(Kr.X)", 4: "-gcn fwd", 8: "-dU loop", 16: "-Kt.Z", 32: "-unstage dZ", 64: "-gcn adj",
             128: "-epilogue", 255: "nothing but barriers", 8 + 16 + 2: "-all conv", 4 + 64: "-both gcn", 1 + 32 + 128: "-all global io except dU"}
    if len(sys.argv) > 1 and sys.argv[1] == "apply":   # k_layer_apply_m phases (runtime env COSKAD_DBG, no special build)
        names = {0: "full", 4: "-stage", 1: "-gcn", 2: "-conv", 8: "-store", 16: "-conv X part (global)", 32: "-conv Z part (LDS)",
                 7: "skeleton", 1 + 4: "-stage -gcn"}
        os.environ["ABL_KID"] = "1"
    if len(sys.argv) > 1 and sys.argv[1] == "gcn":   # k_bwd_gcn_params phases
        names = {0: "full", 64: "-stage X", 1: "-stage dZ", 2: "-temporal", 4: "-dA", 8: "-spatial adj", 16: "-restage X", 32: "-dT",
                 127: "barriers only", 4 + 32: "-dA -dT", 2 + 8: "-both mixing", 1 + 16 + 64: "-all staging"}
        os.environ["ABL_KID"] = "5"
    for m, nm in names.items():
        key = {"5": "COSKAD_ABLG", "1": "COSKAD_DBG"}.get(os.environ.get("ABL_KID", "2"), "COSKAD_ABL")
        env = dict(os.environ, **{key: str(m)})
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"{nm:28s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
