"""Scratch: per-kernel average launch time through the library's HIP-event probe (coskad_probe_*).
usage: python tools/probe_kernels.py [lib.so]   -- prints avg us for each (kernel id, C_in, C_out) of the B=4096 train step."""
import ctypes, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    shutil.copy(sys.argv[1], os.path.join(ROOT, "coskad_amd", "libcoskad_hip.so"))
import torch
from coskad_amd import _lib
from coskad_amd.trainer import STSETrainStep
from coskad_amd.models.sts.ae import STSE
from oracle import ref_cpu as R   # synthetic inputs only

B = 4096
torch.manual_seed(0)
model = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda().train()
eng = STSETrainStep(model, lr=1e-4, alpha=1e-6, head='euclidean')
x = R.synthetic_clips(B, 2, 12, 17, seed=100).cuda()
for _ in range(3):
    eng.step(x)
torch.cuda.synchronize()
lib = _lib.lib()
names = {1: "layer_apply", 2: "bwd_data", 3: "bwd_reduce", 4: "fwd_moments", 5: "gcn_params"}
chans = [2, 32, 16, 32, 64]
out = []
for kid in (1, 2, 3, 4, 5):
    for l in range(4):
        lib.coskad_probe_begin(kid, chans[l], chans[l + 1])
        for _ in range(5):
            eng.step(x)
        torch.cuda.synchronize()
        ms, n = ctypes.c_float(0), ctypes.c_int(0)
        lib.coskad_probe_end(ctypes.byref(ms), ctypes.byref(n))
        out.append(f"{names[kid]} L{l+1}: {ms.value*1e3:.1f}" if n.value else f"{names[kid]} L{l+1}: -")
print(" | ".join(out))
