"""Train step of the default-width autoencoder (euclidean_autoencoder.py:106-118) at B = 4096, by the switches of the decoder's
structural paths: coskad_amd.lowrank.MODE (never / wide / always) x trainer.NARROW_OUT.  usage: python tools/bench_ae.py [V]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coskad_amd import lowrank, trainer                      # noqa: E402
from coskad_amd.models.sts.ae import STSAE                    # noqa: E402
from coskad_amd.utils.synthetic import synthetic_clips        # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 17
x = synthetic_clips(4096, 2, 12, V, seed=1).cuda()
for mode in ("never", "wide", "always"):
    for narrow in (False, True):
        lowrank.MODE, trainer.NARROW_OUT = mode, narrow
        torch.manual_seed(0)
        m = STSAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda().train()
        eng = trainer.STSAETrainStep(m, mode='ae', lr=1e-4, alpha=1e-6, lambda_=0.01)
        for _ in range(10):
            eng.step(x)
        torch.cuda.synchronize()
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(30):
                eng.step(x)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 30)
        print(f"V={V} lowrank={mode:6s} narrow={narrow!s:5s} folded={eng.lowrank is not None!s:5s}: {sorted(reps)[1] * 1e3:.3f} ms/step")
        del eng, m
        torch.cuda.empty_cache()
