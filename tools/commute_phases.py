"""Phase costs of k_commute_bwd (timing-only builds of the skip mask, COSKAD_CM_SKIP): python tools/commute_phases.py"""
import os, subprocess, sys
for mask in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 511):
    env = dict(os.environ, COSKAD_CM_SKIP=str(mask), PYTHONPATH=".")
    out = subprocess.run([sys.executable, "tools/bench_commute.py"], env=env, capture_output=True, text=True).stdout.strip()
    print(f"skip {mask:4d}: {out}", flush=True)
