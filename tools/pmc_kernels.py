"""Summarises rocprofv3 --pmc csv output per kernel symbol: python tools/pmc_kernels.py <dir> [substring]"""
import csv, glob, sys
from collections import defaultdict
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_")
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        i = k.find("k_")
        k = k[i:i + 34].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        avg = {n: sum(v) / len(v) for n, v in c.items()}
        if "SQ_WAVE_CYCLES" in avg:
            wc = avg["SQ_WAVE_CYCLES"]
            print(f"{k:36s}", {n.replace("SQ_", ""): round(100 * v / wc, 1) for n, v in avg.items() if n not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES")})
        else:
            print(f"{k:36s}", {n.replace("SQ_", ""): round(v, 1) for n, v in avg.items()})
