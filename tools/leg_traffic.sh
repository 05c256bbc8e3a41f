#!/bin/bash
# HBM traffic per kernel of one bench leg (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes + a kernel trace for the
# durations) -> profiles/<tag>_<leg>_hbm_traffic.csv.  bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, gfx950 correction).
# usage (repo root, GPU box): bash tools/leg_traffic.sh <leg> <tag>
leg=${1:-v25_encoder}; tag=${2:-r04}
export TMPDIR=/tmp
out=gpurun_out/traffic_$leg; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 tools/bench_leg.py $leg 3 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 tools/bench_leg.py $leg 3 > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 tools/bench_leg.py $leg 3 > $out/trace.log 2>&1
python3 - <<PY
import csv, glob, re, collections
def short(n):
    m = re.search(r"coskad::(?:\w+::)*(\w+(?:<[^>]*>)?)", n)
    return m.group(1).replace(" ", "") if m else n.split("(")[0][:60]
cnt = {}
for c, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c: acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    cnt[c] = {k: sum(v) / len(v) for k, v in acc.items()}
f = glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)[0]
dur = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
rows = []
for k in cnt["FETCH_SIZE"]:
    if not k.startswith("k_"): continue
    b = (2 * cnt["FETCH_SIZE"][k] + cnt["WRITE_SIZE"].get(k, 0.0)) * 1024
    rows.append((dur.get(k, 0.0), k, cnt["FETCH_SIZE"][k], cnt["WRITE_SIZE"].get(k, 0.0), b))
with open("profiles/${tag}_${leg}_hbm_traffic.csv", "w") as o:
    o.write("kernel,avg_us,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_MB_per_launch,TB_per_s\n")
    for d, k, fe, wr, b in sorted(rows, reverse=True):
        o.write(f"\"{k}\",{d:.1f},{fe:.1f},{wr:.1f},{b / 1e6:.1f},{(b / 1e12) / (d / 1e6) if d else 0:.2f}\n")
print(open("profiles/${tag}_${leg}_hbm_traffic.csv").read())
PY
