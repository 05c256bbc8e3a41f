#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/time_step.py (the bench workload's train step) -> gpurun_out/prof_step/stats.txt
out=$PWD/gpurun_out/prof_step
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/tools/time_step.py ${1:-20} > $out/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/run_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:60]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} share {100*float(r["TotalDurationNs"])/tot:5.1f}')
PY
