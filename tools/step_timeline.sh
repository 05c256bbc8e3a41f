#!/bin/bash
# Launch-by-launch timeline of ONE train step of the bench workload (rocprofv3 --kernel-trace): kernel, workgroups, start offset,
# duration and the gap to the previous launch's end -> gpurun_out/timeline/step.txt
# usage (repo root, on the GPU box): [LEG=<leg of bench.py>] bash tools/step_timeline.sh [extra bench.py flags]
out=$PWD/gpurun_out/timeline
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$LEG" ]; then      # LEG=v25_encoder|v25_spherical_vae|mlp_projector|poincare_head: one of bench.py's legs instead
  out=$PWD/gpurun_out/timeline_$LEG
  mkdir -p $out
  rocprofv3 --kernel-trace --output-format csv -d $out/raw -o run -- python3 tools/bench_leg.py $LEG 4 > $out/log.txt 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d $out/raw -o run -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-legs --profile-only "$@" > $out/log.txt 2>&1
fi
python3 - <<PY
import csv, glob, re
f = glob.glob("$out/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"coskad::(?:\w+::)*(\w+(?:<[^>]*>)?)", n)
    return m.group(1).replace(" ", "") if m else n.split("(")[0][:60]
names = [short(r["Kernel_Name"]) for r in rows]
ad = [i for i, n in enumerate(names) if n.startswith("k_adam")]
lo, hi = ad[-2] + 1, ad[-1] + 1                      # the last whole step
t0 = int(rows[lo]["Start_Timestamp"]); prev = None; tot = 0; gaps = 0
with open("$out/step.txt", "w") as o:
    for r, n in zip(rows[lo:hi], names[lo:hi]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
        o.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  wg {wg:5d}  lds {r.get('LDS_Block_Size', '?'):>6s} vgpr {r.get('VGPR_Count', '?'):>4s}  {n}\n")
        tot += (e - s) / 1e3; gaps += max(gap, 0); prev = e
    o.write(f"launches {hi - lo}  kernel time {tot:.1f} us  gaps {gaps:.1f} us  span {(prev - t0) / 1e3:.1f} us\n")
print(open("$out/step.txt").read())
PY
