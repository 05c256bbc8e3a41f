#!/bin/bash
# rocprofv3 PMC passes of tools/bench_btl_chain.py (the bottleneck backward + top-layer reductions kernel against the pair it replaces)
set -e
out=gpurun_out/pmc_btl
mkdir -p $out
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "MfmaUtil" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -o run -- python3 tools/bench_btl_chain.py > $out/$tag.log 2>&1 || echo "pass $tag failed"
  echo "pass $tag done"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_btl/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "btlnk" in k or "stats" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
