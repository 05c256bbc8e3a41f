#!/bin/bash
# rocprofv3 passes for the fused eval-mode forward (tools/bench_fused.py): kernel stats + SQ / LDS / MFMA counters.
# usage (repo root): gpurun -- 'bash tools/prof_fused.sh r02_fused'
set -e
tag=${1:-r02_fused}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 tools/bench_fused.py > $out/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/sq -o run -- python3 tools/bench_fused.py > $out/sq.log 2>&1
echo "sq pass done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $out/lds -o run -- python3 tools/bench_fused.py > $out/lds.log 2>&1
echo "lds pass done"
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $out/mfma -o run -- python3 tools/bench_fused.py > $out/mfma.log 2>&1
echo "mfma pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 tools/bench_fused.py > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 tools/bench_fused.py > $out/write.log 2>&1
echo "hbm passes done"
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "lds", "mfma", "fetch", "write"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if "fused" in k or "btlnk" in k:
                print(sub, k, {c: round(v / n[(k, c)], 1) for c, v in d.items()})
PY
