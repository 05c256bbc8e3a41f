#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers ON THE GPU BOX:
#   pass 1: --kernel-trace --stats          (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE  (own pass)    pass 3: --pmc WRITE_SIZE (own pass)
# (counters are never combined with hip/hsa/sys traces -- see MI355X_MICROARCH.md, HBM/rocprofv3 section)
# usage (from the repo root):  gpurun -- 'bash tools/collect_profiles.sh r01'
# then here:                   python tools/summarize_profiles.py r01
set -e
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
args="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs --profile-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $args > $out/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 $args > $out/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 $args > $out/write.log 2>&1
echo "write pass done"
# matrix-pipe and LDS counters (their own passes): rocprofv3's derived MfmaUtil (MFMA-busy cycles / (active cycles x SIMDs)),
# LDS bank-conflict cycles vs LDS-active cycles
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $out/mfma -o run -- python3 $args > $out/mfma.log 2>&1
echo "mfma pass done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/lds -o run -- python3 $args > $out/lds.log 2>&1
echo "lds pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/sq -o run -- python3 $args > $out/sq.log 2>&1
echo "sq pass done"
tail -1 $out/stats.log
