#!/bin/bash
# phase costs of k_bwd_data_bpc<25,..,true> (timing-only builds -DBDB_SKIP=mask -> tools/libf_bd<mask>.so): per-kernel time in the V = 25 encoder step
export TMPDIR=/tmp
for m in 0 1 2 4 8 16 32 63; do
  lib=""; [ $m != 0 ] && lib=$PWD/tools/libf_bd$m.so
  out=gpurun_out/bdph/$m; mkdir -p $out
  COSKAD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tools/bench_leg.py v25_encoder 3 > $out/log.txt 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv",recursive=True)[0]
r={x["Name"]:x for x in csv.DictReader(open(f))}
a=[float(v["AverageNs"])/1e3 for k,v in r.items() if "k_bwd_data_bpc<25, 2, 4" in k]
b=[float(v["AverageNs"])/1e3 for k,v in r.items() if "k_bwd_data_bpc<25, 1, 2" in k]
print("skip %3d: <2,4> %.1f us   <1,2> %.1f us" % ($m, a[0] if a else -1, b[0] if b else -1), flush=True)
PY
done
