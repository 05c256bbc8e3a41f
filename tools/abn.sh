#!/bin/bash
# A/B/... of several builds inside ONE gpurun session: tools/abn.sh a b c  (expects tools/lib_<x>.so)
for r in 1 2; do
  for v in "$@"; do
    cp tools/lib_$v.so coskad_amd/libcoskad_hip.so
    echo -n "$v: "
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
