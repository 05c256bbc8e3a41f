#!/bin/bash
# A/B two builds of libcoskad_hip.so inside ONE gpurun session (same device, interleaved rounds).
# usage: tools/ab.sh   (expects tools/lib_a.so and tools/lib_b.so)
for r in 1 2; do
  for v in a b; do
    cp tools/lib_$v.so coskad_amd/libcoskad_hip.so
    echo -n "$v: "
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
