#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py's legs ON THE GPU BOX, one process per leg -> gpurun_out/<tag>_legs/<leg>/...kernel_stats.csv
# usage (repo root): gpurun -- 'bash tools/collect_leg_profiles.sh r03';  then copy the *_kernel_stats.csv into profiles/<tag>_<leg>_kernel_stats.csv
set -e
tag=${1:-r03}
export TMPDIR=/tmp
for leg in v25_encoder v25_spherical_vae mlp_projector wide_c256; do
  out=gpurun_out/${tag}_legs/$leg
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tools/bench_leg.py $leg 5 > $out/log.txt 2>&1
  echo "$leg done"
done
