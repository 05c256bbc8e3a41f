#!/bin/bash
# A/B builds of csrc/fused_fwd.hip only (the rest of the library is linked as built).
#   tools/ab_fused.sh build <name> [hipcc flags...]   -> tools/libf_<name>.so      (here, no GPU needed)
#   tools/ab_fused.sh run <name> <name> ...            -> times tools/bench_fused.py per variant, two rounds (GPU box)
cmd=$1; shift
if [ "$cmd" = build ]; then
  name=$1; shift
  src=${AB_SRC:-fused_fwd}     # AB_SRC=fused_bwd tools/ab_fused.sh build ...
  cd coskad_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-variable "$@" -c $src.hip -o /tmp/ff_$name.o || exit 1
  objs=$(ls *.o | grep -v $src.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ff_$name.o -o ../../tools/libf_$name.so
else
  # variants are selected through COSKAD_LIB (coskad_amd/_lib.py): the shipped library is never overwritten
  for r in 1 2; do
    for v in "$@"; do
      echo -n "$v: "
      COSKAD_LIB=$PWD/tools/libf_$v.so timeout -k 10 120 python ${AB_BENCH:-tools/bench_fused.py} 2>&1 | tail -1
    done
  done
fi
