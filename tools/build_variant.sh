#!/bin/bash
# tools/build_variant.sh <name> [extra -D flags...]  -> tools/lib_<name>.so  (scratch A/B builds)
name=$1; shift
touch coskad_amd/csrc/*.hip
make -C coskad_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp $*" 2>&1 | grep -i "error" -A5
cp coskad_amd/libcoskad_hip.so tools/lib_$name.so
