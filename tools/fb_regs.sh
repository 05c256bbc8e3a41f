#!/bin/bash
# register / spill summary of the fused backward kernels (cross-compile only): tools/fb_regs.sh [extra hipcc flags]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -mllvm -amdgpu-sched-strategy=max-ilp "$@" /root/repo/coskad_amd/csrc/fused_bwd.hip -o /tmp/fb_cur.s 2>&1 | grep -E "error" -A3
grep -E "\.vgpr_count|vgpr_spill|\.name:" /tmp/fb_cur.s | grep -A2 "name:.*k_layer_bwd_bpc" | grep -v "^--" | paste - - - | awk '{print substr($2,32,12), $4, $6}'
