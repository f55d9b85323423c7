#!/bin/bash
# Register budget of every kernel instantiation (no GPU needed): compile each .hip with -save-temps and list name / VGPRs / AGPRs / spills.
# Budgets that matter (two workgroups per CU): 8-wave (512-thread) kernels <= 128 VGPRs, 4-wave kernels <= 256; attention <2,3> <= 96
# (five waves per SIMD).  Run after any edit of a kernel template: one extra live array silently halves the occupancy of an instantiation
# (round 2: the tap-inner conv staging pushed the LayerNorm-fused 128x128 GEMM from 128 to 134 VGPRs, 82 -> 117 us per GEGLU launch).
set -e
cd "$(dirname "$0")/../neurons_amd/csrc"
T=$(mktemp -d)
for f in gemm rowpanel norm attention elementwise; do
  extra=""
  [ $f = attention ] && extra="-fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops $extra -I../../include -c $f.hip -o $T/$f.o -save-temps=obj 2>/dev/null
  grep -E "^\s+\.name:|\.vgpr_count|\.agpr_count|\.vgpr_spill_count" $T/$f-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | \
    sed -e 's/_ZN12_GLOBAL__N_1[0-9]*//' -e 's/\s\+/ /g' | awk -v f=$f '{print f": "$0}'
done
rm -rf $T
