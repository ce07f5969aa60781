#!/bin/bash
# static instruction counts of render_kernel<false> for the working tree: tools/static_counts.sh [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc -std=c++17 -Os -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -enable-post-misched=0 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I$R/include -I$R/cpuvox_amd/csrc -I$R/cpuvox_amd/csrc/host --cuda-device-only -S -o /tmp/sc.s $R/cpuvox_amd/csrc/cvx_gpu.hip "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "render_kernelILb0" | grep "VGPRs:\|Scratch" | sed 's/.*remark: *//' | tr '\n' ' '
awk '/^_ZN4cvxk13render_kernelILb0E.*:/{p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' /tmp/sc.s > /tmp/sc_rk.s
echo "| v_mov $(grep -c '^\s*v_mov' /tmp/sc_rk.s) valu $(grep -c '^\s*v_' /tmp/sc_rk.s) salu $(grep -c '^\s*s_' /tmp/sc_rk.s) branch $(grep -c '^\s*s_cbranch\|^\s*s_branch' /tmp/sc_rk.s)"
