#!/bin/bash
# Build libcpuvox_gpu of another revision as an A/B partner: tools/build_at.sh <git rev> <name>  ->  cpuvox_amd/libcpuvox_gpu_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/cvx_wt_$2
rm -rf "$WT"; git -C "$R" worktree prune
git -C "$R" worktree add -f "$WT" "$1" > /dev/null 2>&1
make -C "$WT/cpuvox_amd/csrc" gpu > /dev/null
cp "$WT/cpuvox_amd/libcpuvox_gpu.so" "$R/cpuvox_amd/libcpuvox_gpu_$2.so"
git -C "$R" worktree remove --force "$WT"
echo "built cpuvox_amd/libcpuvox_gpu_$2.so from $1"
