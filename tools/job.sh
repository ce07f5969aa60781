#!/bin/bash
# round-5 final library: long parity soak (20 000 random poses, counting + rendering build) + the 1000 benchmark poses
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05s
timeout 1100 python3 tools/soak.py 5000 > gpurun_out/r05s/soak_long.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r05s/soak_long.txt
