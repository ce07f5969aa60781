#!/bin/bash
# round-5: full GPU suite + parity soaks of the final library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05t
timeout 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05t/gputests.log 2>&1; echo "gputests rc=$?"; tail -3 gpurun_out/r05t/gputests.log
timeout 1000 python3 tools/soak.py 2500 > gpurun_out/r05t/soak.txt 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r05t/soak.txt
timeout 1000 python3 tools/soak.py bench > gpurun_out/r05t/soak_bench.txt 2>&1; echo "soak bench rc=$?"; tail -2 gpurun_out/r05t/soak_bench.txt
