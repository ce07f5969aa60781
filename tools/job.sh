#!/bin/bash
# round-5 final: full GPU suite + soak + the whole profile collection with the FINAL library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05z
timeout 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05z/gputests.log 2>&1; echo "gputests rc=$?"; tail -3 gpurun_out/r05z/gputests.log
timeout 900 python3 tools/soak.py 2500 > gpurun_out/r05z/soak.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r05z/soak.txt
timeout 600 python3 tools/soak.py bench > gpurun_out/r05z/soak_bench.txt 2>&1; echo "soak bench rc=$?"; tail -1 gpurun_out/r05z/soak_bench.txt
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1; tail -1 gpurun_out/profile_round_r05.log | cut -c1-100
