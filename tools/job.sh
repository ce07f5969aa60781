#!/bin/bash
# everything profiles/r05_* is made from (tools/profile_round.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
tail -40 gpurun_out/profile_round_r05.log
