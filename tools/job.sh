bash tools/profile_round.sh r02g > gpurun_out/profile_round_r02g.log 2>&1
tail -5 gpurun_out/profile_round_r02g.log
