bash tools/profile_round.sh r02h > gpurun_out/profile_round_r02h.log 2>&1
tail -3 gpurun_out/profile_round_r02h.log
