bash tools/profile_round.sh r02f > gpurun_out/profile_round_r02f.log 2>&1
tail -30 gpurun_out/profile_round_r02f.log
