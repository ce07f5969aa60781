#!/bin/bash
# scratch job script of the round (what `gpurun -- 'bash tools/job.sh'` last ran for the committed artefacts): GPU tests, parity soaks, the profile collection
mkdir -p gpurun_out/final
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/gputests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/final/gputests.log
timeout -k 10 1000 python3 tools/soak.py 2500 > gpurun_out/final/soak.txt 2>&1; tail -1 gpurun_out/final/soak.txt
timeout -k 10 600 python3 tools/soak.py bench > gpurun_out/final/soak_bench.txt 2>&1; tail -1 gpurun_out/final/soak_bench.txt
bash tools/profile_round.sh r05 > gpurun_out/final/profile.log 2>&1; tail -1 gpurun_out/final/profile.log | cut -c1-100
