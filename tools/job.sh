#!/bin/bash
mkdir -p gpurun_out/r05y
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05y/gputests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05y/gputests.log
timeout -k 10 1000 python3 tools/soak.py 2500 > gpurun_out/r05y/soak.txt 2>&1; tail -1 gpurun_out/r05y/soak.txt
timeout -k 10 600 python3 tools/soak.py bench > gpurun_out/r05y/soak_bench.txt 2>&1; tail -1 gpurun_out/r05y/soak_bench.txt
bash tools/profile_round.sh r05 > gpurun_out/r05y/profile.log 2>&1; tail -1 gpurun_out/r05y/profile.log | cut -c1-100
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
