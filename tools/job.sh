#!/bin/bash
mkdir -p gpurun_out/r05h
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05h/gputests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05h/gputests.log
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
bash tools/profile_round.sh r05 > gpurun_out/r05h/profile.log 2>&1; tail -3 gpurun_out/r05h/profile.log | cut -c1-200
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
