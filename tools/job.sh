#!/bin/bash
mkdir -p gpurun_out/r05i
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
timeout -k 10 1000 python3 tools/soak.py 2500 > gpurun_out/r05i/soak.txt 2>&1; tail -1 gpurun_out/r05i/soak.txt
timeout -k 10 600 python3 tools/soak.py bench > gpurun_out/r05i/soak_bench.txt 2>&1; tail -1 gpurun_out/r05i/soak_bench.txt
bash tools/bbprof.sh bb21 32 > gpurun_out/r05i/bb.log 2>&1; tail -3 gpurun_out/r05i/bb.log
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
