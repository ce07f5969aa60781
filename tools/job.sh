#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05y
timeout 900 python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_straddle.so" --frames 256 --steps 3 --rounds 5 --contexts 3 --check-frames 16 --latency 100 > gpurun_out/r05y/straddle.txt 2>&1
tail -7 gpurun_out/r05y/straddle.txt
