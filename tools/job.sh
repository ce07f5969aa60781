#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05v
timeout 900 python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_skyskip.so" --frames 256 --steps 3 --rounds 5 --contexts 3 --check-frames 16 --latency 100 > gpurun_out/r05v/runyy.txt 2>&1
tail -5 gpurun_out/r05v/runyy.txt
