#!/bin/bash
mkdir -p gpurun_out/r05s
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05s/t3.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05s/t3.log
timeout -k 10 600 python3 tools/ab_fast.py "libcpuvox_gpu_base.so libcpuvox_gpu.so" --contexts 3 --latency 200 > gpurun_out/r05s/ab2.txt 2>&1
tail -8 gpurun_out/r05s/ab2.txt
timeout -k 10 1000 python3 tools/soak.py 1500 > gpurun_out/r05s/soak.txt 2>&1; tail -1 gpurun_out/r05s/soak.txt
