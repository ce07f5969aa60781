#!/bin/bash
mkdir -p gpurun_out/r05o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05o/gputests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05o/gputests.log
timeout -k 10 600 python3 tools/ab_fast.py "libcpuvox_gpu_base.so libcpuvox_gpu.so" --contexts 3 --latency 200 > gpurun_out/r05o/ab.txt 2>&1
tail -7 gpurun_out/r05o/ab.txt
