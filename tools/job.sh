#!/bin/bash
mkdir -p gpurun_out/r05d
timeout -k 10 600 python3 tools/ab_fast.py "libcpuvox_gpu_base.so libcpuvox_gpu_v2a.so libcpuvox_gpu.so" --contexts 3 --latency 200 > gpurun_out/r05d/ab4.txt 2>&1
tail -10 gpurun_out/r05d/ab4.txt
