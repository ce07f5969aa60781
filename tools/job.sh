#!/bin/bash
mkdir -p gpurun_out/r05x
timeout -k 10 900 python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_minreg.so libcpuvox_gpu_maxocc.so libcpuvox_gpu_ilp_postra.so" --contexts 3 --latency 100 > gpurun_out/r05x/flags2.txt 2>&1
tail -11 gpurun_out/r05x/flags2.txt
