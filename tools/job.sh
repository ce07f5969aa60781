#!/bin/bash
mkdir -p gpurun_out/r05g
timeout -k 10 900 python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_cb8x4.so libcpuvox_gpu_cb2x16.so libcpuvox_gpu_cb1x32.so libcpuvox_gpu_cb8x8.so libcpuvox_gpu_cb4x4.so" --contexts 2 > gpurun_out/r05g/abcb.txt 2>&1
tail -8 gpurun_out/r05g/abcb.txt
