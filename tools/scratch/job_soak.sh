#!/bin/bash
mkdir -p gpurun_out/final
sha256sum cpuvox_amd/libcpuvox_gpu.so | cut -c1-16
timeout -k 10 1150 python3 tools/soak.py 8000 > gpurun_out/final/soak_long.txt 2>&1; echo rc=$?; tail -2 gpurun_out/final/soak_long.txt
