#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05x
timeout 900 python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_pix4.so" --width 3840 --height 2160 --frames 64 --steps 2 --rounds 4 --contexts 2 --check-frames 4 --oracle-frames 1 > gpurun_out/r05x/pix4_4k.txt 2>&1
tail -4 gpurun_out/r05x/pix4_4k.txt
timeout 900 python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_pix4.so" --frames 256 --steps 2 --rounds 4 --contexts 2 --check-frames 8 --oracle-frames 0 > gpurun_out/r05x/pix4_1080.txt 2>&1
tail -3 gpurun_out/r05x/pix4_1080.txt
