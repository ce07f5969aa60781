#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05o
timeout 900 python3 tools/ab_fast.py "libcpuvox_gpu_head.so libcpuvox_gpu.so libcpuvox_gpu_q_noflush.so libcpuvox_gpu_q_noface.so libcpuvox_gpu_q_both.so" --frames 256 --steps 3 --rounds 3 --check-frames 2 --oracle-frames 0 > gpurun_out/r05o/spans.txt 2>&1
tail -8 gpurun_out/r05o/spans.txt
