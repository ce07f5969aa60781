#!/bin/bash
mkdir -p gpurun_out/r05z4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cheap or scene or fuzz or run_rich or foreign or sparse or long_world" > gpurun_out/r05z4/t.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05z4/t.log; grep -n "^E  *Assert" gpurun_out/r05z4/t.log | cut -c1-300
timeout -k 10 600 python3 tools/ab_fast.py "libcpuvox_gpu_base.so libcpuvox_gpu.so" --contexts 3 --latency 200 > gpurun_out/r05z4/ab.txt 2>&1
tail -7 gpurun_out/r05z4/ab.txt
timeout -k 10 600 python3 tools/ab_fast.py "libcpuvox_gpu_base.so libcpuvox_gpu.so" --contexts 2 --width 3840 --height 2160 --frames 64 > gpurun_out/r05z4/ab4k.txt 2>&1
tail -3 gpurun_out/r05z4/ab4k.txt
