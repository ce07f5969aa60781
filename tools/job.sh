#!/bin/bash
mkdir -p gpurun_out/r05s
timeout -k 10 600 python3 tools/section_counts.py 128 > gpurun_out/r05s/counts.txt 2>&1
head -12 gpurun_out/r05s/counts.txt
