#!/bin/bash
mkdir -p gpurun_out/r05g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05g/gputests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05g/gputests.log
timeout -k 10 900 python3 tools/soak.py 1000 > gpurun_out/r05g/soak.txt 2>&1; tail -1 gpurun_out/r05g/soak.txt
