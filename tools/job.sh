#!/bin/bash
mkdir -p gpurun_out/r05l
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uploaded_again or device_built" > gpurun_out/r05l/t.log 2>&1; echo rc=$?; tail -5 gpurun_out/r05l/t.log
