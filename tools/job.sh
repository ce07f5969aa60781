#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05d
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lod_chain or downsample or device_built or scan" > gpurun_out/r05d/ds_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r05d/ds_tests.log
timeout 600 python3 tools/downsample_bench.py 2048 > gpurun_out/r05d/ds_bench.jsonl 2> gpurun_out/r05d/ds_bench.err; cat gpurun_out/r05d/ds_bench.jsonl; tail -3 gpurun_out/r05d/ds_bench.err
