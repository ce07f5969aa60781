#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "downsample or device_built" 2>&1 | tail -3
cd /tmp; timeout 600 python3 $R/tools/downsample_bench.py 2048 > $O/ds_bench.jsonl 2> $O/ds.err; cut -c1-220 $O/ds_bench.jsonl
