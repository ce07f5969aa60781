#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_rich or long_world or scene_bit_exact or fuzz" 2>&1 | tail -5
timeout 900 python3 -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "bench_launch_512" 2>&1 | tail -5
timeout 600 python3 bench.py --cpu-seconds 0 --latency-frames 100 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; python3 -c "import json; d=json.load(open('$O/bench.json')); print(d['roofline'], d['latency'])"
