#!/bin/bash
mkdir -p gpurun_out/r05z2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05z2/gputests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05z2/gputests.log
for i in 1 2; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --latency-frames 200 > gpurun_out/r05z2/bench$i.json 2> gpurun_out/r05z2/bench$i.err; python3 -c "
import json;d=json.load(open('gpurun_out/r05z2/bench$i.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms_avg'],d['latency']['ms'],d['latency']['pipelined_2deep']['ms'],d.get('parity_checked'))"; done
