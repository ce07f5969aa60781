#!/bin/bash
# round-4 GPU job: drain / budget rule / classes A/B + dispatch overlap trace
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04d; mkdir -p $O
ROUNDS=4 bash $R/tools/variants.sh "libcpuvox_gpu_r3like.so libcpuvox_gpu_drain.so libcpuvox_gpu_newrule1.so libcpuvox_gpu.so" > $O/ab_1080.log 2>&1; tail -6 $O/ab_1080.log
ROUNDS=3 bash $R/tools/variants_latency.sh "libcpuvox_gpu_r3like.so libcpuvox_gpu_drain.so" > $O/ab_lat.log 2>&1; cat $O/ab_lat.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 256 --steps 2 --warmup 1 > $O/trace_bench.json 2> $O/trace.err
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \; ; rm -rf $O/trace
grep render_kernel $O/kernel_trace.csv | cut -c1-300 | head -12; head -1 $O/kernel_trace.csv
