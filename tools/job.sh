#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
ROUNDS=4 bash $R/tools/variants.sh "libcpuvox_gpu.so libcpuvox_gpu_btab.so" > $O/ab.log 2>&1; head -4 $O/ab.log; tail -3 $O/ab.log
ROUNDS=2 bash $R/tools/variants_latency.sh "libcpuvox_gpu.so libcpuvox_gpu_btab.so" > $O/ab_lat.log 2>&1; cat $O/ab_lat.log
