#!/bin/bash
# round-4 evidence: full GPU test-suite, gloo rehearsals of the N > 1 bench path (incl. --frames auto and the failure line), parity soaks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
timeout 1100 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 400 python3 bench.py --gpus 2 --backend gloo --frames 16 --steps 2 --warmup 1 --cpu-seconds 0 --latency-frames 0 > $O/bench_gloo_2ranks_raybuffer_gather.json 2> $O/gloo_rb.err; echo "gloo raybuffer rc=$?"; cut -c1-200 $O/bench_gloo_2ranks_raybuffer_gather.json
timeout 400 python3 bench.py --gpus 2 --backend gloo --gather image --frames auto --hbm-budget-gb 0.5 --steps 2 --warmup 1 --cpu-seconds 0 --latency-frames 0 > $O/bench_gloo_2ranks_image_gather_auto.json 2> $O/gloo_img.err; echo "gloo image auto rc=$?"; grep "frames auto" $O/gloo_img.err; cut -c1-200 $O/bench_gloo_2ranks_image_gather_auto.json
timeout 900 python3 tools/soak.py 1500 > $O/parity_soak.txt 2>&1; tail -2 $O/parity_soak.txt
timeout 900 python3 tools/soak.py bench > $O/parity_soak_bench.txt 2>&1; tail -2 $O/parity_soak_bench.txt
