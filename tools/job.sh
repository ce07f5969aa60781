#!/bin/bash
# round-5: rehearsal of the N > 1 bench path on one GPU over gloo (both gathers, --frames auto), smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05r
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05r/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05r/smoke.txt
timeout 900 python3 bench.py --gpus 2 --backend gloo --frames 16 --steps 3 --warmup 1 --cpu-seconds 0 --latency-frames 0 > gpurun_out/r05r/gloo2_raybuffer.json 2> gpurun_out/r05r/gloo2_raybuffer.err; echo "gloo raybuffer rc=$?"; tail -c 600 gpurun_out/r05r/gloo2_raybuffer.json
timeout 900 python3 bench.py --gpus 2 --backend gloo --frames auto --hbm-budget-gb 0.5 --gather image --steps 3 --warmup 1 --cpu-seconds 0 --latency-frames 0 > gpurun_out/r05r/gloo2_image_auto.json 2> gpurun_out/r05r/gloo2_image_auto.err; echo "gloo image rc=$?"; tail -c 400 gpurun_out/r05r/gloo2_image_auto.json; grep -i "auto" gpurun_out/r05r/gloo2_image_auto.err | tail -2
