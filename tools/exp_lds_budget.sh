#!/bin/bash
# LDS budget per wave (= resident waves per CU) against kernel time, experiment build: tools/exp_lds_budget.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
run() { # label, env, bench args...
  local label=$1 words=$2; shift 2
  if [ "$words" = auto ]; then E="A=1"; else E="CVX_MAX_WAVE_MASK_WORDS=$words"; fi
  env $E CVX_GPU_LIB=$R/cpuvox_amd/libcpuvox_gpu_exp.so timeout 300 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --steps 6 --warmup 2 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label words/lane $words:', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
}
for round in 1 2; do
for w in auto 2560 3072 3840 4352; do run "1080p 256 frames" $w --frames 256; done
for w in auto 2560 3072 3840 4352 5120 6144; do run "4K 2048^3 128 frames" $w --frames 128 --width 3840 --height 2160; done
done
