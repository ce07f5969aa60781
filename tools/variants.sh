#!/bin/bash
# A/B several builds of libcpuvox_gpu on ONE box: quick parity of each against the oracle, then interleaved bench rounds.
# usage: tools/variants.sh "<lib> <lib> ..." [bench args]      (lib = file name under cpuvox_amd/, e.g. libcpuvox_gpu.so)
LIBS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for L in $LIBS; do
  echo "== parity $L"
  CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scene_bit_exact or fuzz or batch_equals or sub_tile" 2>&1 | tail -3
done
cd /tmp; export TMPDIR=/tmp
for round in 1 2 3; do
  for L in $LIBS; do
    CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 256 --steps 6 --warmup 2 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
  done
done
