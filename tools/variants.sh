#!/bin/bash
# A/B several builds of libcpuvox_gpu on ONE box: quick parity of each against the oracle, then interleaved bench rounds.
# usage: [ROUNDS=5] tools/variants.sh "<lib> <lib> ..." [bench args]      (lib = file name under cpuvox_amd/, e.g. libcpuvox_gpu.so)
LIBS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for L in $LIBS; do
  echo "== parity $L"
  CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scene_bit_exact or fuzz or batch_equals or sub_tile" 2>&1 | tail -1
done
cd /tmp; export TMPDIR=/tmp
LOG=/tmp/variants_$$.log; : > $LOG
for round in $(seq 1 ${ROUNDS:-5}); do
  for L in $LIBS; do
    CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 256 --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])" | tee -a $LOG
  done
done
echo "== kernel ms per build: median / min over the rounds"
python3 - $LOG <<'PY'
import sys, statistics, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    d[p[0]].append(float(p[-1]))
for k, v in d.items():
    print(f"{k:34s} median {statistics.median(v):.3f}  min {min(v):.3f}  ({len(v)} rounds)")
PY
