#!/bin/bash
# single-frame latency A/B of several builds on ONE box (the `latency` leg of bench.py): tools/variants_latency.sh "<lib> <lib> ..."
LIBS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for round in $(seq 1 ${ROUNDS:-3}); do
  for L in $LIBS; do
    CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 300 python3 $R/bench.py --cpu-seconds 0 --frames 32 --steps 1 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); l=d['latency']; print('$L', 'latency ms', l['ms'], 'max', l['ms_max'], 'kernel', l['kernel_ms'])"
  done
done
