#!/bin/bash
# A/B two builds of libcpuvox_gpu on the SAME box, interleaved (cdna guide rule 24): tools/ab.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for round in 1 2 3; do
  for L in "$A" "$B"; do
    CVX_GPU_LIB=$R/cpuvox_amd/$L python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
  done
done
