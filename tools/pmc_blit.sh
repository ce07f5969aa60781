#!/bin/bash
# HBM traffic and instruction counters of the Phase-2 kernel (blit_batch_kernel) in the default bench command: tools/pmc_blit.sh <outdir>
# (separate --pmc passes, as tools/pmc_passes.sh; aggregated per launch with tools/pmc_aggregate.py)
set -u
OUT=$(realpath -m "$1"); shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pass$i" -- python3 "$R/bench.py" --cpu-seconds 0 --latency-frames 0 --steps 1 --warmup 0 > "$OUT/pass$i.log" 2>&1
  echo "pass$i ($counters) rc=$?"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
LIST
python3 "$R/tools/pmc_aggregate.py" "$OUT" "blit_batch_kernel" > "$OUT/pmc_blit_batch_kernel.csv"
rm -rf "$OUT"/pass*/runc "$OUT"/pass*/*/*agent_info.csv 2>/dev/null
cat "$OUT/pmc_blit_batch_kernel.csv"
