#!/bin/bash
# FETCH_SIZE calibration on the GPU box: tools/fetch_calibration.sh <tag>  -> gpurun_out/<tag>/{plain.csv, pass*/, counters_tcc.txt, report.md}
TAG=${1:-fetchcal}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_cal "$R/tools/fetch_calibration.hip" 2> "$OUT/build.log" || { tail -5 "$OUT/build.log"; exit 1; }
timeout 300 /tmp/fetch_cal > "$OUT/plain.csv" 2> "$OUT/plain.err" || { echo "plain run failed"; tail -3 "$OUT/plain.err"; exit 1; }
cat "$OUT/plain.csv"
timeout 120 rocprofv3 -L > "$OUT/counters_all.txt" 2>&1; grep -i "TCC_EA0_RD\|TCC_EA0_WR\|FETCH_SIZE\|WRITE_SIZE\|TCC_HIT\|TCC_MISS\|TCC_REQ\|TCC_READ\|TCC_BUBBLE\|RDREQ" "$OUT/counters_all.txt" | sort -u | head -80 > "$OUT/counters_tcc.txt"
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pass$i" -- /tmp/fetch_cal > "$OUT/pass$i.log" 2>&1
  echo "pass$i ($counters) rc=$?"
done <<'LIST'
FETCH_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_REQ_sum TCC_READ_sum
TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
LIST
python3 "$R/tools/fetch_calibration.py" "$OUT" > "$OUT/report.md" 2> "$OUT/report.err"; cat "$OUT/report.md"
find "$OUT" -name "*agent_info.csv" -delete 2>/dev/null
