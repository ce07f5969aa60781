#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py in small passes (one hardware block / few slots per pass; a pass that asks
# for more than the hardware can collect aborts and hangs, so every pass runs under its own timeout).
# usage: [PMC_ONLY_TRAFFIC=1] tools/pmc_passes.sh <outdir> [bench args...]     (PMC_ONLY_TRAFFIC: just the passes roofline.traffic needs)
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  if [ -n "${PMC_ONLY_TRAFFIC:-}" ]; then case "$counters" in FETCH_SIZE*|WRITE_SIZE*|TCC_EA0_*) ;; *) continue ;; esac; fi
  i=$((i+1))
  timeout ${PMC_PASS_TIMEOUT:-240} rocprofv3 --pmc $counters --output-format csv -d "$OUT/pass$i" -- python3 "$R/bench.py" "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass$i ($counters) rc=$?"
done <<'LIST'
TA_TA_BUSY_sum GRBM_GUI_ACTIVE
TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
LIST
