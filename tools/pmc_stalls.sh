#!/bin/bash
# Where the render kernel's wave cycles go (SQ wait / active counters), two passes: tools/pmc_stalls.sh <lib under cpuvox_amd/> [bench args]
L=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/stalls_$L
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
export CVX_GPU_LIB=$R/cpuvox_amd/$L
i=0
for counters in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
                "SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INST_CYCLES_SALU" \
                "SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pass$i" -- python3 "$R/bench.py" --cpu-seconds 0 --latency-frames 0 --frames 128 --steps 2 --warmup 1 "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass$i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(float); n = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "render_kernel<false>" in r["Kernel_Name"]:
            d[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add((p, r["Dispatch_Id"]))
for k in sorted(d):
    print(f"{k:28s} {d[k] / len(n[k]):.5g} per launch")
PY
