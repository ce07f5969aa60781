#!/bin/bash
# Executed-instruction counts of the render kernel for one build: tools/pmc_insts.sh <lib name under cpuvox_amd/> [bench args]
# (compare two builds by the SQ_INSTS_* per launch; same frames => same work)
L=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/insts_$L
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
export CVX_GPU_LIB=$R/cpuvox_amd/$L
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT" -- python3 "$R/bench.py" --cpu-seconds 0 --latency-frames 0 --frames 128 --steps 2 --warmup 1 "$@" > "$OUT/log.txt" 2>&1
echo "== $L rc=$?"
python3 "$R/tools/pmc_aggregate.py" "$OUT/.." "render_kernel<false>" 2>/dev/null | head -0
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(float); n = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "render_kernel<false>" in r["Kernel_Name"]:
            d[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(d):
    print(f"{k:28s} {d[k] / len(n[k]):.5g} per launch ({len(n[k])} launches)")
PY
