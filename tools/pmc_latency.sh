#!/bin/bash
# SQ counters of the single-frame path for one build: tools/pmc_latency.sh <lib under cpuvox_amd/> [poses [width height [world [lod-error]]]]
# (per launch = per frame; every kernel of the program that matches *_kernel is listed, so render_kernel and lone_kernel builds compare directly)
L=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmclat_$L
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
export CVX_GPU_LIB=$R/cpuvox_amd/$L
i=0
for counters in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES" \
                "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/single_frames.py" "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass$i rc=$? $(tail -1 $OUT/pass$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(float); n = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0]
        if "render_kernel" in k or "lone_kernel" in k:
            d[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add((p, r["Dispatch_Id"]))
for k in sorted(d):
    print(f"{k[0][-40:]:40s} {k[1]:26s} {d[k] / len(n[k]):.5g} per launch ({len(n[k])})")
PY
