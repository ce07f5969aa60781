#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05ic
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|SQC_" | head -40 > $R/gpurun_out/r05ic/avail.txt
cat $R/gpurun_out/r05ic/avail.txt | cut -c1-200 | head -40
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $R/gpurun_out/r05ic/p1 -- python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 128 --steps 2 --warmup 1 > $R/gpurun_out/r05ic/p1.log 2>&1; echo rc=$?
python3 - $R/gpurun_out/r05ic/p1 <<'PY'
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
