R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/icache
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc ${CVX_PMC:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY} --output-format csv -d $OUT/p1 -- python3 $R/tools/single_frames.py 50 > $OUT/p1.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(float); n = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0]
        if "lone_kernel" in k or "render_kernel" in k:
            d[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(d): print(k[0][-30:], k[1], f"{d[k]/len(n[k]):.5g}")
PY
