mkdir -p gpurun_out/r2f
cd /tmp; export TMPDIR=/tmp
for L in libcpuvox_gpu.so libcpuvox_gpu_lanemajor.so; do
CVX_GPU_LIB=$GRAFT_REPO_ROOT/cpuvox_amd/$L python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
for c in "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
CVX_GPU_LIB=$GRAFT_REPO_ROOT/cpuvox_amd/$L timeout 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2f/pmc_$L -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 2 --warmup 1 > /dev/null 2>&1
done
python3 - $GRAFT_REPO_ROOT/gpurun_out/r2f/pmc_$L <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(float); n = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "render_kernel<false>" in r["Kernel_Name"]:
            d[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add((p, r["Dispatch_Id"]))
for k in sorted(d): print(f"   {k:16s} {d[k] / len(n[k]):.5g} per launch")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2f/pmc_$L
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r2f/store_layout.txt
