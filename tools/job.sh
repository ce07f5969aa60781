R=$(pwd); OUT=$R/gpurun_out/r2i/blitpmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for counters in "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $counters --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 256 --steps 1 --warmup 1 > $OUT/pass$i.log 2>&1
  echo pass$i rc=$?
done
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(float); n = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "blit_batch_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add((p, r["Dispatch_Id"]))
for k in sorted(d):
    print(f"{k:32s} {d[k] / len(n[k]):.5g} per launch ({len(n[k])} launches)")
PY
