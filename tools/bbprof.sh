#!/bin/bash
# Basic-block profile of render_kernel<false> on the GPU box: tools/bbprof.sh <tag> [frames]   -> gpurun_out/<tag>/bbprof.txt
TAG=${1:-bbprof}; FRAMES=${2:-32}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; WORK=/tmp/bbprof_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
python3 "$R/tools/bbprof.py" build "$WORK" -j 16 > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
tail -1 "$OUT/build.log"
timeout 1500 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d "$WORK/pmc" -- python3 "$R/tools/bbprof.py" drive "$WORK" "$FRAMES" > "$OUT/drive.log" 2>&1
echo "drive rc=$?"; tail -2 "$OUT/drive.log"
python3 "$R/tools/bbprof.py" report "$WORK" "$WORK/pmc" > "$OUT/bbprof.txt" 2> "$OUT/report.err"
cp "$WORK/blocks.json" "$WORK/profile.json" "$WORK/order.json" "$OUT/" 2>/dev/null
mkdir -p "$OUT/pmc"; find "$WORK/pmc" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc/" \;
cp "$WORK/base/cvx_gpu-hip-amdgcn-amd-amdhsa-gfx950.s" "$OUT/device.s"
head -30 "$OUT/bbprof.txt"
