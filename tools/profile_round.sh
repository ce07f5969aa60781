#!/bin/bash
# Everything profiles/rNN_* is made from, in one go on the GPU box: tools/profile_round.sh <tag, e.g. r02>
# (kernel trace + stats, PMC passes, SQ stall / instruction counters of the default bench command; results under gpurun_out/<tag>/)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
# 1. kernel trace + stats of the default command (no CPU leg: the profiler would only see it as idle time)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --cpu-seconds 0 --latency-frames 0 > "$OUT/trace_bench.json" 2> "$OUT/trace.log"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# 2. counter passes (separate runs, --pmc only) with the DRIVER's step counts (every BENCH_rNN.json: --steps 20 --warmup 5): the summary lists every
# launch, so bench.py assembles the timed steps of any run these 25 launches cover -- the default 10 / 2 as well (round 4 collected 10 / 2 and the
# driver's line went without its traffic)
DRV="--steps 20 --warmup 5"
bash "$R/tools/pmc_passes.sh" "$OUT/pmc" --cpu-seconds 0 --latency-frames 0 $DRV > "$OUT/pmc_passes.log" 2>&1
python3 "$R/tools/pmc_aggregate.py" "$OUT/pmc" "render_kernel<false>" --cpu-seconds 0 --latency-frames 0 $DRV > "$OUT/pmc_render_kernel.csv"
# 3. the bench line of this build with the measured traffic attached (and the CPU leg, parity check, latency legs)
timeout 900 python3 "$R/bench.py" --pmc-csv "$OUT/pmc_render_kernel.csv" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
# 3b. the driver's own command line, counters looked up as the driver's run will (no --pmc-csv: the file must first be copied to profiles/; here it is only a rehearsal of the timing)
timeout 900 python3 "$R/bench.py" --gpus 1 $DRV --pmc-csv "$OUT/pmc_render_kernel.csv" > "$OUT/bench_driver_args.json" 2> "$OUT/bench_driver_args.err"
# 4. SQ counters (128 frames per launch)
bash "$R/tools/pmc_stalls.sh" libcpuvox_gpu.so > "$OUT/sq_counters.txt" 2>&1
bash "$R/tools/pmc_insts.sh" libcpuvox_gpu.so >> "$OUT/sq_counters.txt" 2>&1
# 5. the 4K configurations (BASELINE.json configs 4 and 5 on one GPU): their own stamped traffic counters, then the bench line WITH the CPU leg (parity_checked)
# (256 frames per launch: ~2 - 3 M rays, like the 512 frames of the 1080p default; with the 128 / 64 frames of the round's first collections the launches were too short --
# config 5 gains 10 % from 64 -> 256 frames, config 4 2 % from 128 -> 256, the 1080p default 1 % from 512 -> 1024)
C4="--frames 256 --steps 4 --warmup 1 --width 3840 --height 2160"
C5="--frames 256 --steps 4 --warmup 1 --width 3840 --height 2160 --world proc4096 --lod-error 4"
PMC_ONLY_TRAFFIC=1 bash "$R/tools/pmc_passes.sh" "$OUT/pmc4" --cpu-seconds 0 --latency-frames 0 $C4 > "$OUT/pmc4_passes.log" 2>&1
python3 "$R/tools/pmc_aggregate.py" "$OUT/pmc4" "render_kernel<false>" --cpu-seconds 0 --latency-frames 0 $C4 > "$OUT/pmc_render_kernel_config4.csv"
timeout 900 python3 "$R/bench.py" --cpu-seconds 15 --latency-frames 0 $C4 --pmc-csv "$OUT/pmc_render_kernel_config4.csv" > "$OUT/bench_config4_1gpu.json" 2> "$OUT/bench_config4.err"
PMC_ONLY_TRAFFIC=1 PMC_PASS_TIMEOUT=400 bash "$R/tools/pmc_passes.sh" "$OUT/pmc5" --cpu-seconds 0 --latency-frames 0 $C5 > "$OUT/pmc5_passes.log" 2>&1
python3 "$R/tools/pmc_aggregate.py" "$OUT/pmc5" "render_kernel<false>" --cpu-seconds 0 --latency-frames 0 $C5 > "$OUT/pmc_render_kernel_config5.csv"
timeout 1100 python3 "$R/bench.py" --cpu-seconds 15 --latency-frames 0 $C5 --pmc-csv "$OUT/pmc_render_kernel_config5.csv" > "$OUT/bench_config5_1gpu.json" 2> "$OUT/bench_config5.err"
rm -rf "$OUT/trace" "$OUT"/pmc*/pass*/runc "$OUT"/pmc*/pass*/*/*agent_info.csv 2>/dev/null
ls -la "$OUT"
cat "$OUT/kernel_stats.csv" | head -8
cat "$OUT/pmc_render_kernel.csv"
# 6. single-frame latency (the reference's call pattern): kernel trace of 50 blocking draws + the SQ counters of the latency kernel
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_latency" -- python3 "$R/tools/single_frames.py" 200 > "$OUT/trace_latency.log" 2>&1
find "$OUT/trace_latency" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_single_frames.csv" \;
rm -rf "$OUT/trace_latency"
bash "$R/tools/pmc_latency.sh" libcpuvox_gpu.so 50 > "$OUT/sq_counters_single_frames.txt" 2>&1
cat "$OUT/kernel_stats_single_frames.csv" | head -5
