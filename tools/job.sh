mkdir -p gpurun_out/r2e
cd /tmp
for cfg in "--frames 512" "--frames 128 --width 3840 --height 2160" "--frames 64 --width 3840 --height 2160 --world proc4096 --lod-error 4" "--frames 256 --width 2560 --height 1440" "--frames 512 --width 1280 --height 720"; do
for w in auto 2560 4352; do
echo "$cfg budget $w:"; if [ $w != auto ]; then export CVX_MAX_WAVE_MASK_WORDS=$w; else unset CVX_MAX_WAVE_MASK_WORDS; fi; python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --latency-frames 0 --steps 3 --warmup 1 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r2e/lds_budget_auto2.txt
