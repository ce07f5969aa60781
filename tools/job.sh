mkdir -p gpurun_out/r2b
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r2b/gpu_tests.txt
cd /tmp
for r in 1 2; do
for w in 2560 3072 8192; do
  echo "max wave mask words $w:"; CVX_MAX_WAVE_MASK_WORDS=$w python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --frames 256 --steps 5 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
done; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r2b/lds_split.txt
for w in 2560 8192; do
echo "4K config4 max wave mask words $w:"; CVX_MAX_WAVE_MASK_WORDS=$w python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --frames 128 --steps 4 --warmup 1 --width 3840 --height 2160 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r2b/lds_split.txt
