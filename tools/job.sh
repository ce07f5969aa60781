mkdir -p gpurun_out/r2h
cd /tmp
for r in 1 2; do
python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
for w in 2240 2560; do
CVX_MAX_WAVE_MASK_WORDS=$w CVX_GPU_LIB=$GRAFT_REPO_ROOT/cpuvox_amd/libcpuvox_gpu_lb5.so python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lb5 budget $w', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
done; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r2h/lb5.txt
