mkdir -p gpurun_out/r2f
cd /tmp
for r in 1 2; do for L in libcpuvox_gpu.so libcpuvox_gpu_v100.so libcpuvox_gpu_pk100.so libcpuvox_gpu_slow100.so; do
CVX_GPU_LIB=$GRAFT_REPO_ROOT/cpuvox_amd/$L python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
done; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r2f/sensitivity_pk.txt
