R=$(pwd); cd /tmp; export TMPDIR=/tmp
for r in 1 2; do
for L in libcpuvox_gpu.so libcpuvox_gpu_age256.so libcpuvox_gpu_age1024.so; do
  CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 200 --frames 64 --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); l=d['latency']; print('$L', 'latency ms', l['ms'], 'kernel', l['kernel_ms'], 'max', l['ms_max'])"
done
done
