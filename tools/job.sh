R=$(pwd)
bash tools/variants.sh "libcpuvox_gpu_b0.so libcpuvox_gpu.so" --frames 512 2>&1 | grep -v "^Traceback\|^  File\|^    \|^json"
cd /tmp; export TMPDIR=/tmp
for L in libcpuvox_gpu_b0.so libcpuvox_gpu.so libcpuvox_gpu_b0.so libcpuvox_gpu.so; do
CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --world mill512 --frames 256 --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mill512 $L', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
done
