mkdir -p gpurun_out/r2i
R=$(pwd)
echo "== blit tests"; timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blit" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
for blk in 64x4 16x16 8x8 32x8 16x4 8x32 128x2; do
  CVX_BLIT_BLOCK=$blk timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$blk', d['value'], json.dumps(d['phase2']))"
done
