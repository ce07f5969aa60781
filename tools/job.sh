R=$(pwd)
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_order or batch_equals or sharded or zero_copy" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
b() { timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 6 --warmup 2 "${@:2}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'])"; }
for r in 1 2; do
CVX_TILE_COST_MIDDLE_RAY=1 CVX_TILE_COST_PIXELS=1 b middle+pixels
b edge_max
CVX_TILE_COST_PIXELS=0.25 b edge_max+0.25pix
done
CVX_TILE_COST_MIDDLE_RAY=1 CVX_TILE_COST_PIXELS=1 b 4k_middle --frames 128 --width 3840 --height 2160 --steps 3 --warmup 1
b 4k_edge --frames 128 --width 3840 --height 2160 --steps 3 --warmup 1
