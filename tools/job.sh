R=$(pwd); cd /tmp; export TMPDIR=/tmp
b() { timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"; }
for r in 1 2; do
b base
for w in 0.5 1 1.5; do CVX_TILE_COST_PIXELS=$w b pix$w; done
done
