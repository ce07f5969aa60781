R=$(pwd)
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blit" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['phase2'])[:200])"; done
