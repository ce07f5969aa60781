R=$(pwd)
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "blit or render_manager or baseline or example" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 --frames 512 --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['phase2']))"
