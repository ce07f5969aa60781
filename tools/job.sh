R=$(pwd); cd /tmp; export TMPDIR=/tmp
b() { timeout 600 python3 $R/bench.py --cpu-seconds 0 --latency-frames 0 "${@:2}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'])"; }
A="--frames 128 --width 3840 --height 2160 --steps 3 --warmup 1"
b 4k_auto $A
CVX_MAX_WAVE_MASK_WORDS=2560 b 4k_10KB $A
CVX_MAX_WAVE_MASK_WORDS=3072 b 4k_12KB $A
CVX_MAX_WAVE_MASK_WORDS=5120 b 4k_20KB $A
CVX_MAX_WAVE_MASK_WORDS=7680 b 4k_30KB $A
b 4k_auto_256f --frames 256 --width 3840 --height 2160 --steps 2 --warmup 1
b 1080_1024f --frames 1024 --steps 3 --warmup 1
b 1080_256f --frames 256 --steps 6 --warmup 1
b 1080_128f --frames 128 --steps 8 --warmup 1
