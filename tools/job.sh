mkdir -p gpurun_out/r2i
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --world mill256 --width 640 --height 480 --frames 512 --steps 4 --warmup 1 --cpu-seconds 8 --latency-frames 100 > $R/gpurun_out/r2i/bench_config1.json 2>/dev/null; echo rc=$?
timeout 600 python3 $R/bench.py --world mill512 --frames 512 --steps 4 --warmup 1 --cpu-seconds 8 --latency-frames 100 > $R/gpurun_out/r2i/bench_config2.json 2>/dev/null; echo rc=$?
python3 -c "
import json
for n in (1,2):
    d=json.load(open('$R/gpurun_out/r2i/bench_config%d.json'%n)); print(n, d['value'], d['fps'], d['roofline']['frac'], d['latency']['ms'], d['cpu_baseline']['value'], d['parity_checked'])"
