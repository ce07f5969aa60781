R=$(pwd)
timeout 1100 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $R/gpurun_out/r2i/bench_final.json 2> $R/gpurun_out/r2i/bench_final.err; echo bench rc=$?
python3 -c "import json; d=json.load(open('$R/gpurun_out/r2i/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity_checked'], d['latency']['ms'], d['phase2']['blit_ms_per_frame'], d['cpu_baseline']['value'])"
