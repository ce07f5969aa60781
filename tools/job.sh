#!/bin/bash
# final check of the tree as the driver will run it: smoke, the default bench line (traffic attached from profiles/, eight frames checked), GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], d['parity_checked'], d['parity']['frames'], d['parity']['pixels_compared'], d['parity']['pixels_differing'], d['latency']['ms'])"
