R=$(pwd)
timeout 1100 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 2 --backend gloo --frames 16 --steps 2 --warmup 1 2>/dev/null | grep '^{' > $R/gpurun_out/r2i/bench_gloo2.json; echo gloo rc=$?
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r2i/bench_gloo2.json')); print(d['n_gpus'], d['value'], d['config']['parallelism'], d['config']['exchange_verified'], d['config']['exchange_path'], d['config']['ranks_seen'])"
