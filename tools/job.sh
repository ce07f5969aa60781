mkdir -p gpurun_out/r2h
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tee gpurun_out/r2h/gpu_tests.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29513 $GRAFT_REPO_ROOT/bench.py --gpus 3 --steps 2 --warmup 1 --frames 12 --backend gloo > $GRAFT_REPO_ROOT/gpurun_out/r2h/bench_gloo3.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2h/bench_gloo3.err; echo rc=$?; python3 -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/r2h/bench_gloo3.json')); print(d['n_gpus'], d['config']['ranks_seen'], d['config']['exchange_verified'], d['config']['exchange_path'], d['value'])"
