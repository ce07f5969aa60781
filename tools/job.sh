mkdir -p gpurun_out/r2e
timeout 900 python3 -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "rccl" 2>&1 | tail -5
cd /tmp
timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 5 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_default.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_default.err; echo rc=$?; tail -3 $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_default.err; cat $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_default.json
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 2 --warmup 1 --frames 16 --backend gloo > $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_gloo2.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_gloo2.err; echo rc=$?; tail -5 $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_gloo2.err; cat $GRAFT_REPO_ROOT/gpurun_out/r2e/bench_gloo2.json
