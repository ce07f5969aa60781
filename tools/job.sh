mkdir -p gpurun_out/r2g
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tee gpurun_out/r2g/gpu_tests.txt
