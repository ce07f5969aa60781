mkdir -p gpurun_out/r2b
bash tools/variants.sh "libcpuvox_gpu_flat.so libcpuvox_gpu.so" > gpurun_out/r2b/variants_global.txt 2>&1
cat gpurun_out/r2b/variants_global.txt
timeout 1500 python3 -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r2b/baseline_configs.txt
