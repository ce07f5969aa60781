mkdir -p gpurun_out/r2d
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r2d/gpu_tests.txt
bash tools/variants.sh "libcpuvox_gpu_base.so libcpuvox_gpu.so" --frames 512 2>&1 | tee gpurun_out/r2d/variants_arena.txt
