mkdir -p gpurun_out/r2h
bash tools/variants.sh "libcpuvox_gpu_base.so libcpuvox_gpu.so" --frames 512 2>&1 | tee gpurun_out/r2h/variants_cost.txt | tail -7
