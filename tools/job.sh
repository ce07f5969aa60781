mkdir -p gpurun_out/r2c
bash tools/variants.sh "libcpuvox_gpu_base.so libcpuvox_gpu.so libcpuvox_gpu_o2.so libcpuvox_gpu_o3.so" --frames 512 2>&1 | tee gpurun_out/r2c/variants_latency1.txt
