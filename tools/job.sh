mkdir -p gpurun_out/r2e
bash tools/variants.sh "libcpuvox_gpu.so libcpuvox_gpu_ifcvt.so" --frames 512 2>&1 | tee gpurun_out/r2e/variants_ifcvt.txt
