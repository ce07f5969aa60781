mkdir -p gpurun_out/r2g
bash tools/variants.sh "libcpuvox_gpu_base.so libcpuvox_gpu.so" --frames 512 2>&1 | tee gpurun_out/r2g/variants_scan.txt | tail -8
