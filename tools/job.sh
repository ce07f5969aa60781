mkdir -p gpurun_out/r2i
bash tools/variants.sh "libcpuvox_gpu.so libcpuvox_gpu_age256.so libcpuvox_gpu_age512.so libcpuvox_gpu_age1024.so libcpuvox_gpu_age2048.so" --frames 512
