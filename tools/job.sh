mkdir -p gpurun_out/r2i
bash tools/variants.sh "libcpuvox_gpu_nodefer.so libcpuvox_gpu.so" --frames 512
