bash tools/variants.sh "libcpuvox_gpu.so libcpuvox_gpu_notm.so libcpuvox_gpu_tdp4.so libcpuvox_gpu_tdp8.so libcpuvox_gpu_o2.so" --frames 512 2>&1 | grep -v "^Traceback\|^  File\|^    \|^json"
