bash tools/variants.sh "libcpuvox_gpu_h2.so libcpuvox_gpu_h3.so" --frames 512 2>&1 | grep -v "^Traceback\|^  File\|^    \|^json"
