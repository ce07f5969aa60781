bash tools/variants.sh "libcpuvox_gpu_b0.so libcpuvox_gpu_loophints.so" --frames 512 2>&1 | grep -v "^Traceback\|^  File\|^    \|^json"
