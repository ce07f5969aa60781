mkdir -p gpurun_out/r2i
R=$(pwd)
echo "== SM parity test"; timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "state_machine" 2>&1 | tail -3
echo "== section profile, 1 frame"; python3 tools/section_profile.py --frames 1
echo "== section profile, 4 frames"; python3 tools/section_profile.py --frames 4
bash tools/variants.sh "libcpuvox_gpu.so libcpuvox_gpu_fold1.so libcpuvox_gpu_fold8.so libcpuvox_gpu_pnf6.so" --frames 512
