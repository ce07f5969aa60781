mkdir -p gpurun_out/r2g
timeout 2400 python3 tools/soak.py 5000 2>&1 | tee gpurun_out/r2g/parity_soak.txt | tail -8
