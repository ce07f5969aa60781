mkdir -p gpurun_out/r2d
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip 2>/dev/null
timeout 120 /tmp/valu_rate 2>&1 | tee gpurun_out/r2d/valu_rate.txt
