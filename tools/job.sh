#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "downsample or device_built" 2>&1 | tail -3
cd /tmp
for L in libcpuvox_gpu_ds0.so libcpuvox_gpu_ds2k.so libcpuvox_gpu.so libcpuvox_gpu_ds4k.so; do
  echo "== $L"; CVX_GPU_LIB=$R/cpuvox_amd/$L timeout 600 python3 $R/tools/downsample_bench.py 2048 2> $O/$L.err | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'lod' in d: print('  lod',d['lod'],'device_ms',d['device_ms'],'identical',d['identical_to_host_build'])
    else: print('  chain device_ms',d['build_lods_device_ms'])
"
done 2>&1 | tee $O/ds_ab.log
