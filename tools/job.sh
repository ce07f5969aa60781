#!/bin/bash
mkdir -p gpurun_out/r05l
for g in raybuffer image; do
timeout -k 10 500 python3 bench.py --gpus 2 --backend gloo --gather $g --steps 3 --warmup 1 --frames 64 --cpu-seconds 0 > gpurun_out/r05l/gloo_$g.json 2> gpurun_out/r05l/gloo_$g.err; echo "rc=$?"; python3 -c "
import json;d=json.load(open('gpurun_out/r05l/gloo_$g.json'));print(d.get('value'),d.get('exchange_verified'),d.get('exchange_path'),d.get('n_gpus'))"
done
python3 - <<'PY'
import time,sys
sys.path.insert(0,'.')
from cpuvox_amd import gpu,host
ws=host.WorldSet.procedural(2048,2048,2048,0x5EED2048)
ctx=gpu.Context(0)
t=time.time(); ctx.upload_world(ws); print('upload_world proc2048 (6 levels, host side): %.2f s'%(time.time()-t))
ctx.close()
PY
