"""World.DownSample on the device against the host build (BASELINE config 3 world by default).
Usage: python tools/downsample_bench.py [dim] ; prints one JSON line per LOD and a total."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cpuvox_amd import gpu, host  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
t0 = time.perf_counter()
ws = host.WorldSet.procedural(dim, dim, dim)
build_s = time.perf_counter() - t0
ctx = gpu.Context(0)
ctx.downsample(ws, 0, 5)  # warm-up: code object load, allocator
tot_host = tot_dev = tot_wall = 0.0
src_bytes = ws.info(0).byteLength
for extra in range(1, ws.lod_count):
    host_s, voxels = ws.downsample_host_seconds(extra)
    t0 = time.perf_counter()
    blob, columns, dev_voxels, dev_ms = ctx.downsample(ws, 0, extra)
    wall_s = time.perf_counter() - t0
    same = bool(np.array_equal(np.frombuffer(blob, dtype=np.uint8), ws.storage(extra)))
    tot_host += host_s
    tot_dev += dev_ms / 1e3
    tot_wall += wall_s
    print(json.dumps({"lod": extra, "voxels": dev_voxels, "host_voxels": voxels, "identical_to_host_build": same, "host_s": round(host_s, 3),
                      "host_threads": host.default_threads(), "device_ms": round(dev_ms, 2), "call_s_incl_validation_pcie": round(wall_s, 3),
                      "source_GBps_device": round(src_bytes / (dev_ms / 1e3) / 1e9, 1), "out_bytes": len(blob)}))
rebuilt = ctx.build_lods(ws)  # warm-up of the chain's kernels
chain_ms = []
for _ in range(3):
    t0 = time.perf_counter()
    rebuilt = ctx.build_lods(ws)
    chain_s = time.perf_counter() - t0
    chain_ms.append(ctx.last_build_lods_ms)
chain_same = all(bool(np.array_equal(rebuilt.storage(lod), ws.storage(lod))) for lod in range(1, ws.lod_count))
print(json.dumps({"world": f"proc{dim}", "build_lods_call_s": round(chain_s, 3), "build_lods_device_ms": round(min(chain_ms), 2), "build_lods_device_ms_runs": [round(v, 2) for v in chain_ms], "chain_identical_to_host_build": chain_same, "lod0_bytes": src_bytes, "lod0_voxels": ws.lod0_voxels, "world_build_s_host": round(build_s, 1),
                  "downsample_1_5_host_s": round(tot_host, 2), "downsample_1_5_device_s": round(tot_dev, 3),
                  "downsample_1_5_call_s": round(tot_wall, 2)}))
