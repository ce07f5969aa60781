#!/usr/bin/env python3
"""Renders N single frames (one blocking cvx_draw_segments each, the poses of tools/ab_latency.py) with the build $CVX_GPU_LIB selects: the
program tools/pmc_latency.sh profiles.  python3 tools/single_frames.py [poses] [width height] [world] [lod-error]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: F401,E402

from cpuvox_amd import gpu, host  # noqa: E402

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
world = sys.argv[4] if len(sys.argv) > 4 else "proc2048"
lod_error = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
dim = int(world[4:])
ws = host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, lod_error)
ctx = gpu.Context(0)
ctx.upload_world(ws)
ctx.set_resolution(W, H)
frames = []
for g in range(poses):
    pos, eul = host.sample_benchmark_path(((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH, ws.dims)
    frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
ctx.draw_segments(frames[0], 0)
t = time.perf_counter()
for fr in frames:
    ctx.draw_segments(fr, 0)
print(f"{poses} single frames {W}x{H} {world}: {(time.perf_counter() - t) / poses * 1e3:.4f} ms per call", flush=True)
ctx.close()
