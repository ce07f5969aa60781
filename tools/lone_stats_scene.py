#!/usr/bin/env python3
"""Event counts of the latency kernel for one test scene (tests/scenes.py): python3 tools/lone_stats_scene.py <scene>   (needs the lonestats variant, see lone_stats.py)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CVX_GPU_LIB"] = os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_lonestats.so")
import scenes  # noqa: E402
from cpuvox_amd import gpu  # noqa: E402

name = sys.argv[1]
ws, fr, W, H = scenes.scene_frame(name)
ctx = gpu.Context(0)
ctx.upload_world(ws)
ctx.set_resolution(W, H)
ctx.set_latency_kernel(gpu.LATENCY_ALWAYS)
lib = ctypes.CDLL(os.environ["CVX_GPU_LIB"])
out = (ctypes.c_uint64 * 96)()
ctx.draw_segments(fr, 0)
lib.cvx_debug_lone_stats(out, 1)
ctx.draw_segments(fr, 0)
lib.cvx_debug_lone_stats(out, 0)
names = ["windows", "columns", "run projections", "side trips", "side pixels", "face trips", "face pixels", "side overlaps", "face overlaps", "processColumn", "... listed", "clipColumn",
         "... general form", "... window touched", "cullAndFilter", "runs inside the world bounds", "rays", "clean window", "", "", "", "clipped column hits"]
print(name, f"kernel {ctx.last_draw_ms():.3f} ms")
for i, n in enumerate(names):
    if n:
        print(f"  {n:30s} {out[i]:10d}   longest wave {out[48 + i]:8d}")
ctx.close()
