#!/usr/bin/env python3
"""Event counts of the latency kernel (cvx_lone.h) per frame: python3 tools/lone_stats.py [poses] [width height] [world] [lod-error]   (CVX_POSE_INDEX=i: that sample of the benchmark path, `poses` times)
needs the counting variant: make -C cpuvox_amd/csrc variant NAME=lonestats DEFS=-DCVX_LONE_STATS"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CVX_GPU_LIB"] = os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_lonestats.so")

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
lib = ctypes.CDLL(os.environ["CVX_GPU_LIB"])
out = (ctypes.c_uint64 * 96)()
lib.cvx_debug_lone_stats(out, 1)
acc = [0] * 48
longest = [0] * 48
lives = []
for g in range(poses):
    index = int(os.environ["CVX_POSE_INDEX"]) if "CVX_POSE_INDEX" in os.environ else (g * 37) % 1000
    pos, eul = host.sample_benchmark_path(index / 1000 * host.BENCHMARK_PATH_LENGTH, ws.dims)
    ctx.draw_segments(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]), 0)
    lib.cvx_debug_lone_stats(out, 1)
    for i in range(48):
        acc[i] += out[i]
        longest[i] += out[48 + i]
    lives.append((ctx.last_draw_ms(), out[19], out[18], out[16], out[20]))
for i in range(48):
    out[i] = acc[i]
ms = sum(l[0] for l in lives) / poses
ticks_per_ms = sum(l[1] for l in lives) / sum(l[0] for l in lives)  # (an upper bound of the clock: the longest wave cannot outlive the kernel)
print(f"kernel {ms:.3f} ms per frame; longest wave / kernel: clock ticks per ms >= {ticks_per_ms:.0f}; sum of wave lives / (longest life x waves) = "
      f"{sum(l[2] for l in lives) / sum(l[1] * l[3] for l in lives):.3f}; columns of the longest wave ~{sum(l[4] for l in lives) / poses:.0f}")
names_extra = {29: "reduces that raise nextFreePixelMin", 46: "reduces that lower nextFreePixelMax"}
names = ["windows", "columns", "run projections (per window and run index)", "side trips", "side pixels", "face trips", "face pixels", "side overlaps (:505)", "face overlaps (:581)",
         "processColumn", "... listed", "clipColumn", "... general form", "... window touched", "cullAndFilter", "... with the window bounds of the clip before it", "rays", "processColumn with a clean window", "(sum of lives)", "(longest life)", "(columns of the longest)", "clipped column is itself a hit", "-", "-", "passes over windows with a second run", "... whose remaining columns have only one", "passes over windows with a third run", "... whose remaining columns have at most two", "hits of the pass that wrote no pixel"]
print(f"per frame ({poses} frames {W}x{H} {world}):")
for i, n in enumerate(names):
    print(f"  {n:50s} {out[i] / poses:12.1f}     longest wave: {longest[i] / poses:10.1f}")
for i, n in names_extra.items():
    print(f"  {n:50s} {out[i] / poses:12.1f}     longest wave: {longest[i] / poses:10.1f}")
sections = ["event loop / other", "window: DDA", "window: records + projections", "clip", "clip: window touched", "cull + filter", "column glue", "side: horizon", "side: pixels",
            "face: horizon", "face: pixels", "skybox pass", "window: waiting for the records", "window: waiting for the face colours"]
total = sum(out[32 + i] for i in range(len(sections)))
if total:
    print("share of wave cycles per section (s_memtime, -DCVX_LONE_TIMES):")
    ltotal = sum(longest[32 + i] for i in range(len(sections)))
    for i, n in enumerate(sections):
        print(f"  {n:50s} {out[32 + i] / total * 100:6.2f} %   longest wave: {longest[32 + i] / ltotal * 100:6.2f} %  {longest[32 + i] / poses / 1e3:8.1f} k ticks")
ctx.close()
