"""Diagnostic: how the state-machine render kernel schedules its blocks (needs `make -C cpuvox_amd/csrc variant NAME=smstats DEFS=-DCVX_SM_STATS`).
Usage: python tools/sm_stats.py [frames]     (CVX_SM_THRESHOLD etc. apply)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CVX_GPU_LIB", os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_smstats.so"))
from cpuvox_amd import gpu, host  # noqa: E402

frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
ws = host.WorldSet.procedural(2048, 2048, 2048)
lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
frames = []
for g in range(frames_n):
    t = ((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH
    pos, eul = host.sample_benchmark_path(t, ws.dims)
    frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
ctx = gpu.Context(0, buffer_count=frames_n)
ctx.upload_world(ws)
ctx.set_resolution(W, H)
ctx.draw_segments_batch(frames, 0)
ctx.debug_section_cycles(reset=True)
ctx.draw_segments_batch(frames, 0)
ms = ctx.last_draw_ms()
c = ctx.debug_section_cycles()
names = ["ADV", "CLIP", "WALK", "SIDE", "SPIX", "TB"]
waves = max(1, c[14])
print(f"{frames_n} frames, {waves} waves, kernel {ms:.2f} ms (stats build), threshold env {os.environ.get('CVX_SM_THRESHOLD', '-')}")
print(f"passes per wave {c[6] / waves:9.1f}   of which nothing reached its threshold {100.0 * c[7] / max(1, c[6]):5.1f} %")
tot = 0
for k, n in enumerate(names):
    tot += c[k]
    print(f"{n:5s} executions per wave {c[k] / waves:9.1f}   lanes per execution {c[8 + k] / max(1, c[k]):5.1f}   lane-executions per wave {c[8 + k] / waves:10.1f}")
print(f"block executions per pass {tot / max(1, c[6]):.2f}")
