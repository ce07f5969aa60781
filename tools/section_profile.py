"""Diagnostic: where the render kernel's wave cycles go (needs `make -C cpuvox_amd/csrc gpu-prof` and a GPU).
Usage: CVX_GPU_LIB=cpuvox_amd/libcpuvox_gpu_prof.so python tools/section_profile.py [--frames 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CVX_GPU_LIB", os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_prof.so"))

from cpuvox_amd import gpu, host  # noqa: E402

NAMES = ["(unused)", "look-ahead + cull", "frustum clip", "element walk", "side setup", "side pixels",
         "top/bottom setup", "top/bottom pixels", "skybox pass", "memory wait at the loop top", "memory wait before the walk",
         "-", "-", "-", "-", "(stamp bookkeeping)"]

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--dim", type=int, default=2048)
args = ap.parse_args()
W, H = 1920, 1080
ws = host.WorldSet.procedural(args.dim, args.dim, args.dim)
lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
frames = []
for g in range(args.frames):
    t = ((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH
    pos, eul = host.sample_benchmark_path(t, ws.dims)
    frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
ctx = gpu.Context(0, buffer_count=args.frames)
ctx.upload_world(ws)
ctx.set_resolution(W, H)
ctx.draw_segments_batch(frames, 0)
ctx.debug_section_cycles(reset=True)
ctx.draw_segments_batch(frames, 0)
cyc = ctx.debug_section_cycles()
total = sum(cyc[:16])
print(f"kernel {ctx.last_draw_ms():.2f} ms for {args.frames} frames (instrumented build; read the shares, not the time)")
for i, (n, c) in enumerate(zip(NAMES, cyc)):
    lanes = 64.0 * cyc[16 + i] / max(1, c)
    print(f"{n:28s} {c:16d} {100.0 * c / total:6.2f} %   mean active lanes {lanes:5.1f}")

