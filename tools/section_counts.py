"""Diagnostic: how often a WAVE executes each part of the render kernel (needs `make -C cpuvox_amd/csrc gpu-count`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CVX_GPU_LIB", os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_count.so"))
from cpuvox_amd import gpu, host  # noqa: E402

NAMES = {1: "column visits (wave-steps)", 8: "drawColumn entered", 2: "frustum clip executed", 3: "element-walk iterations",
         4: "run projected (side block entered)", 9: "side visible (6 divisions)", 10: "side overlaps window", 5: "side pixel iterations",
         6: "top/bottom attempted", 11: "top/bottom visible (2 divisions)", 12: "top/bottom overlaps window", 7: "top/bottom pixel iterations"}
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
ctx.debug_section_cycles(reset=True)
ctx.debug_section_histogram(reset=True)
ctx.enable_counters(True)
ctx.draw_segments_batch(frames, 0)
c = ctx.counters()
cyc = ctx.debug_section_cycles()
print(f"{frames_n} frames: lane-level S={c.S} E={c.E} C={c.C} P={c.P} R={c.R}")
NAMES.update({13: "drawColumn: ALL lanes could carry camSpace*Last", 14: "clip: ALL lanes could reuse the Last half", 15: "clip: ALL lanes same frustumBoundsMax"})
for k in (1, 8, 13, 2, 14, 15, 3, 4, 9, 10, 5, 6, 11, 12, 7):
    print(f"{NAMES[k]:40s} {cyc[k]:12d}  per wave-step {cyc[k] / max(1, cyc[1]):6.3f}   active lanes per execution {cyc[16 + k] / max(1, cyc[k]):5.1f}")
hist = ctx.debug_section_histogram()
print("active lanes per execution, share of executions in buckets 1-8, 9-16, ..., 57-64:")
for k in (1, 8, 2, 3, 4, 10, 5, 6, 12, 7):
    tot = max(1, sum(hist[k]))
    print(f"{NAMES[k]:40s} " + " ".join(f"{100.0 * v / tot:5.1f}" for v in hist[k]))
