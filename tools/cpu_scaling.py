"""Thread scaling of the CPU oracle on this host (which thread count should bench.py's cpu_baseline use?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cpuvox_amd import host  # noqa: E402
import oraclelib as O  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W, H = 1920, 1080
ws = host.WorldSet.procedural(dim, dim, dim)
lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
frames = []
for g in range(24):
    t = ((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH
    pos, eul = host.sample_benchmark_path(t, ws.dims)
    frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
(td_rays, td_w), (lr_rays, lr_w) = O.raybuffer_shapes(W, H)
bufs = (np.zeros((td_rays, td_w), dtype=np.uint32), np.zeros((lr_rays, lr_w), dtype=np.uint32))
print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)), "omp max", O.lib().orc_max_threads())
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError:
    pass
for thr in (1, 8, 16, 32, 64, 128, 256):
    if thr > os.cpu_count():
        break
    O.draw_segments(ws, frames[0], W, H, threads=thr, counters=False, out=bufs)
    t0 = time.perf_counter()
    rays = 0
    for f in frames if thr > 1 else frames[:6]:
        O.draw_segments(ws, f, W, H, threads=thr, counters=False, out=bufs)
        rays += f.totalRays
    dt = time.perf_counter() - t0
    print(f"{thr:4d} threads: {rays / dt / 1e6:.4f} Mrays/s")
