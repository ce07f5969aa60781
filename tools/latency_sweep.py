#!/usr/bin/env python3
"""Single-frame latency against the sub-tile split (lanes per wave) -- the evidence behind the automatic choice in DrawBatch.
The reference's caller issues ONE blocking DrawSegments per frame (UnityManager.cs:182, RenderManager.cs:363); with one frame in
flight the chip is nearly empty, so a tile is cut into 64 / split narrower waves (a wave's cost per column step is the union of
what its rays need).  CVX_TILE_SPLIT pins the factor in the EXPERIMENT build (libcpuvox_gpu_exp.so); `auto` is the product library.

usage: python tools/latency_sweep.py [poses]      (on the GPU box; prints a markdown table)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    os.environ.setdefault("CVX_NO_TORCH_PRELOAD", "1")
    sys.path.insert(0, ROOT)
    import numpy as np
    from cpuvox_amd import gpu, host

    K = int(sys.argv[2])
    W, H = 1920, 1080
    ws = host.WorldSet.procedural(2048, 2048, 2048, 0x5EED2048)
    lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
    frames = []
    for g in range(K):
        pos, eul = host.sample_benchmark_path(((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH, ws.dims)
        frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
    ctx = gpu.Context(0, buffer_count=2)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    packed = [ctx.pack_batch([f]) for f in frames]
    for k in range(8):
        ctx.draw_packed(packed[k], k % 2, gpu.DRAW_SYNC)
    t = []
    ctx.draw_time_stats(reset=True)
    for k in range(K):
        t0 = time.perf_counter()
        ctx.draw_packed(packed[k], k % 2, gpu.DRAW_SYNC)
        t.append((time.perf_counter() - t0) * 1e3)
    kms, n = ctx.draw_time_stats(reset=True)
    t = np.array(t)
    print(f"RESULT {t.mean():.3f} {np.median(t):.3f} {t.max():.3f} {int(t.argmax() * 37 % 1000)} {kms / max(1, n):.3f}")
    sys.exit(0)

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
exp = os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_exp.so")
rows = []
for split in ["auto", 64, 32, 16, 8, 4, 1]:
    env = dict(os.environ)
    if split != "auto":
        env["CVX_GPU_LIB"] = exp
        env["CVX_TILE_SPLIT"] = str(split)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(K)], env=env, capture_output=True, text=True, timeout=900).stdout
    r = [l for l in out.split("\n") if l.startswith("RESULT")]
    rows.append((split, r[0].split()[1:] if r else None))
print(f"| rays per wave (sub-tile split) | wall ms mean | median | max (pose) | kernel ms mean | fps |")
print("|---|---|---|---|---|---|")
for split, r in rows:
    lanes = "automatic (product library)" if split == "auto" else f"{64 // split} (split {split})"
    if r:
        print(f"| {lanes} | {r[0]} | {r[1]} | {r[2]} ({r[3]}) | {r[4]} | {1000 / float(r[0]):.0f} |")
    else:
        print(f"| {lanes} | failed | | | | |")
