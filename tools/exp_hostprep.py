import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cpuvox_amd import gpu, host
W, H, F = 1920, 1080, 512
ws = host.WorldSet.procedural(2048, 2048, 2048, 0x5EED2048)
lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
frames = []
for g in range(F):
    pos, eul = host.sample_benchmark_path(((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH, ws.dims)
    frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
ctx = gpu.Context(0, buffer_count=F)
ctx.upload_world(ws)
ctx.set_resolution(W, H)
pk = ctx.pack_batch(frames)
ctx.draw_packed(pk, 0, gpu.DRAW_SYNC)
for i in range(4):
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.draw_packed(pk, 0, gpu.DRAW_ASYNC)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"host part of an async 512-frame draw: {(t1 - t0) * 1e3:.2f} ms; until done: {(t2 - t0) * 1e3:.2f} ms")
