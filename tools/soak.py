"""Parity soak (not part of the test-suite): many random camera poses, GPU raybuffers and counters against the CPU oracle.
Usage: python tools/soak.py [poses per case]   random poses over four worlds: the counting build, then the shipped build pinned to the batch kernel AND
                                                  to the latency kernel (cvx_set_latency_kernel), each against the oracle
       python tools/soak.py bench                 the 1000 benchmark poses as batches (batch kernel) and as single blocking draws (latency kernel)
       python tools/soak.py 4k [poses per case]   the same random poses at 3840x2160 and 4096x2304: windows of more than 2048 pixels, the latency kernel's two-register instance"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oraclelib as O  # noqa: E402
import scenes  # noqa: E402
from cpuvox_amd import gpu  # noqa: E402

big = len(sys.argv) > 1 and sys.argv[1] == "4k"
poses = int(sys.argv[2]) if big and len(sys.argv) > 2 else (300 if big else (int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "bench" else 100))
if len(sys.argv) > 1 and sys.argv[1] == "bench":
    # every pose of the benchmark path (the 1000 samples bench.py cycles through) on the benchmark world at the benchmark resolution, as ONE
    # batch per 100 poses through the rendering build (what bench.py times) -- raybuffers against the oracle
    from cpuvox_amd import host

    W, H = 1920, 1080
    ws = scenes.load_world("proc2048")
    lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
    ctx = gpu.Context(0, buffer_count=100)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    bad = 0
    for first in range(0, 1000, 100):
        frames = []
        for i in range(first, first + 100):
            pos, eul = host.sample_benchmark_path(i / 1000 * host.BENCHMARK_PATH_LENGTH, ws.dims)
            frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
        for b in range(100):
            ctx.clear_raybuffers(b, 0x9314FFFF)
        ctx.draw_segments_batch(frames, 0)
        for b, fr in enumerate(frames):
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0x9314FFFF, counters=False)
            n_td, n_lr = scenes.used_rows(fr)
            g_td = ctx.read_raybuffer(b, gpu.RAYBUFFER_TOPDOWN, 0, n_td)
            g_lr = ctx.read_raybuffer(b, gpu.RAYBUFFER_LEFTRIGHT, 0, n_lr)
            if not (np.array_equal(g_td, o_td[:n_td]) and np.array_equal(g_lr, o_lr[:n_lr])):
                bad += 1
                print("MISMATCH benchmark pose", first + b)
            # ... and the reference's call pattern: one blocking draw of this frame alone (the latency kernel, cvx_lone.h)
            ctx.set_latency_kernel(gpu.LATENCY_ALWAYS)
            ctx.clear_raybuffers(b, 0x9314FFFF)
            ctx.draw_segments(fr, b)
            ctx.set_latency_kernel(gpu.LATENCY_AUTO)
            l_td = ctx.read_raybuffer(b, gpu.RAYBUFFER_TOPDOWN, 0, n_td)
            l_lr = ctx.read_raybuffer(b, gpu.RAYBUFFER_LEFTRIGHT, 0, n_lr)
            if not (np.array_equal(l_td, o_td[:n_td]) and np.array_equal(l_lr, o_lr[:n_lr])):
                bad += 1
                print("MISMATCH benchmark pose (latency kernel)", first + b)
        print(f"benchmark poses {first}..{first + 99}: {bad} mismatches so far", flush=True)
    print(f"soak (benchmark path, proc2048 @ {W}x{H}): 1000 poses as batches (batch kernel) + 1000 single draws (latency kernel), {bad} mismatches")
    sys.exit(1 if bad else 0)
CLEAR = 0x9314FFFF
rng = np.random.default_rng(20261003)
bad = total = 0
CASES = (("proc1024", 3840, 2160, 1.0), ("mill512", 4096, 2304, 2.0)) if big else (("proc1024", 1920, 1080, 1.0), ("proc512", 1280, 720, 6.0), ("mill512", 1024, 768, 1.0), ("proc256x1024x512", 801, 603, 3.0))
for world, W, H, lod_error in CASES:
    ws = scenes.load_world(world)
    ctx = gpu.Context(0, buffer_count=4)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    for i in range(poses):
        frac = rng.uniform(-0.3, 1.3, size=3)
        pos = [frac[k] * ws.dims[k] for k in range(3)]
        eul = [rng.uniform(-89.5, 89.5), rng.uniform(0, 360), rng.choice([0.0, rng.uniform(0, 360)])]
        fr = scenes.make_frame(ws, W, H, pos, eul, lod_error=lod_error)
        ctx.enable_counters(True)
        ctx.clear_raybuffers(0, CLEAR)
        ctx.draw_segments(fr, 0)
        c = ctx.counters()
        g_td = ctx.read_raybuffer(0, gpu.RAYBUFFER_TOPDOWN)
        g_lr = ctx.read_raybuffer(0, gpu.RAYBUFFER_LEFTRIGHT)
        o_td, o_lr, oc = O.draw_segments(ws, fr, W, H, clear=CLEAR)
        n_td, n_lr = scenes.used_rows(fr)
        ok = np.array_equal(g_td[:n_td], o_td[:n_td]) and np.array_equal(g_lr[:n_lr], o_lr[:n_lr]) and \
            (c.S, c.E, c.C, c.P, c.R) == (oc.S, oc.E, oc.C, oc.P, oc.R)
        # ... and the shipped build: the batch kernel (render_kernel<false>, which leaves a finished ray at other points) and the latency kernel (lone_kernel)
        ctx.enable_counters(False)
        for mode in (gpu.LATENCY_NEVER, gpu.LATENCY_ALWAYS):
            ctx.set_latency_kernel(mode)
            ctx.clear_raybuffers(1, CLEAR)
            ctx.draw_segments(fr, 1)
            r_td = ctx.read_raybuffer(1, gpu.RAYBUFFER_TOPDOWN)
            r_lr = ctx.read_raybuffer(1, gpu.RAYBUFFER_LEFTRIGHT)
            ok = ok and np.array_equal(r_td[:n_td], o_td[:n_td]) and np.array_equal(r_lr[:n_lr], o_lr[:n_lr])
        ctx.set_latency_kernel(gpu.LATENCY_AUTO)
        total += 1
        if (i + 1) % (100 if big else 1000) == 0:
            print(f"  {world}: {i + 1} poses, {bad} mismatches so far", flush=True)
        if not ok:
            bad += 1
            print("MISMATCH", world, W, H, pos, eul)
    ctx.close()
    print(f"{world} {W}x{H} lodError {lod_error}: {poses} poses done, {bad} mismatches so far", flush=True)
print(f"soak: {total} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
