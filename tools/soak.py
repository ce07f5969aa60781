"""Parity soak (not part of the test-suite): many random camera poses, GPU raybuffers and counters against the CPU oracle.
Usage: python tools/soak.py [poses per case]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oraclelib as O  # noqa: E402
import scenes  # noqa: E402
from cpuvox_amd import gpu  # noqa: E402

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 100
CLEAR = 0x9314FFFF
rng = np.random.default_rng(20261003)
bad = total = 0
for world, W, H, lod_error in (("proc1024", 1920, 1080, 1.0), ("proc512", 1280, 720, 6.0), ("mill512", 1024, 768, 1.0), ("proc256x1024x512", 801, 603, 3.0)):
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
        # ... and the rendering build (render_kernel<false>; render_sm_kernel under CVX_RENDER_SM=1), which leaves a finished ray at other points
        ctx.enable_counters(False)
        ctx.clear_raybuffers(1, CLEAR)
        ctx.draw_segments(fr, 1)
        r_td = ctx.read_raybuffer(1, gpu.RAYBUFFER_TOPDOWN)
        r_lr = ctx.read_raybuffer(1, gpu.RAYBUFFER_LEFTRIGHT)
        ok = ok and np.array_equal(r_td[:n_td], o_td[:n_td]) and np.array_equal(r_lr[:n_lr], o_lr[:n_lr])
        total += 1
        if (i + 1) % 1000 == 0:
            print(f"  {world}: {i + 1} poses, {bad} mismatches so far", flush=True)
        if not ok:
            bad += 1
            print("MISMATCH", world, W, H, pos, eul)
    ctx.close()
    print(f"{world} {W}x{H} lodError {lod_error}: {poses} poses done, {bad} mismatches so far", flush=True)
print(f"soak: {total} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
