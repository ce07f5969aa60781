#!/usr/bin/env python3
"""Single-frame kernel time of every test scene (tests/scenes.py: BASELINE configs 1 / 2 and the small worlds) through the batch kernel and through the latency
kernel (cvx_set_latency_kernel NEVER / ALWAYS), warm, median of 30 blocking draws each: does AUTO's choice hold outside the benchmark world?"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from cpuvox_amd import gpu  # noqa: E402

ctxs = {}
print(f"{'scene':28s} {'rays':>6s} {'batch ms':>9s} {'latency ms':>11s}  ratio  {'AUTO ms':>8s}")
for name in scenes.SCENES:
    ws, fr, W, H = scenes.scene_frame(name)
    world = scenes.SCENES[name][0]
    if world not in ctxs:
        ctxs[world] = gpu.Context(0)
        ctxs[world].upload_world(ws)
    ctx = ctxs[world]
    ctx.set_resolution(W, H)
    out = []
    for mode in (gpu.LATENCY_NEVER, gpu.LATENCY_ALWAYS, gpu.LATENCY_AUTO):
        ctx.set_latency_kernel(mode)
        for _ in range(5):
            ctx.draw_segments(fr, 0)
        t = []
        for _ in range(30):
            ctx.draw_segments(fr, 0)
            t.append(ctx.last_draw_ms())
        out.append(statistics.median(t))
    print(f"{name:28s} {fr.totalRays:6d} {out[0]:9.4f} {out[1]:11.4f}  {out[1] / out[0]:5.2f}  {out[2]:8.4f}", flush=True)
for c in ctxs.values():
    c.close()
