"""Diagnostic: how much a perfect launch-order estimate would be worth (needs `make -C cpuvox_amd/csrc variant NAME=tiletimes DEFS=-DCVX_TILE_TIMES`).
The batch is rendered with the library's own longest-first estimate, the clock ticks every tile really took are recorded, and the same batch is
rendered again in the order of those measured costs.  Usage: python tools/lpt_oracle.py [frames] [first frame]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CVX_GPU_LIB", os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_tiletimes.so"))
from cpuvox_amd import gpu, host  # noqa: E402

frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1920, 1080
ws = host.WorldSet.procedural(2048, 2048, 2048)
lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, 1.0)
frames = []
for g in range(first, first + frames_n):
    t = ((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH
    pos, eul = host.sample_benchmark_path(t, ws.dims)
    frames.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
ctx = gpu.Context(0, buffer_count=frames_n)
ctx.upload_world(ws)
ctx.set_resolution(W, H)
packed = ctx.pack_batch(frames)


def draw(n=3):
    ms = []
    for _ in range(n):
        ctx.draw_packed(packed, 0)
        ms.append(ctx.last_draw_ms())
    return min(ms), ms


base, all_base = draw(4)
print(f"{frames_n} frames from {first}: estimate order {base:.3f} ms  {['%.2f' % m for m in all_base]}")
out = "/tmp/cvx_tile_times.bin"
os.environ["CVX_TILE_TIMES_OUT"] = out
ctx.draw_packed(packed, 0)
del os.environ["CVX_TILE_TIMES_OUT"]
ticks = np.fromfile(out, dtype=np.float32)
print(f"tiles {ticks.size}: ticks mean {ticks.mean():.0f} max {ticks.max():.0f}  sum / 4096 slots = {ticks.sum() / 4096:.0f} ticks; longest tile = {ticks.max():.0f}")
os.environ["CVX_TILE_COST_FILE"] = out
oracle, all_oracle = draw(4)
print(f"measured-cost order {oracle:.3f} ms  {['%.2f' % m for m in all_oracle]}   ({100.0 * (oracle / base - 1.0):+.1f} %)")
# how good is the estimate?  rank correlation between the library's estimate order and the measured costs is not available here (the estimate
# stays inside the library); what can be said: the share of the total cost carried by the longest tiles
srt = np.sort(ticks)[::-1]
for q in (0.01, 0.05, 0.2, 0.5):
    k = max(1, int(q * srt.size))
    print(f"longest {100 * q:4.0f} % of the tiles carry {100.0 * srt[:k].sum() / srt.sum():5.1f} % of the ticks (shortest of them {srt[k - 1]:.0f})")
# a second batch (other frames) rendered in the order measured for the first: does a stale order hurt?
del os.environ["CVX_TILE_COST_FILE"]


# ---- temporal coherence: the costs measured for this batch applied to a batch whose frame i is the NEIGHBOURING pose of frame i
# (frame g + 27 has pose (g * 37 + 999) % 1000 = one sample earlier on the path)
def tiles_per_frame(frs):
    return [[(max(0, s.RayCount) + 63) // 64 for s in f.segments] for f in frs]


def frames_from(start):
    out = []
    for g in range(start, start + frames_n):
        t = ((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH
        pos, eul = host.sample_benchmark_path(t, ws.dims)
        out.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
    return out


for shift, label in ((0, "same batch through the mapping (check)"), (1000, "same poses, 1000 frames later (check)"), (27, "neighbouring pose (1 / 1000 of the path away)"), (135, "five poses away")):
    other = frames_from(first + shift)
    tp_a, tp_b = tiles_per_frame(frames), tiles_per_frame(other)
    costs, cursor = [], 0
    for fa, fb in zip(tp_a, tp_b):
        for na, nb in zip(fa, fb):
            seg = ticks[cursor:cursor + na]
            cursor += na
            if nb == 0:
                continue
            if na == 0:
                costs.append(np.full(nb, ticks.mean(), dtype=np.float32))
            else:
                costs.append(np.interp(np.linspace(0.0, 1.0, nb), np.linspace(0.0, 1.0, na), seg).astype(np.float32))
    costs = np.concatenate(costs)
    packed_b = ctx.pack_batch(other)

    def draw_b(n=3):
        ms = []
        for _ in range(n):
            ctx.draw_packed(packed_b, 0)
            ms.append(ctx.last_draw_ms())
        return min(ms)

    os.environ.pop("CVX_TILE_COST_FILE", None)
    est = draw_b()
    costs.tofile("/tmp/cvx_tile_costs_b.bin")
    os.environ["CVX_TILE_COST_FILE"] = "/tmp/cvx_tile_costs_b.bin"
    hist = draw_b()
    os.environ.pop("CVX_TILE_COST_FILE", None)
    os.environ["CVX_TILE_TIMES_OUT"] = "/tmp/cvx_tile_times_b.bin"
    ctx.draw_packed(packed_b, 0)
    del os.environ["CVX_TILE_TIMES_OUT"]
    own = np.fromfile("/tmp/cvx_tile_times_b.bin", dtype=np.float32)
    corr = float(np.corrcoef(own, costs)[0, 1]) if own.size == costs.size else float("nan")
    rel = float(np.median(np.abs(own - costs) / np.maximum(own, 1.0))) if own.size == costs.size else float("nan")
    print(f"{label}: estimate order {est:.3f} ms, order from the other batch's measured costs {hist:.3f} ms ({100.0 * (hist / est - 1.0):+.1f} %), tiles {costs.size}, "
          f"correlation with its own measured costs {corr:.4f}, median relative difference {rel:.3f}")


# ---- the library's estimate, rescaled per (frame, segment) by what the NEIGHBOURING pose's segment measured (ticks per estimated step):
# keeps the estimate's ranking inside a segment (an upper bound per ray), corrects the systematic part (pitch, height, how much is drawn)
def estimates_of(pk):
    os.environ["CVX_TILE_EST_OUT"] = "/tmp/cvx_tile_est.bin"
    ctx.draw_packed(pk, 0)
    del os.environ["CVX_TILE_EST_OUT"]
    return np.fromfile("/tmp/cvx_tile_est.bin", dtype=np.float32)


est_a = estimates_of(packed)
print(f"estimate vs measured ticks, this batch: correlation {float(np.corrcoef(est_a, ticks)[0, 1]):.4f}")
for shift, label in ((27, "neighbouring pose"), (135, "five poses away")):
    other = frames_from(first + shift)
    packed_b = ctx.pack_batch(other)
    est_b = estimates_of(packed_b)
    tp_a, tp_b = tiles_per_frame(frames), tiles_per_frame(other)
    pred = np.zeros_like(est_b)
    ca = cb = 0
    glob = ticks.sum() / max(1.0, est_a.sum())
    for fa, fb in zip(tp_a, tp_b):
        for na, nb in zip(fa, fb):
            ta, ea = ticks[ca:ca + na], est_a[ca:ca + na]
            ca += na
            ratio = ta.sum() / ea.sum() if na and ea.sum() > 0 else glob
            pred[cb:cb + nb] = est_b[cb:cb + nb] * ratio
            cb += nb

    def draw_b2(n=3):
        ms = []
        for _ in range(n):
            ctx.draw_packed(packed_b, 0)
            ms.append(ctx.last_draw_ms())
        return min(ms)

    os.environ.pop("CVX_TILE_COST_FILE", None)
    e0 = draw_b2()
    pred.astype(np.float32).tofile("/tmp/cvx_tile_costs_c.bin")
    os.environ["CVX_TILE_COST_FILE"] = "/tmp/cvx_tile_costs_c.bin"
    e1 = draw_b2()
    os.environ.pop("CVX_TILE_COST_FILE", None)
    print(f"{label}: estimate order {e0:.3f} ms, estimate x per-segment ticks/step of the other batch {e1:.3f} ms ({100.0 * (e1 / e0 - 1.0):+.1f} %)")


# ---- data for offline analysis (which tiles the estimate ranks too low)
if os.environ.get("CVX_LPT_DUMP"):
    tp = tiles_per_frame(frames)
    fidx, sidx, tidx = [], [], []
    for i, per_seg in enumerate(tp):
        for s_, n_ in enumerate(per_seg):
            fidx += [i] * n_
            sidx += [s_] * n_
            tidx += list(range(n_))
    poses = []
    for g in range(first, first + frames_n):
        t = ((g * 37) % 1000) / 1000 * host.BENCHMARK_PATH_LENGTH
        pos, eul = host.sample_benchmark_path(t, ws.dims)
        poses.append(list(pos) + list(eul))
    np.savez_compressed(os.environ["CVX_LPT_DUMP"], est=est_a, ticks=ticks, frame=np.array(fidx, dtype=np.int32), seg=np.array(sidx, dtype=np.int8),
                        tile=np.array(tidx, dtype=np.int16), poses=np.array(poses, dtype=np.float32), tiles_per_seg=np.array(tp, dtype=np.int32))
    print("dumped", os.environ["CVX_LPT_DUMP"])
