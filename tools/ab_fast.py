#!/usr/bin/env python3
"""A/B of several builds of libcpuvox_gpu in ONE process on one box: the world is built once, every build gets its own context, the
timed rounds are interleaved (build A step s, build B step s, ...), and before anything is timed every build's raybuffers of a sample of
frames are compared bit for bit with the first build's (and the first build's with the CPU oracle).

    python3 tools/ab_fast.py "libcpuvox_gpu.so libcpuvox_gpu_x.so ..." [--frames 256] [--steps 3] [--rounds 5] [--latency 100]
                             [--width 1920 --height 1080 --world proc2048 --lod-error 1] [--oracle-frames 2] [--check-frames 24]

tools/variants.sh does the same with one bench.py process per build and round (~30 s each for the world); this is the quick instrument
for many small kernel variants.  Prints kernel ms per launch (HIP events around the launch, cvx_draw_time_stats): median / min per build
and the ratio to the first build; with --latency also the single-frame blocking draw (mean over the poses)."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--latency", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--world", default="proc2048")
    ap.add_argument("--lod-error", type=float, default=1.0)
    ap.add_argument("--oracle-frames", type=int, default=2)
    ap.add_argument("--check-frames", type=int, default=24)
    ap.add_argument("--contexts", type=int, default=1, help="contexts (= sets of raybuffer allocations) per build: two contexts of ONE build differ by up to ~1.5 % (where "
                    "their pools landed), so small effects need several per build; the table then gives the median over all of a build's contexts and their spread")
    args = ap.parse_args()

    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process: torch's first)

    from cpuvox_amd import gpu, host

    W, H, F = args.width, args.height, args.frames
    dim = int(args.world[4:])
    t0 = time.time()
    ws = host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
    dims = ws.dims
    lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, args.lod_error)

    def frame_for(g):
        i = (g * 37) % 1000
        pos, eul = host.sample_benchmark_path(i / 1000 * host.BENCHMARK_PATH_LENGTH, dims)
        return host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, dims[1])

    steps_frames = [[frame_for(s * F + i) for i in range(F)] for s in range(args.steps)]
    packed = [gpu.pack_frames(fr) for fr in steps_frames]
    print(f"world + frames: {time.time() - t0:.1f} s", flush=True)

    names = args.libs.split()
    builds = []
    bound = {}
    for c in range(args.contexts):
        for name in names:
            path = name if os.path.isabs(name) else os.path.join(ROOT, "cpuvox_amd", name)
            if name not in bound:
                bound[name] = gpu._bind(path)
            L = bound[name]
            gpu._lib = L
            ctx = gpu.Context(0, buffer_count=F)
            ctx.upload_world(ws)
            ctx.set_resolution(W, H)
            builds.append((name, L, ctx))

    def use(b):
        gpu._lib = b[1]
        return b[2]

    # ---- parity: a sample of the frames of step 0, every build against the first, the first against the oracle
    sample = sorted({(F * i) // args.check_frames for i in range(args.check_frames)})
    reference = None
    ok = True
    for b in builds:
        ctx = use(b)
        ctx.draw_packed(packed[0], 0, gpu.DRAW_SYNC)
        got = []
        for f in sample:
            rc = [max(0, sg.RayCount) for sg in steps_frames[0][f].segments]
            got.append((ctx.read_raybuffer(f, gpu.RAYBUFFER_TOPDOWN, 0, rc[0] + rc[1]), ctx.read_raybuffer(f, gpu.RAYBUFFER_LEFTRIGHT, 0, rc[2] + rc[3])))
        if reference is None:
            reference = got
            if args.oracle_frames > 0:
                import oraclelib as O

                bad = 0
                for k in range(min(args.oracle_frames, len(sample))):
                    idx = (k * len(sample)) // max(1, args.oracle_frames)
                    fr = steps_frames[0][sample[idx]]
                    o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False)
                    g_td, g_lr = got[idx]
                    bad += _differing(np, fr, g_td, g_lr, o_td, o_lr)
                print(f"parity {b[0]} vs the CPU oracle on {args.oracle_frames} frames: {bad} differing pixels", flush=True)
                ok = ok and bad == 0
        else:
            bad = sum(int((a[0] != r[0]).sum() + (a[1] != r[1]).sum()) for a, r in zip(got, reference))
            print(f"parity {b[0]} vs {builds[0][0]} on {len(sample)} frames: {bad} differing words", flush=True)
            ok = ok and bad == 0

    # ---- timing, interleaved
    times = {b[0]: [] for b in builds}
    per_context = {id(b[2]): [] for b in builds}
    for b in builds:  # warm-up
        ctx = use(b)
        for s in range(args.steps):
            ctx.draw_packed(packed[s], 0, gpu.DRAW_ASYNC)
        ctx.synchronize()
        ctx.draw_time_stats(reset=True)
    for r in range(args.rounds):
        for b in (builds if r % 2 == 0 else builds[::-1]):
            ctx = use(b)
            for s in range(args.steps):
                ctx.draw_packed(packed[s], 0, gpu.DRAW_ASYNC)
            ctx.synchronize()
            ms, n = ctx.draw_time_stats(reset=True)
            times[b[0]].append(ms / max(1, n))
            per_context[id(b[2])].append(ms / max(1, n))
    base = statistics.median(times[builds[0][0]])
    print(f"== kernel ms per launch of {F} frames ({args.steps} launches per round, {args.rounds} rounds, interleaved)")
    for name in names:
        v = times[name]
        ctx_medians = [statistics.median(per_context[id(b[2])]) for b in builds if b[0] == name]
        spread = f"  contexts {min(ctx_medians):.3f} .. {max(ctx_medians):.3f}" if len(ctx_medians) > 1 else ""
        print(f"{name:40s} median {statistics.median(v):8.3f}  min {min(v):8.3f}  vs first {statistics.median(v) / base * 100 - 100:+6.2f} %{spread}", flush=True)

    if args.latency > 0:
        lat = {b[0]: [] for b in builds}
        for b in builds[len(names):]:  # (closed through their OWN library: a context collected later would be destroyed by whichever build is current)
            use(b).close()
        builds = builds[:len(names)]
        frames = [frame_for(i * 5) for i in range(args.latency)]
        for r in range(3):
            for b in builds:
                ctx = use(b)
                t = time.perf_counter()
                for fr in frames:
                    ctx.draw_segments(fr, 0)
                lat[b[0]].append((time.perf_counter() - t) / len(frames) * 1e3)
        base = min(lat[builds[0][0]])
        print(f"== single frame, blocking cvx_draw_segments, mean over {args.latency} poses (best of 3)")
        for name in names:
            print(f"{name:40s} {min(lat[name]):7.4f} ms  vs first {min(lat[name]) / base * 100 - 100:+6.2f} %", flush=True)
    for b in builds:
        use(b).close()
    if not ok:
        print("PARITY FAILURE", flush=True)
        raise SystemExit(3)


def _differing(np, fr, g_td, g_lr, o_td, o_lr):
    import scenes

    n_td, n_lr = scenes.used_rows(fr)
    bad = 0
    for g, o, n in ((g_td, o_td, n_td), (g_lr, o_lr, n_lr)):
        m = min(n, g.shape[0])
        written = o[:m] != 0
        bad += int((g[:m][written] != o[:m][written]).sum())
    return bad


if __name__ == "__main__":
    main()
