#!/usr/bin/env python3
"""Single-frame latency A/B of several builds of libcpuvox_gpu in ONE process on one box (the reference's call pattern: one blocking
cvx_draw_segments per frame, RenderManager.cs:358-363).  The world is built once; every build renders the same poses, interleaved by rounds;
a sample of the poses is compared bit for bit between every build and the first, and the first with the CPU oracle.

    python3 tools/ab_latency.py "libcpuvox_gpu_base.so libcpuvox_gpu.so ..." [--poses 200] [--rounds 3] [--width 1920 --height 1080 --world proc2048
                                --lod-error 1] [--check 16] [--oracle 2] [--frames-per-launch 1,2,4,8]

Prints per build: wall ms per call (mean / median / p95 / max over the poses, best round) and the kernel's own ms (HIP events).  With
--frames-per-launch also the `latency_curve`: ms per launch of 1 / 2 / 4 ... frames (cvx_draw_segments_batch, blocking)."""
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
    ap.add_argument("--poses", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--world", default="proc2048")
    ap.add_argument("--lod-error", type=float, default=1.0)
    ap.add_argument("--check", type=int, default=16)
    ap.add_argument("--oracle", type=int, default=2)
    ap.add_argument("--frames-per-launch", default="")
    ap.add_argument("--dump", default=None, help="write the per-pose wall ms of every build (best round) and the poses' ray counts to this JSON file")
    ap.add_argument("--pose-range", default=None, help="lo:hi -- only benchmark-path samples lo <= i < hi (0:450 = one clamped top / bottom segment per frame, the shape of BASELINE config 5)")
    args = ap.parse_args()

    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process: torch's first)

    from cpuvox_amd import gpu, host

    W, H = args.width, args.height
    dim = int(args.world[4:])
    t0 = time.time()
    ws = host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
    dims = ws.dims
    lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, args.lod_error)

    def frame_for(g):
        i = (g * 37) % 1000
        if args.pose_range:
            lo, hi = (int(v) for v in args.pose_range.split(":"))
            i = lo + i % (hi - lo)
        pos, eul = host.sample_benchmark_path(i / 1000 * host.BENCHMARK_PATH_LENGTH, dims)
        return host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, dims[1])

    frames = [frame_for(i) for i in range(args.poses)]
    print(f"world + frames: {time.time() - t0:.1f} s", flush=True)
    curve = [int(v) for v in args.frames_per_launch.split(",") if v]
    names = args.libs.split()
    builds = []
    for name in names:
        path = name if os.path.isabs(name) else os.path.join(ROOT, "cpuvox_amd", name)
        L = gpu._bind(path)
        gpu._lib = L
        ctx = gpu.Context(0, buffer_count=max([2] + curve))
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        builds.append((name, L, ctx))

    def use(b):
        gpu._lib = b[1]
        return b[2]

    # ---- parity of the single-frame path
    sample = sorted({(args.poses * i) // args.check for i in range(args.check)})
    reference = None
    ok = True
    for b in builds:
        ctx = use(b)
        got = []
        for f in sample:
            fr = frames[f]
            ctx.clear_raybuffers(0, 0)
            ctx.draw_segments(fr, 0)
            rc = [max(0, sg.RayCount) for sg in fr.segments]
            got.append((ctx.read_raybuffer(0, gpu.RAYBUFFER_TOPDOWN, 0, rc[0] + rc[1]), ctx.read_raybuffer(0, gpu.RAYBUFFER_LEFTRIGHT, 0, rc[2] + rc[3])))
        if reference is None:
            reference = got
            if args.oracle > 0:
                import oraclelib as O
                import scenes

                bad = 0
                for k in range(min(args.oracle, len(sample))):
                    idx = (k * len(sample)) // max(1, args.oracle)
                    fr = frames[sample[idx]]
                    o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False)
                    n_td, n_lr = scenes.used_rows(fr)
                    for g, o, n in ((got[idx][0], o_td, n_td), (got[idx][1], o_lr, n_lr)):
                        m = min(n, g.shape[0])
                        bad += int((g[:m] != o[:m]).sum())
                print(f"parity {b[0]} vs the CPU oracle on {args.oracle} poses: {bad} differing pixels", flush=True)
                ok = ok and bad == 0
        else:
            bad = sum(int((a[0] != r[0]).sum() + (a[1] != r[1]).sum()) for a, r in zip(got, reference))
            print(f"parity {b[0]} vs {builds[0][0]} on {len(sample)} poses: {bad} differing words", flush=True)
            ok = ok and bad == 0

    # ---- timing
    wall = {b[0]: [] for b in builds}
    kern = {b[0]: [] for b in builds}
    for b in builds:  # warm-up
        ctx = use(b)
        for fr in frames[:20]:
            ctx.draw_segments(fr, 0)
    for r in range(args.rounds):
        for b in (builds if r % 2 == 0 else builds[::-1]):
            ctx = use(b)
            w, k = [], []
            for fr in frames:
                t = time.perf_counter()
                ctx.draw_segments(fr, 0)
                w.append((time.perf_counter() - t) * 1e3)
                k.append(ctx.last_draw_ms())
            wall[b[0]].append(w)
            kern[b[0]].append(k)

    def stats(v):
        s = sorted(v)
        return statistics.mean(v), statistics.median(v), s[int(0.95 * (len(s) - 1))], s[-1]

    print(f"== single frame {W}x{H}, blocking cvx_draw_segments, {args.poses} poses, best of {args.rounds} rounds: wall ms mean / median / p95 / max | kernel ms mean / max")
    base = None
    for name in names:
        best = min(range(args.rounds), key=lambda r: statistics.mean(wall[name][r]))
        m, md, p95, mx = stats(wall[name][best])
        km, _, _, kmx = stats(kern[name][best])
        base = base or m
        worst = max(range(args.poses), key=lambda i: wall[name][best][i])
        print(f"{name:36s} {m:7.4f} {md:7.4f} {p95:7.4f} {mx:7.4f} (pose {worst}) | {km:7.4f} {kmx:7.4f}   vs first {m / base * 100 - 100:+6.2f} %", flush=True)

    if args.dump:
        import json

        per_pose = {}
        for name in names:
            best = min(range(args.rounds), key=lambda r: statistics.mean(wall[name][r]))
            per_pose[name] = [round(v, 5) for v in wall[name][best]]
        rays = [sum(max(0, sg.RayCount) for sg in fr.segments) for fr in frames]
        with open(args.dump, "w") as f:
            json.dump({"rays": rays, "wall_ms": per_pose}, f)
    if curve:
        print("== latency curve: wall ms per blocking launch of N frames (mean over the launches of one pass over the poses, best of the rounds)")
        for b in builds:
            ctx = use(b)
            row = []
            for n in curve:
                packs = [gpu.pack_frames(frames[i:i + n]) for i in range(0, args.poses - n + 1, n)]
                best = None
                for r in range(args.rounds):
                    t = time.perf_counter()
                    for p in packs:
                        ctx.draw_packed(p, 0, gpu.DRAW_SYNC)
                    dt = (time.perf_counter() - t) / len(packs) * 1e3
                    best = dt if best is None else min(best, dt)
                row.append(f"{n}: {best:.3f}")
            print(f"{b[0]:36s} " + "  ".join(row), flush=True)
    for b in builds:
        use(b).close()
    if not ok:
        print("PARITY FAILURE", flush=True)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
