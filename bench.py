#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mrays/s (+ fps) of the Phase-1 raybuffer
renderer (RenderManager.DrawSegments -> DrawSegmentRayJob) at 1920x1080 on a procedural 2048^3
world with the full LOD chain, camera poses from the reference's built-in benchmark fly-through
(Assets/Code/BenchmarkPath.anim, UnityManager.cs:79-97).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch: N*F frames (F = --frames per GPU).  Every
frame's 64-ray tiles are dealt round-robin to the N GPUs (cvx_set_shard), so each GPU renders F
frames' worth of rays per step (weak scaling); with N > 1 the rendered tiles are then exchanged
over RCCL (all_to_all, frame f is assembled on GPU f % N) inside the timed region.
Rank 0 prints ONE JSON line.  World and camera inputs are synthetic and resident in HBM before the
timed region; the per-step host->device traffic is the frame parameters (a few KB).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
POSES = 1000           # fixed number of benchmark-path samples
POSE_STRIDE = 37       # coprime with POSES: consecutive frames spread over the whole path


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step (kept in flight in one launch)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--world", default="proc2048", help="proc<dim> | mill512 | mill256")
    ap.add_argument("--lod-error", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: skip the RCCL tile exchange (replica-style throughput)")
    return ap.parse_args()


def load_world(name: str, rank: int, world_size: int, barrier):
    """Procedural worlds are built once (rank 0) and shared through a cache file in /tmp."""
    from cpuvox_amd import host

    if name.startswith("mill"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenes

        return scenes.load_world(name)
    dim = int(name[4:])
    cache = f"/tmp/cpuvox_{name}_5EED2048.world"
    if world_size == 1:
        return host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
    if rank == 0 and not os.path.exists(cache):
        ws = host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
        ws.save(cache + ".tmp")
        os.replace(cache + ".tmp", cache)
    barrier()
    return host.WorldSet.load(cache)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world_size

    import torch
    import torch.distributed as dist

    from cpuvox_amd import gpu, host

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    N = args.gpus
    if N > 1:
        dist.init_process_group("nccl", device_id=device)

    def barrier():
        if N > 1:
            dist.barrier()

    W, H, F = args.width, args.height, args.frames
    ws = load_world(args.world, rank, N, barrier)
    dims = ws.dims

    # ---- frames: global frame g uses benchmark pose (g * stride) % POSES ------------------------
    pose0 = host.camera_pose((0, 0, 0), (0, 0, 0), W, H)
    lods, far = host.setup_lods(pose0, ws.max_dimension, W, H, args.lod_error)
    total_steps = args.warmup + args.steps
    G = N * F  # frames per step, whole job

    def frame_for(g: int):
        t = ((g * POSE_STRIDE) % POSES) / POSES * host.BENCHMARK_PATH_LENGTH
        pos, eul = host.sample_benchmark_path(t, dims)
        return host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, dims[1])

    steps_frames = [[frame_for(s * G + i) for i in range(G)] for s in range(total_steps)]

    # ---- device context -----------------------------------------------------------------------------
    ctx = gpu.Context(local_rank, buffer_count=G)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    ctx.set_shard(rank, N)
    exchange = None
    if N > 1 and not args.no_exchange:
        from cpuvox_amd import dist as cdist

        lay_td, lay_lr = ctx.raybuffer_layout(0), ctx.raybuffer_layout(1)
        pools = cdist.allocate_pools(G, lay_td, lay_lr, device)
        ctx.bind_raybuffers(pools.td.data_ptr(), pools.td.numel() * 4, pools.lr.data_ptr(), pools.lr.numel() * 4)
        exchange = [cdist.TileExchange(frames, W, H, rank, N, pools, device, ctx) for frames in steps_frames]
        cdist.TileExchange.allocate_staging(exchange, device)
    packed = [ctx.pack_batch(frames) for frames in steps_frames]
    rays_per_step = [sum(f.totalRays for f in frames) for frames in steps_frames]

    # ---- algorithmic bytes per launch: instrumented pass, outside the timed region -------------------
    ctx.enable_counters(True)
    alg_bytes = []
    visits = []
    for s in range(total_steps):
        ctx.draw_packed(packed[s], 0, gpu.DRAW_SYNC)
        c = ctx.counters()
        alg_bytes.append(c.algorithmic_bytes())
        visits.append(c.S)
    ctx.enable_counters(False)

    def run_step(s: int):
        ctx.draw_packed(packed[s], 0, gpu.DRAW_ASYNC)
        if exchange is not None:
            # the render stream belongs to the context, the exchange runs on torch's stream: order them on the host
            ctx.synchronize()
            exchange[s].run()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        run_step(s)
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()

    ctx.draw_time_stats(reset=True)
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        run_step(s)
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if N > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- N > 1: check, outside the timed region, that the exchanged frames are complete: a frame assembled on its
    # display rank from N ranks' tiles must equal the same frame rendered whole by that rank alone (GPU vs GPU; the
    # GPU path itself is pinned to the CPU oracle by tests/).
    exchange_verified = None
    if exchange is not None:
        s_last = total_steps - 1
        mine = [b for b in range(G) if b % N == rank][:2]
        assembled = [(ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1)) for b in mine]
        ctx.set_shard(0, 1)
        ok = True
        for (a_td, a_lr), b in zip(assembled, mine):
            fr = steps_frames[s_last][b]
            ctx.draw_segments(fr, b)
            n_td = max(0, fr.segments[0].RayCount) + max(0, fr.segments[1].RayCount)
            n_lr = max(0, fr.segments[2].RayCount) + max(0, fr.segments[3].RayCount)
            w_td, w_lr = ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1)
            ok = ok and bool((a_td[:n_td] == w_td[:n_td]).all() and (a_lr[:n_lr] == w_lr[:n_lr]).all())
        ctx.set_shard(rank, N)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        exchange_verified = bool(flag.item())

    timed = range(args.warmup, total_steps)
    total_rays = sum(rays_per_step[s] for s in timed)  # whole job: every ray of every frame is rendered by exactly one GPU
    total_frames = G * args.steps
    value = total_rays / elapsed / 1e6
    # roofline of the dominant kernel (render_kernel) on this rank: algorithmic bytes of its launches / their duration
    k_bytes = sum(alg_bytes[s] for s in timed)
    k_ms_total, k_draws = ctx.draw_time_stats(reset=True)  # HIP events around each launch, on the launch stream
    kernel_ms = [k_ms_total / max(1, k_draws)] * max(1, k_draws)
    k_sec = k_ms_total / 1e3
    achieved = k_bytes / k_sec / 1e9

    result = {
        "metric": "Mrays/s, Phase-1 raybuffer rendering (DrawSegmentRayJob) at 1080p, 2048^3 world",
        "value": round(value, 3),
        "unit": "Mrays/s",
        "n_gpus": N,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "fps": round(total_frames / elapsed, 2),
        "config": {
            "workload": f"procedural {args.world} world seed 0x5EED2048, {W}x{H}, full LOD chain (6 levels), benchmark-path poses "
                        f"({POSES} samples, stride {POSE_STRIDE}), lodError {args.lod_error}" if args.world.startswith("proc") else
                        f"{args.world} (mill.obj voxelised), {W}x{H}, benchmark-path poses, lodError {args.lod_error}",
            "frames_per_gpu_per_step": F,
            "frames_per_step": G,
            "rays_per_frame_mean": round(total_rays / total_frames, 1),
            "parallelism": f"ray-tile sharding x{N}" + (" + RCCL all_to_all tile exchange" if exchange is not None else ""),
            "exchange_verified": exchange_verified,
            "world_dims": list(dims),
            "lod_distances": lods,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None,
            "kernel": "cvxk::render_kernel",
            "kernel_ms_avg": round(sum(kernel_ms) / len(kernel_ms), 4),
            "algorithmic_bytes_per_launch": int(k_bytes / len(kernel_ms)),
            "bytes_per_ray": round(k_bytes / max(1, total_rays / N), 1),
            "column_visits_per_s": round(sum(visits[s] for s in timed) / k_sec, 1),
        },
    }

    if rank == 0 and N == 1 and args.cpu_seconds > 0:  # the CPU baseline is reported at N = 1 only
        result["cpu_baseline"] = cpu_baseline(ws, steps_frames[args.warmup], W, H, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)
    barrier()
    ctx.close()
    if N > 1:
        dist.destroy_process_group()


def cpu_baseline(ws, frames, W, H, budget_s: float):
    """The CPU oracle (oracle/cvx_oracle.c: the same algorithm, OpenMP parallel-for over rays, grain 1 like
    RenderJob) timed on this box's host cores on a bounded sample of the same frames.  kind = "port":
    the reference itself (C#/Unity/Burst) cannot run here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O

    import numpy as np

    threads = O.lib().orc_max_threads()
    (td_rays, td_w), (lr_rays, lr_w) = O.raybuffer_shapes(W, H)
    bufs = (np.zeros((td_rays, td_w), dtype=np.uint32), np.zeros((lr_rays, lr_w), dtype=np.uint32))
    O.draw_segments(ws, frames[0], W, H, counters=False, out=bufs)  # warm-up (page in the world)
    t0 = time.perf_counter()
    O.draw_segments(ws, frames[0], W, H, counters=False, out=bufs)
    per_frame = max(1e-4, time.perf_counter() - t0)
    del per_frame
    rays = 0
    n = 0
    t0 = time.perf_counter()
    while True:  # bounded by wall time: frames differ a lot in cost
        f = frames[n % len(frames)]
        O.draw_segments(ws, f, W, H, counters=False, out=bufs)
        rays += f.totalRays
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 4096:
            break
    return {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "fps": round(n / dt, 2),
        "sample": f"{n} frames cycling over the first timed step ({rays} rays), {dt:.1f} s wall, OpenMP {threads} threads, wall clock of orc_draw_segments only",
    }


if __name__ == "__main__":
    main()
