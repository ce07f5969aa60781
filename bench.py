#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mrays/s (+ fps) of the Phase-1 raybuffer
renderer (RenderManager.DrawSegments -> DrawSegmentRayJob) at 1920x1080 on a procedural 2048^3
world with the full LOD chain, camera poses from the reference's built-in benchmark fly-through
(Assets/Code/BenchmarkPath.anim, UnityManager.cs:79-97).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch: N*F frames (F = --frames per GPU).
N = 1: one launch renders the F frames (cvx_draw_segments_batch).
N > 1: every frame's 64-ray tiles are dealt round-robin to the N GPUs, so each GPU renders F frames' worth of rays
per step (weak scaling); frame f is displayed on GPU f % N.  The kernel writes every tile straight into the buffer it
has to end up in (cvx_draw_segments_placed + cpuvox_amd.dist.ShardPlan): tiles of frames displayed elsewhere go into a
per-destination send section, and the exchange is ONE send and ONE receive per peer (grouped ncclSend/ncclRecv on
RCCL, each pair on its own xGMI link), overlapped with the next step's render on a second stream.  The exchange is
inside the timed region; afterwards frames assembled from N ranks' tiles are compared with the same frames rendered
whole by their display rank.
Rank 0 prints ONE JSON line.  World and camera inputs are synthetic and resident in HBM before the timed region; the
per-step host->device traffic is the frame / tile descriptors (a few hundred KB).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# the host driver of the GPU boxes only supports dmabuf IPC: RCCL peer buffers need this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    ap.add_argument("--frames", type=lambda v: v if v == "auto" else int(v), default=512,
                    help="frames per GPU per step (kept in flight in one launch); 'auto': the largest of 512, 256, ... whose raybuffer areas (N > 1: send + display, two parities) fit --hbm-budget-gb")
    ap.add_argument("--hbm-budget-gb", type=float, default=96.0, help="--frames auto: HBM this run may spend on raybuffer / exchange areas per GPU (a third of the 288 GB)")
    ap.add_argument("--allow-fallback", action="store_true", help="N > 1: if frames assembled through cvx_exchange fail verification, re-time with the torch.distributed exchange (labelled) "
                    "instead of printing the failure line and exiting with code 5")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--world", default="proc2048", help="proc<dim> | mill512 | mill256")
    ap.add_argument("--lod-error", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall time of the cpu_baseline sample at N = 1 (0 = skip)")
    ap.add_argument("--pose-range", default=None, help="diagnostics: lo:hi -- only benchmark-path samples lo <= i < hi (e.g. 0:450 = one top/bottom segment per frame)")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: skip the RCCL tile exchange (replica-style throughput)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: finish each step's exchange before the next render")
    ap.add_argument("--torch-exchange", action="store_true", help="N > 1: exchange with torch.distributed P2P ops instead of cvx_exchange (the C-ABI path)")
    ap.add_argument("--latency-frames", type=int, default=200, help="N = 1: frames of the single-frame latency legs (0 = skip)")
    ap.add_argument("--pmc-csv", default=None, help="counter summary of a rocprofv3 --pmc run of THIS command (tools/pmc_passes.sh + "
                    "tools/pmc_aggregate.py): fills roofline.traffic from FETCH_SIZE + WRITE_SIZE; without it traffic is null")
    ap.add_argument("--comm-timeout", type=float, default=180.0, help="N > 1: seconds cvx_comm_create may wait for the peers (then exit code 4)")
    ap.add_argument("--gather", choices=("raybuffer", "image", "auto"), default="raybuffer",
                    help="N > 1: what travels to the display rank of a frame -- the other ranks' raybuffer tile rows (BASELINE.json's north_star; default) or "
                         "their pixels of the finished image (cvx_image_*: each rank runs Phase 2 for its own tiles; W*H*4 bytes per frame in total); "
                         "auto = whichever of the two puts fewer bytes on the wire for the frames of step 0 (rank 0 decides, every rank follows)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' = single-GPU rehearsal of the N > 1 path "
                    "(all ranks share the visible GPUs, tiles travel through host memory)")
    return ap.parse_args()


def load_world(name: str, rank: int, world_size: int, barrier):
    """Procedural worlds are built once (rank 0) and shared with the other ranks through a cache file in a private
    directory; the file name carries a hash of the generator's source, and dims / LOD count are checked after loading."""
    import hashlib

    from cpuvox_amd import host

    if name.startswith("mill"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenes

        return scenes.load_world(name)
    dim = int(name[4:])
    if world_size == 1:
        return host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
    with open(os.path.join(ROOT, "cpuvox_amd", "csrc", "host", "cvx_world.cpp"), "rb") as fh:
        tag = hashlib.sha256(fh.read()).hexdigest()[:12]
    cache_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cpuvox_bench_{os.getuid()}")
    os.makedirs(cache_dir, mode=0o700, exist_ok=True)
    if os.stat(cache_dir).st_uid != os.getuid():
        raise SystemExit(f"{cache_dir} is not mine")
    cache = os.path.join(cache_dir, f"{name}_5EED2048_{tag}.world")
    if rank == 0 and not os.path.exists(cache):
        ws = host.WorldSet.procedural(dim, dim, dim, 0x5EED2048)
        ws.save(cache + f".tmp{os.getpid()}")
        os.replace(cache + f".tmp{os.getpid()}", cache)
    barrier()
    ws = host.WorldSet.load(cache)
    if tuple(ws.dims) != (dim, dim, dim) or ws.lod_count != host.LOD_LEVELS:
        raise SystemExit(f"{cache}: unexpected world {tuple(ws.dims)} with {ws.lod_count} LODs")
    return ws


def self_launch(n: int) -> int:
    """Starts `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process, relays its
    output (rank 0 prints the JSON line) and returns its exit code."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE, launching: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as ONE process (how the driver starts it): become the launcher.  Nothing in this process has
        # touched torch or HIP yet, and it never will -- the ranks are fresh children of torch.distributed.run, never an exec of a
        # process that has initialised the GPU.
        raise SystemExit(self_launch(args.gpus))
    if world_size != args.gpus:
        args.gpus = world_size

    import numpy as np
    import torch
    import torch.distributed as dist

    from cpuvox_amd import gpu, host

    if world_size > 1:
        print(f"bench.py: rank {rank} of {world_size} up (local rank {local_rank}, backend {args.backend})", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py (rank {rank} of {world_size}) needs a GPU: the product path has no CPU fallback")
    if args.backend == "nccl" and world_size > torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world_size}: only {torch.cuda.device_count()} GPU(s) visible; RCCL needs one device per rank "
                         "(--backend gloo rehearses the N > 1 path on fewer devices)")
    local_rank %= torch.cuda.device_count()  # only differs from LOCAL_RANK in a --backend gloo rehearsal
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    N = args.gpus
    if N > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if N > 1:
            dist.barrier()

    W, H = args.width, args.height
    ws = load_world(args.world, rank, N, barrier)
    dims = ws.dims

    # ---- frames: global frame g uses benchmark pose (g * stride) % POSES ------------------------
    pose0 = host.camera_pose((0, 0, 0), (0, 0, 0), W, H)
    lods, far = host.setup_lods(pose0, ws.max_dimension, W, H, args.lod_error)
    total_steps = args.warmup + args.steps

    def frame_for(g: int):
        i = (g * POSE_STRIDE) % POSES
        if args.pose_range:
            lo, hi = (int(v) for v in args.pose_range.split(":"))
            i = lo + i % (hi - lo)
        t = i / POSES * host.BENCHMARK_PATH_LENGTH
        pos, eul = host.sample_benchmark_path(t, dims)
        return host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, dims[1])

    # ---- frames per GPU and step.  `auto`: the largest F whose areas fit the stated HBM budget -- one GPU: F raybuffer pairs;
    # N > 1: per-destination send sections + the display area, two parities each (the shard plan of step 0 gives their exact sizes) -- and the
    # payload every xGMI link would carry per step, printed before anything is allocated
    predicted = None
    if args.frames == "auto":
        predicted = predict_frames_auto(frame_for, W, H, rank, N, args.hbm_budget_gb * 1e9)
        args.frames = predicted["frames_per_gpu"]
        if rank == 0:
            if not predicted["fits_budget"]:  #  not even the smallest batch fits: go on with it and say so
                print(f"bench.py: WARNING --frames auto: no candidate fits --hbm-budget-gb {args.hbm_budget_gb:g}; running {args.frames} frames per GPU OVER the budget", file=sys.stderr, flush=True)
            print(f"bench.py: --frames auto -> {args.frames} per GPU ({predicted['area_bytes_per_gpu'] / 1e9:.2f} GB of areas per GPU against a budget of {args.hbm_budget_gb:.1f} GB; "
                  f"{predicted['payload_bytes_per_link_per_step'] / 1e6:.1f} MB per peer and step = {predicted['link_ms_at_153_GBps']:.2f} ms on one 153 GB/s xGMI link)", file=sys.stderr, flush=True)
    F = args.frames
    G = N * F  # frames per step, whole job

    steps_frames = [[frame_for(s * G + i) for i in range(G)] for s in range(total_steps)]
    rays_per_step = [sum(f.totalRays for f in frames) for frames in steps_frames]

    # ---- device context -----------------------------------------------------------------------------
    sharded = N > 1
    ctx = gpu.Context(local_rank, buffer_count=1 if sharded else G)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    packed = [ctx.pack_batch(frames) for frames in steps_frames]

    gather_choice = None
    if sharded and args.gather == "auto":
        # what each gather would put on the wire for the frames of step 0 (host arithmetic + one tile census on the device): the smaller one is taken
        from cpuvox_amd import dist as cdist0

        rb_bytes = cdist0.ShardPlan(steps_frames[0], W, H, rank, N).send_total * 256
        ip0 = gpu.ImagePlan(ctx, packed[0], W, H, rank, N)
        choice = [None]
        if rank == 0:
            choice[0] = choose_gather(rb_bytes, ip0.send_pixels * 4)
        dist.broadcast_object_list(choice, src=0)
        gather_choice = dict(choice[0], raybuffer_bytes_this_rank=int(rb_bytes), image_bytes_this_rank=int(ip0.send_pixels * 4))
        args.gather = choice[0]["gather"]
        del ip0

    plans = tile_outs = None
    send = disp = None
    s_render = s_exchange = None
    native_plans = None
    comm = None
    exchange_path = None
    ranks_seen = [0]
    if sharded:
        from cpuvox_amd import dist as cdist

        # census: which ranks / devices take part (the first multi-GPU run checks itself)
        me = {"rank": rank, "device": local_rank, "gpu": torch.cuda.get_device_name(local_rank), "host": os.uname().nodename}
        census = [None] * N
        dist.all_gather_object(census, me)
        ranks_seen = sorted(c["rank"] for c in census)
        plans = [cdist.ShardPlan(frames, W, H, rank, N) for frames in steps_frames]
        # The exchange behind the C ABI (cvx_shard_plan_* + cvx_exchange on a communicator the library owns); the plans above
        # stay for verification (assemble) and as the torch.distributed fallback.
        if args.backend == "nccl" and not args.torch_exchange:
            # Every rank runs the SAME sequence of collectives whatever fails locally: (1) build + verify the native plans, (2) rank 0
            # makes the id, (3) broadcast it, (4) all-reduce "everything fine so far", and only if that holds on every rank
            # (5) cvx_comm_create -- so either all ranks enter ncclCommInitRank or none does -- then (6) all-reduce its outcome.
            why = None
            try:
                native_plans = [gpu.NativeShardPlan(pk, W, H, rank, N) for pk in packed]
                for a, b in zip(native_plans, plans):
                    if not (a.tile_count == b.tile_count and list(a.send_start) == list(b.send_start) and list(a.disp_start) == list(b.disp_start)):
                        raise RuntimeError("native shard plan differs from the Python plan")
            except Exception as e:  # noqa: BLE001
                why = f"shard plan: {e}"
            uid = [None]
            if rank == 0:
                try:
                    uid[0] = gpu.comm_unique_id()
                except Exception as e:  # noqa: BLE001
                    uid[0] = f"error: {e}"
            dist.broadcast_object_list(uid, src=0)
            if why is None and not isinstance(uid[0], bytes):
                why = f"rank 0 could not make a RCCL id ({uid[0]})"
            flag = torch.tensor([1 if why is None else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if bool(flag.item()):
                try:
                    comm = gpu.comm_create(ctx, uid[0], rank, N, timeout_s=args.comm_timeout)
                except Exception as e:  # noqa: BLE001  (a timeout means a peer is gone: the all-reduce below would hang too)
                    if "did not return within" in str(e):
                        print(f"bench.py rank {rank}: {e}", file=sys.stderr, flush=True)
                        os._exit(4)
                    why = f"cvx_comm_create: {e}"
                flag = torch.tensor([1 if comm else 0], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            elif why is None:
                why = "a peer could not prepare the C-ABI exchange"
            if bool(flag.item()):
                exchange_path = "cvx_exchange (grouped ncclSend/ncclRecv inside libcpuvox_gpu, library-owned communicator)"
            else:
                if comm:
                    gpu.comm_destroy(comm)
                native_plans, comm = None, None
                exchange_path = f"torch.distributed batch_isend_irecv (C-ABI exchange unavailable: {why or 'a peer could not create the communicator'})"
        else:
            exchange_path = "torch.distributed batch_isend_irecv"
        raybuffer_areas = args.gather != "image"  # (the image gather has its own, smaller buffers: below)
        send_rows = max(1, max(p.send_total for p in plans)) if raybuffer_areas else 1
        disp_rows = max(1, max(p.disp_total for p in plans)) if raybuffer_areas else 1
        # two parities: step s+1 renders into the other pair while step s is still on the wire
        send = [torch.zeros((send_rows, cdist.TILE_RAYS), dtype=torch.int32, device=device) for _ in range(2)]
        disp = [torch.zeros((disp_rows, cdist.TILE_RAYS), dtype=torch.int32, device=device) for _ in range(2)]
        tile_outs = [plans[s].tile_out(send[s % 2].data_ptr(), disp[s % 2].data_ptr()) for s in range(total_steps)] if raybuffer_areas else None
        s_render, s_exchange = torch.cuda.Stream(device), torch.cuda.Stream(device)
        ctx.set_stream(s_render.cuda_stream)

    image_mode = sharded and args.gather == "image"
    img_plans = img_store = img_send = img_recv = img_images = None
    if image_mode:
        # image gather: compact local tile store + pixel streams + the images this rank displays, two parities like the raybuffer areas
        img_plans = [gpu.ImagePlan(ctx, pk, W, H, rank, N) for pk in packed]
        words = lambda n: max(1, int(n))  # noqa: E731
        img_store = [torch.zeros(words(max(p.local_store_bytes for p in img_plans) // 4), dtype=torch.int32, device=device) for _ in range(2)]
        img_send = [torch.zeros(words(max(p.send_pixels for p in img_plans)), dtype=torch.int32, device=device) for _ in range(2)]
        img_recv = [torch.zeros(words(max(p.recv_pixels for p in img_plans)), dtype=torch.int32, device=device) for _ in range(2)]
        img_images = [torch.zeros((words(max(p.images for p in img_plans)), H, W), dtype=torch.int32, device=device) for _ in range(2)]
        tile_outs = [img_plans[s].tile_out(img_store[s % 2].data_ptr()) for s in range(total_steps)]
        if not comm:
            exchange_path = "torch.distributed batch_isend_irecv of the pixel streams" + (f" ({exchange_path})" if exchange_path and "unavailable" in exchange_path else "")
        else:
            exchange_path = "cvx_image_exchange (grouped ncclSend/ncclRecv of pixel streams inside libcpuvox_gpu, library-owned communicator)"

    def image_exchange_torch(plan, send, recv):
        """The pixel streams with torch.distributed P2P ops (fallback / gloo rehearsal: staged through host memory)."""
        via_host = dist.get_backend() == "gloo"
        if via_host:
            torch.cuda.current_stream().synchronize()
        ops, received = [], []
        for peer in range(N):
            if peer == rank:
                continue
            s0, sn, r0, rn = plan.transfer(peer)
            if sn:
                ops.append(dist.P2POp(dist.isend, send[s0:s0 + sn].cpu() if via_host else send[s0:s0 + sn], peer))
            if rn:
                buf = torch.empty(rn, dtype=recv.dtype) if via_host else recv[r0:r0 + rn]
                received.append((r0, rn, buf))
                ops.append(dist.P2POp(dist.irecv, buf, peer))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        if via_host:
            for r0, rn, buf in received:
                recv[r0:r0 + rn].copy_(buf)
            torch.cuda.current_stream().synchronize()

    def draw(s: int, flags: int):
        if sharded:
            ctx.draw_placed(packed[s], tile_outs[s], flags)
        else:
            ctx.draw_packed(packed[s], 0, flags)

    # ---- algorithmic bytes per launch: instrumented pass, outside the timed region -------------------
    ctx.enable_counters(True)
    alg_bytes, visits, lod_visits, pixels = [], [], [], []
    for s in range(total_steps):
        draw(s, gpu.DRAW_SYNC)
        c = ctx.counters()
        alg_bytes.append(c.algorithmic_bytes())
        visits.append(c.S)
        pixels.append(c.P)
        lod_visits.append(list(c.lodVisits))
    ctx.enable_counters(False)

    def exchange_step(s: int, par: int):
        """The exchange of step s (its tiles lie in the areas of parity `par`), enqueued on the current stream (s_exchange)."""
        if image_mode:
            ip = img_plans[s]
            ip.pack(ctx, s_exchange.cuda_stream, img_store[par].data_ptr(), img_send[par].data_ptr(), img_images[par].data_ptr())
            if comm:
                ip.exchange(ctx, comm, s_exchange.cuda_stream, img_send[par].data_ptr(), img_recv[par].data_ptr())
            else:
                image_exchange_torch(ip, img_send[par], img_recv[par])
            ip.unpack(ctx, s_exchange.cuda_stream, img_recv[par].data_ptr(), img_images[par].data_ptr())
        elif comm:
            native_plans[s].exchange(ctx, comm, s_exchange.cuda_stream, send[par].data_ptr(), disp[par].data_ptr())
        else:
            for req in plans[s].exchange(send[par], disp[par]):
                req.wait()

    def alone(what: str, s: int, repeats: int = 2):
        """One step's render, or one step's exchange, with nothing else on the GPU: wall ms between barriers, max over the ranks, best of `repeats`
        (untimed extras after the measured region: what the step costs when the two do NOT overlap)."""
        best = None
        for _ in range(repeats):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if what == "render":
                draw(s, gpu.DRAW_ASYNC)
                ctx.synchronize()
            else:
                with torch.cuda.stream(s_exchange):
                    exchange_step(s, s % 2)
                s_exchange.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            best = float(tmax.item()) if best is None else min(best, float(tmax.item()))
        return best * 1e3

    def run_region(first: int, last: int, overlap: bool):
        """Steps [first, last): render (+ exchange).  Returns after everything has completed on this rank."""
        ev_done = {}
        for s in range(first, last):
            if not sharded or args.no_exchange:
                draw(s, gpu.DRAW_ASYNC)
                continue
            par = s % 2
            if s - 2 in ev_done:
                s_render.wait_event(ev_done[s - 2])  # the exchange that read / filled this parity has finished
            draw(s, gpu.DRAW_ASYNC)                   # enqueued on s_render (the context's stream)
            ev_render = torch.cuda.Event()
            ev_render.record(s_render)
            s_exchange.wait_event(ev_render)
            with torch.cuda.stream(s_exchange):
                exchange_step(s, par)
                ev_done[s] = torch.cuda.Event()
                ev_done[s].record(s_exchange)
            if not overlap:
                ev_done[s].synchronize()
        ctx.synchronize()
        torch.cuda.synchronize()

    def verify_exchange(step: int) -> bool:
        """Frames assembled on this rank from N ranks' tiles == the same frames rendered whole here (GPU vs GPU;
        the GPU path itself is pinned to the CPU oracle by tests/)."""
        ok = True
        mine = [b for b in range(G) if b % N == rank][:2]
        for b in mine:
            fr = steps_frames[step][b]
            if image_mode:  # the gathered image == Phase 2 of the same frame rendered whole on this rank
                ctx.clear_raybuffers(0, 0)
                ctx.draw_segments(fr, 0)
                ok = ok and bool((img_images[step % 2][b // N].cpu().numpy().view(np.uint32) == ctx.blit_segments(0)).all())
                continue
            rc = [s.RayCount for s in fr.segments]
            a_td, a_lr = plans[step].assemble(disp[step % 2], b, rc, W, H)
            ctx.clear_raybuffers(0, 0)
            ctx.draw_segments(fr, 0)
            n_td, n_lr = max(0, rc[0]) + max(0, rc[1]), max(0, rc[2]) + max(0, rc[3])
            w_td = ctx.read_raybuffer(0, gpu.RAYBUFFER_TOPDOWN, 0, n_td)
            w_lr = ctx.read_raybuffer(0, gpu.RAYBUFFER_LEFTRIGHT, 0, n_lr)
            ok = ok and bool((a_td[:n_td] == w_td).all() and (a_lr[:n_lr] == w_lr).all())
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def timed(overlap: bool):
        run_region(0, args.warmup, overlap)
        barrier()
        ctx.draw_time_stats(reset=True)
        t0 = time.perf_counter()
        run_region(args.warmup, total_steps, overlap)
        barrier()
        dt = time.perf_counter() - t0
        if N > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    overlap = sharded and not args.no_exchange and not args.no_overlap
    elapsed = timed(overlap)
    k_ms_total, k_draws = ctx.draw_time_stats(reset=True)  # HIP events around each launch, on the launch stream
    exchange_verified = None
    if sharded and not args.no_exchange:
        exchange_verified = verify_exchange(total_steps - 1)
        if not exchange_verified and not args.allow_fallback:
            # A run whose assembled frames are wrong has no number: rank 0 prints what was measured up to here as a FAILURE
            # line -- no `value` -- and every rank exits non-zero.  (--allow-fallback re-times with the other exchange / without overlap instead.)
            if rank == 0:
                print(json.dumps({"metric": "Mrays/s, Phase-1 raybuffer rendering (DrawSegmentRayJob)", "value": None, "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
                                  "failure": "frames assembled from the ranks' tiles differ from the same frames rendered whole", "exchange_verified": False,
                                  "exchange_path": exchange_path, "gather": args.gather, "overlap": overlap, "ranks_seen": ranks_seen,
                                  "unverified_ms_per_step": round(elapsed / args.steps * 1e3, 4)}), flush=True)
            ctx.close()
            if N > 1:
                dist.destroy_process_group()
            raise SystemExit(5)
        if not exchange_verified and comm:
            # never report a number from a run whose frames are wrong: first fall back to the torch.distributed exchange
            gpu.comm_destroy(comm)
            comm = None
            exchange_path = "torch.distributed batch_isend_irecv (cvx_exchange produced wrong frames: reported as a failure of that path)"
            elapsed = timed(overlap)
            k_ms_total, k_draws = ctx.draw_time_stats(reset=True)
            exchange_verified = verify_exchange(total_steps - 1)
        if not exchange_verified and overlap:
            # ... then redo the region without overlap
            overlap = False
            elapsed = timed(False)
            k_ms_total, k_draws = ctx.draw_time_stats(reset=True)
            exchange_verified = verify_exchange(total_steps - 1)

    breakdown = None
    if sharded and not args.no_exchange:
        # The first multi-GPU line has to explain itself: the last step's render alone, its exchange alone, and what the overlap hid
        last_step = total_steps - 1
        render_alone = alone("render", last_step)
        exchange_alone = alone("exchange", last_step)
        per_gpu_bytes = (int(np.mean([p.send_pixels for p in img_plans]) * 4) if image_mode else int(np.mean([p.send_total for p in plans]) * 256))
        breakdown = scaling_breakdown(elapsed / args.steps * 1e3, render_alone, exchange_alone, per_gpu_bytes, N)

    steps = range(args.warmup, total_steps)
    total_rays = sum(rays_per_step[s] for s in steps)  # whole job: every ray of every frame is rendered by exactly one GPU
    total_frames = G * args.steps
    value = total_rays / elapsed / 1e6
    # roofline of the dominant kernel (render_kernel) on this rank: algorithmic bytes of its launches / their duration
    k_bytes = sum(alg_bytes[s] for s in steps)
    k_sec = k_ms_total / 1e3
    achieved = k_bytes / k_sec / 1e9
    if image_mode and not args.no_exchange:
        parallelism = f"ray-tile sharding x{N}, per-rank Phase 2 + P2P gather of the finished pixels (image gather)" + (" overlapped with the next render" if overlap else "")
    elif sharded and not args.no_exchange:
        parallelism = f"ray-tile sharding x{N}, zero-copy placement + RCCL P2P tile exchange" + (" overlapped with the next render" if overlap else "")
    else:
        parallelism = f"ray-tile sharding x{N}" + (" (no exchange)" if sharded else "")

    if args.world.startswith("proc"):
        workload = (f"procedural {args.world} world seed 0x5EED2048, {W}x{H}, full LOD chain (6 levels), benchmark-path poses "
                    f"({POSES} samples, stride {POSE_STRIDE}), lodError {args.lod_error}")
    else:
        workload = f"{args.world} (mill.obj voxelised), {W}x{H}, benchmark-path poses, lodError {args.lod_error}"

    result = {
        # BASELINE.json's metric on its configuration; other --width/--height/--world runs are labelled with what they ran
        "metric": ("Mrays/s, Phase-1 raybuffer rendering (DrawSegmentRayJob) at 1080p, 2048^3 world" if (W, H, args.world) == (1920, 1080, "proc2048")
                   else f"Mrays/s, Phase-1 raybuffer rendering (DrawSegmentRayJob) at {W}x{H}, {args.world} world"),
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
            "workload": workload,
            "frames_per_gpu_per_step": F,
            "frames_auto": predicted,
            "frames_per_step": G,
            "rays_per_frame_mean": round(total_rays / total_frames, 1),
            "parallelism": parallelism,
            "ranks_seen": ranks_seen,
            "exchange_verified": exchange_verified,
            "exchange_path": exchange_path,
            "gather": args.gather if sharded else None,
            "gather_auto": gather_choice,
            "exchange_bytes_per_step_per_gpu": (int(np.mean([p.send_pixels for p in img_plans]) * 4) if image_mode else int(np.mean([p.send_total for p in plans]) * 256)) if sharded else 0,
            "world_dims": list(dims),
            "lod_distances": lods,
            "lod_visits_per_step": [int(np.mean([lod_visits[s][l] for s in steps])) for l in range(6)],
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None,
            "kernel": "cvxk::render_kernel",
            "kernel_ms_avg": round(k_ms_total / max(1, k_draws), 4),
            "algorithmic_bytes_per_launch": int(k_bytes / max(1, k_draws)),
            "bytes_per_ray": round(k_bytes / max(1, total_rays / N), 1),
            "column_visits_per_s": round(sum(visits[s] for s in steps) / k_sec, 1),
        },
    }

    if breakdown:
        result.update(breakdown)
    if not args.pmc_csv and N == 1:
        args.pmc_csv = find_counter_summary(args)
    if args.pmc_csv:
        # Counter summary of a rocprofv3 --pmc run of THIS command and build (tools/pmc_passes.sh + tools/pmc_aggregate.py); never a
        # committed file of another build.  `traffic` = bytes the L2 exchanged with the fabric (Infinity Cache / HBM) per launch, calibrated
        # (profiles/r04_fetch_calibration.md, tools/fetch_calibration.hip): on gfx950 EVERY read request of the L2 is a 128-byte line, also for a
        # scattered 4-, 16- or 32-byte access, and FETCH_SIZE tallies each at 64 bytes -- so fetched bytes = 128 x TCC_EA0_RDREQ_128B + 64 x (the
        # 64-byte requests) + 32 x TCC_EA0_RDREQ_32B when the summary holds those counters, else 2 x FETCH_SIZE (the same number when all requests
        # are 128-byte ones, which is what this kernel produces).  WRITE_SIZE is exact.
        try:
            counters = read_counter_summary(args.pmc_csv, args.warmup, args.steps)
            if counters is None:
                raise ValueError("its launches do not cover the timed steps of this run")
            fetch_raw, write = counters["FETCH_SIZE"] * 1024, counters["WRITE_SIZE"] * 1024
            if all(k in counters for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum")):
                n, n32, n128 = counters["TCC_EA0_RDREQ_sum"], counters["TCC_EA0_RDREQ_32B_sum"], counters["TCC_EA0_RDREQ_128B_sum"]
                fetch, how = 128.0 * n128 + 64.0 * max(0.0, n - n32 - n128) + 32.0 * n32, "request counters by size (TCC_EA0_RDREQ_32B / _128B / total)"
            else:
                fetch, how = 2.0 * fetch_raw, "2 x FETCH_SIZE (every read request of this kernel is a 128-byte line tallied at 64 bytes)"
            launches = max(1, k_draws)
            if all(k in counters for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VALU_TRANS_F32", "GRBM_GUI_ACTIVE")):
                # The roof the kernel is actually under: instruction ISSUE.  Executed instructions per launch (SQ_INSTS_*, the same counter
                # passes) x the issue prices tools/valu_rate.hip measures on gfx950 at this occupancy (profiles/r04_valu_rate.txt, r05_valu_rate_salu_mix.txt:
                # a vector instruction 2.25 cycles of its SIMD, v_rcp_f32 8.1 .. 13.2, a scalar instruction or branch 1.0 .. 2.25 -- hidden behind another
                # wave's vector issue or not --, an LDS / vector-memory instruction 2.3) against the SIMD cycles the launch had: GRBM_GUI_ACTIVE / 8 XCDs x
                # 1024 SIMDs.  (profiles/r05_issue_model.md is the same model on the exact per-block instruction counts.)
                trans, valu = counters["SQ_INSTS_VALU_TRANS_F32"], counters["SQ_INSTS_VALU"] - counters["SQ_INSTS_VALU_TRANS_F32"]
                scalar = counters["SQ_INSTS_SALU"] + counters["SQ_INSTS_BRANCH"]
                mem = sum(counters.get(k, 0.0) for k in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
                simd_cycles = counters["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
                lo = (2.25 * valu + 8.1 * trans + 1.0 * scalar + 2.3 * mem) / simd_cycles
                hi = (2.25 * valu + 13.2 * trans + 2.25 * scalar + 2.3 * mem) / simd_cycles
                result["roofline"]["secondary"] = {
                    "bound": "valu_issue", "frac": round((lo + hi) / 2.0, 4), "frac_range": [round(lo, 4), round(hi, 4)],
                    "instructions_per_launch": {"vector": int(valu + trans), "v_rcp_f32": int(trans), "scalar_and_branch": int(scalar), "lds_and_vector_memory": int(mem)},
                    "simd_cycles_per_launch": int(simd_cycles),
                    "what": "executed instructions x measured issue prices / SIMD cycles of the launch: the share of the launch the SIMDs' issue ports are busy (the kernel's "
                            "arithmetic is the reference's IEEE-exact f32 with divisions; no dense contraction, so no MFMA roof applies)",
                }
            pixel_bytes = 4 * sum(pixels[s] for s in steps) / launches
            result["roofline"]["traffic"] = int(fetch + write)
            result["roofline"]["traffic_detail"] = {
                "fetch_bytes": int(fetch), "fetch_calibration": how, "fetch_size_counter_raw": int(fetch_raw), "write_bytes": int(write),
                "algorithmic_pixel_bytes": int(pixel_bytes), "write_amplification": round(write / max(1.0, pixel_bytes), 3),
                "fabric_TB_per_s": round((fetch + write) / (k_ms_total / max(1, k_draws)) / 1e9, 3),
                "tcc_hit_rate": (round(counters["TCC_HIT_sum"] / max(1.0, counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"]), 4)
                                 if "TCC_HIT_sum" in counters and "TCC_MISS_sum" in counters else None),
                "source": f"{os.path.relpath(args.pmc_csv, ROOT)}: rocprofv3 --pmc passes of this workload with this library (stamped with its sha-256 and the bench arguments), per launch of render_kernel<false>",
            }
        except (OSError, KeyError, ValueError) as e:
            result["roofline"]["traffic_detail"] = {"error": f"unreadable counter summary {args.pmc_csv}: {e}"}

    parity_frames = []
    if N == 1:
        # what the GPU rendered for eight frames spread over the LAST timed step (compared with the CPU oracle in the cpu_baseline leg below)
        last = steps_frames[total_steps - 1]
        for b in sorted({(F * i) // 8 for i in range(8)}):
            rc = [max(0, sg.RayCount) for sg in last[b].segments]
            parity_frames.append((b, last[b], ctx.read_raybuffer(b, gpu.RAYBUFFER_TOPDOWN, 0, rc[0] + rc[1]),
                                  ctx.read_raybuffer(b, gpu.RAYBUFFER_LEFTRIGHT, 0, rc[2] + rc[3])))

    if N == 1:
        # Phase 2 (RenderManager.BlitSegments) over the frames of the last step, image left in HBM: reported beside
        # the Phase-1 metric, never part of `value` (and never allowed to take the Phase-1 line down with it).
        try:
            for b in range(min(F, 8)):
                ctx.blit_segments(b, to_host=False)
            ctx.synchronize()
            t0 = time.perf_counter()
            for b in range(F):
                ctx.blit_segments(b, to_host=False)
            ctx.synchronize()
            blit_ms = (time.perf_counter() - t0) * 1e3 / F
            phase1_ms = elapsed * 1e3 / total_frames
            # ... and as ONE launch over the step's frames (cvx_blit_segments_batch, F images left in HBM): 4 B gathered + 4 B stored per pixel
            images = torch.empty((F, H, W), dtype=torch.int32, device=device)
            ctx.blit_segments_batch(0, F, images.data_ptr())
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.blit_segments_batch(0, F, images.data_ptr())
            ctx.synchronize()
            batch_ms = (time.perf_counter() - t0) * 1e3 / (3 * F)
            del images
            result["phase2"] = {"blit_ms_per_frame": round(batch_ms, 4), "fps_phase1_plus_phase2": round(1e3 / (phase1_ms + batch_ms), 2),
                                "blit_gbps": round(8.0 * W * H / (batch_ms * 1e-3) / 1e9, 1),
                                "single_launch_blit_ms_per_frame": round(blit_ms, 4), "fps_phase1_plus_single_launch_blits": round(1e3 / (phase1_ms + blit_ms), 2),
                                "what": f"cvx_blit_segments_batch over the {F} frames of a step ({W}x{H} ARGB32 images, device resident; blit_gbps = 8 B per screen pixel / time) "
                                        f"and cvx_blit_segments, one launch per frame"}
        except Exception as e:  # noqa: BLE001
            result["phase2"] = {"error": str(e)}

    if N == 1 and args.latency_frames > 0:
        # The reference's own call pattern: ONE blocking DrawSegments per frame (RenderManager.cs:155-167,363), and the same with its two
        # raybuffer pairs (BUFFER_COUNT = 2, RenderManager.cs:14,53-56) used as a 2-deep pipeline: frame k is submitted (CVX_DRAW_ASYNC)
        # while the host still holds frame k - 1.  Reported beside the batch metric, never as `value`.
        try:
            K = args.latency_frames
            singles = [frame_for(i) for i in range(K)]
            packed1 = [ctx.pack_batch([f]) for f in singles]
            rays1 = sum(f.totalRays for f in singles)
            for k in range(min(K, 8)):
                ctx.draw_packed(packed1[k], k % 2, gpu.DRAW_SYNC)
            ctx.draw_time_stats(reset=True)
            per_frame = []
            t0 = time.perf_counter()
            for k in range(K):
                t1 = time.perf_counter()
                ctx.draw_packed(packed1[k], k % 2, gpu.DRAW_SYNC)
                per_frame.append(time.perf_counter() - t1)
            dt = time.perf_counter() - t0
            k_ms1, n1 = ctx.draw_time_stats(reset=True)
            worst = max(range(K), key=lambda k: per_frame[k])
            st = torch.cuda.Stream(device)
            ctx.set_stream(st.cuda_stream)
            events = []
            t0 = time.perf_counter()
            for k in range(K):
                ctx.draw_packed(packed1[k], k % 2, gpu.DRAW_ASYNC)
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
                if k >= 1:
                    events[k - 1].synchronize()  # the host consumes frame k - 1 while frame k renders
            events[-1].synchronize()
            dt2 = time.perf_counter() - t0
            ctx.set_stream(None)
            # ... the same blocking calls pinned to the batch kernel (what rounds 1-5 ran a single frame on), and the curve: ms per blocking launch of n frames
            # with the library's own kernel choice (the latency kernel up to 12288 rays per launch, the batch kernel beyond)
            ctx.set_latency_kernel(gpu.LATENCY_NEVER)
            t0 = time.perf_counter()
            for k in range(K):
                ctx.draw_packed(packed1[k], k % 2, gpu.DRAW_SYNC)
            dt_batch = time.perf_counter() - t0
            ctx.set_latency_kernel(gpu.LATENCY_AUTO)
            curve = {}
            for n in (1, 2, 4, 8, 16, 64):
                if n > F or n > K:
                    continue
                packs = [ctx.pack_batch(singles[i:i + n]) for i in range(0, K - n + 1, n)]
                ctx.draw_packed(packs[0], 0, gpu.DRAW_SYNC)
                t0 = time.perf_counter()
                for pk in packs:
                    ctx.draw_packed(pk, 0, gpu.DRAW_SYNC)
                curve[str(n)] = round((time.perf_counter() - t0) / len(packs) * 1e3, 4)
            # two of the single frames back through the latency kernel for the parity leg below
            for k in (0, K // 2):
                ctx.set_latency_kernel(gpu.LATENCY_ALWAYS)
                ctx.draw_packed(packed1[k], 0, gpu.DRAW_SYNC)
                ctx.set_latency_kernel(gpu.LATENCY_AUTO)
                rc = [max(0, sg.RayCount) for sg in singles[k].segments]
                parity_frames.append((f"latency kernel, pose {(k * POSE_STRIDE) % POSES}", singles[k], ctx.read_raybuffer(0, gpu.RAYBUFFER_TOPDOWN, 0, rc[0] + rc[1]),
                                      ctx.read_raybuffer(0, gpu.RAYBUFFER_LEFTRIGHT, 0, rc[2] + rc[3])))
            result["fps_per_frame_latency"] = round(K / dt, 1)
            result["latency_curve"] = {"ms_per_launch_by_frames": curve, "what": "wall ms of ONE blocking cvx_draw_segments_batch of n frames (the bench poses), mean over the launches; "
                                       "automatic kernel choice (cvx_set_latency_kernel AUTO)"}
            result["latency"] = {
                "frames": 1, "ms": round(dt / K * 1e3, 4), "fps": round(K / dt, 1), "mrays": round(rays1 / dt / 1e6, 3),
                "kernel": "cvxk::lone_kernel (one wave per ray, lanes = the ray's next 64 columns: csrc/cvx_lone.h)",
                "kernel_ms": round(k_ms1 / max(1, n1), 4),
                "ms_batch_kernel": round(dt_batch / K * 1e3, 4),
                "ms_max": round(per_frame[worst] * 1e3, 4), "fps_min": round(1.0 / per_frame[worst], 1), "worst_pose": (worst * POSE_STRIDE) % POSES,
                "ms_p95": round(sorted(per_frame)[min(K - 1, (95 * K) // 100)] * 1e3, 4),  # (ms_max is one wall-clock sample: a host hiccup shows up there)
                "pipelined_2deep": {"ms": round(dt2 / K * 1e3, 4), "fps": round(K / dt2, 1), "mrays": round(rays1 / dt2 / 1e6, 3)},
                "what": f"{K} single-frame cvx_draw_segments calls (first {K} poses of the bench), blocking / 2-deep CVX_DRAW_ASYNC over two raybuffer pairs; wall clock",
            }
        except Exception as e:  # noqa: BLE001
            result["latency"] = {"error": str(e)}

    parity_failed = False
    if rank == 0 and N == 1 and args.cpu_seconds > 0:  # the CPU baseline is reported at N = 1 only
        try:
            result["cpu_baseline"], parity = cpu_baseline(ws, steps_frames[args.warmup], W, H, args.cpu_seconds, parity_frames)
            result["parity_checked"] = parity["ok"]
            result["parity"] = parity
            parity_failed = not parity["ok"]
        except Exception as e:  # noqa: BLE001  (the GPU measurement above stands on its own)
            result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
            result["parity_checked"] = False
    else:
        result["parity_checked"] = False  # no oracle leg in this run (N > 1 checks the exchange against single-GPU renders instead)
    if rank == 0:
        print(json.dumps(result), flush=True)
    barrier()
    if comm:
        gpu.comm_destroy(comm)
    ctx.close()
    if N > 1:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the GPU raybuffers of the timed frames differ from the CPU oracle")


XGMI_LINK_GBPS = 153.0  # one xGMI link of an MI355X (7 per GPU, point to point)


def scaling_breakdown(ms_per_step, render_ms_alone, exchange_ms_alone, exchange_bytes_per_gpu, n_gpus):
    """Keys an N > 1 line carries so that it explains itself: what a step's render and a step's exchange cost on their own, how much of the cheaper one
    the overlap hid (1 = the step took max(render, exchange), 0 = their sum), and what the payload would take on the wire -- every peer pair has its
    own xGMI link, so a GPU's (N - 1) transfers run side by side and the bound is ONE peer's share over ONE link."""
    hidden = render_ms_alone + exchange_ms_alone - ms_per_step
    smaller = min(render_ms_alone, exchange_ms_alone)
    per_peer = exchange_bytes_per_gpu / max(1, n_gpus - 1)
    return {
        "render_ms_alone": round(render_ms_alone, 4),
        "exchange_ms_alone": round(exchange_ms_alone, 4),
        "overlap_efficiency": round(max(0.0, min(1.0, hidden / smaller)), 4) if smaller > 0 else None,
        "payload_ms_per_link_predicted": round(per_peer / (XGMI_LINK_GBPS * 1e9) * 1e3, 4),
        "exchange_effective_gbps_per_link": round(per_peer / (exchange_ms_alone * 1e-3) / 1e9, 2) if exchange_ms_alone > 0 else None,
        "scaling_notes": "render_ms_alone / exchange_ms_alone: one untimed extra step each after the measured region (max over ranks); overlap_efficiency = "
                         "(render + exchange - ms_per_step) / min(render, exchange); payload_ms_per_link_predicted = this GPU's bytes per peer and step / 153 GB/s.  "
                         "RCCL's send / receive kernels take CUs from a render kernel that is bound by instruction issue: NCCL_MAX_NCHANNELS (e.g. 4 .. 8) caps how many",
    }


def choose_gather(raybuffer_bytes, image_bytes):
    """--gather auto: the gather that puts fewer bytes on the wire (the image gather pays a per-rank Phase 2 for it; BASELINE.json's north_star names the raybuffer gather)."""
    pick = "image" if image_bytes < raybuffer_bytes else "raybuffer"
    return {"gather": pick, "raybuffer_bytes": int(raybuffer_bytes), "image_bytes": int(image_bytes),
            "why": f"{pick} gather: {min(raybuffer_bytes, image_bytes) / max(1, max(raybuffer_bytes, image_bytes)):.2f} x the bytes of the other (rank 0's share of step 0)"}


def predict_frames_auto(frame_for, W, H, rank, N, budget_bytes, candidates=(512, 256, 128, 64, 32, 16, 8)):
    """--frames auto: the largest F whose raybuffer areas fit the stated HBM budget -- one GPU: F raybuffer pairs; N > 1: the
    per-destination send sections + the display area, two parities each (exact sizes from the shard plan of step 0) -- and the payload every xGMI
    link would carry per step, all BEFORE anything is allocated.  Host arithmetic only (tests/test_dist.py runs it for the 8-rank 4K shapes)."""
    choice = None
    for cand in candidates:
        if N > 1:
            from cpuvox_amd import dist as cdist_

            plan0 = cdist_.ShardPlan([frame_for(i) for i in range(N * cand)], W, H, rank, N)
            area = 2.0 * (plan0.send_total + plan0.disp_total) * 256.0
            per_link = plan0.send_total * 256.0 / max(1, N - 1)
        else:
            area = cand * 4.0 * (H * (W + 2 * H) + W * (2 * W + H))  # RenderManager.cs:35-36 capacities per raybuffer pair
            per_link = 0.0
        choice = (cand, area, per_link)
        if area <= budget_bytes:
            break
    return {"frames_per_gpu": choice[0], "area_bytes_per_gpu": int(choice[1]), "hbm_budget_bytes": int(budget_bytes), "fits_budget": bool(choice[1] <= budget_bytes),
            "payload_bytes_per_link_per_step": int(choice[2]), "link_ms_at_153_GBps": round(choice[2] / 153e9 * 1e3, 3)}


def read_counter_summary(path, warmup, steps):
    """{counter: mean per launch over the TIMED steps [warmup, warmup + steps)} from a tools/pmc_aggregate.py summary, or None when the
    summary cannot speak for those steps.  Launch i of a bench.py run is step i (warm-up first) and a step's frames depend on its index
    only, so a summary that lists every launch (`per_launch` column, round 5) serves any --steps / --warmup its launches cover -- the
    driver's 20 / 5 as well as the default 10 / 2.  A summary without that column (rounds 1-4) only holds the mean over the timed
    steps of the run that collected it: usable for exactly the same --steps / --warmup."""
    import csv

    with open(path, newline="") as fh:
        first = fh.readline()
        stamp = json.loads(first[1:]) if first.startswith("#") else {}
        rows = list(csv.DictReader(l for l in ([] if first.startswith("#") else [first]) + fh.readlines() if not l.startswith("#")))
    theirs = stamped_bench_args(stamp)
    out = {}
    for row in rows:
        series = [float(v) for v in (row.get("per_launch") or "").split(";") if v]
        if series:
            if len(series) != theirs["steps"] + theirs["warmup"] or warmup + steps > len(series):
                return None  # (other launches were counted too, e.g. latency legs, or the run was shorter than this one)
            out[row["counter"]] = sum(series[warmup:warmup + steps]) / steps
        elif (theirs["steps"], theirs["warmup"]) == (steps, warmup):
            out[row["counter"]] = float(row["mean_per_launch"])
        else:
            return None
    return out or None


def stamped_bench_args(stamp):
    """The workload arguments of the bench.py runs a counter summary was collected with (defaults filled in)."""
    ap = argparse.ArgumentParser()
    for flag, typ in (("--frames", int), ("--width", int), ("--height", int), ("--world", str), ("--lod-error", float), ("--pose-range", str), ("--steps", int), ("--warmup", int)):
        ap.add_argument(flag, type=typ, default=None)
    theirs, _ = ap.parse_known_args(stamp.get("bench_args", []))
    return {k: (getattr(theirs, k) if getattr(theirs, k) is not None else parse_default(k)) for k in ("frames", "width", "height", "world", "lod_error", "pose_range", "steps", "warmup")}


def find_counter_summary(args):
    """profiles/rNN_pmc_render_kernel*.csv of THIS build and THIS workload whose launches cover the timed steps of this run, or None.
    tools/profile_round.sh stamps the file with the sha-256 of the library the counter passes loaded and with their bench arguments
    (tools/pmc_aggregate.py); counters of another build or of another shape are never attached to the line (roofline.traffic stays null)."""
    import glob

    if os.environ.get("CVX_GPU_LIB"):
        return None  # another build of the ABI was selected: no committed counters belong to it
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from cpuvox_amd import gpu
        from pmc_aggregate import library_sha256

        mine_lib = library_sha256(gpu.lib_path())
    except Exception:  # noqa: BLE001
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_render_kernel*.csv")), reverse=True):
        try:
            with open(path) as fh:
                first = fh.readline()
            if not first.startswith("#"):
                continue
            stamp = json.loads(first[1:])
            theirs = stamped_bench_args(stamp)
            same = all(theirs[k] == getattr(args, k) for k in ("frames", "width", "height", "world", "lod_error", "pose_range"))
            # the library ITSELF must be the one the counters were collected with (the build is deterministic), not just its sources
            if stamp.get("library_sha256") == mine_lib and same and read_counter_summary(path, args.warmup, args.steps) is not None:
                return path
        except (OSError, ValueError, KeyError):
            continue
    return None


def parse_default(name):
    return {"frames": 512, "width": 1920, "height": 1080, "world": "proc2048", "lod_error": 1.0, "pose_range": None, "steps": 10, "warmup": 2}[name]


def cpu_baseline(ws, frames, W, H, budget_s: float, parity_frames=()):
    """The CPU oracle (oracle/cvx_oracle.c: the same algorithm, OpenMP parallel-for over rays, grain 1 like
    RenderJob) timed on this box's host cores on a bounded sample of the same frames.  kind = "port":
    the reference itself (C#/Unity/Burst) cannot run here.  The same leg also checks the GPU: `parity_frames` are
    raybuffers of the last timed step as read back from the device; the oracle renders the same frames and every pixel
    the frame writes (rows of the used rays x [origMin, origMax] of their segment) must be identical."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oraclelib as O

    threads = O.lib().orc_max_threads()
    (td_rays, td_w), (lr_rays, lr_w) = O.raybuffer_shapes(W, H)
    bufs = (np.zeros((td_rays, td_w), dtype=np.uint32), np.zeros((lr_rays, lr_w), dtype=np.uint32))
    O.draw_segments(ws, frames[0], W, H, counters=False, out=bufs)  # warm-up (page in the world)
    def sample(library, seconds):
        r = k = 0
        t_start = time.perf_counter()
        while True:  # bounded by wall time: frames differ a lot in cost
            f = frames[k % len(frames)]
            O.draw_segments(ws, f, W, H, counters=False, out=bufs, library=library)
            r += f.totalRays
            k += 1
            t = time.perf_counter() - t_start
            if t >= seconds or k >= 4096:
                return r, k, t

    rays, n, dt = sample(None, budget_s)
    # A second, fairer number for the reference's real build: Burst compiles these jobs with FloatMode.Fast (DrawSegmentRayJob.cs:11,
    # 48,86,155), i.e. reassociation / contraction allowed.  Same C file, -O3 -march=native -ffast-math, same threads, same frames;
    # never parity-checked (its pixels may legitimately differ) and never `value`.
    fast = None
    try:
        fl = O.lib_fast()
        if fl is not None:
            O.draw_segments(ws, frames[0], W, H, counters=False, out=bufs, library=fl)
            fr, fn, fdt = sample(fl, max(3.0, budget_s / 3))
            fast = {"value": round(fr / fdt / 1e6, 4), "unit": "Mrays/s", "fps": round(fn / fdt, 2), "cores": threads, "parity_checked": False,
                    "build": "gcc -O3 -march=native -ffast-math (Burst FloatMode.Fast analogue), built on this host",
                    "sample": f"{fn} frames ({fr} rays), {fdt:.1f} s wall"}
    except Exception as e:  # noqa: BLE001
        fast = {"value": None, "error": str(e)}
    baseline = {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "fps": round(n / dt, 2),
        "sample": f"{n} frames cycling over the first timed step ({rays} rays), {dt:.1f} s wall, OpenMP {threads} threads, wall clock of orc_draw_segments only",
        "build": "gcc -O2 -ffp-contract=off -fno-fast-math (strict IEEE: the parity contract)",
        "fast_math": fast,
    }
    from cpuvox_amd import dist as cdist

    parity = {"ok": True, "frames": [], "pixels_compared": 0, "pixels_differing": 0}
    for b, fr, g_td, g_lr in parity_frames:
        o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, counters=False)
        rc = [max(0, sg.RayCount) for sg in fr.segments]
        ranges = cdist.segment_pixel_ranges(fr.vanishingPointScreenSpace, W, H)
        row0 = [0, rc[0], 0, rc[2]]
        for sg in range(4):
            g, o = (g_td, o_td) if sg < 2 else (g_lr, o_lr)
            lo, hi = ranges[sg]
            a = g[row0[sg]: row0[sg] + rc[sg], lo: hi + 1]
            e = o[row0[sg]: row0[sg] + rc[sg], lo: hi + 1]
            parity["pixels_compared"] += int(a.size)
            parity["pixels_differing"] += int((a != e).sum())
        parity["frames"].append(b)
    parity["ok"] = parity["pixels_differing"] == 0 and parity["pixels_compared"] > 0
    parity["what"] = ("frames of the last timed step (buffer indices above: the batch kernel) and two of the single-frame draws (the latency kernel): device raybuffers vs the CPU "
                      "oracle, every pixel the frame writes")
    return baseline, parity


if __name__ == "__main__":
    main()
