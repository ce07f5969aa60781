"""CPU, gloo, world_size 2: the multi-GPU path's sharding + tile exchange (cpuvox_amd/dist.py).

Each rank holds only the tiles it "rendered" (tile t of every frame with t % N == rank; the pixel data comes
from the oracle, re-laid-out into the device's tile-major format by the test) and after TileExchange.run()
frame f must be complete, bit for bit, on rank f % N.

This is a test of the sharding plan and of the transport (which rows travel where and land where), not of the renderer:
no GPU output is involved here -- what the GPU renders into these buffers is pinned to the oracle by the -m gpu tests
(test_zero_copy_sharding_emulated_on_one_gpu, test_two_rank_exchange_emulated_on_one_gpu), and the plan the library
itself computes (cvx_shard_plan_*) is checked against this Python plan in tests/test_abi.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oraclelib as O
import scenes
from cpuvox_amd import dist as cdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 320, 200
TIMES = (0.1, 0.5, 0.75, 0.9, 1.1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _tile_major(frame, td, lr, tiles_td, tiles_lr):
    """Oracle ray-major buffers -> the tile-major pools libcpuvox_gpu renders into (cvx_device.h)."""
    pool_td = np.zeros((tiles_td, H * 64), dtype=np.uint32)
    pool_lr = np.zeros((tiles_lr, W * 64), dtype=np.uint32)
    rc = [max(0, s.RayCount) for s in frame.segments]
    for s in range(4):
        pool, buf, col = (pool_td, td, H) if s < 2 else (pool_lr, lr, W)
        row0 = rc[s - 1] if s in (1, 3) else 0
        base = (rc[s - 1] + 63) // 64 if s in (1, 3) else 0
        for plane in range(rc[s]):
            pool[base + plane // 64].reshape(col, 64)[:, plane % 64] = buf[row0 + plane]
    return pool_td, pool_lr


def _worker(rank, world_size, port, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        ws = scenes.load_world("proc256")
        frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in TIMES]
        tiles_td, tiles_lr = cdist.tile_capacity(W, H)
        G = len(frames)
        pools = cdist.Pools(torch.zeros((G * tiles_td, H * 64), dtype=torch.int32), torch.zeros((G * tiles_lr, W * 64), dtype=torch.int32), tiles_td, tiles_lr)
        full = []
        for b, fr in enumerate(frames):
            td, lr, _ = O.draw_segments(ws, fr, W, H, threads=2)
            ptd, plr = _tile_major(fr, td, lr, tiles_td, tiles_lr)
            full.append((ptd, plr))
            rc = [s.RayCount for s in fr.segments]
            for t, (kind, tile, _seg) in enumerate(cdist.frame_tiles(rc)):
                if t % world_size == rank:  # what cvx_set_shard(rank, N) renders on this rank
                    if kind == 0:
                        pools.td[b * tiles_td + tile] = torch.from_numpy(ptd[tile].view(np.int32))
                    else:
                        pools.lr[b * tiles_lr + tile] = torch.from_numpy(plr[tile].view(np.int32))
        ex = cdist.TileExchange(frames, W, H, rank, world_size, pools, torch.device("cpu"))
        ex.run()
        ok = True
        for b in range(G):
            if b % world_size != rank:
                continue
            got_td = pools.td[b * tiles_td:(b + 1) * tiles_td].numpy().view(np.uint32)
            got_lr = pools.lr[b * tiles_lr:(b + 1) * tiles_lr].numpy().view(np.uint32)
            ok = ok and bool((got_td == full[b][0]).all() and (got_lr == full[b][1]).all())
        result_queue.put((rank, ok, ex.sent_rows, ex.recv_rows))
    finally:
        dist.destroy_process_group()


def test_tile_exchange_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert all(ok for _, ok, _, _ in results), results
    # every tile sent by one rank is received by the other
    assert results[0][2] == results[1][3] and results[1][2] == results[0][3] and results[0][2] > 0


def test_only_writable_rows_travel():
    """The payload of a tile is rows [origMin, origMax] of its segment, not the whole tile."""
    ranges = cdist.segment_pixel_ranges((100.4, 50.5), 320, 200)
    assert ranges == [(50, 199), (0, 50), (100, 319), (0, 100)]  # RoundToInt(50.5) = 50 (half to even)
    assert cdist.segment_pixel_ranges((-5000.0, 1e9), 320, 200) == [(199, 199), (0, 199), (0, 319), (0, 0)]


def test_frame_tiles_matches_the_library_numbering():
    """frame_tiles() mirrors BuildFrame() in cvx_gpu.hip: segment-major, segment 1/3 tiles follow 0/2 in their pool."""
    tiles = cdist.frame_tiles([130, 64, 0, 65])
    assert [(k, t) for k, t, _ in tiles] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1)]
    assert [s for _, _, s in tiles] == [0, 0, 0, 1, 3, 3]
    assert cdist.frame_tiles([0, 0, 0, 0]) == []
    assert cdist.tile_capacity(1920, 1080) == ((1920 + 2160 + 63) // 64 + 2, (3840 + 1080 + 63) // 64 + 2)


def _tile_rows(frame, td, lr, kind, tile, seg, lo, rows):
    """What the kernel writes for one tile: pixel rows [lo, lo+rows) x 64 lanes, from the oracle's ray-major buffers."""
    rc = [max(0, s.RayCount) for s in frame.segments]
    seg_tile0 = [0, (rc[0] + 63) // 64, 0, (rc[2] + 63) // 64]
    seg_row0 = [0, rc[0], 0, rc[2]]
    plane0 = (tile - seg_tile0[seg]) * 64
    lanes = min(64, rc[seg] - plane0)
    out = np.zeros((rows, 64), dtype=np.uint32)
    buf = td if kind == 0 else lr
    out[:, :lanes] = buf[seg_row0[seg] + plane0: seg_row0[seg] + plane0 + lanes, lo: lo + rows].T
    return out


def _shard_worker(rank, world_size, port, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        ws = scenes.load_world("proc256")
        frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in TIMES]
        plan = cdist.ShardPlan(frames, W, H, rank, world_size)
        send = torch.zeros((max(1, plan.send_total), 64), dtype=torch.int32)
        disp = torch.zeros((max(1, plan.disp_total), 64), dtype=torch.int32)
        oracle = [O.draw_segments(ws, fr, W, H, threads=2)[:2] for fr in frames]
        # "render": put my tiles where ShardPlan.tile_out tells the kernel to put them
        out = plan.tile_out(1 << 40, 1 << 41)  # fake bases (the address may lie in front of the slot: rows < origMin)
        index = 0
        areas = (send.numpy().view(np.uint32), disp.numpy().view(np.uint32))
        for b, fr in enumerate(frames):
            ranges = cdist.segment_pixel_ranges(fr.vanishingPointScreenSpace, W, H)
            for kind, tile, seg in cdist.frame_tiles([s.RayCount for s in fr.segments]):
                addr = int(out[index])
                index += 1
                if addr == 0:
                    continue
                lo, hi = ranges[seg]
                area = 1 if addr >= (3 << 39) else 0
                row = (addr - ((1 << 41) if area else (1 << 40))) // 256 + lo  # tile_out points at pixel row 0
                areas[area][row: row + hi - lo + 1] = _tile_rows(fr, oracle[b][0], oracle[b][1], kind, tile, seg, lo, hi - lo + 1)
        for req in plan.exchange(send, disp):
            req.wait()
        ok = True
        for b, fr in enumerate(frames):
            if b % world_size != rank:
                continue
            td, lr = plan.assemble(disp, b, [s.RayCount for s in fr.segments], W, H)
            n_td, n_lr = scenes.used_rows(fr)
            ok = ok and bool((td[:n_td] == oracle[b][0][:n_td]).all() and (lr[:n_lr] == oracle[b][1][:n_lr]).all())
        result_queue.put((rank, ok, plan.send_total, plan.disp_total))
    finally:
        dist.destroy_process_group()


def test_zero_copy_shard_plan_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results), results
    assert results[0][2] > 0 and results[1][2] > 0


def test_bench_self_launches_one_process_per_gpu():
    """`python bench.py --gpus 2` started as ONE process (the driver's command) must become a torch.distributed.run launcher:
    two ranks come up (here they stop at "needs a GPU"), nothing dies in argument handling, the exit code is the children's."""
    import subprocess
    import sys

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert "launching:" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert "rank 0 of 2 up" in r.stderr and "rank 1 of 2 up" in r.stderr, r.stderr[-2000:]
    import torch

    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr


def _rccl_worker(rank, world_size, port, gather, result_queue):
    """One rank of the real thing: its own GPU, cvx_comm_create over the id broadcast by gloo, cvx_exchange / cvx_image_exchange on RCCL."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from cpuvox_amd import gpu

    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        ws = scenes.load_world("proc256")
        frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in TIMES]
        ctx = gpu.Context(rank)
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        packed = ctx.pack_batch(frames)
        uid = [gpu.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = gpu.comm_create(ctx, uid[0], rank, world_size, timeout_s=120.0)
        ok = True
        if gather == "raybuffer":
            plan = gpu.NativeShardPlan(packed, W, H, rank, world_size)
            pyplan = cdist.ShardPlan(frames, W, H, rank, world_size)
            send = torch.zeros((max(1, plan.send_total), 64), dtype=torch.int32, device=dev)
            disp = torch.zeros((max(1, plan.disp_total), 64), dtype=torch.int32, device=dev)
            ctx.draw_placed(packed, plan.tile_out(send.data_ptr(), disp.data_ptr()))
            plan.exchange(ctx, comm, None, send.data_ptr(), disp.data_ptr())
            ctx.synchronize()
            for b, fr in enumerate(frames):
                if b % world_size != rank:
                    continue
                g_td, g_lr = pyplan.assemble(disp, b, [s.RayCount for s in fr.segments], W, H)
                o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False, threads=2)
                n_td, n_lr = scenes.used_rows(fr)
                ok = ok and bool((g_td[:n_td] == o_td[:n_td]).all() and (g_lr[:n_lr] == o_lr[:n_lr]).all())
            sent = plan.send_total
        else:
            plan = gpu.ImagePlan(ctx, packed, W, H, rank, world_size)
            store = torch.zeros(max(1, plan.local_store_bytes // 4), dtype=torch.int32, device=dev)
            send = torch.zeros(max(1, plan.send_pixels), dtype=torch.int32, device=dev)
            recv = torch.zeros(max(1, plan.recv_pixels), dtype=torch.int32, device=dev)
            images = torch.zeros((max(1, plan.images), H, W), dtype=torch.int32, device=dev)
            ctx.draw_placed(packed, plan.tile_out(store.data_ptr()))
            plan.pack(ctx, None, store.data_ptr(), send.data_ptr(), images.data_ptr())
            plan.exchange(ctx, comm, None, send.data_ptr(), recv.data_ptr())
            plan.unpack(ctx, None, recv.data_ptr(), images.data_ptr())
            ctx.synchronize()
            got = images.cpu().numpy().view(np.uint32)
            for b, fr in enumerate(frames):
                if b % world_size != rank:
                    continue
                o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False, threads=2)
                ok = ok and bool((got[b // world_size] == O.blit_reference(fr, o_td, o_lr, W, H)).all())
            sent = plan.send_pixels
        gpu.comm_destroy(comm)
        ctx.close()
        result_queue.put((rank, ok, int(sent)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["raybuffer", "image"])
def test_exchange_on_real_rccl_two_gpus(gather):
    """cvx_comm_create + cvx_exchange / cvx_image_exchange with a real peer (ncclCommInitRank, grouped ncclSend / ncclRecv over xGMI),
    two ranks on two GPUs, frames against the oracle.  Skipped on hosts with fewer than two devices (the one-GPU boxes of this pool)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, gather, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results.sort()
    assert all(ok for _, ok, _ in results), results
    assert results[0][2] > 0 and results[1][2] > 0


@pytest.mark.parametrize("name, dim, lod_error", [("config4", 2048, 1.0), ("config5", 4096, 4.0)])
def test_eight_rank_plan_of_the_4k_shapes_fits_its_hbm_budget(name, dim, lod_error):
    """VERDICT r4 item 7 (RenderManager.cs:35-36): a pinned dry run of what the first real 8-GPU run of BASELINE configs 4 / 5
    (3840x2160, 2048^3 / 4096^3) will allocate and put on the wire -- `bench.py --gpus 8 --frames auto` as host arithmetic, no GPU, no world
    (a frame only needs the world's dimensions).  Asserts, for every rank: the chosen frame count's send + display areas (two parities) fit the
    stated budget (default 96 GB of the 288 GB) AND the device; what leaves a rank is what its peers expect; every frame's rows are displayed
    exactly once; the per-link payload bench.py prints is the plan's; the native plan the exchange uses (cvx_shard_plan_*) agrees."""
    import importlib.util

    from cpuvox_amd import gpu, host

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    W, H, N = 3840, 2160, 8
    dims = (dim, dim, dim)
    lods, far = host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), dim, W, H, lod_error)
    cache = {}

    def frame_for(g):  # bench.py's frame_for: global frame g -> benchmark pose (g * stride) % POSES
        i = (g * bench.POSE_STRIDE) % bench.POSES
        if i not in cache:
            pos, eul = host.sample_benchmark_path(i / bench.POSES * host.BENCHMARK_PATH_LENGTH, dims)
            cache[i] = host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, dims[1])
        return cache[i]

    budget = 96e9
    # the candidates a CPU test can afford (plans are Python loops over every tile): the shape of the answer, not the largest batch
    predictions = [bench.predict_frames_auto(frame_for, W, H, r, N, budget, candidates=(32, 8)) for r in range(N)]
    F = predictions[0]["frames_per_gpu"]
    assert all(p["frames_per_gpu"] == F and p["fits_budget"] for p in predictions), predictions
    frames = [frame_for(i) for i in range(N * F)]
    plans = [cdist.ShardPlan(frames, W, H, r, N) for r in range(N)]
    rows_per_frame_total = 0
    for fr in frames:
        rc = [s.RayCount for s in fr.segments]
        ranges = cdist.segment_pixel_ranges(fr.vanishingPointScreenSpace, W, H)
        rows_per_frame_total += sum(ranges[seg][1] - ranges[seg][0] + 1 for _, _, seg in cdist.frame_tiles(rc))
    assert sum(p.disp_total for p in plans) == rows_per_frame_total  # every writable row of every frame lands on exactly one display rank
    for r, (p, pred) in enumerate(zip(plans, predictions)):
        area = 2 * (p.send_total + p.disp_total) * 256
        assert pred["area_bytes_per_gpu"] == area <= budget < 288e9
        assert pred["payload_bytes_per_link_per_step"] == p.send_total * 256 // (N - 1)
        for q in range(N):  # what r sends to q is what q keeps for r
            assert p.send_start[q + 1] - p.send_start[q] == (plans[q].disp_start[r + 1] - plans[q].disp_start[r] if q != r else 0)
    # scaled to the 512 frames per GPU the bench would pick first: areas and per-link payload are linear in the frame count (poses repeat every 1000)
    worst = max(2 * (p.send_total + p.disp_total) * 256 for p in plans) * (512 / F)
    link = max(p.send_total * 256 / (N - 1) for p in plans) * (512 / F)
    assert worst < 288e9 * 0.9, f"{name}: 512 frames per GPU would need {worst / 1e9:.1f} GB of areas"
    print(f"{name}: {F} frames/GPU -> {predictions[0]['area_bytes_per_gpu'] / 1e9:.2f} GB areas, {predictions[0]['payload_bytes_per_link_per_step'] / 1e6:.1f} MB per link and step; "
          f"at 512 frames/GPU ~{worst / 1e9:.1f} GB areas, ~{link / 1e6:.0f} MB = {link / 153e9 * 1e3:.1f} ms per link and step")
    # the plan the C-ABI exchange uses is the same plan (host arithmetic of libcpuvox_gpu, no device needed)
    ctx_free_pack = gpu.pack_frames(frames) if hasattr(gpu, "pack_frames") else None
    if ctx_free_pack is not None:
        for r in (0, N - 1):
            native = gpu.NativeShardPlan(ctx_free_pack, W, H, r, N)
            assert native.tile_count == plans[r].tile_count and list(native.send_start) == list(plans[r].send_start) and list(native.disp_start) == list(plans[r].disp_start)


def test_multi_gpu_line_explains_itself():
    """the N > 1 bench line carries the step's render alone, its exchange alone, what the overlap hid and what the payload takes on one
    xGMI link (bench.scaling_breakdown); --gather auto takes the gather with fewer bytes (bench.choose_gather).  Host arithmetic: checked here, on
    numbers of the shape an 8-GPU run will produce (render 22 ms, exchange 20 ms, 6.9 GB per GPU and step)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module_breakdown", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b = bench.scaling_breakdown(24.0, 22.0, 20.0, 6.9e9, 8)
    assert b["render_ms_alone"] == 22.0 and b["exchange_ms_alone"] == 20.0
    assert abs(b["overlap_efficiency"] - 0.9) < 1e-9          # 18 of the exchange's 20 ms hidden behind the render
    assert abs(b["payload_ms_per_link_predicted"] - 6.9e9 / 7 / 153e9 * 1e3) < 1e-3
    assert abs(b["exchange_effective_gbps_per_link"] - 6.9e9 / 7 / 20e-3 / 1e9) < 0.01
    assert bench.scaling_breakdown(42.0, 22.0, 20.0, 1e9, 2)["overlap_efficiency"] == 0.0   # serialised
    assert bench.scaling_breakdown(22.0, 22.0, 20.0, 1e9, 2)["overlap_efficiency"] == 1.0   # entirely hidden
    assert bench.choose_gather(44e6, 33e6)["gather"] == "image" and bench.choose_gather(30e6, 33e6)["gather"] == "raybuffer"
