"""-m gpu: BASELINE.json configs 3, 4 and 5 at their FULL sizes, HIP path (through the C ABI) against the CPU oracle.

  config 3  procedural 2048^3 world, 1920x1080, lodError 1 (the bench workload): benchmark-path poses covering both
            element iteration directions and LOD 0, 1 and 2 -- once as single cvx_draw_segments calls, once as ONE
            64-frame cvx_draw_segments_batch launch (the shape bench.py times)
  config 4  procedural 2048^3 world, 3840x2160, vanishing point on screen (12 000 rays)
  config 5  procedural 4096^3 world (full height), 3840x2160, forward.y = +-0.001 (UnityManager.cs:193-201),
            lodError 4 so that LOD 0..4 are visited (DrawSegmentRayJob.cs:237-243)

Bar: bit-exact ARGB32 raybuffers, identical work counters (S, E, C, P, R and the per-LOD column visits).
"""
import numpy as np
import pytest

import oraclelib as O
import scenes
from cpuvox_amd import gpu, host

pytestmark = pytest.mark.gpu

CLEAR = 0xDEADBEEF
POSES = 1000  # bench.py: pose index i -> clip time i / POSES * BENCHMARK_PATH_LENGTH
# indices along the benchmark path: 0..450 look up (inverse iteration, one clamped segment), 500.. look down
# (forward iteration); 550..950 have the vanishing point on or near the screen (up to 4 segments, 6000 rays)
CONFIG3_POSES = [0, 50, 200, 300, 450, 500, 550, 650, 700, 800, 950]


def _path_frame(ws, W, H, index, lods, far):
    t = index / POSES * host.BENCHMARK_PATH_LENGTH
    pos, eul = host.sample_benchmark_path(t, ws.dims)
    return host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1])


def _lods(ws, W, H, lod_error):
    return host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), ws.max_dimension, W, H, lod_error)


def _compare(name, fr, g_td, g_lr, o_td, o_lr):
    n_td, n_lr = scenes.used_rows(fr)
    for label, g, o, n in (("topdown", g_td, o_td, n_td), ("leftright", g_lr, o_lr, n_lr)):
        diff = g[:n] != o[:n]
        if diff.any():
            rows, cols = np.nonzero(diff)
            raise AssertionError(f"{name}/{label}: {diff.sum()} of {diff.size} pixels differ; first at ray {rows[0]} pixel {cols[0]}: "
                                 f"gpu {g[rows[0], cols[0]]:08x} oracle {o[rows[0], cols[0]]:08x}; rays affected {len(set(rows.tolist()))}")
        assert (g[n:] == CLEAR).all(), f"{name}/{label}: rows beyond the used rays were written"


def _render_build_matches(ctx, name, fr, o_td, o_lr):
    """The SHIPPED kernel (render_kernel<false>, counters off: it leaves a finished ray once per column and takes the
    automatic LDS-budget / sub-tile path) drawn into the same buffer again and compared with the same oracle output."""
    ctx.enable_counters(False)
    # ... through the kernel a caller gets (AUTO), and pinned to each of the two (include/cpuvox_gpu.h, cvx_set_latency_kernel): the batch kernel, lanes =
    # rays, and the latency kernel, one wave per ray with its lanes the ray's next 64 columns (4K rows take its two-register mask)
    for label, mode in (("automatic", gpu.LATENCY_AUTO), ("batch kernel", gpu.LATENCY_NEVER), ("latency kernel", gpu.LATENCY_ALWAYS)):
        ctx.set_latency_kernel(mode)
        ctx.clear_raybuffers(0, CLEAR)
        ctx.draw_segments(fr, 0)
        ctx.set_latency_kernel(gpu.LATENCY_AUTO)
        _compare(f"{name} [rendering build, {label}]", fr, ctx.read_raybuffer(0, 0), ctx.read_raybuffer(0, 1), o_td, o_lr)


def _same_counters(name, gc, oc):
    assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (oc.S, oc.E, oc.C, oc.P, oc.R), (name, gc.as_dict(), oc.as_dict())
    assert list(gc.lodVisits) == list(oc.lodVisits), (name, list(gc.lodVisits), list(oc.lodVisits))


@pytest.fixture(scope="module")
def proc2048():
    ws = scenes.load_world("proc2048")
    ctx = gpu.Context(0, buffer_count=64)
    ctx.upload_world(ws)
    yield ws, ctx
    ctx.close()


def test_config3_single_draws(proc2048):
    ws, ctx = proc2048
    W, H = 1920, 1080
    ctx.set_resolution(W, H)
    lods, far = _lods(ws, W, H, 1.0)
    assert lods[:3] == [1176.0, 2351.0, 8192.0] and far == 4096.0  # LOD 0-2 reachable
    seen_lods = np.zeros(6, dtype=np.int64)
    directions = set()
    for index in CONFIG3_POSES:
        fr = _path_frame(ws, W, H, index, lods, far)
        directions.add(int(fr.camera.InverseElementIterationDirection))
        ctx.enable_counters(True)
        ctx.clear_raybuffers(0, CLEAR)
        ctx.draw_segments(fr, 0)
        gc = ctx.counters()
        o_td, o_lr, oc = O.draw_segments(ws, fr, W, H, clear=CLEAR)
        _compare(f"config3 pose {index}", fr, ctx.read_raybuffer(0, 0), ctx.read_raybuffer(0, 1), o_td, o_lr)
        _same_counters(f"config3 pose {index}", gc, oc)
        _render_build_matches(ctx, f"config3 pose {index}", fr, o_td, o_lr)
        seen_lods += np.array(list(oc.lodVisits))
    assert directions == {0, 1}, "both element iteration directions must occur"
    assert (seen_lods[:3] > 0).all() and (seen_lods[3:] == 0).all(), seen_lods


def test_config3_bench_launch_512_frames():
    """exactly the launch bench.py times -- 512 frames of the benchmark path (pose (g * 37) % 1000 for frame g) in ONE draw of
    the shipped build at 1080p, the LDS budget / sub-tile decisions DrawBatch makes for THAT batch -- with 40 frames spread over the launch
    (every 13th: first, last, all four segment shapes of the path) compared with the oracle, every pixel they write."""
    ws = scenes.load_world("proc2048")
    W, H, F = 1920, 1080, 512
    ctx = gpu.Context(0, buffer_count=F)
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        lods, far = _lods(ws, W, H, 1.0)
        frames = [_path_frame(ws, W, H, (g * 37) % POSES, lods, far) for g in range(F)]
        checked = list(range(0, F, 13)) + [F - 1]
        for b in checked:
            ctx.clear_raybuffers(b, CLEAR)
        ctx.enable_counters(False)
        ctx.draw_segments_batch(frames, 0)
        for b in checked:
            o_td, o_lr, _ = O.draw_segments(ws, frames[b], W, H, clear=CLEAR, counters=False)
            _compare(f"bench launch frame {b} (pose {(b * 37) % POSES})", frames[b], ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1), o_td, o_lr)
    finally:
        ctx.close()


def test_config3_one_64_frame_batch(proc2048):
    """The shape bench.py times: 64 frames in ONE launch (no sub-tile split at this size), every frame against the oracle."""
    ws, ctx = proc2048
    W, H = 1920, 1080
    ctx.set_resolution(W, H)
    lods, far = _lods(ws, W, H, 1.0)
    indices = CONFIG3_POSES + [(g * 37) % POSES for g in range(64 - len(CONFIG3_POSES))]  # + the first bench poses (stride 37)
    frames = [_path_frame(ws, W, H, i, lods, far) for i in indices]
    for b in range(64):
        ctx.clear_raybuffers(b, CLEAR)
    ctx.enable_counters(True)
    ctx.draw_segments_batch(frames, 0)
    gc = ctx.counters()
    total = O.OrcCounters()
    oracle_out = []
    for b, fr in enumerate(frames):
        o_td, o_lr, oc = O.draw_segments(ws, fr, W, H, clear=CLEAR)
        oracle_out.append((o_td, o_lr))
        _compare(f"config3 batch frame {b} (pose {indices[b]})", fr, ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1), o_td, o_lr)
        for k in ("S", "E", "C", "P", "R"):
            setattr(total, k, getattr(total, k) + getattr(oc, k))
        for l in range(6):
            total.lodVisits[l] += oc.lodVisits[l]
    _same_counters("config3 batch", gc, total)
    # ... and the same launch by the shipped kernel (counters off)
    ctx.enable_counters(False)
    for b in range(64):
        ctx.clear_raybuffers(b, CLEAR)
    ctx.draw_segments_batch(frames, 0)
    for b, fr in enumerate(frames):
        _compare(f"config3 batch frame {b} [rendering build]", fr, ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1), *oracle_out[b])


def test_config4_4k_vp_on_screen(proc2048):
    ws, ctx = proc2048
    W, H = 3840, 2160
    ctx.set_resolution(W, H)
    lods, far = _lods(ws, W, H, 1.0)
    for index in (650, 800):
        fr = _path_frame(ws, W, H, index, lods, far)
        assert abs(fr.totalRays - 12000) <= 2 and all(s.RayCount > 0 for s in fr.segments)
        ctx.enable_counters(True)
        ctx.clear_raybuffers(0, CLEAR)
        ctx.draw_segments(fr, 0)
        gc = ctx.counters()
        o_td, o_lr, oc = O.draw_segments(ws, fr, W, H, clear=CLEAR)
        _compare(f"config4 pose {index}", fr, ctx.read_raybuffer(0, 0), ctx.read_raybuffer(0, 1), o_td, o_lr)
        _same_counters(f"config4 pose {index}", gc, oc)
        _render_build_matches(ctx, f"config4 pose {index}", fr, o_td, o_lr)


def test_config5_4096_cubed_horizontal_deep_lods():
    ws = scenes.load_world("proc4096")
    assert tuple(ws.dims) == (4096, 4096, 4096)
    W, H = 3840, 2160
    ctx = gpu.Context(0)
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        poses = [((-0.1, 0.5, -0.1), (0.0, 45.0, 0.0)),      # outside the world, forward.y clamped to +0.001: LOD 0..4
                 ((0.4, 0.9, 0.3), (0.01, 200.0, 0.0)),      # inside, forward.y clamped to -0.001, forward iteration
                 ((0.5, 0.6, 0.5), (0.0, 100.0, 0.0)),       # inside, +0.001
                 ((0.3, 0.35, 0.7), (20.0, 310.0, 0.0))]     # VP on screen at 4K on the deep world (12 000 rays)
        seen = np.zeros(6, dtype=np.int64)
        for i, (frac, eul) in enumerate(poses):
            pos = [frac[k] * ws.dims[k] for k in range(3)]
            fr = scenes.make_frame(ws, W, H, pos, eul, lod_error=4.0)
            if i < 3:
                assert abs(abs(fr.forward[1]) - 0.001) < 1e-6
            ctx.enable_counters(True)
            ctx.clear_raybuffers(0, CLEAR)
            ctx.draw_segments(fr, 0)
            gc = ctx.counters()
            o_td, o_lr, oc = O.draw_segments(ws, fr, W, H, clear=CLEAR)
            _compare(f"config5 pose {i}", fr, ctx.read_raybuffer(0, 0), ctx.read_raybuffer(0, 1), o_td, o_lr)
            _same_counters(f"config5 pose {i}", gc, oc)
            _render_build_matches(ctx, f"config5 pose {i}", fr, o_td, o_lr)
            seen += np.array(list(oc.lodVisits))
        assert (seen[:5] > 0).all(), f"LOD 0..4 must be visited: {seen}"
    finally:
        ctx.close()


def test_new_lod0_dimensions_invalidate_the_old_lod_chain():
    """LOD 0 of other dimensions starts a new world; the old LOD 1..5 tables must not be indexed with it."""
    small, big = scenes.load_world("proc256"), scenes.load_world("proc512")
    ctx = gpu.Context(0)
    try:
        ctx.upload_world(small)
        ctx.set_resolution(320, 200)
        fr = scenes.benchmark_frame(big, 320, 200, 0.5)
        i = big.info(0)
        ctx._check(gpu.lib().cvx_world_upload(ctx._h, 0, i.storage, i.byteLength, i.dimX, i.dimY, i.dimZ, i.columnCount))
        with pytest.raises(gpu.CvxError, match="LOD 1 has not been uploaded"):
            ctx.draw_segments(fr, 0)
        ctx.upload_world(big)  # the whole chain: fine again
        ctx.clear_raybuffers(0, CLEAR)
        ctx.draw_segments(fr, 0)
        o_td, o_lr, _ = O.draw_segments(big, fr, 320, 200, clear=CLEAR, counters=False)
        _compare("after re-upload", fr, ctx.read_raybuffer(0, 0), ctx.read_raybuffer(0, 1), o_td, o_lr)
        fresh = gpu.Context(0)
        j = small.info(1)
        with pytest.raises(gpu.CvxError, match="LOD 0 first"):
            fresh._check(gpu.lib().cvx_world_upload(fresh._h, 1, j.storage, j.byteLength, j.dimX, j.dimY, j.dimZ, j.columnCount))
        fresh.close()
    finally:
        ctx.close()


def test_downsample_refuses_a_non_lod0_source():
    """World.DownSample is only ever applied to LOD 0 (UnityManager.cs:328-331); other sources are refused."""
    ws = scenes.load_world("proc256")
    ctx = gpu.Context(0)
    try:
        with pytest.raises(gpu.CvxError, match="LOD 0"):
            ctx.downsample(ws, 1, 1)
    finally:
        ctx.close()


def test_two_rank_rehearsal_line_carries_the_scaling_breakdown():
    """The N > 1 path of bench.py as the driver starts it (`python bench.py --gpus 2`), rehearsed on ONE GPU with gloo: the line must verify its exchange
    and explain itself -- render alone, exchange alone, overlap efficiency, predicted per-link payload time -- and `--gather auto`
    must name the gather it took and the bytes of both."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "4", "--steps", "2", "--warmup", "1", "--gather", "auto"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["exchange_verified"] is True and line["config"]["ranks_seen"] == [0, 1]
    for key in ("render_ms_alone", "exchange_ms_alone", "overlap_efficiency", "payload_ms_per_link_predicted", "exchange_effective_gbps_per_link"):
        assert isinstance(line[key], float) and line[key] >= 0.0, key
    auto = line["config"]["gather_auto"]
    assert line["config"]["gather"] == auto["gather"] and auto["gather"] == ("image" if auto["image_bytes"] < auto["raybuffer_bytes"] else "raybuffer")


def test_library_owned_rccl_communicator_single_rank():
    """cvx_comm_unique_id / cvx_comm_create / cvx_exchange / cvx_comm_destroy with one rank (all a one-GPU box can run): librccl
    is found and initialised by the library, and the exchange of a one-rank plan is a no-op that leaves the display area alone."""
    import torch

    ws = scenes.load_world("proc256")
    W, H = 320, 200
    frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in (0.1, 0.75)]
    ctx = gpu.Context(0)
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        comm = gpu.comm_create(ctx, gpu.comm_unique_id(), 0, 1)
        assert comm
        packed = ctx.pack_batch(frames)
        plan = gpu.NativeShardPlan(packed, W, H, 0, 1)
        assert plan.send_total == 0 and plan.disp_total > 0
        disp = torch.zeros((plan.disp_total, 64), dtype=torch.int32, device="cuda:0")
        send = torch.zeros((1, 64), dtype=torch.int32, device="cuda:0")
        ctx.draw_placed(packed, plan.tile_out(send.data_ptr(), disp.data_ptr()))
        before = disp.clone()
        plan.exchange(ctx, comm, None, send.data_ptr(), disp.data_ptr())
        ctx.synchronize()
        assert torch.equal(before, disp) and int((disp != 0).sum()) > 0
        # the placed render equals the oracle (assembled with the python plan, which the native one matches)
        from cpuvox_amd import dist as cdist

        ref = cdist.ShardPlan(frames, W, H, 0, 1)
        for b, fr in enumerate(frames):
            g_td, g_lr = ref.assemble(disp, b, [s.RayCount for s in fr.segments], W, H)
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False)
            n_td, n_lr = scenes.used_rows(fr)
            assert np.array_equal(g_td[:n_td], o_td[:n_td]) and np.array_equal(g_lr[:n_lr], o_lr[:n_lr])
        gpu.comm_destroy(comm)
    finally:
        ctx.close()
