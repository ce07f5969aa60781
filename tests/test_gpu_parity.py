"""-m gpu: the HIP path (through the C ABI of libcpuvox_gpu.so) against the CPU oracle and the committed
golden fixtures.  Bar: bit-exact ARGB32 raybuffers (integer/byte output), identical work counters."""
import json
import os

import numpy as np
import pytest

import oraclelib as O
import scenes
from cpuvox_amd import gpu, host

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(scenes.GOLDEN, "golden.json")))
CLEAR = 0xDEADBEEF


@pytest.fixture(scope="module")
def contexts():
    cache = {}

    def get(world_name, W, H):
        key = world_name
        if key not in cache:
            ctx = gpu.Context(0)
            ctx.upload_world(scenes.load_world(world_name))
            cache[key] = ctx
        ctx = cache[key]
        ctx.set_resolution(W, H)
        return ctx

    yield get
    for ctx in cache.values():
        ctx.close()


@pytest.fixture
def exp_library(monkeypatch):
    """The experiment build (make gpu-exp, -DCVX_EXPERIMENTS): the only build that reads the diagnostic CVX_* environment switches and exports the
    diagnostics of include/cpuvox_gpu_diag.h (the arithmetic self-test among them).  The product library reads no environment (tests/test_abi.py).
    `make all` / __graft_entry__.build() always build it next to the product library, from the same sources with the same HIPFLAGS, so a missing
    file is a broken build, not a reason to skip: the shipped kernels' arithmetic contract is pinned through this build."""
    path = os.path.join(os.path.dirname(gpu.lib_path()), "libcpuvox_gpu_exp.so")
    assert os.path.exists(path), "libcpuvox_gpu_exp.so not built: run `make -C cpuvox_amd/csrc all` (or __graft_entry__.build())"
    gpu.use_library(path)
    yield monkeypatch
    gpu.use_library(None)


def _render_gpu(ctx, fr, counters=False, latency=gpu.LATENCY_AUTO):
    ctx.enable_counters(counters)
    ctx.set_latency_kernel(latency)
    ctx.clear_raybuffers(0, CLEAR)
    ctx.draw_segments(fr, 0)
    ctx.set_latency_kernel(gpu.LATENCY_AUTO)
    td = ctx.read_raybuffer(0, gpu.RAYBUFFER_TOPDOWN)
    lr = ctx.read_raybuffer(0, gpu.RAYBUFFER_LEFTRIGHT)
    return td, lr


# a draw goes to one of two kernels (include/cpuvox_gpu.h, cvx_set_latency_kernel): the batch kernel (lanes = rays, cvx_kernels.h) or the latency kernel
# (one wave per ray, lanes = columns, cvx_lone.h).  Both must give the oracle's raybuffers bit for bit; AUTO is what a caller gets.
BOTH_KERNELS = [("batch kernel", gpu.LATENCY_NEVER), ("latency kernel", gpu.LATENCY_ALWAYS)]


def _compare(name, fr, g_td, g_lr, o_td, o_lr):
    n_td, n_lr = scenes.used_rows(fr)
    for label, g, o, n in (("topdown", g_td, o_td, n_td), ("leftright", g_lr, o_lr, n_lr)):
        diff = g[:n] != o[:n]
        if diff.any():
            rows, cols = np.nonzero(diff)
            raise AssertionError(f"{name}/{label}: {diff.sum()} of {diff.size} pixels differ; first at ray {rows[0]} pixel {cols[0]}: "
                                 f"gpu {g[rows[0], cols[0]]:08x} oracle {o[rows[0], cols[0]]:08x}; rays affected {len(set(rows.tolist()))}")


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_scene_bit_exact_vs_oracle_and_golden(contexts, name):
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = contexts(scenes.SCENES[name][0], W, H)
    g_td, g_lr = _render_gpu(ctx, fr, counters=True)
    o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
    _compare(name, fr, g_td, g_lr, o_td, o_lr)
    # rows beyond the used rays and pixels outside [origMin, origMax] are never touched
    n_td, n_lr = scenes.used_rows(fr)
    assert (g_td[n_td:] == CLEAR).all() and (g_lr[n_lr:] == CLEAR).all()
    assert ((g_td[:n_td] == CLEAR) == (o_td[:n_td] == CLEAR)).all()
    # instrumented kernel counts the same algorithmic work as the instrumented oracle
    gc = ctx.counters()
    assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (cnt.S, cnt.E, cnt.C, cnt.P, cnt.R), (gc.as_dict(), cnt.as_dict())
    assert list(gc.lodVisits) == list(cnt.lodVisits)
    # golden fixture (oracle output committed from the build container)
    gold = GOLDEN[name]
    zero_td = np.where(g_td[:n_td] == CLEAR, 0, g_td[:n_td]).astype(np.uint32)
    zero_lr = np.where(g_lr[:n_lr] == CLEAR, 0, g_lr[:n_lr]).astype(np.uint32)
    assert scenes.crc(zero_td) == gold["crcTopDown"]
    assert scenes.crc(zero_lr) == gold["crcLeftRight"]
    assert gold["counters"]["S"] == gc.S and gold["counters"]["P"] == gc.P


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_scene_bit_exact_through_both_kernels(contexts, name):
    """Every scene with the counters off (the shipped kernels), once pinned to the batch kernel and once to the latency kernel: the oracle's raybuffers
    bit for bit, nothing outside the rows / pixel windows of the frame touched."""
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = contexts(scenes.SCENES[name][0], W, H)
    o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR, counters=False)
    n_td, n_lr = scenes.used_rows(fr)
    for label, mode in BOTH_KERNELS + [("automatic choice", gpu.LATENCY_AUTO)]:
        g_td, g_lr = _render_gpu(ctx, fr, latency=mode)
        _compare(f"{name} [{label}]", fr, g_td, g_lr, o_td, o_lr)
        assert (g_td[n_td:] == CLEAR).all() and (g_lr[n_lr:] == CLEAR).all(), label
        assert ((g_td[:n_td] == CLEAR) == (o_td[:n_td] == CLEAR)).all() and ((g_lr[:n_lr] == CLEAR) == (o_lr[:n_lr] == CLEAR)).all(), label


def test_counters_off_gives_same_pixels(contexts):
    name = "proc256_t04_lod8"
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = contexts(scenes.SCENES[name][0], W, H)
    a = _render_gpu(ctx, fr, counters=True)
    b = _render_gpu(ctx, fr, counters=False)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()


def test_random_poses_fuzz(contexts):
    """Seeded random cameras in and around a procedural world, random lodError: GPU == oracle bit for bit."""
    rng = np.random.default_rng(20241115)
    ws = scenes.load_world("proc256")
    W, H = 320, 200
    ctx = contexts("proc256", W, H)
    for i in range(40):
        frac = rng.uniform(-0.3, 1.3, size=3)
        frac[1] = rng.uniform(0.05, 1.2)
        pos = [frac[k] * ws.dims[k] for k in range(3)]
        eul = [rng.uniform(-89, 89), rng.uniform(0, 360), rng.choice([0.0, 0.0, rng.uniform(0, 360)])]
        fr = scenes.make_frame(ws, W, H, pos, eul, lod_error=float(rng.choice([1.0, 3.0, 9.0])))
        o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR, counters=False)
        for label, mode in BOTH_KERNELS:
            g_td, g_lr = _render_gpu(ctx, fr, latency=mode)
            _compare(f"fuzz{i} [{label}] pos={pos} eul={eul}", fr, g_td, g_lr, o_td, o_lr)


def test_random_poses_fuzz_mill_and_odd_resolutions(contexts):
    """Sparse model world (empty columns, StepToWorldIntersection from outside) at non-multiple-of-32/64 resolutions."""
    rng = np.random.default_rng(7731)
    ws = scenes.load_world("mill256")
    for W, H in ((333, 217), (640, 360), (97, 401)):
        ctx = contexts("mill256", W, H)
        for i in range(12):
            frac = rng.uniform(-0.5, 1.5, size=3)
            pos = [frac[k] * ws.dims[k] for k in range(3)]
            eul = [rng.uniform(-89, 89), rng.uniform(0, 360), rng.choice([0.0, rng.uniform(0, 360)])]
            fr = scenes.make_frame(ws, W, H, pos, eul)
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR, counters=False)
            for label, mode in BOTH_KERNELS:
                g_td, g_lr = _render_gpu(ctx, fr, latency=mode)
                _compare(f"millfuzz {W}x{H} #{i} [{label}] pos={pos} eul={eul}", fr, g_td, g_lr, o_td, o_lr)


def test_latency_kernel_wide_windows_fuzz(contexts):
    """The latency kernel keeps a ray's seen mask in ONE vector register (windows of up to 2048 pixels) or TWO (up to 4096: lone_kernel<true>).  Random poses at
    resolutions whose pixel windows end just below / above 2048 pixels and at odd offsets inside a mask word, both kernels against the oracle."""
    rng = np.random.default_rng(60606)
    ws = scenes.load_world("proc256")
    for W, H in ((2047, 1031), (2049, 2050), (2500, 2113), (4096, 1500)):
        ctx = contexts("proc256", W, H)
        for i in range(4):
            frac = rng.uniform(-0.2, 1.2, size=3)
            frac[1] = rng.uniform(0.1, 1.1)
            pos = [frac[k] * ws.dims[k] for k in range(3)]
            eul = [rng.uniform(-89, 89), rng.uniform(0, 360), rng.choice([0.0, rng.uniform(0, 360)])]
            fr = scenes.make_frame(ws, W, H, pos, eul, lod_error=float(rng.choice([1.0, 6.0])))
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR, counters=False)
            for label, mode in BOTH_KERNELS:
                g_td, g_lr = _render_gpu(ctx, fr, latency=mode)
                _compare(f"wide {W}x{H} #{i} [{label}] pos={pos} eul={eul}", fr, g_td, g_lr, o_td, o_lr)


def test_batch_equals_single_frames(contexts):
    """cvx_draw_segments_batch (many frames per launch, mixed iteration directions) == frame-by-frame draws."""
    ws = scenes.load_world("proc256")
    W, H = 320, 200
    ctx = gpu.Context(0, buffer_count=6)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in (0.0, 0.3, 0.55, 0.75, 0.9, 1.1)]
    for b in range(6):
        ctx.clear_raybuffers(b, CLEAR)
    ctx.draw_segments_batch(frames, 0)
    for b, fr in enumerate(frames):
        o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR, counters=False)
        _compare(f"batch{b}", fr, ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1), o_td, o_lr)
    ctx.close()


def test_sharded_render_covers_every_tile_once(contexts):
    """cvx_set_shard: the union of the shards' tiles equals the unsharded raybuffer, shards are disjoint."""
    name = "proc256_t075_lod8"
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = contexts(scenes.SCENES[name][0], W, H)
    full = _render_gpu(ctx, fr)
    acc = [np.full_like(full[0], CLEAR), np.full_like(full[1], CLEAR)]
    for shard in range(3):
        ctx.set_shard(shard, 3)
        part = _render_gpu(ctx, fr)
        for k in range(2):
            written = part[k] != CLEAR
            assert (acc[k][written] == CLEAR).all(), "tile rendered by two shards"
            acc[k][written] = part[k][written]
    ctx.set_shard(0, 1)
    assert (acc[0] == full[0]).all() and (acc[1] == full[1]).all()


def test_blit_matches_pixel_centre_rule(contexts):
    for name in ("mill256_t075", "mill256_t09_roll", "proc256_t0_lod8"):
        ws, fr, W, H = scenes.scene_frame(name)
        ctx = contexts(scenes.SCENES[name][0], W, H)
        g_td, g_lr = _render_gpu(ctx, fr)
        img = ctx.blit_segments(0)
        ref = O.blit_reference(fr, g_td, g_lr, W, H, clear=0)
        assert (img == ref).all(), f"{name}: {(img != ref).sum()} screen pixels differ"
        # ... and an independent check (float64 barycentrics, none of the kernel's edge-function arithmetic): the images may differ only where a
        # weight or a ray coordinate lies within rounding distance of a boundary, and those pixels are a sliver of the screen
        ref64, margin = O.blit_reference_f64(fr, g_td, g_lr, W, H, clear=0)
        differ = img != ref64
        assert not (differ & (margin > 1e-4)).any(), f"{name}: {(differ & (margin > 1e-4)).sum()} pixels away from every boundary differ from the float64 rule"
        assert differ.mean() < 2e-3, f"{name}: {differ.sum()} pixels differ from the float64 rule"


def test_batch_blit_equals_single_blits():
    """cvx_blit_segments_batch (Phase 2 of a whole batch in one launch, images left on the device) == cvx_blit_segments frame by frame,
    both into a caller's device buffer (a torch tensor) and into the array the context owns."""
    import torch

    names = ["proc256_t0_lod8", "proc256_t04_lod8", "proc256_t075_lod8", "proc256_t075_lod1"]  # one world, one resolution
    frames = []
    for n in names:
        ws, fr, W, H = scenes.scene_frame(n)
        frames.append(fr)
    ctx = gpu.Context(0, buffer_count=len(frames) + 1)
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        for b in range(len(frames) + 1):
            ctx.clear_raybuffers(b, 0)
        ctx.draw_segments_batch(frames, 1)  # buffers 1 .. n
        singles = [ctx.blit_segments(1 + i) for i in range(len(frames))]
        dst = torch.zeros((len(frames), H, W), dtype=torch.int32, device="cuda:0")
        p = ctx.blit_segments_batch(1, len(frames), dst.data_ptr())
        assert p == dst.data_ptr()
        ctx.synchronize()
        got = dst.cpu().numpy().view(np.uint32)
        for i, n in enumerate(names):
            assert (got[i] == singles[i]).all(), f"{n}: batch blit differs from the single blit in {(got[i] != singles[i]).sum()} pixels"
            td = ctx.read_raybuffer(1 + i, gpu.RAYBUFFER_TOPDOWN)
            lr = ctx.read_raybuffer(1 + i, gpu.RAYBUFFER_LEFTRIGHT)
            assert (got[i] == O.blit_reference(frames[i], td, lr, W, H, clear=0)).all(), n
        # context-owned image array: same pixels (read back through a torch view of the returned address is not possible, so blit a
        # sub-range twice and compare the two device arrays on the device)
        own = ctx.blit_segments_batch(2, 2)
        assert own and own != dst.data_ptr()
        with pytest.raises(RuntimeError):
            ctx.blit_segments_batch(1, len(frames) + 1)  # past the last buffer
        with pytest.raises(RuntimeError):
            ctx.blit_segments_batch(0, 1)                # buffer 0 was never drawn into
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["mill256", "proc256", "proc128x512x64", "proc64x4096x32"])  # the last: occupied spans taller than one LDS chunk
def test_downsample_matches_host_build(contexts, name):
    """cvx_world_downsample (World.DownSample on the device) against the host build of the same level: the storage blobs
    (headers, guards, runs, averaged colours, element offsets) must be byte-identical for every LOD the reference builds."""
    ws = scenes.load_world(name)
    ctx = contexts(name, 320, 200)
    for extra in range(1, ws.lod_count):
        blob, columns, voxels, ms = ctx.downsample(ws, 0, extra)
        want = ws.storage(extra)
        info = ws.info(extra)
        assert columns == info.columnCount, f"{name} LOD {extra}: ColumnCount {columns} vs {info.columnCount}"
        got = np.frombuffer(blob, dtype=np.uint8)
        assert got.size == want.size, f"{name} LOD {extra}: {got.size} bytes vs {want.size}"
        diff = np.flatnonzero(got != want)
        assert diff.size == 0, f"{name} LOD {extra}: first differing byte at {diff[0]} of {want.size} ({diff.size} differ)"
        assert voxels > 0 and ms >= 0.0


def test_world_with_device_built_lods_renders_identically(contexts):
    """LOD 1..5 from cvx_world_downsample, assembled with cvxh_world_from_blobs, uploaded and rendered: same raybuffers as the
    host-built chain (lodError 8 reaches the higher levels)."""
    name = "proc256_t075_lod8"
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = contexts(scenes.SCENES[name][0], W, H)
    rebuilt = ctx.build_lods(ws)
    assert rebuilt.lod_count == ws.lod_count
    for lod in range(ws.lod_count):
        assert np.array_equal(rebuilt.storage(lod), ws.storage(lod)), f"LOD {lod} blob differs"
    ctx.upload_world(rebuilt)
    try:
        g_td, g_lr = _render_gpu(ctx, fr)
        o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR, counters=False)
        _compare(name, fr, g_td, g_lr, o_td, o_lr)
    finally:
        ctx.upload_world(ws)


def _random_alpha_world(dims, seed, columns, segments):
    """Columns of several vertical voxel segments (multi-run columns, segments that cross bucket boundaries of every level) with a random ARGB --
    random ALPHA included -- per voxel: what the reference keeps of it at LOD j is the alpha of the voxel it inserts first (World.cs:85-94,101-127,
    WordBuilder.cs:199-214), which the one-pass LOD chain has to carry through its sums (cvx_downsample.h, SumVoxel)."""
    rng = np.random.default_rng(seed)
    xs, ys, zs = [], [], []
    for _ in range(columns):
        x, z = int(rng.integers(0, dims[0])), int(rng.integers(0, dims[2]))
        for _ in range(int(rng.integers(1, segments + 1))):
            lo = int(rng.integers(0, dims[1]))
            hi = min(dims[1] - 1, lo + int(rng.integers(0, 40)))
            ys.extend(range(lo, hi + 1))
            xs.extend([x] * (hi - lo + 1))
            zs.extend([z] * (hi - lo + 1))
    n = len(xs)
    argb = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    return np.array(xs, np.int32), np.array(ys, np.int32), np.array(zs, np.int32), argb


@pytest.mark.parametrize("dims,seed,columns,segments", [((64, 256, 64), 11, 2500, 4), ((32, 1024, 128), 12, 1500, 6), ((128, 64, 32), 13, 3000, 2)])
def test_lod_chain_reads_lod0_once_and_matches_the_host_build(dims, seed, columns, segments):
    """cvx_world_build_lods (round 5: level 1 from LOD 0, every further level from the sums of the one before it) against the host build of the same
    chain -- which tests/test_world_model.py pins to an independent Python model -- on worlds whose voxels all differ in alpha: byte-identical
    blobs for LOD 1 .. 5, i.e. integer averages from exact sums AND the reference's first-inserted alpha at every level."""
    x, y, z, argb = _random_alpha_world(dims, seed, columns, segments)
    ws = host.WorldSet.from_voxels(dims, x, y, z, argb, threads=4)
    assert ws.lod_count == 6
    ctx = gpu.Context(0)
    try:
        rebuilt = ctx.build_lods(ws)
        assert rebuilt.lod_count == 6
        for lod in range(1, 6):
            got, want = rebuilt.storage(lod), ws.storage(lod)
            assert got.size == want.size, f"LOD {lod}: {got.size} bytes vs {want.size}"
            diff = np.flatnonzero(got != want)
            assert diff.size == 0, f"LOD {lod}: first differing byte at {diff[0]} of {want.size} ({diff.size} differ)"
            # the single-level entry point (LOD 0 -> LOD j directly, the kernels of rounds 1-4) agrees as well
            blob, columns_, voxels, ms = ctx.downsample(ws, 0, lod)
            assert np.array_equal(np.frombuffer(blob, dtype=np.uint8), want), f"cvx_world_downsample LOD {lod}"
    finally:
        ctx.close()


def test_lod_chain_of_a_blob_whose_columns_share_one_pool_region():
    """the chain sized its pools from the SIZE of the LOD 0 pool; a blob whose headers share one storageOffset (the reference's loader
    accepts it, and so does cvx_world_upload) has a tiny pool and needs big levels -- the write pass then ran past its buffers.  The bound now counts
    every column by itself.  Every column of this world IS one column (all headers point at the same run list); the chain must give the bytes of the
    level-by-level entry point (which allocates from the scanned totals) for every level."""
    dims = (64, 128, 64)
    x, y, z, argb = _random_alpha_world(dims, 31, 600, 4)
    base = host.WorldSet.from_voxels(dims, x, y, z, argb, threads=4)
    blob = np.array(base.storage(0), dtype=np.uint8, copy=True)
    columns = dims[0] * dims[2]
    header = np.dtype({"names": ["off", "rc", "mn", "mx"], "formats": ["<i4", "<u2", "<u2", "<u2"], "offsets": [0, 4, 6, 8], "itemsize": 12})
    hdr = blob[:columns * 12].view(header)
    el = blob[columns * 12:].view(np.dtype([("ci", "<i2"), ("len", "<i2")]))
    tallest = int(np.argmax(hdr["rc"]))
    h = hdr[tallest].copy()
    runs = el[h["off"] + 1: h["off"] + 1 + h["rc"]]
    colours = int(max((r["ci"] + r["len"] for r in runs if r["ci"] >= 0), default=0))
    block = int(h["rc"]) + 2 + colours  # guard, runs, guard, colours
    assert h["rc"] >= 3 and colours >= 20
    pool = np.array(el[h["off"]: h["off"] + block], copy=True)
    aliased = np.zeros(columns * 12 + block * 4, dtype=np.uint8)
    ah = aliased[:columns * 12].view(header)
    ah["off"], ah["rc"], ah["mn"], ah["mx"] = 0, h["rc"], h["mn"], h["mx"]  # 4096 headers, ONE pool region of `block` entries
    aliased[columns * 12:] = pool.view(np.uint8)
    assert columns * block > 4 * (block + columns), "the old bound (pool + columns) must be far too small for level 1"
    ws = host.WorldSet.from_blobs(dims, [aliased])
    ctx = gpu.Context(0)
    try:
        rebuilt = ctx.build_lods(ws, levels=6)
        for lod in range(1, 6):
            want, columns_, voxels, ms = ctx.downsample(ws, 0, lod)
            got = rebuilt.storage(lod)
            assert np.array_equal(got, np.frombuffer(want, dtype=np.uint8)), f"LOD {lod}: chain and level-by-level build differ"
            assert voxels > 0
    finally:
        ctx.close()


def test_lod_chain_deeper_than_its_sums_reach_and_its_fallback(exp_library):
    """cvx_world_build_lods chains levels 1 .. 7 (the channel sums of deeper levels do not fit 32 bits) and builds the rest from LOD 0 directly;
    when the chain's pools do not fit the device it builds every level that way.  Both hand-overs against the level-by-level entry point: a 256^3 world
    with all 8 levels, and the same with the chain made to fail (CVX_LOD_CHAIN_FAILS, experiment build)."""
    ws = scenes.load_world("proc256")
    blobs = {}
    for fails in (False, True):
        if fails:
            exp_library.setenv("CVX_LOD_CHAIN_FAILS", "1")
        ctx = gpu.Context(0)
        try:
            rebuilt = ctx.build_lods(ws, levels=9)  # LOD 1 .. 8 (256 >> 8 = 1 column)
            assert rebuilt.lod_count == 9
            for lod in range(1, 9):
                got = np.array(rebuilt.storage(lod), copy=True)
                if not fails:
                    want, columns_, voxels, ms = ctx.downsample(ws, 0, lod)
                    assert np.array_equal(got, np.frombuffer(want, dtype=np.uint8)), f"LOD {lod}: chain and level-by-level build differ"
                    blobs[lod] = got
                else:
                    assert np.array_equal(got, blobs[lod]), f"LOD {lod}: the fallback differs from the chain"
        finally:
            ctx.close()
    exp_library.delenv("CVX_LOD_CHAIN_FAILS")


@pytest.mark.parametrize("split", [1, 2, 16, 64])
def test_sub_tile_split_is_invisible(split, exp_library):
    """Small batches are rendered with tiles cut into sub-tiles of 64 / split rays per wave (DrawBatch); the raybuffers and the
    counters must not depend on the cut (CVX_TILE_SPLIT pins the factor; the other tests run with the automatic choice)."""
    exp_library.setenv("CVX_TILE_SPLIT", str(split))
    ctx = gpu.Context(0)
    try:
        for name in ("mill256_t075", "proc256_t04_lod8"):
            ws, fr, W, H = scenes.scene_frame(name)
            ctx.upload_world(ws)
            ctx.set_resolution(W, H)
            g_td, g_lr = _render_gpu(ctx, fr, counters=True)
            c = ctx.counters()
            o_td, o_lr, oc = O.draw_segments(ws, fr, W, H, clear=CLEAR)
            _compare(f"{name} split {split}", fr, g_td, g_lr, o_td, o_lr)
            assert (c.S, c.E, c.C, c.P, c.R) == (oc.S, oc.E, oc.C, oc.P, oc.R)
    finally:
        ctx.close()


@pytest.mark.parametrize("order", ["reverse", "random", "pixels8", "middle"])
def test_launch_order_is_invisible(order, exp_library):
    monkeypatch = exp_library
    """The tiles of a batch are launched longest-first by an estimate (column visits of the tile's longer edge ray, EstimateTileCost); the estimate and
    the order are scheduling only -- every tile writes its own rows -- so any order gives the same raybuffers."""
    if order.startswith("pixels"):
        monkeypatch.setenv("CVX_TILE_COST_PIXELS", order[len("pixels"):])
    elif order == "middle":
        monkeypatch.setenv("CVX_TILE_COST_MIDDLE_RAY", "1")
    else:
        monkeypatch.setenv("CVX_TILE_ORDER", order)
    names = ["proc256_t0_lod8", "proc256_t04_lod8", "proc256_t075_lod8", "proc256_t075_lod1"]  # one world, one resolution
    frames = [scenes.scene_frame(n)[1] for n in names]
    ws, _, W, H = scenes.scene_frame(names[0])
    ctx = gpu.Context(0, buffer_count=len(frames))
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        for b in range(len(frames)):
            ctx.clear_raybuffers(b, CLEAR)
        ctx.draw_segments_batch(frames, 0)
        for b, (n, fr) in enumerate(zip(names, frames)):
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=CLEAR)
            _compare(f"{n} order {order}", fr, ctx.read_raybuffer(b, gpu.RAYBUFFER_TOPDOWN), ctx.read_raybuffer(b, gpu.RAYBUFFER_LEFTRIGHT), o_td, o_lr)
    finally:
        ctx.close()


def test_cpp_example_on_the_c_abis(tmp_path):
    """examples/flythrough.cpp: world building, camera, the RenderManager twin and the GPU library used from plain C++ through
    the two C ABIs (no Python in the loop); the image it writes for path key t = 0 equals the Python-driven render of the same pose."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    built = subprocess.run(["make", "-C", os.path.join(root, "cpuvox_amd", "csrc"), "examples"], capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("the example could not be built here: " + built.stderr[-300:])
    W, H = 320, 200
    prefix = str(tmp_path / "fly")
    out = subprocess.run([os.path.join(root, "cpuvox_amd", "flythrough"), "proc:128", "3", str(W), str(H), prefix],
                         check=True, capture_output=True, text=True, timeout=300).stdout
    assert "3 frames at 320x200" in out and "fps" in out, out
    with open(prefix + "_0.ppm", "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"320 200\n" and f.readline() == b"255\n"
        rgb = np.frombuffer(f.read(), dtype=np.uint8).reshape(H, W, 3)
    from cpuvox_amd import host
    from cpuvox_amd.render_manager import RenderManager

    ws = host.WorldSet.procedural(128, 128, 128)
    rm = RenderManager(W, H)
    rm.upload_world(ws)
    pos, eul = host.sample_benchmark_path(0.0, ws.dims)
    lods, far = host.setup_lods(host.camera_pose(pos, eul, W, H), ws.max_dimension, W, H, 1.0)
    rm.swap_buffers()
    img = rm.draw_world(host.camera_pose(pos, eul, W, H), lods, far)
    want = np.stack([(img >> 8) & 255, (img >> 16) & 255, img >> 24], axis=-1).astype(np.uint8)[::-1]
    assert np.array_equal(rgb, want)


@pytest.fixture
def diag_context(exp_library):
    """A context of the experiment build: cvx_selftest_math (include/cpuvox_gpu_diag.h) is not in the product library.  Same sources, same
    compiler flags, same device functions (cvx_kernels.h) as the product build."""
    ctx = gpu.Context(0)
    yield ctx
    ctx.close()


def test_device_float_contract(diag_context):
    """IEEE binary32 on the device: correctly rounded / and sqrt, no contraction, denormals kept, x86 (int) rule."""
    ctx = diag_context
    rng = np.random.default_rng(7)
    n = 1 << 16
    a = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1.4e-45, -1.4e-45, 1e-40, 3.4e38, 2147483648.0, -2147483648.0,
                        2147483520.0, -2147483904.0, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 1e10, -1e10], dtype=np.float32)
    a[: special.size] = special
    b[: special.size] = special[::-1]
    with np.errstate(all="ignore"):
        assert np.array_equal(ctx.selftest_math(0, a, b).view(np.uint32), (a / b).view(np.uint32))
        pa = np.abs(a)
        assert np.array_equal(ctx.selftest_math(1, pa, b).view(np.uint32), np.sqrt(pa).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(2, pa, b).view(np.uint32), (np.float32(1.0) / np.sqrt(pa)).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(3, a, b).view(np.uint32), (a + b * (b - a)).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(4, a, b).view(np.uint32), np.floor(a).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(5, a, b).view(np.uint32), np.ceil(a).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(6, a, b).view(np.uint32), np.rint(a).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(8, a, b).view(np.uint32), (a * b).view(np.uint32))
        assert np.array_equal(ctx.selftest_math(9, a, b).view(np.uint32), (a + b).view(np.uint32))
        oor = np.isnan(a) | (a >= np.float32(2147483648.0)) | (a < np.float32(-2147483648.0))
        want = np.where(oor, np.int64(-2147483648), np.trunc(np.where(oor, 0, a)).astype(np.int64)).astype(np.int32)
        assert np.array_equal(ctx.selftest_math(7, a, b).view(np.int32), want)


def _float_soup(rng, n):
    """Random binary32 values over the whole exponent range, plus every special the kernel's guards have to route."""
    bits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32).copy()
    # a third of the values in the range the renderer actually produces (pixels, depths, run lengths)
    k = n // 3
    x[:k] = (rng.standard_normal(k) * 10.0 ** rng.uniform(-6, 6, k)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1.4e-45, -1.4e-45, 1e-40, 3.4e38, -3.4e38, 2.0 ** -30, 2.0 ** 30,
                        np.nextafter(np.float32(2.0 ** -30), np.float32(0)), np.nextafter(np.float32(2.0 ** 30), np.float32(np.inf)),
                        2147483648.0, -2147483648.0, 2147483520.0, -2147483904.0, 0.5, -0.5, 1.9999999, 16777216.0, 65535.0, 0.499, 16385.6],
                       dtype=np.float32)
    x[k:k + special.size] = special
    return x


def test_device_scan_tails_and_many_chunks(diag_context):
    """the three-launch prefix sum of cvx_world_downsample (cvx_downsample.h) against numpy on lengths the power-of-two worlds of the
    other tests never produce: a ragged last chunk (`first + i < n`, `at < n`), a single short chunk, exactly one chunk, and more than 256 chunks
    of 4096 (the carry loop of scan_chunk_offsets_kernel runs more than once: a 4096^2 world's LOD 1 has 1024)."""
    ctx = diag_context
    rng = np.random.default_rng(99)
    for n in (1, 5, 4095, 4096, 4097, 6800, 256 * 4096, 256 * 4096 + 17, 3_000_001):
        values = rng.integers(0, 70, size=n, dtype=np.uint32)
        got, total = ctx.selftest_scan(values)
        want = np.concatenate(([0], np.cumsum(values[:-1], dtype=np.uint64))).astype(np.uint64)
        assert total == int(values.sum(dtype=np.uint64)), n
        assert np.array_equal(got, (want & 0xFFFFFFFF).astype(np.uint32)), n
    # sums beyond 2^32: the total is exact in 64 bits (callers refuse > 2^31 elements before they use the 32-bit offsets)
    big = np.full(70_000, 0xFFFF_0000, dtype=np.uint32)
    got, total = ctx.selftest_scan(big)
    assert total == 70_000 * 0xFFFF_0000


def test_short_division_is_ieee_division(diag_context):
    """quot_safe / recip_safe (the division without v_div_scale / v_div_fmas / v_div_fixup that the side-face and frustum blocks use
    for operands in [2^-30, 2^30]) against IEEE division on 2^28 operand pairs -- random bit patterns, renderer-range values, all
    specials, all-ones mantissas (the hard case of a Newton-refined reciprocal) -- and the helpers' other contracts."""
    ctx = diag_context
    rng = np.random.default_rng(2026)
    n = 1 << 24
    hard = ((rng.integers(0, 254, 4096).astype(np.uint32) << 23) | np.uint32(0x7FFFFF)).view(np.float32)  # 1.11...1 x 2^e
    for chunk in range(16):
        a, b = _float_soup(rng, n), _float_soup(rng, n)
        if chunk % 2 == 0:  # both operands inside the short form's range: it is the path under test
            a = (rng.uniform(-1, 1, n) * 2.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
            b = (rng.uniform(-1, 1, n) * 2.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
        b[-4096:] = hard * np.float32(2.0 ** -100) if chunk < 8 else hard
        a[-8192:-4096] = hard
        with np.errstate(all="ignore"):
            want = (a / b).view(np.uint32)
            got = ctx.selftest_math(10, a, b).view(np.uint32)
            nan = np.isnan(a / b)
            assert np.array_equal(got[~nan], want[~nan]) and np.isnan(got.view(np.float32)[nan]).all(), f"chunk {chunk}: {(got != want).sum()} quotients differ"
            want = (np.float32(1.0) / b).view(np.uint32)
            got = ctx.selftest_math(11, a, b).view(np.uint32)
            nan = np.isnan(b)
            assert np.array_equal(got[~nan], want[~nan]), f"chunk {chunk}: reciprocals differ"
    a, b = _float_soup(rng, 1 << 20), _float_soup(rng, 1 << 20)
    with np.errstate(all="ignore"):
        # f2i_floor == (int)floorf with the x86 rule
        fl = np.floor(a)
        oor = np.isnan(fl) | (fl >= np.float32(2147483648.0)) | (fl < np.float32(-2147483648.0))
        want = np.where(oor, np.int64(-2147483648), np.where(oor, 0, fl).astype(np.int64)).astype(np.int32)
        assert np.array_equal(ctx.selftest_math(12, a, b).view(np.int32), want)
        # v_min_f32 / v_max_f32 == Unity's min / max except for the sign of a zero result and NaN payloads
        for op, ref in ((13, lambda x, y: np.where(np.isnan(y) | (x < y), x, y)), (14, lambda x, y: np.where(np.isnan(y) | (x > y), x, y))):
            got, want = ctx.selftest_math(op, a, b), ref(a, b)
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)) | ((got == 0) & (want == 0))
            # (a SIGNALLING NaN operand comes back quieted instead of being dropped; arithmetic never produces one and the
            # library rejects non-finite inputs, so the kernel cannot meet one)
            snan = lambda v: np.isnan(v) & ((v.view(np.uint32) & np.uint32(0x00400000)) == 0)
            assert (same | snan(a) | snan(b)).all()
        # div_safe: magnitude in [2^-30, 2^30]
        mag = np.abs(a)
        assert np.array_equal(ctx.selftest_math(15, a, b) == 1.0, (mag >= np.float32(2.0 ** -30)) & (mag <= np.float32(2.0 ** 30)))


def test_latency_kernel_assembly_primitives(diag_context):
    """The two inline-assembly pieces of the latency kernel (csrc/cvx_lone.h) on their own, one wavefront per case:
    the DDA's crossing chains -- 63 `v_add_f32 wave_shr:1` per chain, lane n = n of the reference's rounded additions (SegmentDDAData.cs:135-150) --
    against the additions done one after the other in binary32, and `v_writelane` with its lane select in M0 against an array assignment.  The
    scenes cover both through the raybuffers; here the wait states the compiler cannot see are pinned by themselves."""
    ctx = diag_context
    rng = np.random.default_rng(11)
    waves = 4096
    start = (rng.random(waves, dtype=np.float32) * np.float32(10.0) ** rng.integers(-6, 6, waves).astype(np.float32)).astype(np.float32)
    step = (rng.random(waves, dtype=np.float32) * np.float32(10.0) ** rng.integers(-7, 7, waves).astype(np.float32) + np.float32(1e-7)).astype(np.float32)
    start[:4] = np.array([0.0, 1.4e-45, 3.0e38, 0.33333334], dtype=np.float32)   # zero, a denormal, near overflow, an ordinary fraction
    step[:4] = np.array([1.4e-45, 1.4e-45, 3.0e37, 10000000.0], dtype=np.float32)  # (step 1 / 1e-7: the DDA's largest tDelta)
    got = ctx.selftest_lone(0, start, step).reshape(waves, 2, 64)
    want = np.empty_like(got)
    x, z = start.copy(), start.copy()
    with np.errstate(over="ignore"):
        for n in range(64):
            want[:, 0, n], want[:, 1, n] = x, z
            x = (x + step).astype(np.float32)
            z = (z - step).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    lanes = rng.standard_normal(waves * 64).astype(np.float32)
    values = rng.standard_normal(waves).astype(np.float32)
    got = ctx.selftest_lone(1, lanes, values).reshape(waves, 64)
    want = lanes.reshape(waves, 64).copy()
    want[np.arange(waves), np.arange(waves) % 64] = values
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cheap_texture_row_is_certified(diag_context):
    """Round 5: the side pixels take their texture row from a cheap computation (hardware reciprocals, fused multiply-adds) wherever a bound on its
    distance from the reference's value proves the floor equal (`tex_row_cheap`, cvx_kernels.h); the others take the reference's two IEEE divisions.
    The proof is on paper; this is the experiment: 2^26 samples on the device -- renderer-shaped ones (screen bounds, 1 / z and u / z of two ends, both
    orders, thin and tall runs, ends close to the near plane), their extremes, and float soup with every special value -- and NOT ONE certain row may
    differ from the exact one.  Renderer-shaped samples must be certain almost always (the shortcut is worth its instructions)."""
    ctx = diag_context
    rng = np.random.default_rng(516)
    groups = 1 << 20
    for chunk in range(16):
        if chunk < 10:
            span = np.exp(rng.uniform(np.log(1e-4 if chunk >= 6 else 1e-2), np.log(3000.0), groups))
            bx = rng.uniform(-200.0, 2300.0, groups)
            by = bx + span
            y = np.rint(rng.uniform(bx - 1.0, by + 1.0)).clip(-2, 16385)
            zb = np.exp(rng.uniform(np.log(0.06 if chunk >= 6 else 0.5), np.log(6000.0), groups))
            zt = zb * np.exp(rng.uniform(-1.5, 1.5, groups)) if chunk % 2 else np.exp(rng.uniform(np.log(0.06), np.log(6000.0), groups))
            ua = np.rint(np.exp(rng.uniform(0.0, np.log(3000.0), groups)))
            uvax, uvay, uvbx, uvby = 1.0 / zb, ua / zb, 1.0 / zt, np.zeros(groups)
            if chunk % 3 == 0:  # the swapped order (:496-499)
                uvax, uvbx, uvay, uvby = uvbx, uvax, uvby, uvay
            cols = [y, bx, by, uvax, uvbx, uvay, uvby, np.zeros(groups)]
            cols = [np.asarray(c, dtype=np.float32) for c in cols]
        else:
            cols = [_float_soup(rng, groups) for _ in range(8)]
            cols[0] = np.rint(rng.uniform(-2, 16385, groups)).astype(np.float32)
            if chunk >= 13:  # soup in the bounds only / in the texture coordinates only
                keep = slice(1, 3) if chunk == 13 else slice(3, 7)
                real = [np.asarray(c, dtype=np.float32) for c in (rng.uniform(0, 1080, groups), rng.uniform(0, 1080, groups) + 1081, 1.0 / rng.uniform(1, 900, groups),
                                                                  1.0 / rng.uniform(1, 900, groups), rng.uniform(1, 60, groups) / rng.uniform(1, 900, groups), np.zeros(groups))]
                for k in range(1, 7):
                    if not (keep.start <= k < keep.stop):
                        cols[k] = real[k - 1]
        a = np.stack(cols[:4], axis=1).reshape(-1)
        b = np.stack(cols[4:], axis=1).reshape(-1)
        out = ctx.selftest_math(16, a, b).view(np.int32).reshape(-1, 4)
        exact, cheap, certain = out[:, 0], out[:, 1], out[:, 2] == 1
        wrong = certain & (exact != cheap)
        assert not wrong.any(), f"chunk {chunk}: {wrong.sum()} certain rows differ, e.g. inputs {[c[np.flatnonzero(wrong)[0]] for c in cols]} exact {exact[wrong][0]} cheap {cheap[wrong][0]}"
        if chunk < 6:
            assert certain.mean() > 0.97, f"chunk {chunk}: only {certain.mean():.4f} of the renderer-shaped rows are certain"


def test_run_rich_world_slow_paths(contexts):
    """the paths the ordinary scenes rarely take, forced.  World `stripes128x256x128`: every column a stack of 8 .. 30 solid
    bands (~30 RLE elements per column against ~3 in the terrain worlds), so nearly every drawn column has runs beyond the two a device record
    holds (the run-list scan, `ovPending`, top-down AND bottom-up) and many runs are drawn per column.  Cameras above the world looking down and
    below it looking up: the world's columns then project INSIDE the pixel window instead of straddling it, so every frustum clip takes the
    reference's own path (`windowUntouched == false`: the four projections, floor / ceil, the window update and the window-closed exit) instead
    of the proven shortcut; cameras inside the world for the straddling case next to it.  Both iteration directions, counting and rendering
    build, two resolutions, against the oracle."""
    name = "stripes128x256x128"
    ws = scenes.load_world(name)
    poses = [((64.3, 128.0, 20.2), (0.0, 10.0, 0.0)), ((64.3, 400.0, 64.2), (60.0, 30.0, 0.0)), ((20.3, 420.0, 30.2), (35.0, 45.0, 0.0)),
             ((64.3, 520.0, 64.2), (86.0, 120.0, 0.0)), ((64.3, -120.0, 64.2), (-55.0, 200.0, 0.0)), ((100.3, -200.0, 90.2), (-30.0, 300.0, 0.0)),
             ((64.3, 100.0, 64.2), (25.0, 77.0, 0.0)), ((5.3, 200.0, 120.2), (-20.0, 135.0, 0.0)), ((64.3, 300.0, 64.2), (89.0, 0.0, 33.0))]
    directions = set()
    for W, H in ((320, 200), (517, 333)):
        ctx = contexts(name, W, H)
        for pos, eul in poses:
            fr = scenes.make_frame(ws, W, H, pos, eul)
            directions.add(bool(fr.camera.InverseElementIterationDirection))
            o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
            assert cnt.E > 15 * cnt.S, "the world is not run-rich for this pose"
            for counting in (True, False):
                g_td, g_lr = _render_gpu(ctx, fr, counters=counting)
                _compare(f"stripes {W}x{H} pos={pos} eul={eul} counting={counting}", fr, g_td, g_lr, o_td, o_lr)
                if counting:
                    gc = ctx.counters()
                    assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (cnt.S, cnt.E, cnt.C, cnt.P, cnt.R), (gc.as_dict(), cnt.as_dict())
    assert directions == {False, True}, "both element iteration directions must be covered"


def test_foreign_blob_columns_go_through_the_run_list():
    """Round 5: a 16-byte device record holds a column's runs only under the invariants of the reference's builder (the top run ends at WorldMax,
    the lowest stands on WorldMin, every ColorsIndex is the sum of the lengths above it; cvx_device.h); any other column is "listed".  The C ABI
    accepts every blob the reference's loader would, so here a built world's LOD-0 blob is edited into shapes no builder emits -- header bounds
    wider than the runs, a second run that shares the first one's colours, columns of air runs only (RunCount > 0, nothing solid) -- and
    rendered against the oracle, which walks the blob as the reference does: counting and rendering build, both iteration directions."""
    dims = (64, 128, 64)
    x, y, z, argb = _random_alpha_world(dims, 21, 3500, 3)
    base = host.WorldSet.from_voxels(dims, x, y, z, argb, threads=4)
    blobs = [np.array(base.storage(lod), dtype=np.uint8, copy=True) for lod in range(base.lod_count)]
    blob = blobs[0]
    columns = dims[0] * dims[2]
    header = np.dtype({"names": ["off", "rc", "mn", "mx"], "formats": ["<i4", "<u2", "<u2", "<u2"], "offsets": [0, 4, 6, 8], "itemsize": 12})
    hdr = blob[:columns * 12].view(header)
    el = blob[columns * 12:].view(np.dtype([("ci", "<i2"), ("len", "<i2")]))
    edited = {"bounds": 0, "shared colours": 0, "air only": 0}
    for c in range(columns):
        h = hdr[c]
        if h["rc"] == 0:
            continue
        runs = el[h["off"] + 1: h["off"] + 1 + h["rc"]]
        solid = np.flatnonzero(runs["ci"] >= 0)
        if c % 5 == 0:
            hdr[c]["mx"] = min(dims[1], int(h["mx"]) + 2)
            hdr[c]["mn"] = max(0, int(h["mn"]) - 1)
            edited["bounds"] += 1
        elif c % 5 == 1 and solid.size >= 2:
            runs["ci"][solid[1]] = 0  # (its colours [0, len) lie inside the column's colour array: len1 <= len0 + len1)
            edited["shared colours"] += 1
        elif c % 5 == 2:
            runs["ci"][solid] = -1
            edited["air only"] += 1
    assert min(edited.values()) > 50, edited
    ws = host.WorldSet.from_blobs(dims, blobs)
    W, H = 320, 200
    ctx = gpu.Context(0)
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        directions = set()
        for pos, eul in (((32.3, 70.0, 5.2), (5.0, 10.0, 0.0)), ((32.3, 140.0, 32.2), (70.0, 30.0, 0.0)), ((10.3, -20.0, 50.2), (-40.0, 120.0, 0.0)),
                         ((60.3, 64.0, 60.2), (0.0, 225.0, 0.0)), ((32.3, 100.0, 32.2), (-25.0, 300.0, 0.0))):
            fr = scenes.make_frame(ws, W, H, pos, eul)
            directions.add(bool(fr.camera.InverseElementIterationDirection))
            o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
            for counting, label, mode in [(True, "counting build", gpu.LATENCY_AUTO)] + [(False, l, m) for l, m in BOTH_KERNELS]:
                g_td, g_lr = _render_gpu(ctx, fr, counters=counting, latency=mode)
                _compare(f"foreign blob pos={pos} eul={eul} [{label}]", fr, g_td, g_lr, o_td, o_lr)
                if counting:
                    gc = ctx.counters()
                    assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (cnt.S, cnt.E, cnt.C, cnt.P, cnt.R), (gc.as_dict(), cnt.as_dict())
        assert directions == {False, True}
    finally:
        ctx.close()


def test_sparse_deep_world_keeps_its_colours_column_after_column():
    """The device keeps a level's colours in blocks of 4 x 8 columns, each as deep as its deepest column (cvx_device.h) -- unless that would take more
    than four times the colours themselves: a hundred deep columns in an empty world, as here, keep theirs column after column (`colorShift` 2
    instead of 7).  Same pictures and counters as the oracle either way."""
    dims = (64, 256, 64)
    x, y, z, argb = _random_alpha_world(dims, 31, 100, 4)
    ws = host.WorldSet.from_voxels(dims, x, y, z, argb, threads=4)
    W, H = 320, 200
    ctx = gpu.Context(0)
    try:
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        for pos, eul in (((32.3, 130.0, -20.2), (5.0, 0.0, 0.0)), ((32.3, 300.0, 32.2), (80.0, 30.0, 0.0)), ((-10.3, 100.0, 70.2), (-10.0, 120.0, 0.0)), ((32.3, 128.0, 32.2), (0.0, 45.0, 0.0))):
            fr = scenes.make_frame(ws, W, H, pos, eul)
            o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
            assert cnt.P > 0
            for counting in (True, False):
                g_td, g_lr = _render_gpu(ctx, fr, counters=counting)
                _compare(f"sparse world pos={pos} eul={eul} counting={counting}", fr, g_td, g_lr, o_td, o_lr)
                if counting:
                    gc = ctx.counters()
                    assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (cnt.S, cnt.E, cnt.C, cnt.P, cnt.R), (gc.as_dict(), cnt.as_dict())
    finally:
        ctx.close()


def test_one_level_uploaded_again_keeps_the_others(contexts):
    """cvx_world_upload of ONE level after a draw: the next draw lays out a new arena, takes the uploaded level from the host and every other
    level (records, run list, counts, colours) from the old arena, device to device.  Same pictures and counters as before, against the oracle."""
    name = "proc256_t075_lod8"  # lodError 8: rays reach the higher levels
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = contexts(scenes.SCENES[name][0], W, H)
    o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
    try:
        for lod in (1, 0, 3):
            i = ws.info(lod)
            ctx._check(gpu.lib().cvx_world_upload(ctx._h, lod, i.storage, i.byteLength, i.dimX, i.dimY, i.dimZ, i.columnCount))
            for counting in (True, False):
                g_td, g_lr = _render_gpu(ctx, fr, counters=counting)
                _compare(f"{name} after level {lod} again, counting={counting}", fr, g_td, g_lr, o_td, o_lr)
                if counting:
                    gc = ctx.counters()
                    assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (cnt.S, cnt.E, cnt.C, cnt.P, cnt.R), (gc.as_dict(), cnt.as_dict())
    finally:
        ctx.upload_world(ws)


def test_long_world_far_edge_checkpoints(contexts):
    """a ray that makes more than ~9 000 crossings on one axis at one LOD level.  The column loop tests the ray's position
    only beyond a stop distance `tMax + (n - 4) tDelta`; the DDA's own additions drift by ~n^2 2^-24 tDelta, so over 16 384 crossings the three
    crossings of slack would be used up five times over.  The kernel therefore renews the stop distance every 1024 crossings.  World: 16384 x 1024
    x 256 columns, LOD distances and far clip beyond the world (the caller's choice at the boundary: CameraData.LODDistances / FarClip), cameras
    above the terrain looking along x: about a dozen rays per frame walk the whole length and leave through the far edge, the others through the
    sides.  Rendering AND counting build against the oracle (pixels; S / P of the counting build)."""
    name = "proc16384x1024x256"
    ws = scenes.load_world(name)
    W, H = 1280, 160
    ctx = contexts(name, W, H)
    poses = [((3.5, 900.0, 128.5), (0.0, 90.0, 0.0)), ((16380.5, 880.0, 100.2), (0.2, 270.01, 0.0)), ((-3000.0, 870.0, 127.3), (-0.1, 90.0, 0.0)),
             ((8000.0, 860.0, 30.3), (0.1, 89.2, 0.0)), ((100.5, 850.0, 250.0), (0.0, 91.0, 0.0))]
    longest = 0
    for pos, eul in poses:
        pose = host.camera_pose(pos, eul, W, H)
        fr = host.setup_frame(pose, [200000.0] * 6, 100000.0, W, H, ws.dims[1], True)
        o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
        for counting in (True, False):
            g_td, g_lr = _render_gpu(ctx, fr, counters=counting)
            _compare(f"long world pos={pos} eul={eul} counting={counting}", fr, g_td, g_lr, o_td, o_lr)
            if counting:
                gc = ctx.counters()
                assert (gc.S, gc.E, gc.C, gc.P, gc.R) == (cnt.S, cnt.E, cnt.C, cnt.P, cnt.R), (gc.as_dict(), cnt.as_dict())
        longest = max(longest, cnt.S)
    assert longest > 10 * 16000, "no frame with ~a dozen rays along the whole world: the scenario no longer exercises the checkpoints"


def test_errors_are_reported_not_swallowed():
    ctx = gpu.Context(0)
    ws = scenes.load_world("proc256")
    fr = scenes.benchmark_frame(ws, 320, 200, 0.5)
    ctx.set_resolution(320, 200)
    with pytest.raises(gpu.CvxError):  # no world uploaded
        ctx.draw_segments(fr, 0)
    ctx.upload_world(ws)
    ctx.draw_segments(fr, 0)
    with pytest.raises(gpu.CvxError):  # resolution mismatch
        ctx.width = 640
        ctx.draw_segments(fr, 0)
    ctx.width = 320
    for mode in (gpu.LATENCY_ALWAYS, gpu.LATENCY_NEVER):  # (the checks sit in front of both kernels)
        ctx.set_latency_kernel(mode)
        fr.camera.PositionY = float("nan")
        with pytest.raises(gpu.CvxError):  # non-finite camera
            ctx.draw_segments(fr, 0)
    with pytest.raises(gpu.CvxError):  # not a latency-kernel mode
        ctx.set_latency_kernel(7)
    ctx.set_latency_kernel(gpu.LATENCY_AUTO)
    # a world wider than the column loop's packed position (x * 65536 + z, cvx_kernels.h ColumnCursor) is refused at upload
    import ctypes as C
    blob = (C.c_uint8 * (65536 * 12))()
    assert gpu.lib().cvx_world_upload(ctx._h, 0, blob, len(blob), 65536, 2, 1, 65536) == -1  # CVX_ERR_INVALID_ARGUMENT
    assert b"32768" in gpu.lib().cvx_last_error(ctx._h)
    ctx.close()


def test_copy_rows_pack_unpack_round_trip(contexts):
    """cvx_copy_rows (payload packing of the multi-GPU tile exchange): pool -> staging -> pool restores the rows."""
    import torch

    from cpuvox_amd import dist as cdist

    name = "proc256_t075_lod8"
    ws, fr, W, H = scenes.scene_frame(name)
    ctx = gpu.Context(0, buffer_count=2)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    lay_td, lay_lr = ctx.raybuffer_layout(0), ctx.raybuffer_layout(1)
    dev = torch.device("cuda", 0)
    pools = cdist.allocate_pools(2, lay_td, lay_lr, dev)
    ctx.bind_raybuffers(pools.td.data_ptr(), pools.td.numel() * 4, pools.lr.data_ptr(), pools.lr.numel() * 4)
    ctx.draw_segments(fr, 1)
    ctx.synchronize()
    # "rank 1 of 2" view: what it would send to rank 0 for a frame in buffer 0 ... use buffer 1 -> root 1, owner 0 sends
    ex = cdist.TileExchange([fr, fr], W, H, 0, 2, pools, dev, ctx)
    assert ex.sent_rows > 0 and ex.recv_rows > 0
    cdist.TileExchange.allocate_staging([ex], dev)
    before_td, before_lr = pools.td.clone(), pools.lr.clone()
    ex._copy_rows(ex.send_spans, ex._send_spans_dev, ex.send_buf, True)
    # staging holds exactly the spans' rows
    rows_td = pools.td.view(-1, 64)
    rows_lr = pools.lr.view(-1, 64)
    for sp in ex.send_spans[:: max(1, len(ex.send_spans) // 7)]:
        src = (rows_td if sp["kind"] == 0 else rows_lr)[int(sp["poolRow"]): int(sp["poolRow"]) + int(sp["rows"])]
        assert torch.equal(ex.send_buf[int(sp["packedRow"]): int(sp["packedRow"]) + int(sp["rows"])], src)
    # wipe the span rows, unpack, compare
    for sp in ex.send_spans:
        (rows_td if sp["kind"] == 0 else rows_lr)[int(sp["poolRow"]): int(sp["poolRow"]) + int(sp["rows"])] = 0
    ex._copy_rows(ex.send_spans, ex._send_spans_dev, ex.send_buf, False)
    assert torch.equal(pools.td, before_td) and torch.equal(pools.lr, before_lr)
    # the render through bound (external) pools is still bit exact
    o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False)
    n_td, n_lr = scenes.used_rows(fr)
    assert (ctx.read_raybuffer(1, 0, 0, n_td) == o_td[:n_td]).all() and (ctx.read_raybuffer(1, 1, 0, n_lr) == o_lr[:n_lr]).all()
    ctx.close()


def test_render_manager_twin_draw_world():
    """The C++ RenderManager twin (SetResolution / SwapBuffers / ClearRayBuffer / DrawWorld) end to end:
    the screen image equals the Phase-2 rule applied to the ORACLE's raybuffers of the same frame."""
    from cpuvox_amd import host
    from cpuvox_amd.render_manager import RAYBUFFER_TOPDOWN, RenderManager

    ws = scenes.load_world("mill256")
    W, H = 640, 480
    rm = RenderManager(W, H)
    rm.upload_world(ws)
    assert rm.set_resolution(W, H) is False and rm.swap_buffers() == 1 and rm.swap_buffers() == 0
    for t in (0.0, 0.75, 0.9):
        pos, eul = host.sample_benchmark_path(t, ws.dims)
        pose = host.camera_pose(pos, eul, W, H)
        lods, far = host.setup_lods(pose, ws.max_dimension, W, H)
        rm.swap_buffers()
        rm.clear_raybuffer(RAYBUFFER_TOPDOWN)
        img = rm.draw_world(pose, lods, far)
        fr = rm.last_frame
        ref_fr = scenes.make_frame(ws, W, H, pos, eul)
        assert bytes(fr.camera) == bytes(ref_fr.camera) and [s.RayCount for s in fr.segments] == [s.RayCount for s in ref_fr.segments]
        o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0x9314FFFF, counters=False)  # ClearRayBuffer pink: bytes FF FF 14 93
        n_td, n_lr = scenes.used_rows(fr)
        assert (rm.read_raybuffer(gpu.RAYBUFFER_TOPDOWN, 0, n_td) == o_td[:n_td]).all()
        ref = O.blit_reference(fr, o_td, o_lr, W, H, clear=0)
        assert (img == ref).all(), f"t={t}: {(img != ref).sum()} screen pixels differ"
    assert rm.set_resolution(320, 240) is True
    rm.close()
    with pytest.raises(RuntimeError):
        import cpuvox_amd.gpu as g
        old = g.lib_path
        try:
            g.lib_path = lambda: "/nonexistent/libcpuvox_gpu.so"
            RenderManager(W, H)
        finally:
            g.lib_path = old


def test_two_rank_exchange_emulated_on_one_gpu():
    """The multi-GPU path minus RCCL: two contexts render the two shards of every frame, the tile exchange's pack ->
    (peer-to-peer transfer emulated by a device copy) -> unpack assembles frame f on "rank" f % 2, bit-identical to
    the oracle.  Exercises cvx_set_shard, cvx_bind_raybuffers, cvx_copy_rows and cpuvox_amd.dist.TileExchange."""
    import torch

    from cpuvox_amd import dist as cdist

    ws = scenes.load_world("proc256")
    W, H = 320, 200
    frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in (0.05, 0.45, 0.75, 0.9)]
    G, N = len(frames), 2
    dev = torch.device("cuda", 0)
    ranks = []
    for r in range(N):
        ctx = gpu.Context(0, buffer_count=G)
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        ctx.set_shard(r, N)
        pools = cdist.allocate_pools(G, ctx.raybuffer_layout(0), ctx.raybuffer_layout(1), dev)
        ctx.bind_raybuffers(pools.td.data_ptr(), pools.td.numel() * 4, pools.lr.data_ptr(), pools.lr.numel() * 4)
        ex = cdist.TileExchange(frames, W, H, r, N, pools, dev, ctx)
        cdist.TileExchange.allocate_staging([ex], dev)
        ctx.draw_segments_batch(frames, 0)
        ranks.append((ctx, pools, ex))
    for r in range(N):  # pack on every rank
        ctx, pools, ex = ranks[r]
        ex._copy_rows(ex.send_spans, ex._send_spans_dev, ex.send_buf, True)
    for r in range(N):  # "P2P": rank r's section for peer p lands in p's section for r
        _, _, ex = ranks[r]
        for p in range(N):
            if p == r:
                continue
            _, _, exp = ranks[p]
            s0, s1 = ex.send_off[p], ex.send_off[p + 1]
            r0, r1 = exp.recv_off[r], exp.recv_off[r + 1]
            assert s1 - s0 == r1 - r0 > 0
            exp.recv_buf[r0:r1].copy_(ex.send_buf[s0:s1])
    torch.cuda.synchronize()
    for r in range(N):  # unpack, then frame b with b % N == r must be complete on rank r
        ctx, pools, ex = ranks[r]
        ex._copy_rows(ex.recv_spans, ex._recv_spans_dev, ex.recv_buf, False)
        for b, fr in enumerate(frames):
            if b % N != r:
                continue
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False)
            _compare(f"emulated exchange, frame {b} on rank {r}", fr, ctx.read_raybuffer(b, 0), ctx.read_raybuffer(b, 1), o_td, o_lr)
    for ctx, _, _ in ranks:
        ctx.close()


@pytest.mark.parametrize("case", ["4k_vp_on_screen", "4k_4096_horizontal_lod4"])
def test_baseline_configs_4_and_5_shapes(case):
    """BASELINE.json configs 4 / 5 as parity cases (the bench runs config 3): 3840x2160 (120 mask words per lane),
    VP on screen with 12000 rays; and a 4096-wide world, forward.y = +-0.001 (worst-case precision, single clamped
    segment), lodError = 4 so that LOD 0-4 are reached.  Bit-exact against the oracle."""
    from cpuvox_amd import host

    W, H = 3840, 2160
    if case == "4k_vp_on_screen":
        ws = scenes.load_world("proc512")
        frames = [scenes.benchmark_frame(ws, W, H, 0.75, 2.0)]
        assert abs(frames[0].totalRays - 12000) <= 2
    else:
        ws = scenes.load_world("proc4096x512x4096")
        frames = [scenes.make_frame(ws, W, H, (-0.1 * 4096, 0.5 * 512, -0.1 * 4096), (0.0, 45.0, 0.0), lod_error=4.0),
                  scenes.make_frame(ws, W, H, (0.4 * 4096, 0.9 * 512, 0.3 * 4096), (0.01, 200.0, 0.0), lod_error=4.0)]
        assert [abs(f.forward[1]) for f in frames] == pytest.approx([0.001, 0.001], abs=1e-6)
    ctx = gpu.Context(0)
    ctx.upload_world(ws)
    ctx.set_resolution(W, H)
    for i, fr in enumerate(frames):
        ctx.enable_counters(True)
        ctx.clear_raybuffers(0, CLEAR)
        ctx.draw_segments(fr, 0)
        o_td, o_lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
        _compare(f"{case}[{i}]", fr, ctx.read_raybuffer(0, 0), ctx.read_raybuffer(0, 1), o_td, o_lr)
        gc = ctx.counters()
        assert list(gc.lodVisits) == list(cnt.lodVisits) and gc.P == cnt.P
        if case != "4k_vp_on_screen" and i == 0:
            assert sum(1 for v in cnt.lodVisits if v > 0) >= 5, list(cnt.lodVisits)  # LOD 0-4 reached
    ctx.close()


def test_zero_copy_sharding_emulated_on_one_gpu():
    """cvx_draw_segments_placed + cpuvox_amd.dist.ShardPlan: every tile is rendered straight into the send or display
    buffer it belongs to; "P2P" (emulated: device copy of a peer's send section) completes the frames on their display
    ranks, bit-identical to the oracle.  Three emulated ranks on one GPU."""
    import torch

    from cpuvox_amd import dist as cdist

    ws = scenes.load_world("proc256")
    W, H = 320, 200
    frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in (0.05, 0.3, 0.45, 0.75, 0.9, 1.1, 0.6)]
    N = 3
    dev = torch.device("cuda", 0)
    ranks = []
    for r in range(N):
        ctx = gpu.Context(0)
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        plan = cdist.ShardPlan(frames, W, H, r, N)
        send = torch.zeros((max(1, plan.send_total), 64), dtype=torch.int32, device=dev)
        disp = torch.zeros((max(1, plan.disp_total), 64), dtype=torch.int32, device=dev)
        ctx.draw_placed(ctx.pack_batch(frames), plan.tile_out(send.data_ptr(), disp.data_ptr()))
        ranks.append((ctx, plan, send, disp))
    # every tile is rendered by exactly one rank
    assert sum(len(p.my_tiles) for _, p, _, _ in ranks) == ranks[0][1].tile_count
    for r, (_, plan, send, _) in enumerate(ranks):  # the transfers
        for p, (_, pplan, _, pdisp) in enumerate(ranks):
            if p == r:
                continue
            s0, s1 = int(plan.send_start[p]), int(plan.send_start[p + 1])
            r0, r1 = int(pplan.disp_start[r]), int(pplan.disp_start[r + 1])
            assert s1 - s0 == r1 - r0
            if s1 > s0:
                pdisp[r0:r1].copy_(send[s0:s1])
    torch.cuda.synchronize()
    for r, (ctx, plan, _, disp) in enumerate(ranks):
        for b, fr in enumerate(frames):
            if b % N != r:
                continue
            g_td, g_lr = plan.assemble(disp, b, [s.RayCount for s in fr.segments], W, H)
            o_td, o_lr, _ = O.draw_segments(ws, fr, W, H, clear=0, counters=False)
            _compare(f"zero-copy sharding, frame {b} on rank {r}", fr, g_td, g_lr, o_td, o_lr)
        ctx.close()


@pytest.mark.parametrize("N", [1, 2, 3])
def test_image_gather_emulated_on_one_gpu(N):
    """The IMAGE gather (cvx_image_plan_* / cvx_image_pack / cvx_image_unpack): every emulated rank renders its tiles into its compact
    local store, blits only its own pixels, the peers' pixel streams are "sent" (device copies of the plan's transfer ranges) and the
    display rank of every frame ends up with exactly the image cvx_blit_segments makes from the whole frame on one GPU -- which in
    turn is pinned to the numpy rule over the oracle's raybuffers."""
    import torch

    ws = scenes.load_world("proc256")
    W, H = 320, 200
    frames = [scenes.benchmark_frame(ws, W, H, t, 6.0) for t in (0.05, 0.3, 0.45, 0.75, 0.9, 1.1, 0.6)]
    dev = torch.device("cuda", 0)
    whole = gpu.Context(0, buffer_count=len(frames))
    whole.upload_world(ws)
    whole.set_resolution(W, H)
    whole.draw_segments_batch(frames, 0)
    expected = [whole.blit_segments(b) for b in range(len(frames))]
    o_td, o_lr, _ = O.draw_segments(ws, frames[3], W, H, clear=0, counters=False)
    assert np.array_equal(expected[3], O.blit_reference(frames[3], o_td, o_lr, W, H))
    ranks = []
    for r in range(N):
        ctx = gpu.Context(0)
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        packed = ctx.pack_batch(frames)
        plan = gpu.ImagePlan(ctx, packed, W, H, r, N)
        store = torch.zeros(max(1, plan.local_store_bytes // 4), dtype=torch.int32, device=dev)
        send = torch.full((max(1, plan.send_pixels),), 0x55, dtype=torch.int32, device=dev)
        recv = torch.full((max(1, plan.recv_pixels),), 0x66, dtype=torch.int32, device=dev)
        images = torch.full((max(1, plan.images), H, W), 0x77, dtype=torch.int32, device=dev)
        ctx.draw_placed(packed, plan.tile_out(store.data_ptr()))
        plan.pack(ctx, None, store.data_ptr(), send.data_ptr(), images.data_ptr())
        ctx.synchronize()
        ranks.append((ctx, plan, send, recv, images))
    assert sum(p.images for _, p, _, _, _ in ranks) == len(frames)
    # every pixel of every frame travels at most once: what the peers send == what the display ranks expect, W * H per frame in total
    for r, (_, plan, send, _, _) in enumerate(ranks):
        for q, (_, qplan, _, qrecv, _) in enumerate(ranks):
            s0, sn, _, _ = plan.transfer(q)
            _, _, r0, rn = qplan.transfer(r)
            assert sn == rn and (q != r or sn == 0)
            if sn:
                qrecv[r0:r0 + rn].copy_(send[s0:s0 + sn])
    torch.cuda.synchronize()
    total_sent = sum(p.send_pixels for _, p, _, _, _ in ranks)
    assert total_sent <= len(frames) * W * H and (N == 1) == (total_sent == 0)
    for r, (ctx, plan, _, recv, images) in enumerate(ranks):
        plan.unpack(ctx, None, recv.data_ptr(), images.data_ptr())
        ctx.synchronize()
        got = images.cpu().numpy().view(np.uint32)
        for b in range(len(frames)):
            if b % N == r:
                assert np.array_equal(got[b // N], expected[b]), f"image gather: frame {b} on rank {r} of {N} differs in {(got[b // N] != expected[b]).sum()} pixels"
        plan.close()
        ctx.close()
    whole.close()
