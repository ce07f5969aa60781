"""CPU: host-side mirror of the managed code (camera / VP / segments / LOD distances / world building)."""
import math
import os
import tempfile

import numpy as np
import pytest

import scenes
from cpuvox_amd import host


def _frame(ws, W, H, pos, eul, **kw):
    return scenes.make_frame(ws, W, H, pos, eul, **kw)


@pytest.fixture(scope="module")
def ws():
    return scenes.load_world("proc256")


def test_ray_count_is_2w_plus_2h_when_vp_on_screen(ws):
    """Sum RayCount = 2(W+H) whenever the VP is inside the screen (RenderManager.cs:128-142,419-420,482)."""
    for W, H in ((640, 480), (1920, 1080), (800, 450)):
        for pitch in (50.0, 70.0, 85.0, -60.0, -80.0):
            fr = _frame(ws, W, H, (100, 200, 100), (pitch, 33.0, 0.0))
            vp = fr.vanishingPointScreenSpace
            if 0 <= vp[0] <= W and 0 <= vp[1] <= H:
                assert abs(fr.totalRays - 2 * (W + H)) <= 2, (W, H, pitch, fr.totalRays)


def test_vanishing_point_matches_pinhole_geometry(ws):
    """Looking down by pitch p (no roll) the VP is at x = W/2, y = H/2 - (H/2) / (tan(fov/2) tan(p))."""
    W, H, fov = 640, 480, 85.0
    for pitch in (30.0, 59.12, 80.0):
        fr = _frame(ws, W, H, (10, 100, 10), (pitch, 123.0, 0.0))
        want_y = H / 2 - (H / 2) / (math.tan(math.radians(fov / 2)) * math.tan(math.radians(pitch)))
        assert abs(fr.vanishingPointScreenSpace[0] - W / 2) < 1e-2
        assert abs(fr.vanishingPointScreenSpace[1] - want_y) < 0.05, (pitch, fr.vanishingPointScreenSpace[1], want_y)
        assert fr.camera.InverseElementIterationDirection == 0
    fr = _frame(ws, W, H, (10, 100, 10), (-40.0, 0.0, 0.0))
    assert fr.camera.InverseElementIterationDirection == 1 and fr.vanishingPointScreenSpace[1] > H


def test_horizon_clamp(ws):
    """LimitRotationHorizon (UnityManager.cs:193-201): |forward.y| < 0.001 -> +-0.001, Mathf.Sign(0) = +1."""
    fr = _frame(ws, 640, 480, (0, 100, 0), (0.0, 45.0, 0.0))
    assert abs(fr.forward[1] - 0.001) < 1e-6 and fr.camera.InverseElementIterationDirection == 1
    fr = _frame(ws, 640, 480, (0, 100, 0), (0.01, 45.0, 0.0))
    assert abs(fr.forward[1] + 0.001) < 1e-6 and fr.camera.InverseElementIterationDirection == 0
    assert [s.RayCount > 0 for s in fr.segments] == [True, False, False, False]
    fr = _frame(ws, 640, 480, (0, 100, 0), (0.01, 45.0, 0.0), limit_horizon=False)
    assert abs(fr.forward[1] + math.sin(math.radians(0.01))) < 1e-7


def test_world_to_screen_matrix_projects_a_known_point(ws):
    """CameraData.WorldToScreenMatrix (CameraData.cs:24-29): a point on the optical axis lands on the screen centre,
    z' = 0 on the near plane, w' = view depth."""
    W, H = 640, 480
    pos, eul = np.array([50.0, 80.0, 60.0]), (20.0, 30.0, 0.0)
    fr = _frame(ws, W, H, pos, eul)
    M = np.array(list(fr.camera.WorldToScreenMatrix), dtype=np.float64).reshape(4, 4).T  # column major
    f = np.array(list(fr.forward), dtype=np.float64)
    for depth in (0.05, 1.0, 100.0):
        p = M @ np.append(pos + f * depth, 1.0)
        assert abs(p[3] - depth) < 1e-3 * max(1, depth)
        assert abs(p[0] / p[3] - W / 2) < 0.1 and abs(p[1] / p[3] - H / 2) < 0.1  # float32 matrix, positions ~1e2
    p = M @ np.append(pos + f * 0.05, 1.0)
    assert abs(p[2]) < 1e-3  # near plane


def test_plane_rays_point_through_the_segment_corners(ws):
    """CamLocalPlaneRayMin/Max (RenderManager.cs:480-500) are world-axis XZ offsets whose projection lands on
    MinScreen/MaxScreen's column (x for top/bottom segments)."""
    W, H = 640, 480
    pos = np.array([50.0, 200.0, 60.0])
    fr = _frame(ws, W, H, pos, (59.12, -135.0, 0.0))
    M = np.array(list(fr.camera.WorldToScreenMatrix), dtype=np.float64).reshape(4, 4).T
    seg = fr.segments[0]
    for ray, screen in ((seg.CamLocalPlaneRayMin, seg.MinScreen), (seg.CamLocalPlaneRayMax, seg.MaxScreen)):
        p = M @ np.array([pos[0] + ray[0], 0.0, pos[2] + ray[1], 1.0])  # any height: vertical lines pass through the VP
        # the projected point, the VP and the corner are collinear
        a = np.array(list(fr.vanishingPointScreenSpace)); b = np.array(list(screen)); c = p[:2] / p[3]
        cross = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
        assert abs(cross) / (np.linalg.norm(b - a) * np.linalg.norm(c - a)) < 2e-3


def test_setup_lods_values():
    """SURVEY.md Appendix A.18 table (fov 85, square pixels)."""
    def lods(W, H, dim, err=1.0):
        return host.setup_lods(host.camera_pose((0, 0, 0), (0, 0, 0), W, H), dim, W, H, err)

    d, far = lods(640, 480, 256)
    assert far == 512.0 and d == [1024.0] * 6
    d, far = lods(1920, 1080, 2048)
    assert far == 4096.0 and d == [1176.0, 2351.0, 8192.0, 8192.0, 8192.0, 8192.0]
    d, far = lods(3840, 2160, 4096, 4.0)
    assert far == 8192.0 and d[:3] == [589.0, 1176.0, 2351.0] and abs(d[3] - 4701.0) <= 1 and d[4:] == [16384.0, 16384.0]


def test_benchmark_path_keys():
    """BenchmarkPath.anim keys (:16-148) are hit exactly, positions scale with the world (UnityManager.cs:87)."""
    dims = (256, 128, 512)
    pos, eul = host.sample_benchmark_path(0.0, dims)
    assert np.allclose(pos, [-25.6, 64.0, -51.2], atol=1e-4) and np.allclose(eul, [0, 45, 0])
    pos, eul = host.sample_benchmark_path(0.75, dims)
    assert np.allclose(pos, [230.4, 121.6, 460.8], atol=1e-3) and np.allclose(eul, [59.12, -135, 0], atol=1e-4)
    pos, eul = host.sample_benchmark_path(0.875, dims)
    assert np.allclose(eul, [59.12, -135, 180], atol=1e-3)
    pos, eul = host.sample_benchmark_path(1.15, dims)
    assert np.allclose(eul, [85, -225.5, 360], atol=1e-3)


def _columns(ws, lod):
    info = ws.info(lod)
    raw = ws.storage(lod)
    hdr = np.frombuffer(raw[: info.columnCount * 12].tobytes(), dtype=np.dtype([("off", "<i4"), ("runs", "<u2"), ("wmin", "<u2"), ("wmax", "<u2"), ("pad", "<u2")]))
    elems = np.frombuffer(raw[info.columnCount * 12:].tobytes(), dtype=np.dtype([("ci", "<i2"), ("len", "<i2")]))
    return info, hdr, elems


@pytest.mark.parametrize("lod", [0, 1, 3])
def test_rle_column_invariants(ws, lod):
    """World.cs:190-234 / WordBuilder.cs:232-258: guards are (0,0), runs fill the column height exactly, colour
    indices are consecutive, worldMin/Max bound the solid runs in LOD-0 units."""
    info, hdr, elems = _columns(ws, lod)
    used = (info.dimX >> lod) * (info.dimZ >> lod)
    height = info.dimY >> lod
    rng = np.random.default_rng(lod)
    for i in rng.integers(0, used, 300):
        h = hdr[i]
        if h["runs"] == 0:
            continue
        e = elems[h["off"]: h["off"] + h["runs"] + 2]
        assert e[0]["len"] == 0 and e[0]["ci"] == 0 and e[-1]["len"] == 0 and e[-1]["ci"] == 0
        runs = e[1:-1]
        assert (runs["len"] > 0).all() and runs["len"].sum() == height
        solid = runs[runs["ci"] >= 0]
        assert (solid["ci"] == np.concatenate([[0], np.cumsum(solid["len"])[:-1]])).all()
        top = height
        lo, hi = 1 << 30, -1
        for r in runs:
            bottom = top - r["len"]
            if r["ci"] >= 0:
                lo, hi = min(lo, bottom), max(hi, top)
            top = bottom
        assert h["wmin"] == lo << lod and h["wmax"] == hi << lod


def test_downsample_preserves_occupancy(ws):
    """World.DownSample (World.cs:45-127): a LOD-1 voxel is solid iff any of its 8 LOD-0 children is."""
    def occupancy(lod, x, z):
        info, hdr, elems = _columns(ws, lod)
        h = hdr[(x >> lod) * (info.dimZ >> lod) + (z >> lod)]
        occ = np.zeros(info.dimY >> lod, dtype=bool)
        if h["runs"]:
            top = info.dimY >> lod
            for r in elems[h["off"] + 1: h["off"] + 1 + h["runs"]]:
                if r["ci"] >= 0:
                    occ[top - r["len"]: top] = True
                top -= r["len"]
        return occ

    for x, z in ((10, 20), (128, 64), (254, 254), (77, 200)):
        x &= ~1; z &= ~1
        child = np.zeros(ws.dims[1], dtype=bool)
        for dx in (0, 1):
            for dz in (0, 1):
                child |= occupancy(0, x + dx, z + dz)
        assert (occupancy(1, x, z) == child.reshape(-1, 2).any(axis=1)).all()


def test_save_load_round_trip(ws):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.world")
        ws.save(path)
        head = np.fromfile(path, dtype="<i4", count=6)
        assert head[0] == 0 and head[1] == 0 and list(head[2:5]) == list(ws.dims) and head[5] == 6  # WorldSaveFile.Header
        back = host.WorldSet.load(path)
        for lod in range(6):
            assert (back.storage(lod) == ws.storage(lod)).all()


def test_procedural_world_is_deterministic_and_seeded():
    a = host.WorldSet.procedural(64, 64, 64, 123)
    b = host.WorldSet.procedural(64, 64, 64, 123, threads=3)
    c = host.WorldSet.procedural(64, 64, 64, 124)
    assert (a.storage(0) == b.storage(0)).all() and a.lod0_voxels == b.lod0_voxels
    assert a.storage(0).shape != c.storage(0).shape or (a.storage(0) != c.storage(0)).any()


def test_dedupe_averages_colours():
    """RLEColumnBuilder.ToFinalColumn (WordBuilder.cs:192-228): voxels sharing a Y are merged, r/g/b averaged."""
    ws = host.WorldSet.from_voxels((64, 64, 64), [5, 5, 5], [7, 7, 8], [9, 9, 9], [0x0A141EFF, 0x1E2832FF, 0x646464FF])
    info, hdr, elems = _columns(ws, 0)
    h = hdr[5 * 64 + 9]
    assert h["runs"] == 3 and h["wmin"] == 7 and h["wmax"] == 9
    colors = ws.storage(0)[info.columnCount * 12:].view("<u4")[h["off"] + h["runs"] + 2: h["off"] + h["runs"] + 4]
    assert colors[0] == 0x646464FF  # y = 8 (top)
    assert colors[1] == 0x141E28FF  # y = 7: bytes A=FF, R=(0x1E+0x32)/2.. averaged per channel
    assert ws.lod0_voxels == 2


def test_obj_import_and_voxelise(tmp_path):
    """ObjModel.Import + SimpleMesh.Rescale + VoxelizerHelper on a two-triangle quad with vertex colours."""
    obj = tmp_path / "quad.obj"
    obj.write_text("o q\nv 0 0 0 1 0 0\nv 4 0 0 1 0 0\nv 4 0 4 1 0 0\nv 0 0 4 1 0 0\nv 0 2 0 0 1 0\ns 1\nf 1 2 3\nf 1 3 4\nf 1/1 2/1 5/1\n")
    ws = host.WorldSet.from_obj(str(obj), 32, flip=(False, False, False))
    assert ws.dims == (32, 16, 32)
    assert ws.lod0_voxels >= 32 * 32  # the floor quad fills every column
    info, hdr, elems = _columns(ws, 0)
    assert (hdr["runs"][: 32 * 32] > 0).all()
    with pytest.raises(RuntimeError):
        host.WorldSet.from_obj(str(tmp_path / "missing.obj"), 32)


def test_builder_rejects_bad_input():
    with pytest.raises(RuntimeError):
        host.WorldSet.from_voxels((48, 64, 64), [], [], [], [])  # x not a power of two (WordBuilder.cs:30-32)
    with pytest.raises(RuntimeError):
        host.WorldSet.from_voxels((64, 64, 64), [64], [0], [0], [0xFFFFFFFF])  # voxel out of bounds


def test_world_from_blobs_round_trip():
    """cvxh_world_from_blobs: a set assembled from the storage blobs of another is the same world (WorldSaveFile.cs:86-92)."""
    ws = scenes.load_world("proc64") if hasattr(scenes, "load_world") else None
    blobs = [bytes(ws.storage(lod)) for lod in range(ws.lod_count)]
    copy = host.WorldSet.from_blobs(ws.dims, blobs)
    assert copy.lod_count == ws.lod_count and copy.dims == ws.dims
    for lod in range(ws.lod_count):
        a, b = ws.info(lod), copy.info(lod)
        assert (a.columnCount, a.elementCount, a.byteLength, a.lod) == (b.columnCount, b.elementCount, b.byteLength, b.lod)
        assert np.array_equal(ws.storage(lod), copy.storage(lod))
    with pytest.raises(RuntimeError):
        host.WorldSet.from_blobs(ws.dims, [b"\x00" * 16])


def test_host_downsample_timing_hook():
    ws = scenes.load_world("proc64")
    seconds, voxels = ws.downsample_host_seconds(1, threads=2)
    assert seconds >= 0.0 and voxels > 0


def _write_png(path, rgba):
    """Minimal PNG writer (8-bit RGBA, filter 0) for the texture tests; rgba[y][x] with row 0 = TOP."""
    import struct
    import zlib

    h, w = len(rgba), len(rgba[0])
    raw = b"".join(b"\x00" + bytes(c for px in row for c in px) for row in rgba)

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def _world_colours(ws):
    """Set of (r, g, b) of every voxel of LOD 0 (colour entries of the element pool are bytes a, r, g, b)."""
    info = ws.info(0)
    blob = ws.storage(0)
    hdr = blob[: info.columnCount * 12].view(np.dtype([("off", "<i4"), ("rc", "<u2"), ("mn", "<u2"), ("mx", "<u2"), ("pad", "<u2")]))
    pool = blob[info.columnCount * 12:].view("<u4")
    out, voxels = set(), 0
    for h in hdr[hdr["rc"] > 0]:
        runs = pool[h["off"] + 1: h["off"] + 1 + h["rc"]]
        solid = [int(r >> 16) for r in runs if (int(r) & 0xFFFF) < 0x8000]
        n = sum(solid)
        voxels += n
        for c in pool[h["off"] + h["rc"] + 2: h["off"] + h["rc"] + 2 + n]:
            out.add(((int(c) >> 8) & 255, (int(c) >> 16) & 255, int(c) >> 24))
    return out, voxels


QUAD_OBJ = """mtllib quad.mtl
v 0 0 0
v 8 0 0
v 8 4 8
v 0 4 8
vt 0 0
vt 1 0
vt 1 1
vt 0 1
usemtl painted
f 1/1 2/2 3/3
f 1/1 3/3 4/4
"""


def test_obj_material_textures(tmp_path):
    """mtllib / usemtl / map_Kd (ObjModel.cs:44-49, SimpleMesh.cs:152-218, WordBuilder.cs:78-84): voxel colour = vertex colour * texel,
    texels that are not fully opaque leave no voxel, PNG / TGA / PPM decode to the same world, a broken JPEG is refused."""
    # 3x3 texels, all different; GetDiffusePixel maps uv to floor(uv * (size - 1)), so texel column / row 2 is only hit at uv == 1
    opaque = [[(40 * x + 10, 60 * y + 20, 200 - 30 * x, 255) for x in range(3)] for y in range(3)]  # rows top-down
    palette = {px[:3] for row in opaque for px in row}
    centre = opaque[1][1][:3]
    (tmp_path / "quad.obj").write_text(QUAD_OBJ)
    (tmp_path / "quad.mtl").write_text("# test\nnewmtl other\nKd 1 1 1\nnewmtl painted\nKd 0.5 0.5 0.5\nmap_Kd -bm 1.0 tex.png\n")
    _write_png(tmp_path / "tex.png", opaque)
    ws = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    colours, voxels = _world_colours(ws)
    assert colours <= palette and len(colours) >= 4 and centre in colours, colours
    assert voxels >= 64

    # the same picture as P6 PPM (rows top-down) builds the identical world
    with open(tmp_path / "tex.ppm", "wb") as f:
        f.write(b"P6\n# c\n3 3\n255\n" + bytes(c for row in opaque for px in row for c in px[:3]))
    (tmp_path / "quad.mtl").write_text("newmtl painted\nmap_Kd tex.ppm\n")
    ws_ppm = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    assert np.array_equal(ws_ppm.storage(0), ws.storage(0))
    # ... and as a bottom-up 32-bit TGA
    with open(tmp_path / "tex.tga", "wb") as f:
        f.write(bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 3, 0, 32, 8]) + bytes(c for row in opaque[::-1] for px in row for c in (px[2], px[1], px[0], px[3])))
    (tmp_path / "quad.mtl").write_text("newmtl painted\nmap_Kd tex.tga\n")
    ws_tga = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    assert np.array_equal(ws_tga.storage(0), ws.storage(0))

    # a translucent texel removes the voxels it covers
    holed = [list(row) for row in opaque]
    holed[1][1] = centre + (128,)
    _write_png(tmp_path / "tex.png", holed)
    (tmp_path / "quad.mtl").write_text("newmtl painted\nmap_Kd tex.png\n")
    ws_holed = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    colours_holed, voxels_holed = _world_colours(ws_holed)
    assert voxels_holed < voxels and centre not in colours_holed

    # unknown material name -> vertex colours only (white)
    (tmp_path / "quad.obj").write_text(QUAD_OBJ.replace("usemtl painted", "usemtl missing"))
    ws_plain = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    assert _world_colours(ws_plain)[0] == {(255, 255, 255)}

    # a degenerate (flat) model has a zero-height world: an error, not a crash
    (tmp_path / "flat.obj").write_text("v 0 0 0\nv 8 0 0\nv 8 0 8\nf 1 2 3\n")
    with pytest.raises(RuntimeError):
        host.WorldSet.from_obj(str(tmp_path / "flat.obj"), 8, flip=(False, False, False))

    (tmp_path / "quad.obj").write_text(QUAD_OBJ)
    (tmp_path / "tex.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\x00" * 32)
    (tmp_path / "quad.mtl").write_text("newmtl painted\nmap_Kd tex.jpg\n")
    with pytest.raises(RuntimeError, match="JPEG"):
        host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))


@pytest.mark.parametrize("mode", ["L", "RGB"])
@pytest.mark.parametrize("options", [
    dict(quality=95, subsampling=0),                       # 4:4:4 baseline
    dict(quality=90, subsampling=2),                       # 4:2:0
    dict(quality=85, subsampling=1),                       # 4:2:2
    dict(quality=92, subsampling=0, progressive=True),     # progressive (spectral selection + successive approximation)
    dict(quality=80, subsampling=2, progressive=True),
    dict(quality=90, subsampling=2, restart_marker_blocks=3),  # restart intervals (ignored by Pillow builds that lack the option)
])
def test_jpeg_textures_decode_like_libjpeg(tmp_path, mode, options):
    """map_Kd JPEG textures (Texture2D.LoadImage in the reference, SimpleMesh.cs:186-205): the decoder of cvx_image.cpp against
    Pillow's libjpeg on images Pillow encodes here -- sizes that are not multiples of the MCU, grey and YCbCr, subsampled chroma,
    progressive scans.  Tolerance: +-3 per channel (IDCT rounding differs between decoders; the chroma upsampling is the same filter)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    W, H = 83, 61
    ys, xs = np.mgrid[0:H, 0:W]
    base = np.stack([128 + 100 * np.sin(xs / 9.0) * np.cos(ys / 7.0), 128 + 90 * np.cos(xs / 5.0 + ys / 11.0), 40 + 2.0 * xs + 0.5 * ys], axis=-1)
    base = np.clip(base + rng.normal(0, 3, base.shape), 0, 255).astype(np.uint8)
    src = Image.fromarray(base).convert(mode)
    path = str(tmp_path / "t.jpg")
    try:
        src.save(path, "JPEG", **options)
    except TypeError:
        pytest.skip("this Pillow does not know one of the save options")
    ref = np.asarray(Image.open(path).convert("RGB"), dtype=np.int32)[::-1]  # row 0 = bottom, like the loader
    got = host.load_image(path)
    assert got.shape == (H, W, 4) and (got[..., 3] == 255).all()
    err = np.abs(got[..., :3].astype(np.int32) - ref)
    assert err.max() <= 3 and err.mean() < 0.6, (err.max(), err.mean())


def test_jpeg_texture_on_a_voxelised_quad(tmp_path):
    """The OBJ path end to end with a JPEG map_Kd: same world as with the PNG of the decoded pixels."""
    Image = pytest.importorskip("PIL.Image")
    tex = np.zeros((16, 16, 3), dtype=np.uint8)
    tex[:8, :8] = (200, 40, 40)
    tex[:8, 8:] = (40, 200, 40)
    tex[8:, :8] = (40, 40, 200)
    tex[8:, 8:] = (220, 220, 60)
    Image.fromarray(tex).save(str(tmp_path / "tex.jpg"), "JPEG", quality=95, subsampling=0)
    decoded = host.load_image(str(tmp_path / "tex.jpg"))  # row 0 = bottom
    (tmp_path / "quad.obj").write_text(QUAD_OBJ)
    (tmp_path / "quad.mtl").write_text("newmtl painted\nmap_Kd tex.jpg\n")
    ws_jpg = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    _write_png(tmp_path / "tex.png", [[tuple(int(v) for v in decoded[15 - y, x]) for x in range(16)] for y in range(16)])
    (tmp_path / "quad.mtl").write_text("newmtl painted\nmap_Kd tex.png\n")
    ws_png = host.WorldSet.from_obj(str(tmp_path / "quad.obj"), 8, flip=(False, False, False))
    assert np.array_equal(ws_jpg.storage(0), ws_png.storage(0)) and ws_jpg.lod0_voxels > 0
