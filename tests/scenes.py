"""Shared scene definitions for the parity tests, golden fixtures, smoke() and bench.py.

A scene = (world, resolution, camera pose, lodError).  Worlds are either the
mill.obj voxelisations committed as fixtures (tests/golden/*.world.xz, made by
tests/golden/make_golden.py from the reference's datasets/mill.obj with this
repo's own builder) or procedural worlds generated on the fly by
libcpuvox_host -- nothing here reads /root/reference at run time.
"""
from __future__ import annotations

import lzma
import os
import tempfile
import zlib

import numpy as np

from cpuvox_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

_world_cache: dict = {}


def load_world(name: str) -> host.WorldSet:
    """name: 'mill256' | 'mill512' (fixtures) or 'proc<dim>' / 'proc<X>x<Y>x<Z>[s<seed>]' (procedural)."""
    if name in _world_cache:
        return _world_cache[name]
    if name.startswith("mill"):
        path = os.path.join(GOLDEN, f"{name}.world.xz")
        raw = lzma.decompress(open(path, "rb").read())
        with tempfile.NamedTemporaryFile(suffix=".world", delete=False) as f:
            f.write(raw)
            tmp = f.name
        try:
            ws = host.WorldSet.load(tmp)
        finally:
            os.unlink(tmp)
    elif name.startswith("proc"):
        spec = name[4:]
        seed = 0x5EED2048
        if "s" in spec:
            spec, s = spec.split("s")
            seed = int(s, 0)
        if "x" in spec:
            dx, dy, dz = (int(v) for v in spec.split("x"))
        else:
            dx = dy = dz = int(spec)
        ws = host.WorldSet.procedural(dx, dy, dz, seed)
    elif name.startswith("stripes"):
        # run-rich world for the slow paths of the element walk: every column is a stack of 8..30 solid bands (thickness 1..5, period 6..20, phase and
        # colours hashed from the position), 'stripes<X>x<Y>x<Z>'.  Most columns have far more than the two runs a device record holds.
        dx, dy, dz = (int(v) for v in name[7:].split("x"))
        x, y, z = np.meshgrid(np.arange(dx, dtype=np.int64), np.arange(dy, dtype=np.int64), np.arange(dz, dtype=np.int64), indexing="ij")
        h = (x * 73856093) ^ (z * 19349663)
        period = 6 + (h >> 3) % 15
        thick = 1 + (h >> 9) % 5
        phase = (h >> 14) % period
        solid = ((y + phase) % period) < np.minimum(thick, period - 1)
        solid &= ~(((x + 2 * z) % 11 == 0) & (y > dy // 2))  # some columns lose their upper half: uneven run counts between neighbours
        xs, ys, zs = x[solid], y[solid], z[solid]
        argb = (0xFF000000 | (((xs * 2654435761 + ys * 40503 + zs * 2246822519) >> 7) & 0xFFFFFF)).astype(np.uint32)
        ws = host.WorldSet.from_voxels((dx, dy, dz), xs, ys, zs, argb)
    else:
        raise KeyError(name)
    _world_cache[name] = ws
    return ws


def make_frame(ws: host.WorldSet, width: int, height: int, position, euler, lod_error: float = 1.0, limit_horizon: bool = True) -> host.Frame:
    """UnityManager.LateUpdate + RenderManager.DrawWorld setup for one pose."""
    pose = host.camera_pose(position, euler, width, height)
    lods, far = host.setup_lods(pose, ws.max_dimension, width, height, lod_error)
    return host.setup_frame(pose, lods, far, width, height, ws.dims[1], limit_horizon)


def benchmark_frame(ws: host.WorldSet, width: int, height: int, t: float, lod_error: float = 1.0) -> host.Frame:
    pos, eul = host.sample_benchmark_path(t, ws.dims)
    return make_frame(ws, width, height, pos, eul, lod_error)


# name -> (world, W, H, kind, args, lodError)
#   kind 'path': args = benchmark clip time t
#   kind 'pose': args = (position as fraction of world dims, euler degrees)
SCENES = {
    # BASELINE.json config 1: mill.obj 256^3, 640x480, key t=0: camera outside the world, pitch 0 -> forward.y=+0.001,
    # single segment (index 1), ITERATION_DIRECTION -1, StepToWorldIntersection exercised.
    "mill256_t0": ("mill256", 640, 480, "path", 0.0, 1.0),
    # mirrored pose: forward.y clamped to -0.001 -> segment 0, ITERATION_DIRECTION +1
    "mill256_t0_down": ("mill256", 640, 480, "pose", ((-0.1, 0.5, -0.1), (0.01, 45.0, 0.0)), 1.0),
    "mill256_t03": ("mill256", 640, 480, "path", 0.3, 1.0),
    "mill256_t05": ("mill256", 640, 480, "path", 0.5, 1.0),
    # VP on screen, 4 segments, R = 2(W+H)
    "mill256_t075": ("mill256", 640, 480, "path", 0.75, 1.0),
    # camera roll (euler z) from the benchmark path
    "mill256_t09_roll": ("mill256", 640, 480, "path", 0.9, 1.0),
    "mill256_t11": ("mill256", 640, 480, "path", 1.1, 1.0),
    # inside the model looking up (inverse iteration, VP below... above the screen centre)
    "mill256_inside_up": ("mill256", 320, 240, "pose", ((0.5, 0.3, 0.5), (-40.0, 30.0, 0.0)), 1.0),
    "mill256_inside_down": ("mill256", 320, 240, "pose", ((0.55, 0.9, 0.45), (70.0, 200.0, 0.0)), 1.0),
    # BASELINE.json config 2: mill.obj 512^3, 1920x1080, VP on screen (benchmark key t=0.75), R = 6000
    "mill512_t075_1080p": ("mill512", 1920, 1080, "path", 0.75, 1.0),
    # procedural worlds: every column is non-empty, multi-run columns (floating slabs), LOD chain reached via lodError
    "proc256_t0_lod8": ("proc256", 640, 480, "path", 0.0, 8.0),
    "proc256_t04_lod8": ("proc256", 640, 480, "path", 0.4, 8.0),
    "proc256_t075_lod8": ("proc256", 640, 480, "path", 0.75, 8.0),
    "proc256_t075_lod1": ("proc256", 640, 480, "path", 0.75, 1.0),
    "proc256_low_lod10": ("proc256", 512, 384, "pose", ((0.2, 0.45, 0.3), (8.0, 60.0, 0.0)), 10.0),
    "proc256_up_lod4": ("proc256", 512, 384, "pose", ((0.5, 0.5, 0.5), (-25.0, 300.0, 0.0)), 4.0),
    "proc128x256x64_t06": ("proc128x256x64", 400, 300, "path", 0.6, 4.0),
}


def scene_frame(name: str):
    world, W, H, kind, args, lod_error = SCENES[name]
    ws = load_world(world)
    if kind == "path":
        fr = benchmark_frame(ws, W, H, args, lod_error)
    else:
        frac, eul = args
        pos = [frac[i] * ws.dims[i] for i in range(3)]
        fr = make_frame(ws, W, H, pos, eul, lod_error)
    return ws, fr, W, H


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def used_rows(frame: host.Frame):
    """(rays used in the top-down buffer, rays used in the left-right buffer), RenderManager.cs:322-323."""
    rc = [max(0, s.RayCount) for s in frame.segments]
    return rc[0] + rc[1], rc[2] + rc[3]
