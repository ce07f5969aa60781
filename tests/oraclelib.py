"""Test-side binding of the CPU oracle (oracle/libcvx_oracle.so).

TEST INFRASTRUCTURE ONLY.  The product package (cpuvox_amd/) never imports
this module or anything under oracle/; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, and only as the checker / baseline.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LOD_LEVELS = 6


class OrcWorld(C.Structure):
    _fields_ = [
        ("storage", C.c_void_p),
        ("dimX", C.c_int32),
        ("dimY", C.c_int32),
        ("dimZ", C.c_int32),
        ("lod", C.c_int32),
        ("columnCount", C.c_int32),
    ]


class OrcCounters(C.Structure):
    _fields_ = [
        ("S", C.c_int64),
        ("E", C.c_int64),
        ("C", C.c_int64),
        ("P", C.c_int64),
        ("R", C.c_int64),
        ("lodVisits", C.c_int64 * LOD_LEVELS),
        ("continuations", C.c_int64),
    ]

    def algorithmic_bytes(self) -> int:
        """B = 12*S + 4*E + 4*C + 4*P + 80*R."""
        return 12 * self.S + 4 * self.E + 4 * self.C + 4 * self.P + 80 * self.R

    def as_dict(self):
        return {"S": self.S, "E": self.E, "C": self.C, "P": self.P, "R": self.R,
                "lodVisits": list(self.lodVisits), "continuations": self.continuations,
                "bytes": self.algorithmic_bytes()}


_lib = None


def build(force: bool = False) -> str:
    if os.environ.get("CVX_ORACLE_LIB"):  # another build of the oracle (the sanitizer build, tests/test_sanitizers.py)
        return os.environ["CVX_ORACLE_LIB"]
    path = os.path.join(ORACLE_DIR, "libcvx_oracle.so")
    src = os.path.join(ORACLE_DIR, "cvx_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libcvx_oracle.so"], stdout=subprocess.DEVNULL)
    return path


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_draw_segments.restype = C.c_int
        L.orc_draw_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


_lib_fast = None


def lib_fast():
    """The same C file built the way Burst builds the reference's jobs (FloatMode.Fast, DrawSegmentRayJob.cs:11,48,86,155):
    -O3 -march=native -ffast-math, FMA contraction allowed.  NOT a parity build -- its pixels may differ from the strict
    port's, and it is only ever timed (bench.py cpu_baseline.fast_math).  Compiled on the machine it runs on (-march=native)
    into a private temporary directory; None when no compiler is available."""
    global _lib_fast
    if _lib_fast is None:
        import tempfile

        out_dir = tempfile.mkdtemp(prefix="cvx_oracle_fast_")
        path = os.path.join(out_dir, "libcvx_oracle_fast.so")
        cmd = [os.environ.get("CC", "gcc"), "-std=c11", "-O3", "-march=native", "-ffast-math", "-fopenmp", "-fPIC", "-shared", "-o", path,
               os.path.join(ORACLE_DIR, "cvx_oracle.c"), "-lm"]
        try:
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except (OSError, subprocess.CalledProcessError):
            _lib_fast = False
            return None
        L = C.CDLL(path)
        L.orc_draw_segments.restype = C.c_int
        L.orc_draw_segments.argtypes = lib().orc_draw_segments.argtypes
        L.orc_max_threads.restype = C.c_int
        _lib_fast = L
    return _lib_fast or None


def orc_worlds(world_set):
    """orc_world[6] describing a cpuvox_amd.host.WorldSet (no copies)."""
    arr = (OrcWorld * LOD_LEVELS)()
    for i in range(LOD_LEVELS):
        info = world_set.info(i)
        arr[i].storage = info.storage
        arr[i].dimX, arr[i].dimY, arr[i].dimZ = info.dimX, info.dimY, info.dimZ
        arr[i].lod = info.lod
        arr[i].columnCount = info.columnCount
    return arr


def raybuffer_shapes(width: int, height: int):
    """(rays, pixels-per-ray) of the two raybuffers, RenderManager.cs:35-36."""
    return (width + 2 * height, height), (2 * width + height, width)


def draw_segments(world_set, frame, width: int, height: int, threads: int = 0, clear: int = 0, counters: bool = True, out=None, library=None):
    """Run the oracle's DrawSegments on a cpuvox_amd.host.Frame.

    Returns (topDown[rays, H] uint32, leftRight[rays, W] uint32, OrcCounters).
    Buffers are pre-cleared to `clear` (the reference leaves stale pixels).
    """
    (td_rays, td_w), (lr_rays, lr_w) = raybuffer_shapes(width, height)
    if out is not None:  # preallocated (and not cleared): timing runs
        td, lr = out
        assert td.shape == (td_rays, td_w) and lr.shape == (lr_rays, lr_w)
    else:
        td = np.full((td_rays, td_w), clear, dtype=np.uint32)
        lr = np.full((lr_rays, lr_w), clear, dtype=np.uint32)
    worlds = orc_worlds(world_set)
    cnt = OrcCounters()
    vp = (C.c_float * 2)(*frame.vanishingPointScreenSpace)
    rc = (library or lib()).orc_draw_segments(C.addressof(frame.segments), C.addressof(worlds), C.addressof(frame.camera), width, height,
                                 C.addressof(vp), td.ctypes.data, lr.ctypes.data, threads,
                                 C.addressof(cnt) if counters else None)
    if rc < 0:
        raise RuntimeError("orc_draw_segments failed")
    return td, lr, cnt


def blit_reference(frame, td: np.ndarray, lr: np.ndarray, width: int, height: int, clear: int = 0) -> np.ndarray:
    """Phase-2 rule (RenderManager.BlitSegments RenderManager.cs:199-256 +
    RayBufferBlit.shader:48-64) evaluated at pixel centres, numpy.

    For segment s with triangle (VP = a, MaxScreen = b, MinScreen = q): per-frame edge functions
    inv = 1 / den, A0 = (b.y - q.y) inv, B0 = (q.x - b.x) inv, A1 = (q.y - a.y) inv, B1 = (a.x - q.x) inv (float32),
    per pixel centre c: w_vp = A0 (c.x - q.x) + B0 (c.y - q.y), w_max = A1 (c.x - q.x) + B1 (c.y - q.y),
    w_min = 1 - w_vp - w_max (the same float32 operations, in the same order, as blit_pixel in cvx_kernels.h); inside when all >= 0;
    x = w_max / (w_max + w_min); ray = min(floor(x * RayCount), RayCount-1);
    colour = raybuffer[ray + offset][screen y (segments 0,1) or screen x (2,3)].
    Returns image[H, W] uint32, row 0 = bottom (Unity screen space).
    """
    img = np.full((height, width), clear, dtype=np.uint32)
    done = np.zeros((height, width), dtype=bool)
    ys, xs = np.mgrid[0:height, 0:width]
    cx = xs.astype(np.float32) + np.float32(0.5)
    cy = ys.astype(np.float32) + np.float32(0.5)
    vpx, vpy = np.float32(frame.vanishingPointScreenSpace[0]), np.float32(frame.vanishingPointScreenSpace[1])
    for s in range(4):
        seg = frame.segments[s]
        rcount = seg.RayCount
        if rcount <= 0:
            continue
        ax, ay = vpx, vpy
        bx, by = np.float32(seg.MaxScreen[0]), np.float32(seg.MaxScreen[1])
        qx, qy = np.float32(seg.MinScreen[0]), np.float32(seg.MinScreen[1])
        den = (by - qy) * (ax - qx) + (qx - bx) * (ay - qy)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv = np.float32(1.0) / den
            a0, b0 = (by - qy) * inv, (qx - bx) * inv
            a1, b1 = (qy - ay) * inv, (ax - qx) * inv
            dx, dy = cx - qx, cy - qy
            w_vp = a0 * dx + b0 * dy
            w_max = a1 * dx + b1 * dy
            w_min = np.float32(1.0) - w_vp - w_max
        inside = (w_vp >= 0) & (w_max >= 0) & (w_min >= 0) & ~done
        with np.errstate(divide="ignore", invalid="ignore"):
            x = w_max / (w_max + w_min)
        ray = np.clip(np.floor(x * np.float32(rcount)), 0, rcount - 1)
        ray = np.where(np.isfinite(ray), ray, 0).astype(np.int64)
        offset = 0
        if s == 1:
            offset = frame.segments[0].RayCount
        if s == 3:
            offset = frame.segments[2].RayCount
        if s < 2:
            vals = td[np.clip(ray + offset, 0, td.shape[0] - 1), ys]
        else:
            vals = lr[np.clip(ray + offset, 0, lr.shape[0] - 1), xs]
        img[inside] = vals[inside]
        done |= inside
    return img


def blit_reference_f64(frame, td: np.ndarray, lr: np.ndarray, width: int, height: int, clear: int = 0):
    """Independent statement of the Phase-2 rule in float64 with plain barycentric weights (three cross products divided by the triangle's doubled
    area: none of blit_reference's / the kernel's float32 edge-function arithmetic).  Returns (image, margin): margin[y, x] = the smallest |weight|
    over the four segments' triangles at that pixel centre -- pixels with a margin above ~1e-4 cannot be assigned differently by float32 rounding, so
    the GPU image has to equal this one there (the float32 reference and the kernel share their formula)."""
    img = np.full((height, width), clear, dtype=np.uint32)
    done = np.zeros((height, width), dtype=bool)
    margin = np.full((height, width), np.inf)
    ys, xs = np.mgrid[0:height, 0:width]
    cx, cy = xs + 0.5, ys + 0.5
    ax, ay = float(frame.vanishingPointScreenSpace[0]), float(frame.vanishingPointScreenSpace[1])
    for s in range(4):
        seg = frame.segments[s]
        rcount = seg.RayCount
        if rcount <= 0:
            continue
        bx, by = float(seg.MaxScreen[0]), float(seg.MaxScreen[1])
        qx, qy = float(seg.MinScreen[0]), float(seg.MinScreen[1])
        area = (bx - ax) * (qy - ay) - (qx - ax) * (by - ay)
        if area == 0.0:
            continue
        # weights of a (the vanishing point), b (MaxScreen), q (MinScreen)
        w_a = ((bx - cx) * (qy - cy) - (qx - cx) * (by - cy)) / area
        w_b = ((qx - cx) * (ay - cy) - (ax - cx) * (qy - cy)) / area
        w_q = 1.0 - w_a - w_b
        margin = np.minimum(margin, np.minimum(np.abs(w_a), np.minimum(np.abs(w_b), np.abs(w_q))))
        inside = (w_a >= 0) & (w_b >= 0) & (w_q >= 0) & ~done
        with np.errstate(divide="ignore", invalid="ignore"):
            x = w_b / (w_b + w_q)
        ray = np.clip(np.floor(x * rcount), 0, rcount - 1)
        ray = np.where(np.isfinite(ray), ray, 0).astype(np.int64)
        # a pixel whose ray coordinate sits on a ray boundary may pick the neighbouring ray in float32: part of the margin
        with np.errstate(invalid="ignore"):
            frac = x * rcount - np.floor(x * rcount)
        margin = np.where(inside, np.minimum(margin, np.minimum(frac, 1.0 - frac) / max(1, rcount) * 8.0), margin)
        offset = frame.segments[0].RayCount if s == 1 else (frame.segments[2].RayCount if s == 3 else 0)
        if s < 2:
            vals = td[np.clip(ray + offset, 0, td.shape[0] - 1), ys]
        else:
            vals = lr[np.clip(ray + offset, 0, lr.shape[0] - 1), xs]
        img[inside] = vals[inside]
        done |= inside
    return img, margin


def argb_to_rgb8(img: np.ndarray) -> np.ndarray:
    """uint32 ARGB32 (bytes A,R,G,B in memory) -> uint8[..., 3] RGB."""
    b = img.view(np.uint8).reshape(img.shape + (4,))
    return np.ascontiguousarray(b[..., 1:4])


def write_png(path: str, rgb: np.ndarray) -> None:
    """Minimal PNG writer (zlib) for human inspection of renders; rgb[H, W, 3] with row 0 at the TOP."""
    import struct
    import zlib

    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
