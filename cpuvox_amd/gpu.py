"""ctypes binding of libcpuvox_gpu.so (include/cpuvox_gpu.h): the C-ABI
drop-in for RenderManager.DrawSegments (Assets/Code/RenderManager.cs:258-372)
running as HIP kernels on MI355X.

There is NO CPU fallback: if the library or a HIP device is missing every
call raises.  (The CPU oracle lives under oracle/ and is test infrastructure;
this package never imports it.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .host import CameraData, Frame, LOD_LEVELS, SegmentData, WorldSet

_HERE = os.path.dirname(os.path.abspath(__file__))

RAYBUFFER_TOPDOWN = 0
RAYBUFFER_LEFTRIGHT = 1
DRAW_SYNC = 0
DRAW_ASYNC = 1
LATENCY_AUTO, LATENCY_NEVER, LATENCY_ALWAYS = 0, 1, 2  # cvx_set_latency_kernel

# every symbol include/cpuvox_gpu.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "cvx_create", "cvx_destroy", "cvx_last_error", "cvx_set_stream", "cvx_world_upload", "cvx_set_resolution",
    "cvx_set_buffer_count", "cvx_draw_segments", "cvx_draw_segments_batch", "cvx_set_shard", "cvx_set_latency_kernel", "cvx_synchronize",
    "cvx_clear_raybuffer", "cvx_read_raybuffer", "cvx_blit_segments", "cvx_blit_segments_batch", "cvx_raybuffer_device_ptr",
    "cvx_screen_device_ptr", "cvx_last_draw_ms", "cvx_enable_counters", "cvx_get_counters",
    "cvx_get_raybuffer_layout", "cvx_version", "cvx_bind_raybuffers", "cvx_draw_time_stats", "cvx_copy_rows", "cvx_draw_segments_placed",
    "cvx_world_downsample", "cvx_world_build_lods", "cvx_free",
    "cvx_shard_plan_create", "cvx_shard_plan_destroy", "cvx_shard_plan_tile_count", "cvx_shard_plan_sections", "cvx_shard_plan_tile_out", "cvx_shard_plan_transfer",
    "cvx_comm_unique_id", "cvx_comm_create", "cvx_comm_create_timeout", "cvx_comm_destroy", "cvx_exchange",
    "cvx_image_plan_create", "cvx_image_plan_destroy", "cvx_image_plan_tile_count", "cvx_image_plan_sizes", "cvx_image_plan_transfer",
    "cvx_image_plan_tile_out", "cvx_image_pack", "cvx_image_exchange", "cvx_image_unpack",
]
# include/cpuvox_gpu_diag.h: only the experiment / profiling builds export these (cpuvox_amd.gpu.use_library(".../libcpuvox_gpu_exp.so"))
DIAG_EXPORTS = ["cvx_selftest_math", "cvx_selftest_scan", "cvx_selftest_lone", "cvx_debug_occupancy", "cvx_debug_section_cycles", "cvx_debug_section_histogram"]


class Counters(C.Structure):
    _fields_ = [("S", C.c_int64), ("E", C.c_int64), ("C", C.c_int64), ("P", C.c_int64), ("R", C.c_int64),
                ("lodVisits", C.c_int64 * LOD_LEVELS)]

    def algorithmic_bytes(self) -> int:
        return 12 * self.S + 4 * self.E + 4 * self.C + 4 * self.P + 80 * self.R

    def as_dict(self):
        return {"S": self.S, "E": self.E, "C": self.C, "P": self.P, "R": self.R,
                "lodVisits": list(self.lodVisits), "bytes": self.algorithmic_bytes()}


class RaybufferLayout(C.Structure):
    _fields_ = [("width", C.c_int32), ("rayCapacity", C.c_int32), ("tileRays", C.c_int32),
                ("tileCapacity", C.c_int32), ("tileBytes", C.c_int64)]


_lib = None


def lib_path() -> str:
    # CVX_GPU_LIB selects another build of the same ABI (e.g. the diagnostic libcpuvox_gpu_prof.so); never a CPU path
    return os.environ.get("CVX_GPU_LIB") or os.path.join(_HERE, "libcpuvox_gpu.so")


def _load_torch_hip_runtime_first() -> None:
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 under unversioned file names; its
    libraries ask for them by those names, so a copy of /opt/rocm's runtime that this library pulled in earlier (SONAME
    libamdhip64.so.7) is not recognised as the same thing, a second runtime is loaded and finds no GPU ("No HIP GPUs are
    available").  The other order works: the loader matches this library's libamdhip64.so.7 against the SONAME of torch's copy.
    Hosts that use both (bench.py, cpuvox_amd.dist, the tests that hand torch tensors to the C ABI) therefore need torch loaded
    first; without torch installed nothing happens.  CVX_NO_TORCH_PRELOAD=1 skips this (a host that never touches torch)."""
    import sys

    if "torch" in sys.modules or os.environ.get("CVX_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib() -> C.CDLL:
    """Load libcpuvox_gpu.so (built in-tree by cpuvox_amd/csrc/Makefile); fails loudly when missing."""
    global _lib
    if _lib is None:
        _lib = _bind(lib_path())
    return _lib


def use_library(path: str | None) -> None:
    """Diagnostics / tests: make another build of the same ABI (an experiment or profiling build) the library that contexts
    created FROM NOW ON talk to; None returns to the default.  Contexts of the previous library must be closed first."""
    global _lib
    _lib = _bind(path) if path else None


def _bind(path: str) -> C.CDLL:
    if True:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the HIP extension is required (build with `make -C cpuvox_amd/csrc`); there is no CPU fallback")
        _load_torch_hip_runtime_first()
        L = C.CDLL(path)
        L.cvx_version.restype = C.c_char_p
        L.cvx_last_error.restype = C.c_char_p
        L.cvx_last_error.argtypes = [C.c_void_p]
        L.cvx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.cvx_destroy.argtypes = [C.c_void_p]
        L.cvx_destroy.restype = None
        L.cvx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.cvx_world_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        L.cvx_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cvx_set_buffer_count.argtypes = [C.c_void_p, C.c_int]
        L.cvx_draw_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.cvx_draw_segments_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.cvx_draw_segments_placed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.cvx_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cvx_set_latency_kernel.argtypes = [C.c_void_p, C.c_int]
        L.cvx_synchronize.argtypes = [C.c_void_p]
        L.cvx_clear_raybuffer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32]
        L.cvx_read_raybuffer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.cvx_blit_segments.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cvx_blit_segments_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.cvx_raybuffer_device_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.cvx_screen_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.cvx_last_draw_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.cvx_draw_time_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]
        L.cvx_enable_counters.argtypes = [C.c_void_p, C.c_int]
        L.cvx_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.cvx_get_raybuffer_layout.argtypes = [C.c_void_p, C.c_int, C.POINTER(RaybufferLayout)]
        L.cvx_bind_raybuffers.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.cvx_copy_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        if hasattr(L, "cvx_selftest_math"):  # a diagnostics build (include/cpuvox_gpu_diag.h)
            L.cvx_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
            L.cvx_debug_section_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
            L.cvx_debug_occupancy.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int)]
            L.cvx_selftest_math.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            L.cvx_selftest_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
            L.cvx_selftest_lone.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_world_downsample.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        L.cvx_world_build_lods.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.cvx_free.argtypes = [C.c_void_p]
        L.cvx_free.restype = None
        L.cvx_shard_plan_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cvx_shard_plan_destroy.argtypes = [C.c_void_p]
        L.cvx_shard_plan_destroy.restype = None
        L.cvx_shard_plan_tile_count.argtypes = [C.c_void_p]
        L.cvx_shard_plan_tile_count.restype = C.c_int64
        L.cvx_shard_plan_sections.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_shard_plan_tile_out.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_shard_plan_transfer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cvx_comm_unique_id.argtypes = [C.c_void_p]
        L.cvx_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cvx_comm_create_timeout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)]
        L.cvx_comm_destroy.argtypes = [C.c_void_p]
        L.cvx_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_image_plan_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cvx_image_plan_destroy.argtypes = [C.c_void_p]
        L.cvx_image_plan_destroy.restype = None
        L.cvx_image_plan_tile_count.argtypes = [C.c_void_p]
        L.cvx_image_plan_tile_count.restype = C.c_int64
        L.cvx_image_plan_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.cvx_image_plan_transfer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cvx_image_plan_tile_out.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_image_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_image_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvx_image_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


class CvxError(RuntimeError):
    pass


class Context:
    """What RenderManager owns on the GPU side (RenderManager.cs:12-56): the
    uploaded world LODs and the raybuffer pairs."""

    def __init__(self, device: int = 0, buffer_count: int = 2):
        self._h = C.c_void_p()
        rc = lib().cvx_create(device, C.byref(self._h))
        if rc != 0:
            raise CvxError(f"cvx_create({device}) failed ({rc}): {lib().cvx_last_error(None).decode()}")
        self.width = self.height = 0
        if buffer_count != 2:
            self._check(lib().cvx_set_buffer_count(self._h, buffer_count))
        self.buffer_count = buffer_count

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise CvxError(f"cpuvox_gpu error {rc}: {lib().cvx_last_error(self._h).decode()}")

    def close(self) -> None:
        if self._h:
            lib().cvx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup ------------------------------------------------------------
    def set_stream(self, hip_stream: int | None) -> None:
        self._check(lib().cvx_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def upload_world(self, world_set: WorldSet) -> None:
        """`fixed (World* worldPtr = worldLODs)` (RenderManager.cs:155): all LOD_LEVELS levels."""
        for lod in range(world_set.lod_count):
            i = world_set.info(lod)
            self._check(lib().cvx_world_upload(self._h, lod, i.storage, i.byteLength, i.dimX, i.dimY, i.dimZ, i.columnCount))

    def downsample(self, world_set: WorldSet, lod: int, extra_lods: int):
        """World.DownSample(extraLods) (World.cs:45) of level `lod` on the device.  Returns (blob bytes in the reference's
        storage layout, ColumnCount of the new level, voxel count, device milliseconds)."""
        i = world_set.info(lod)
        out, nbytes, columns, voxels, ms = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_int64(), C.c_float()
        self._check(lib().cvx_world_downsample(self._h, i.storage, i.byteLength, i.dimX, i.dimY, i.dimZ, i.lod, i.columnCount, extra_lods,
                                               C.byref(out), C.byref(nbytes), C.byref(columns), C.byref(voxels), C.byref(ms)))
        try:
            blob = C.string_at(out.value, nbytes.value)
        finally:
            lib().cvx_free(out)
        return blob, columns.value, voxels.value, ms.value

    def build_lods(self, world_set: WorldSet, levels: int = LOD_LEVELS) -> WorldSet:
        """UnityManager.cs:328-331 (`worldLODs[i] = worldLODs[0].DownSample(i)`) with the downsampling on the device:
        a new world set with LOD 0 taken from `world_set` and LOD 1..levels-1 built by cvx_world_downsample."""
        i = world_set.info(0)
        n = levels - 1
        outs, sizes, columns, ms = (C.c_void_p * n)(), (C.c_int64 * n)(), (C.c_int32 * n)(), C.c_float()
        self._check(lib().cvx_world_build_lods(self._h, i.storage, i.byteLength, i.dimX, i.dimY, i.dimZ, i.columnCount, n, outs, sizes, columns, C.byref(ms)))
        try:
            blobs = [world_set.storage(0)] + [C.string_at(outs[k], sizes[k]) for k in range(n)]
        finally:
            for k in range(n):
                lib().cvx_free(outs[k])
        self.last_build_lods_ms = ms.value
        return WorldSet.from_blobs(world_set.dims, blobs)

    def set_resolution(self, width: int, height: int) -> None:
        """RenderManager.SetResolution (RenderManager.cs:94-109)."""
        self._check(lib().cvx_set_resolution(self._h, width, height))
        self.width, self.height = width, height

    def set_shard(self, index: int, count: int) -> None:
        self._check(lib().cvx_set_shard(self._h, index, count))

    def set_latency_kernel(self, mode: int) -> None:
        """Which kernel a draw goes to: LATENCY_AUTO (few rays -> one wave per ray), LATENCY_NEVER, LATENCY_ALWAYS (include/cpuvox_gpu.h)."""
        self._check(lib().cvx_set_latency_kernel(self._h, mode))

    def enable_counters(self, enable: bool) -> None:
        self._check(lib().cvx_enable_counters(self._h, int(enable)))

    # -- the hot path -------------------------------------------------------
    def draw_segments(self, frame: Frame, buffer_index: int = 0, flags: int = DRAW_SYNC) -> None:
        """RenderManager.DrawSegments (RenderManager.cs:258-372)."""
        vp = (C.c_float * 2)(*frame.vanishingPointScreenSpace)
        self._check(lib().cvx_draw_segments(self._h, C.addressof(frame.segments), C.addressof(frame.camera), self.width, self.height,
                                            C.addressof(vp), buffer_index, flags))

    def draw_segments_batch(self, frames, first_buffer_index: int = 0, flags: int = DRAW_SYNC) -> None:
        n = len(frames)
        segs = (SegmentData * (4 * n))()
        cams = (CameraData * n)()
        vps = (C.c_float * (2 * n))()
        for i, f in enumerate(frames):
            for s in range(4):
                segs[4 * i + s] = f.segments[s]
            cams[i] = f.camera
            vps[2 * i] = f.vanishingPointScreenSpace[0]
            vps[2 * i + 1] = f.vanishingPointScreenSpace[1]
        self._batch_keepalive = (segs, cams, vps)
        self._check(lib().cvx_draw_segments_batch(self._h, n, C.addressof(segs), C.addressof(cams), self.width, self.height,
                                                  C.addressof(vps), first_buffer_index, flags))

    def pack_batch(self, frames):
        """Marshal frames once; returns an opaque object for draw_packed (keeps bench loops free of Python overhead)."""
        return pack_frames(frames)

    def draw_packed(self, packed, first_buffer_index: int = 0, flags: int = DRAW_SYNC) -> None:
        n, segs, cams, vps = packed
        self._check(lib().cvx_draw_segments_batch(self._h, n, C.addressof(segs), C.addressof(cams), self.width, self.height,
                                                  C.addressof(vps), first_buffer_index, flags))

    def draw_placed(self, packed, tile_out: np.ndarray, flags: int = DRAW_SYNC) -> None:
        """cvx_draw_segments_placed: packed = pack_batch(frames), tile_out = uint64 device address per tile (0 = skip)."""
        n, segs, cams, vps = packed
        tile_out = np.ascontiguousarray(tile_out, dtype=np.uint64)
        self._check(lib().cvx_draw_segments_placed(self._h, n, C.addressof(segs), C.addressof(cams), self.width, self.height,
                                                   C.addressof(vps), tile_out.size, tile_out.ctypes.data, flags))

    def synchronize(self) -> None:
        self._check(lib().cvx_synchronize(self._h))

    def last_draw_ms(self) -> float:
        ms = C.c_float()
        self._check(lib().cvx_last_draw_ms(self._h, C.byref(ms)))
        return ms.value

    def draw_time_stats(self, reset: bool = False):
        """(total kernel ms, draws) since the last reset; HIP events on the context's stream."""
        ms = C.c_double()
        n = C.c_int()
        self._check(lib().cvx_draw_time_stats(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def counters(self) -> Counters:
        out = Counters()
        self._check(lib().cvx_get_counters(self._h, C.byref(out)))
        return out

    # -- raybuffers ---------------------------------------------------------
    def clear_raybuffer(self, buffer_index: int, which: int, argb: int = 0) -> None:
        """RenderManager.ClearRayBuffer (RenderManager.cs:58-92)."""
        self._check(lib().cvx_clear_raybuffer(self._h, buffer_index, which, argb & 0xFFFFFFFF))

    def clear_raybuffers(self, buffer_index: int = 0, argb: int = 0) -> None:
        self.clear_raybuffer(buffer_index, RAYBUFFER_TOPDOWN, argb)
        self.clear_raybuffer(buffer_index, RAYBUFFER_LEFTRIGHT, argb)

    def read_raybuffer(self, buffer_index: int, which: int, first_ray: int = 0, ray_count: int | None = None) -> np.ndarray:
        """Rows of a raybuffer in the reference's ray-major layout (RayBuffer.cs:121-128)."""
        W, H = self.width, self.height
        width = H if which == RAYBUFFER_TOPDOWN else W
        cap = W + 2 * H if which == RAYBUFFER_TOPDOWN else 2 * W + H
        if ray_count is None:
            ray_count = cap - first_ray
        out = np.empty((ray_count, width), dtype=np.uint32)
        self._check(lib().cvx_read_raybuffer(self._h, buffer_index, which, first_ray, ray_count, out.ctypes.data))
        return out

    def blit_segments(self, buffer_index: int = 0, to_host: bool = True):
        """RenderManager.BlitSegments + RayBufferBlit.shader (Phase 2) -> image[H, W] uint32, row 0 = bottom."""
        if not to_host:
            self._check(lib().cvx_blit_segments(self._h, buffer_index, None))
            return None
        out = np.empty((self.height, self.width), dtype=np.uint32)
        self._check(lib().cvx_blit_segments(self._h, buffer_index, out.ctypes.data))
        return out

    def blit_segments_batch(self, first_buffer: int, frame_count: int, dst_device: int | None = None) -> int:
        """Phase 2 of `frame_count` frames (buffers first_buffer ..) in ONE launch, asynchronous on the context's stream.  Image f goes to
        dst_device + f * H * W * 4 (a device address, e.g. a torch tensor's data_ptr()), or into an array the context owns; returns
        the device address of image 0."""
        p = C.c_void_p()
        self._check(lib().cvx_blit_segments_batch(self._h, first_buffer, frame_count, C.c_void_p(dst_device) if dst_device else None, C.byref(p)))
        return p.value

    def raybuffer_device_ptr(self, buffer_index: int, which: int):
        p = C.c_void_p()
        n = C.c_int64()
        self._check(lib().cvx_raybuffer_device_ptr(self._h, buffer_index, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_raybuffers(self, td_ptr: int, td_bytes: int, lr_ptr: int, lr_bytes: int) -> None:
        """Render into caller-owned device memory (e.g. torch tensors used by a RCCL exchange)."""
        self._check(lib().cvx_bind_raybuffers(self._h, C.c_void_p(td_ptr), td_bytes, C.c_void_p(lr_ptr), lr_bytes))

    def copy_rows(self, hip_stream: int | None, to_packed: bool, span_count: int, spans_ptr: int, packed_ptr: int) -> None:
        """Pack / unpack tile pixel rows between the pools and a staging buffer (multi-GPU exchange payload)."""
        self._check(lib().cvx_copy_rows(self._h, C.c_void_p(hip_stream or 0), int(to_packed), span_count, C.c_void_p(spans_ptr), C.c_void_p(packed_ptr)))

    def raybuffer_layout(self, which: int) -> RaybufferLayout:
        out = RaybufferLayout()
        self._check(lib().cvx_get_raybuffer_layout(self._h, which, C.byref(out)))
        return out

    @staticmethod
    def _diag(name: str):
        """An entry point of include/cpuvox_gpu_diag.h: present in the experiment / profiling builds only."""
        if not hasattr(lib(), name):
            raise RuntimeError(f"{name} is a diagnostic of libcpuvox_gpu_exp.so / the profiling builds (include/cpuvox_gpu_diag.h); "
                               "the product library does not export it: cpuvox_amd.gpu.use_library(<diagnostics build>) first")
        return getattr(lib(), name)

    def debug_occupancy(self, lds_bytes: int) -> int:
        n = C.c_int()
        self._diag("cvx_debug_occupancy")
        self._check(lib().cvx_debug_occupancy(self._h, lds_bytes, C.byref(n)))
        return n.value

    def debug_section_cycles(self, reset: bool = False):
        """Diagnostic build only: wave cycles per render-kernel section (include/cpuvox_gpu_diag.h)."""
        out = (C.c_uint64 * 32)()
        self._check(self._diag("cvx_debug_section_cycles")(self._h, out, int(reset)))
        return list(out)

    def debug_section_histogram(self, reset: bool = False):
        """Counting diagnostic build only: [section][bucket of 8 lanes] executions (include/cpuvox_gpu_diag.h)."""
        out = (C.c_uint64 * 128)()
        self._check(self._diag("cvx_debug_section_histogram")(self._h, out, int(reset)))
        return [list(out[i * 8:(i + 1) * 8]) for i in range(16)]

    def selftest_math(self, op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.empty_like(a)
        self._check(self._diag("cvx_selftest_math")(self._h, op, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def selftest_lone(self, op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        """The latency kernel's inline-assembly primitives on the caller's values (diagnostics build): op 0 the crossing chains, op 1 v_writelane."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        waves = b.size
        out = np.empty(waves * (128 if op == 0 else 64), dtype=np.float32)
        self._check(self._diag("cvx_selftest_lone")(self._h, op, waves, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def selftest_scan(self, values: np.ndarray):
        """Diagnostics build only: (exclusive prefix sums modulo 2^32, 64-bit total) of uint32 values by the device scan of cvx_world_downsample."""
        v = np.ascontiguousarray(values, dtype=np.uint32).copy()
        total = C.c_uint64()
        self._check(self._diag("cvx_selftest_scan")(self._h, v.size, v.ctypes.data, C.byref(total)))
        return v, int(total.value)


def pack_frames(frames):
    """(n, SegmentData[4 n], CameraData[n], float[2 n]) of a list of host frames: the argument arrays of cvx_draw_segments_batch /
    cvx_shard_plan_create.  Needs no context and no GPU."""
    n = len(frames)
    segs = (SegmentData * (4 * n))()
    cams = (CameraData * n)()
    vps = (C.c_float * (2 * n))()
    for i, f in enumerate(frames):
        for s in range(4):
            segs[4 * i + s] = f.segments[s]
        cams[i] = f.camera
        vps[2 * i] = f.vanishingPointScreenSpace[0]
        vps[2 * i + 1] = f.vanishingPointScreenSpace[1]
    return (n, segs, cams, vps)


class NativeShardPlan:
    """cvx_shard_plan_* (include/cpuvox_gpu.h): the shard plan of one batch of frames as the library computes it -- host
    arithmetic only, no GPU needed.  `packed` = Context.pack_batch(frames) or (n, segments, cameras, vps)."""

    def __init__(self, packed, width: int, height: int, rank: int, world_size: int):
        n, segs, _cams, vps = packed
        self._keep = packed
        self._h = C.c_void_p()
        rc = lib().cvx_shard_plan_create(n, C.addressof(segs), C.addressof(vps), width, height, rank, world_size, C.byref(self._h))
        if rc != 0:
            raise CvxError(f"cvx_shard_plan_create failed ({rc}): {lib().cvx_last_error(None).decode()}")
        self.rank, self.N = rank, world_size
        self.tile_count = int(lib().cvx_shard_plan_tile_count(self._h))
        self.send_start = np.zeros(world_size + 1, dtype=np.int64)
        self.disp_start = np.zeros(world_size + 1, dtype=np.int64)
        lib().cvx_shard_plan_sections(self._h, self.send_start.ctypes.data, self.disp_start.ctypes.data)
        self.send_total, self.disp_total = int(self.send_start[-1]), int(self.disp_start[-1])

    def tile_out(self, send_ptr: int, disp_ptr: int) -> np.ndarray:
        out = np.zeros(max(1, self.tile_count), dtype=np.uint64)
        lib().cvx_shard_plan_tile_out(self._h, C.c_void_p(send_ptr), C.c_void_p(disp_ptr), out.ctypes.data)
        return out[: self.tile_count]

    def transfer(self, peer: int):
        """(send row, send rows, receive row, receive rows) between this rank and `peer` -- what cvx_exchange moves."""
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        rc = lib().cvx_shard_plan_transfer(self._h, peer, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        if rc != 0:
            raise CvxError(f"cvx_shard_plan_transfer failed ({rc})")
        return a.value, b.value, c.value, d.value

    def exchange(self, ctx: "Context", comm: int, hip_stream: int | None, send_ptr: int, disp_ptr: int) -> None:
        """cvx_exchange: grouped ncclSend / ncclRecv of this batch's sections, enqueued on hip_stream."""
        ctx._check(lib().cvx_exchange(ctx._h, self._h, C.c_void_p(comm), C.c_void_p(hip_stream or 0), C.c_void_p(send_ptr), C.c_void_p(disp_ptr)))

    def close(self) -> None:
        if self._h:
            lib().cvx_shard_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImagePlan:
    """cvx_image_plan_* (include/cpuvox_gpu.h): the IMAGE gather of one batch of frames -- every rank blits the pixels of the tiles it
    rendered, the display rank receives W * H pixels per frame in total.  Needs the GPU (pixel counts per rank are made on the device)."""

    def __init__(self, ctx: "Context", packed, width: int, height: int, rank: int, world_size: int):
        n, segs, _cams, vps = packed
        self._keep = packed
        self._h = C.c_void_p()
        ctx._check(lib().cvx_image_plan_create(ctx._h, n, C.addressof(segs), C.addressof(vps), width, height, rank, world_size, C.byref(self._h)))
        self.rank, self.N, self.frames = rank, world_size, n
        self.tile_count = int(lib().cvx_image_plan_tile_count(self._h))
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        lib().cvx_image_plan_sizes(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.local_store_bytes, self.send_pixels, self.recv_pixels, self.images = a.value, b.value, c.value, d.value

    def tile_out(self, local_store_ptr: int) -> np.ndarray:
        out = np.zeros(max(1, self.tile_count), dtype=np.uint64)
        lib().cvx_image_plan_tile_out(self._h, C.c_void_p(local_store_ptr), out.ctypes.data)
        return out[: self.tile_count]

    def transfer(self, peer: int):
        """(first send pixel, send pixels, first receive pixel, receive pixels) between this rank and `peer`."""
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        if lib().cvx_image_plan_transfer(self._h, peer, C.byref(a), C.byref(b), C.byref(c), C.byref(d)) != 0:
            raise CvxError("cvx_image_plan_transfer failed")
        return a.value, b.value, c.value, d.value

    def pack(self, ctx: "Context", hip_stream: int | None, local_store_ptr: int, send_ptr: int, images_ptr: int) -> None:
        ctx._check(lib().cvx_image_pack(ctx._h, self._h, C.c_void_p(hip_stream or 0), C.c_void_p(local_store_ptr), C.c_void_p(send_ptr), C.c_void_p(images_ptr)))

    def exchange(self, ctx: "Context", comm: int, hip_stream: int | None, send_ptr: int, recv_ptr: int) -> None:
        ctx._check(lib().cvx_image_exchange(ctx._h, self._h, C.c_void_p(comm), C.c_void_p(hip_stream or 0), C.c_void_p(send_ptr), C.c_void_p(recv_ptr)))

    def unpack(self, ctx: "Context", hip_stream: int | None, recv_ptr: int, images_ptr: int) -> None:
        ctx._check(lib().cvx_image_unpack(ctx._h, self._h, C.c_void_p(hip_stream or 0), C.c_void_p(recv_ptr), C.c_void_p(images_ptr)))

    def close(self) -> None:
        if self._h:
            lib().cvx_image_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = lib().cvx_comm_unique_id(buf)
    if rc != 0:
        raise CvxError(f"cvx_comm_unique_id failed ({rc}): {lib().cvx_last_error(None).decode()}")
    return buf.raw


def comm_create(ctx: Context, unique_id: bytes, rank: int, world_size: int, timeout_s: float = 180.0) -> int:
    """ncclCommInitRank through the C ABI; raises (CVX_ERR_TIMEOUT) when the clique is not complete after `timeout_s`."""
    comm = C.c_void_p()
    ctx._check(lib().cvx_comm_create_timeout(ctx._h, unique_id, rank, world_size, float(timeout_s), C.byref(comm)))
    return comm.value


def comm_destroy(comm: int) -> None:
    lib().cvx_comm_destroy(C.c_void_p(comm))
