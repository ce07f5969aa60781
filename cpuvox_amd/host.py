"""ctypes binding of libcpuvox_host.so (include/cpuvox_host.h).

Thin by design: the host-side mirror of the reference's managed code (world
building, camera / vanishing point / segment setup, LOD distances, benchmark
path) lives in C++ (cpuvox_amd/csrc/host); this module only marshals.
Reference counterparts: UnityManager.cs:163-201,297-343,417-458 and
RenderManager.cs:111-152,374-510.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

LOD_LEVELS = 6  # UnityManager.cs:42

_HERE = os.path.dirname(os.path.abspath(__file__))


class SegmentData(C.Structure):
    """RenderManager.SegmentData (RenderManager.cs:503-510) == cvx_segment_data."""

    _fields_ = [
        ("MinScreen", C.c_float * 2),
        ("MaxScreen", C.c_float * 2),
        ("CamLocalPlaneRayMin", C.c_float * 2),
        ("CamLocalPlaneRayMax", C.c_float * 2),
        ("RayCount", C.c_int32),
    ]


class CameraData(C.Structure):
    """CameraData (CameraData.cs:11-16) == cvx_camera_data."""

    _fields_ = [
        ("WorldToScreenMatrix", C.c_float * 16),
        ("PositionXZ", C.c_float * 2),
        ("PositionY", C.c_float),
        ("InverseElementIterationDirection", C.c_uint8),
        ("pad_", C.c_uint8 * 3),
        ("FarClip", C.c_float),
        ("LODDistances", C.c_float * LOD_LEVELS),
    ]


class WorldInfo(C.Structure):
    _fields_ = [
        ("storage", C.c_void_p),
        ("byteLength", C.c_int64),
        ("dimX", C.c_int32),
        ("dimY", C.c_int32),
        ("dimZ", C.c_int32),
        ("lod", C.c_int32),
        ("columnCount", C.c_int32),
        ("elementCount", C.c_int64),
    ]


class CameraPose(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("eulerAngles", C.c_float * 3),
        ("fieldOfView", C.c_float),
        ("nearClipPlane", C.c_float),
        ("pixelWidth", C.c_int32),
        ("pixelHeight", C.c_int32),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("segments", SegmentData * 4),
        ("camera", CameraData),
        ("vanishingPointScreenSpace", C.c_float * 2),
        ("vanishingPointWorldSpace", C.c_float * 3),
        ("forward", C.c_float * 3),
        ("totalRays", C.c_int32),
    ]


assert C.sizeof(SegmentData) == 36 and C.sizeof(CameraData) == 108

_lib = None


def lib() -> C.CDLL:
    """Load libcpuvox_host.so (built in-tree by cpuvox_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        path = os.environ.get("CVX_HOST_LIB") or os.path.join(_HERE, "libcpuvox_host.so")  # CVX_HOST_LIB: another build of the same ABI (sanitizer build in tests)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C cpuvox_amd/csrc`")
        L = C.CDLL(path)
        L.cvxh_last_error.restype = C.c_char_p
        L.cvxh_version.restype = C.c_char_p
        L.cvxh_world_lod0_voxels.restype = C.c_int64
        L.cvxh_world_lod0_voxels.argtypes = [C.c_void_p]
        L.cvxh_world_from_obj.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cvxh_world_procedural.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.cvxh_world_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.cvxh_world_save.argtypes = [C.c_void_p, C.c_char_p]
        L.cvxh_world_from_blobs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]
        L.cvxh_world_downsample_seconds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.cvxh_world_free.argtypes = [C.c_void_p]
        L.cvxh_world_free.restype = None
        L.cvxh_world_lod_count.argtypes = [C.c_void_p]
        L.cvxh_world_info_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(WorldInfo)]
        L.cvxh_world_builder_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cvxh_world_builder_set_voxels.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvxh_world_builder_finish.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.cvxh_world_builder_free.argtypes = [C.c_void_p]
        L.cvxh_world_builder_free.restype = None
        L.cvxh_setup_lods.argtypes = [C.POINTER(CameraPose), C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.cvxh_setup_frame.argtypes = [C.POINTER(CameraPose), C.c_int, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(Frame)]
        L.cvxh_sample_benchmark_path.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.cvxh_sample_benchmark_path.restype = None
        L.cvxh_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def default_threads() -> int:
    """Worker threads the host library uses for `threads <= 0` (cgroup-quota aware)."""
    return lib().cvxh_default_threads()


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"cpuvox_host error {rc}: {lib().cvxh_last_error().decode()}")


class WorldSet:
    """World[] worldLODs (UnityManager.cs:15): LOD_LEVELS RLE worlds kept in the
    reference's storage layout, ready for cvx_world_upload."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    # -- constructors -----------------------------------------------------
    @staticmethod
    def from_obj(path: str, max_dimension: int, swap_yz=False, flip=(True, False, False), threads=0) -> "WorldSet":
        """UnityManager 'Convert' (UnityManager.cs:297-343); default flips X (UnityManager.cs:27)."""
        h = C.c_void_p()
        _check(lib().cvxh_world_from_obj(path.encode(), max_dimension, int(swap_yz), int(flip[0]), int(flip[1]), int(flip[2]), threads, C.byref(h)))
        return WorldSet(h.value)

    @staticmethod
    def procedural(dim_x: int, dim_y: int, dim_z: int, seed: int = 0x5EED2048, threads=0) -> "WorldSet":
        h = C.c_void_p()
        _check(lib().cvxh_world_procedural(dim_x, dim_y, dim_z, seed & 0xFFFFFFFF, threads, C.byref(h)))
        return WorldSet(h.value)

    @staticmethod
    def load(path: str) -> "WorldSet":
        """WorldSaveFile.Deserialize (WorldSaveFile.cs:57)."""
        h = C.c_void_p()
        _check(lib().cvxh_world_load(path.encode(), C.byref(h)))
        return WorldSet(h.value)

    @staticmethod
    def from_voxels(dims, x, y, z, argb, threads=0) -> "WorldSet":
        """WorldBuilder.SetVoxel per entry, then ToLOD0World + DownSample(1..5)."""
        x = np.ascontiguousarray(x, dtype=np.int32)
        y = np.ascontiguousarray(y, dtype=np.int32)
        z = np.ascontiguousarray(z, dtype=np.int32)
        argb = np.ascontiguousarray(argb, dtype=np.uint32)
        assert x.shape == y.shape == z.shape == argb.shape
        b = C.c_void_p()
        _check(lib().cvxh_world_builder_create(int(dims[0]), int(dims[1]), int(dims[2]), C.byref(b)))
        try:
            _check(lib().cvxh_world_builder_set_voxels(b, x.size, x.ctypes.data, y.ctypes.data, z.ctypes.data, argb.ctypes.data))
            h = C.c_void_p()
            _check(lib().cvxh_world_builder_finish(b, threads, C.byref(h)))
        finally:
            lib().cvxh_world_builder_free(b)
        return WorldSet(h.value)

    @staticmethod
    def from_blobs(dims, blobs) -> "WorldSet":
        """One storage blob (bytes / uint8 array, the reference's layout) per LOD, e.g. LOD 0 from the host build and the
        rest from cpuvox_amd.gpu.Context.downsample."""
        arrays = [np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b, dtype=np.uint8) for b in blobs]
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        lens = (C.c_int64 * len(arrays))(*[a.size for a in arrays])
        h = C.c_void_p()
        _check(lib().cvxh_world_from_blobs(int(dims[0]), int(dims[1]), int(dims[2]), len(arrays), ptrs, lens, C.byref(h)))
        return WorldSet(h.value)

    def downsample_host_seconds(self, extra_lods: int, threads: int = 0):
        """Wall-clock seconds (and voxel count) of World.DownSample(extra_lods) of LOD 0 on the host; the result is discarded."""
        sec, vox = C.c_double(), C.c_int64()
        _check(lib().cvxh_world_downsample_seconds(self._h, extra_lods, threads, C.byref(sec), C.byref(vox)))
        return sec.value, vox.value

    # -- accessors --------------------------------------------------------
    def save(self, path: str) -> None:
        """WorldSaveFile.Serialize (WorldSaveFile.cs:8)."""
        _check(lib().cvxh_world_save(self._h, path.encode()))

    @property
    def lod_count(self) -> int:
        return lib().cvxh_world_lod_count(self._h)

    @property
    def lod0_voxels(self) -> int:
        return lib().cvxh_world_lod0_voxels(self._h)

    def info(self, lod: int) -> WorldInfo:
        out = WorldInfo()
        _check(lib().cvxh_world_info_get(self._h, lod, C.byref(out)))
        return out

    @property
    def dims(self):
        i = self.info(0)
        return (i.dimX, i.dimY, i.dimZ)

    @property
    def max_dimension(self) -> int:
        return max(self.dims)

    def storage(self, lod: int) -> np.ndarray:
        """The raw storage blob of one LOD as a uint8 view (no copy)."""
        i = self.info(lod)
        buf = (C.c_uint8 * i.byteLength).from_address(i.storage)
        return np.frombuffer(buf, dtype=np.uint8)

    def close(self) -> None:
        if self._h:
            lib().cvxh_world_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_pose(position, euler, width: int, height: int, fov: float = 85.0, near: float = 0.05) -> CameraPose:
    """Scene camera defaults: FOV 85, near 0.05 (Assets/Scenes/SampleScene.unity:176-178)."""
    p = CameraPose()
    p.position[:] = [float(v) for v in position]
    p.eulerAngles[:] = [float(v) for v in euler]
    p.fieldOfView = fov
    p.nearClipPlane = near
    p.pixelWidth = width
    p.pixelHeight = height
    return p


def setup_lods(pose: CameraPose, world_max_dimension: int, res_x: int, res_y: int, lod_error: float = 1.0):
    """UnityManager.SetupLods (UnityManager.cs:417-458) -> (LODDistances[6], farClip)."""
    out = (C.c_float * LOD_LEVELS)()
    far = C.c_float()
    _check(lib().cvxh_setup_lods(C.byref(pose), world_max_dimension, res_x, res_y, lod_error, out, C.byref(far)))
    return list(out), far.value


def setup_frame(pose: CameraPose, lod_distances, far_clip: float, width: int, height: int, world_dim_y: int, limit_horizon: bool = True) -> Frame:
    """LimitRotationHorizon + the DrawWorld setup before DrawSegments
    (UnityManager.cs:179-181,193-201; RenderManager.cs:119-152)."""
    lods = (C.c_float * LOD_LEVELS)(*lod_distances)
    out = Frame()
    _check(lib().cvxh_setup_frame(C.byref(pose), int(limit_horizon), far_clip, lods, width, height, world_dim_y, C.byref(out)))
    return out


def sample_benchmark_path(t: float, world_dims):
    """BenchmarkPath.anim at clip time t in [0, 1.15] (UnityManager.cs:86-87)."""
    dims = (C.c_float * 3)(*[float(d) for d in world_dims])
    pos = (C.c_float * 3)()
    eul = (C.c_float * 3)()
    lib().cvxh_sample_benchmark_path(t, dims, pos, eul)
    return list(pos), list(eul)


BENCHMARK_PATH_LENGTH = 1.15  # BenchmarkPath.anim:179


def load_image(path: str) -> np.ndarray:
    """Texture2D.LoadImage + GetPixels32 as ObjModel uses them (SimpleMesh.cs:186-205): uint8[H, W, 4] RGBA, row 0 = BOTTOM row."""
    w, h = C.c_int32(), C.c_int32()
    _check(lib().cvxh_image_load(path.encode(), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    _check(lib().cvxh_image_load(path.encode(), C.byref(w), C.byref(h), out.ctypes.data, out.size))
    return out
