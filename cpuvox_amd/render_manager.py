"""RenderManager -- Python face of the C++ twin of the reference's RenderManager
(Assets/Code/RenderManager.cs:12-256; cpuvox_amd/csrc/host/cvx_render_manager.cpp).

Usage mirrors UnityManager.LateUpdate (UnityManager.cs:163-188):

    rm = RenderManager(width, height)
    rm.upload_world(world_set)
    lods, far = host.setup_lods(pose, world_set.max_dimension, width, height)
    rm.swap_buffers()
    image = rm.draw_world(pose, lods, far)      # uint32[H, W] ARGB32, row 0 = bottom

DrawSegments and BlitSegments run on the GPU (libcpuvox_gpu.so); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import gpu, host

SCREEN_BUFFER, RAYBUFFER_TOPDOWN, RAYBUFFER_LEFTRIGHT = 0, 1, 2  # UnityManager.ERenderMode


class RenderManager:
    def __init__(self, width: int, height: int, device: int = 0):
        L = host.lib()
        L.cvxh_render_manager_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.cvxh_render_manager_destroy.argtypes = [C.c_void_p]
        L.cvxh_render_manager_destroy.restype = None
        L.cvxh_render_manager_upload_world.argtypes = [C.c_void_p, C.c_void_p]
        L.cvxh_render_manager_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.cvxh_render_manager_swap_buffers.argtypes = [C.c_void_p]
        L.cvxh_render_manager_clear_raybuffer.argtypes = [C.c_void_p, C.c_int]
        L.cvxh_render_manager_draw_world.argtypes = [C.c_void_p, C.POINTER(host.CameraPose), C.c_int, C.c_float, C.POINTER(C.c_float),
                                                     C.c_void_p, C.POINTER(host.Frame)]
        L.cvxh_render_manager_read_raybuffer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        self._h = C.c_void_p()
        host._check(L.cvxh_render_manager_create(device, width, height, gpu.lib_path().encode(), C.byref(self._h)))
        self.width, self.height = width, height
        self.last_frame = None

    def close(self) -> None:
        if self._h:
            host.lib().cvxh_render_manager_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_world(self, world_set: host.WorldSet) -> None:
        host._check(host.lib().cvxh_render_manager_upload_world(self._h, world_set._h))

    def set_resolution(self, width: int, height: int) -> bool:
        """RenderManager.SetResolution (RenderManager.cs:94-109): True when the resolution changed."""
        changed = C.c_int()
        host._check(host.lib().cvxh_render_manager_set_resolution(self._h, width, height, C.byref(changed)))
        self.width, self.height = width, height
        return bool(changed.value)

    def swap_buffers(self) -> int:
        """RenderManager.SwapBuffers (RenderManager.cs:53-56)."""
        return host.lib().cvxh_render_manager_swap_buffers(self._h)

    def clear_raybuffer(self, render_mode: int) -> None:
        """RenderManager.ClearRayBuffer (RenderManager.cs:58-92): pink (255, 20, 147) debug fill."""
        host._check(host.lib().cvxh_render_manager_clear_raybuffer(self._h, render_mode))

    def draw_world(self, pose: host.CameraPose, lod_distances, far_clip: float, limit_horizon: bool = True, to_host: bool = True):
        """UnityManager.LateUpdate body: LimitRotationHorizon + RenderManager.DrawWorld (RenderManager.cs:111-194)."""
        lods = (C.c_float * host.LOD_LEVELS)(*lod_distances)
        frame = host.Frame()
        img = np.empty((self.height, self.width), dtype=np.uint32) if to_host else None
        host._check(host.lib().cvxh_render_manager_draw_world(self._h, C.byref(pose), int(limit_horizon), far_clip, lods,
                                                              img.ctypes.data if to_host else None, C.byref(frame)))
        self.last_frame = frame
        return img

    def read_raybuffer(self, which: int, first_ray: int, ray_count: int) -> np.ndarray:
        width = self.height if which == gpu.RAYBUFFER_TOPDOWN else self.width
        out = np.empty((ray_count, width), dtype=np.uint32)
        host._check(host.lib().cvxh_render_manager_read_raybuffer(self._h, which, first_ray, ray_count, out.ctypes.data))
        return out
