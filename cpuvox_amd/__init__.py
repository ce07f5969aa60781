"""cpuvox_amd -- Python side of the MI355X raybuffer renderer (one hot path of pipliz/cpuvox).

Everything that computes lives in two in-tree shared libraries built by `cpuvox_amd/csrc/Makefile`:
`libcpuvox_gpu.so` (HIP kernels for gfx950 + the C ABI of `include/cpuvox_gpu.h`) and `libcpuvox_host.so`
(C++ mirror of the reference's managed code, `include/cpuvox_host.h`).  The modules here only marshal:

* `gpu`            -- ctypes binding of the GPU library (`Context`, `NativeShardPlan`, RCCL communicator helpers); raises when the
                      library or a HIP device is missing: there is no CPU fallback
* `host`           -- ctypes binding of the host library (worlds, camera / segment setup, benchmark path)
* `render_manager` -- the `RenderManager` twin (`SetResolution`, `SwapBuffers`, `DrawWorld`)
* `dist`           -- the multi-GPU shard plan in Python (verification, `torch.distributed` fallback of the exchange)
"""

__version__ = "0.2"
__all__ = ["gpu", "host", "render_manager", "dist"]
