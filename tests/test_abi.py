"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from cpuvox_amd import gpu, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_gpu_library_exports_every_declared_symbol():
    L = gpu.lib()
    declared = _declared("cpuvox_gpu.h", "cvx_")
    assert declared, "no declarations parsed"
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(gpu.EXPORTS) == declared, "cpuvox_amd.gpu.EXPORTS out of sync with include/cpuvox_gpu.h"
    assert b"gfx950" in L.cvx_version()


def test_host_library_exports_every_declared_symbol():
    L = host.lib()
    declared = _declared("cpuvox_host.h", "cvxh_")
    missing = [n for n in declared if not hasattr(L, n)]
    assert declared and not missing, missing


def test_struct_layouts_match_the_reference_blittable_layout():
    # RenderManager.SegmentData (RenderManager.cs:503-510): 4 float2 + int = 36 bytes
    assert C.sizeof(host.SegmentData) == 36
    # CameraData (CameraData.cs:11-16): float4x4 + float2 + float + bool(+3) + float + float[6] = 108 bytes
    assert C.sizeof(host.CameraData) == 108
    assert host.CameraData.InverseElementIterationDirection.offset == 76
    assert host.CameraData.FarClip.offset == 80 and host.CameraData.LODDistances.offset == 84


def test_gpu_code_object_is_gfx950():
    data = open(gpu.lib_path(), "rb").read()
    assert b"gfx950" in data


def test_product_fails_loudly_without_a_device():
    """No CPU fallback: without a HIP device context creation raises (on a GPU box it succeeds)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gpu.CvxError):
        gpu.Context(0)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "cpuvox_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "oraclelib" not in text and "cvx_oracle" not in text and "libcvx_oracle" not in text, os.path.join(root, f)
