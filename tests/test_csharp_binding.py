"""CPU: host/csharp/CpuVoxGpu.cs (the P/Invoke binding a reference maintainer adds) cannot be compiled in this
image (no dotnet / mono / csc, probed on the build container and on the GPU box), so it is kept honest by parsing: every entry
point include/cpuvox_gpu.h declares has a DllImport with the same number of parameters and compatible types, and the blittable
structs have the reference's field lists and sizes."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = open(os.path.join(ROOT, "host", "csharp", "CpuVoxGpu.cs")).read()
HEADER = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cpuvox_gpu.h")).read(), flags=re.S)


def _c_functions():
    out = {}
    for m in re.finditer(r"\b([a-z_0-9 ]+?[ \*]+)(cvx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", HEADER):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        out[name] = (ret, plist)
    return out


def _cs_imports():
    out = {}
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*public static extern\s+([A-Za-z]+)\s+(cvx_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", CS, flags=re.S):
        params = m.group(3).strip()
        out[m.group(2)] = (m.group(1), [] if not params else [p.strip() for p in params.split(",")])
    return out


def _c_kind(decl: str) -> str:
    d = decl.replace("const ", "").strip()
    if "*" in d or "[" in d:
        return "ptr"
    base = d.split()[0]
    return {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "float": "f32", "double": "f64"}.get(base, base)


def _cs_kind(decl: str) -> str:
    d = decl.strip()
    if d.startswith("out ") or "*" in d or d.split()[0] == "IntPtr":
        return "ptr"
    return {"int": "i32", "uint": "i32", "long": "i64", "ulong": "i64", "float": "f32", "double": "f64"}[d.split()[0]]


def test_every_c_entry_point_has_a_dllimport_with_matching_parameters():
    c, cs = _c_functions(), _cs_imports()
    assert len(c) >= 40, sorted(c)
    assert sorted(c) == sorted(cs), f"missing: {sorted(set(c) - set(cs))}, extra: {sorted(set(cs) - set(c))}"
    for name, (ret, params) in c.items():
        cs_ret, cs_params = cs[name]
        assert len(params) == len(cs_params), f"{name}: {len(params)} C parameters, {len(cs_params)} in the DllImport"
        for i, (a, b) in enumerate(zip(params, cs_params)):
            assert _c_kind(a) == _cs_kind(b), f"{name} parameter {i}: C `{a}` vs C# `{b}`"
        want = "ptr" if "*" in ret else _c_kind(ret)
        got = {"int": "i32", "long": "i64", "void": "void", "IntPtr": "ptr"}[cs_ret]
        assert want == got, f"{name}: returns {ret} in C, {cs_ret} in C#"


def _struct(name):
    m = re.search(r"\[StructLayout\(([^\]]*)\)\]\s*public (?:unsafe )?struct " + name + r"\s*\{(.*?)\n\t\}", CS, flags=re.S)
    assert m, f"struct {name} not found"
    return m.group(1), m.group(2)


def _size(body):
    sizes = {"float": 4, "int": 4, "byte": 1, "long": 8, "uint": 4}
    total = 0
    for m in re.finditer(r"(?:public\s+)?(fixed\s+)?(float|int|byte|long|uint)\s+([^;]+);", body):
        names = [n.strip() for n in m.group(3).split(",")]
        for n in names:
            k = re.search(r"\[(\d+)\]", n)
            total += sizes[m.group(2)] * (int(k.group(1)) if k else 1)
    return total


def test_blittable_structs_have_the_reference_layout():
    attrs, body = _struct("SegmentData")  # RenderManager.SegmentData, RenderManager.cs:503-510: 4 x float2 + int
    assert "LayoutKind.Sequential" in attrs and "Pack = 4" in attrs
    assert re.findall(r"(MinScreen|MaxScreen|CamLocalPlaneRayMin|CamLocalPlaneRayMax|RayCount)", body) == ["MinScreen", "MaxScreen", "CamLocalPlaneRayMin", "CamLocalPlaneRayMax", "RayCount"]
    assert _size(body) == 36
    attrs, body = _struct("CameraData")  # CameraData.cs:11-16
    assert "LayoutKind.Sequential" in attrs and "Pack = 4" in attrs
    order = re.findall(r"(WorldToScreenMatrix|PositionXZ|PositionY|InverseElementIterationDirection|pad|FarClip|LODDistances)\b", body)
    assert order[:7] == ["WorldToScreenMatrix", "PositionXZ", "PositionY", "InverseElementIterationDirection", "pad", "FarClip", "LODDistances"], order
    assert _size(body) == 108
    for name, size in (("Counters", 88), ("RaybufferLayout", 24), ("RowSpan", 24)):
        attrs, body = _struct(name)
        assert "LayoutKind.Sequential" in attrs and _size(body) == size, (name, _size(body))


def test_the_header_sizes_agree():
    import ctypes as C

    from cpuvox_amd import gpu, host

    assert C.sizeof(host.SegmentData) == 36 and C.sizeof(host.CameraData) == 108 and C.sizeof(gpu.Counters) == 88 and C.sizeof(gpu.RaybufferLayout) == 24
