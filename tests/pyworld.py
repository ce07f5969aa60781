"""Test infrastructure: a second, independent restatement of the reference's world construction and storage format in plain Python
(dictionaries and lists, no shared code with cpuvox_amd/csrc/host/cvx_world.cpp), used to cross-check the C++ host builder, its
LOD chain and its `.world` files byte for byte (tests/test_world_model.py).  Like the CPU oracle it cannot be pinned against the
reference itself (no vectors ship, the reference cannot run here); two independent restatements agreeing is the evidence.

What it restates (files under /root/reference/Assets/Code):
  WorldBuilder.RLEColumnBuilder.SetVoxel / ToFinalColumn   WordBuilder.cs:157-268  (sort by Y descending, average r/g/b of voxels that share a Y,
                                                                                    air / solid runs from the top, colours in run order)
  World.RLEColumn ctor                                      World.cs:190-234        (guards, WorldMin / WorldMax in LOD-0 voxels, ushort)
  World storage: ColumnCount, column index, blob layout     World.cs:17,145-149,262-283 (headers of 12 bytes, then 4-byte elements)
  World.DownSample / DownSampleColumn / DownSamplePartial   World.cs:45-127
  WorldSaveFile.Serialize                                   WorldSaveFile.cs:8-57,98-104

Conventions the reference leaves open and the host build fixes (documented in cvx_world.cpp), followed here: columns get their elements in
column-index order (the reference allocates from parallel threads), voxels that share a Y keep their insertion order (List.Sort is unstable in
.NET), the blob ends with the last element (the reference's byte length is the allocator's capacity).
"""
from __future__ import annotations

import struct


def _column_count(dim_x: int, dim_z: int, lod: int) -> int:
    return (dim_x * dim_z) // ((lod + 1) * (lod + 1))  # World.cs:17 (not a shift: the table is larger than the columns it holds for lod >= 2)


def final_column(voxels, top_y: int, voxel_scale: int):
    """voxels: [(y, argb)] in insertion order -> None for an empty column, else (runs [(colorsIndex, length)], colours [argb], worldMin, worldMax).
    argb = a | r << 8 | g << 16 | b << 24 (ColorARGB32's byte order a, r, g, b read as a little-endian word)."""
    if not voxels:
        return None
    ordered = sorted(voxels, key=lambda v: -v[0])  # stable: equal Y keep insertion order
    deduped = []
    i = 0
    while i < len(ordered):
        y, first = ordered[i]
        j = i + 1
        r = g = b = 0
        while j < len(ordered) and ordered[j][0] == y:
            c = ordered[j][1]
            r += (c >> 8) & 0xFF
            g += (c >> 16) & 0xFF
            b += (c >> 24) & 0xFF
            j += 1
        weight = j - i
        if weight > 1:
            fr, fg, fb = (first >> 8) & 0xFF, (first >> 16) & 0xFF, (first >> 24) & 0xFF
            first = (first & 0xFF) | ((((fr + r) // weight) & 0xFF) << 8) | ((((fg + g) // weight) & 0xFF) << 16) | ((((fb + b) // weight) & 0xFF) << 24)
        deduped.append((y, first))
        i = j
    runs = []
    top = top_y
    k = 0
    while k < len(deduped):
        air = top - deduped[k][0]
        if air > 0:
            runs.append((-1, air))
            top -= air
        length = 1
        while k + length < len(deduped) and deduped[k + length][0] == top - length:
            length += 1
        runs.append((k, length))
        top -= length
        k += length
    if top >= 0:
        runs.append((-1, top + 1))
    # World.RLEColumn ctor: bounds from the bottom up
    lo, hi = None, None
    bound = 0
    for colors_index, length in reversed(runs):
        nxt = bound + length
        if colors_index >= 0:
            lo = bound if lo is None else min(lo, bound)
            hi = nxt if hi is None else max(hi, nxt)
        bound = nxt
    return runs, [c for _, c in deduped], (lo * voxel_scale) & 0xFFFF, (hi * voxel_scale) & 0xFFFF


class Level:
    def __init__(self, dims, lod):
        self.dims, self.lod = dims, lod
        self.columns = {}  # (cx, cz) in units of this level's columns -> final_column result

    def blob(self) -> bytes:
        dx, dy, dz = self.dims
        count = _column_count(dx, dz, self.lod)
        mul_x = dz >> self.lod
        headers = bytearray(count * 12)
        elements = bytearray()
        cursor = 0
        for cx in range(dx >> self.lod):
            for cz in range(dz >> self.lod):
                col = self.columns.get((cx, cz))
                if col is None:
                    continue
                runs, colours, wmin, wmax = col
                struct.pack_into("<iHHHH", headers, (cx * mul_x + cz) * 12, cursor, len(runs), wmin, wmax, 0)
                elements += struct.pack("<hh", 0, 0)
                for colors_index, length in runs:
                    elements += struct.pack("<hh", colors_index, length)
                elements += struct.pack("<hh", 0, 0)
                for c in colours:
                    elements += struct.pack("<I", c)
                cursor += len(runs) + len(colours) + 2
        return bytes(headers) + bytes(elements)


def build_lod0(dims, voxels) -> Level:
    """voxels: iterable of (x, y, z, argb) in insertion order (WorldBuilder.SetVoxel per entry, then ToLOD0World)."""
    per_column = {}
    for x, y, z, c in voxels:
        per_column.setdefault((x, z), []).append((y, c))
    level = Level(dims, 0)
    for key, vs in per_column.items():
        level.columns[key] = final_column(vs, dims[1] - 1, 1)
    return level


def downsample(src: Level, extra_lods: int) -> Level:
    """World.DownSample(extraLods) of a level (the reference only ever calls it on LOD 0, UnityManager.cs:329)."""
    dx, dy, dz = src.dims
    out = Level(src.dims, src.lod + extra_lods)
    next_lod = src.lod + extra_lods
    steps = 1 << extra_lods
    src_height = dy >> src.lod
    for tx in range(dx >> next_lod):
        for tz in range(dz >> next_lod):
            voxels = []
            for ix in range(steps):
                for iz in range(steps):
                    col = src.columns.get((tx * steps + ix, tz * steps + iz))
                    if col is None:
                        continue
                    runs, colours, _, _ = col
                    bound = src_height
                    for colors_index, length in runs:
                        bound -= length
                        if colors_index < 0:
                            continue
                        for i in range(length):
                            voxels.append(((bound + i) >> next_lod, colours[colors_index + length - i - 1]))
            col = final_column(voxels, (dy >> next_lod) - 1, 1 << next_lod)
            if col is not None:
                out.columns[(tx, tz)] = col
    return out


def world_file(dims, blobs) -> bytes:
    """WorldSaveFile.Serialize: Header {long 0; int X, Y, Z; int WorldCount}, (offset, length) pairs, blobs."""
    head = struct.pack("<qiiii", 0, dims[0], dims[1], dims[2], len(blobs))
    table = bytearray()
    offset = len(head) + 16 * len(blobs)
    for b in blobs:
        table += struct.pack("<qq", offset, len(b))
        offset += len(b)
    return head + bytes(table) + b"".join(blobs)
