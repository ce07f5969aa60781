"""The C++ host builder (libcpuvox_host.so: RLEColumnBuilder.ToFinalColumn, RLEColumn ctor, World.DownSample, WorldSaveFile) against
tests/pyworld.py, an independent plain-Python restatement of the same reference code: storage blobs of all six LODs and the `.world`
file, byte for byte.  CPU only."""
import os
import tempfile

import numpy as np
import pytest

import pyworld
from cpuvox_amd import host


def _random_voxels(dims, seed, n, dup_share=0.3):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, dims[0], n)
    z = rng.integers(0, dims[2], n)
    # heights clustered around a surface so that solid runs of several voxels, gaps and single voxels all occur
    base = (dims[1] * (0.3 + 0.3 * np.sin(x / 5.0) * np.cos(z / 7.0))).astype(np.int64)
    y = np.clip(base + rng.integers(-4, 5, n), 0, dims[1] - 1)
    argb = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    k = int(n * dup_share)  # several triangles hitting one voxel: the colours are averaged, the first one's alpha survives
    pick = rng.integers(0, n, k)
    x, y, z = np.concatenate([x, x[pick]]), np.concatenate([y, y[pick]]), np.concatenate([z, z[pick]])
    argb = np.concatenate([argb, rng.integers(0, 2 ** 32, k, dtype=np.uint64).astype(np.uint32)])
    return x.astype(np.int32), y.astype(np.int32), z.astype(np.int32), argb


@pytest.mark.parametrize("dims,n,seed", [((32, 64, 32), 6000, 1), ((64, 32, 32), 3000, 2), ((32, 128, 64), 9000, 3)])
def test_host_builder_lod_chain_and_file_match_the_python_model(dims, n, seed):
    x, y, z, argb = _random_voxels(dims, seed, n)
    ws = host.WorldSet.from_voxels(dims, x, y, z, argb, threads=3)
    assert ws.lod_count == 6
    lod0 = pyworld.build_lod0(dims, zip(x.tolist(), y.tolist(), z.tolist(), argb.tolist()))
    levels = [lod0] + [pyworld.downsample(lod0, j) for j in range(1, 6)]
    blobs = [lv.blob() for lv in levels]
    for lod in range(6):
        got = ws.storage(lod).tobytes()
        info = ws.info(lod)
        assert info.columnCount == (dims[0] * dims[2]) // ((lod + 1) ** 2)
        assert got == blobs[lod], f"LOD {lod}: host blob ({len(got)} bytes) differs from the model ({len(blobs[lod])} bytes)"
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.world")
        ws.save(path)
        assert open(path, "rb").read() == pyworld.world_file(dims, blobs)
        # and a file written by the model is read back as the same world
        with open(path, "wb") as fh:
            fh.write(pyworld.world_file(dims, blobs))
        back = host.WorldSet.load(path)
        for lod in range(6):
            assert back.storage(lod).tobytes() == blobs[lod]


def test_model_column_rules():
    """The model's own invariants on hand-made columns (so that a wrong model cannot agree with a wrong builder by accident)."""
    # one voxel at the very top of an 8-high column: a solid run of 1, then air down to the floor
    runs, colours, wmin, wmax = pyworld.final_column([(7, 0x112233FF)], 7, 1)
    assert runs == [(0, 1), (-1, 7)] and colours == [0x112233FF] and (wmin, wmax) == (7, 8)
    # two touching voxels + a gap + one at the floor; a duplicate of y = 2 is averaged channel by channel, the first one's alpha stays
    runs, colours, wmin, wmax = pyworld.final_column([(2, 0x10203040), (5, 0x01010101), (4, 0x02020202), (2, 0x30405060), (0, 0xAABBCCDD)], 7, 2)
    assert runs == [(-1, 2), (0, 2), (-1, 1), (2, 1), (-1, 1), (3, 1)]
    assert colours == [0x01010101, 0x02020202, 0x20304040 | 0, 0xAABBCCDD][:2] + [((0x10 + 0x30) // 2) << 24 | ((0x20 + 0x40) // 2) << 16 | ((0x30 + 0x50) // 2) << 8 | 0x40, 0xAABBCCDD]
    assert (wmin, wmax) == (0, 12)
    assert pyworld.final_column([], 7, 1) is None
