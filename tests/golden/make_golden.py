"""Regenerates the committed fixtures under tests/golden/.

Run in the build container (needs /root/reference for datasets/mill.obj):
    python tests/golden/make_golden.py

Produces
  mill256.world.xz / mill512.world.xz
      datasets/mill.obj voxelised by THIS repo's host library
      (cvxh_world_from_obj: X flipped, Y up, max dimension 256 / 512), all 6 LODs,
      in the WorldSaveFile layout, xz-compressed.  Input data for the parity tests
      and bench config 2; the GPU box has no /root/reference.
  golden.json
      per scene (tests/scenes.py): segment ray counts, vanishing point, work
      counters S/E/C/P/R and CRC32 of the two raybuffers' used rows, plus a few
      full rows -- all computed by the CPU oracle (oracle/cvx_oracle.c).
      PARITY UNPINNED: the reference has no vectors of its own for this path; these
      pin the oracle against drift (compiler / platform) and the HIP path against it.
"""
from __future__ import annotations

import json
import lzma
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MILL = "/root/reference/datasets/mill.obj"


def make_worlds():
    from cpuvox_amd import host

    for dim in (256, 512):
        ws = host.WorldSet.from_obj(MILL, dim)
        tmp = os.path.join(HERE, f"mill{dim}.world")
        ws.save(tmp)
        raw = open(tmp, "rb").read()
        os.unlink(tmp)
        with open(os.path.join(HERE, f"mill{dim}.world.xz"), "wb") as f:
            f.write(lzma.compress(raw, preset=9))
        print(f"mill{dim}: dims {ws.dims}, {ws.lod0_voxels} voxels, {len(raw)} bytes raw")


def make_golden():
    import oraclelib as O
    import scenes

    out = {}
    for name in scenes.SCENES:
        ws, fr, W, H = scenes.scene_frame(name)
        td, lr, cnt = O.draw_segments(ws, fr, W, H, clear=0)
        n_td, n_lr = scenes.used_rows(fr)
        rows = {}
        for label, buf, n in (("td", td, n_td), ("lr", lr, n_lr)):
            if n > 0:
                for r in sorted({0, n // 2, n - 1}):
                    rows[f"{label}{r}"] = scenes.crc(buf[r])
        out[name] = {
            "rayCounts": [s.RayCount for s in fr.segments],
            "vanishingPoint": [float(v) for v in fr.vanishingPointScreenSpace],
            "lodDistances": [float(v) for v in fr.camera.LODDistances],
            "inverse": int(fr.camera.InverseElementIterationDirection),
            "counters": cnt.as_dict(),
            "crcTopDown": scenes.crc(td[:n_td]),
            "crcLeftRight": scenes.crc(lr[:n_lr]),
            "rowCrcs": rows,
        }
        print(name, out[name]["rayCounts"], out[name]["counters"]["lodVisits"], hex(out[name]["crcTopDown"]), hex(out[name]["crcLeftRight"]))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if os.path.exists(MILL):
        make_worlds()
    else:
        print("no /root/reference: keeping the committed *.world.xz")
    make_golden()
