"""CPU: the oracle against the committed golden vectors and the path's own invariants."""
import json
import os

import numpy as np
import pytest

import oraclelib as O
import scenes

GOLDEN = json.load(open(os.path.join(scenes.GOLDEN, "golden.json")))
CLEAR = 0xDEADBEEF
FAST = [n for n in scenes.SCENES if "1080p" not in n]


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_oracle_reproduces_golden(name):
    ws, fr, W, H = scenes.scene_frame(name)
    gold = GOLDEN[name]
    assert [s.RayCount for s in fr.segments] == gold["rayCounts"]
    assert [float(v) for v in fr.vanishingPointScreenSpace] == gold["vanishingPoint"]
    td, lr, cnt = O.draw_segments(ws, fr, W, H, clear=0)
    n_td, n_lr = scenes.used_rows(fr)
    assert scenes.crc(td[:n_td]) == gold["crcTopDown"] and scenes.crc(lr[:n_lr]) == gold["crcLeftRight"]
    got = cnt.as_dict()
    for k in ("S", "E", "C", "P", "R", "lodVisits"):
        assert got[k] == gold["counters"][k], k
    for label, buf in (("td", td), ("lr", lr)):
        for key, want in gold["rowCrcs"].items():
            if key.startswith(label):
                assert scenes.crc(buf[int(key[2:])]) == want


@pytest.mark.parametrize("name", FAST)
def test_every_pixel_of_every_ray_written_exactly_the_right_region(name):
    """Pixels [origMin, origMax] of every used ray are written (seen mask + skybox), nothing else is touched,
    and P (pixels stored) equals the region's area: each pixel is written exactly once."""
    ws, fr, W, H = scenes.scene_frame(name)
    td, lr, cnt = O.draw_segments(ws, fr, W, H, clear=CLEAR)
    vp = fr.vanishingPointScreenSpace
    rc = [max(0, s.RayCount) for s in fr.segments]

    def rnd(v, hi):
        return int(min(max(np.rint(np.float32(v)), 0), hi))

    bounds = [(rnd(vp[1], H - 1), H - 1), (0, rnd(vp[1], H - 1)), (rnd(vp[0], W - 1), W - 1), (0, rnd(vp[0], W - 1))]
    area = 0
    for buf, segs in ((td, (0, 1)), (lr, (2, 3))):
        row = 0
        for s in segs:
            lo, hi = bounds[s]
            block = buf[row:row + rc[s]]
            if rc[s]:
                assert (block[:, lo:hi + 1] != CLEAR).all(), f"segment {s}: unwritten pixel inside [origMin, origMax]"
                assert (block[:, :lo] == CLEAR).all() and (block[:, hi + 1:] == CLEAR).all(), f"segment {s}: pixel outside the range touched"
                assert ((block[:, lo:hi + 1] & 0xFF) == 0xFF).all(), "alpha is 255 everywhere (colours and skybox)"
                area += rc[s] * (hi - lo + 1)
            row += rc[s]
        assert (buf[row:] == CLEAR).all(), "rows beyond the used rays touched"
    assert cnt.P == area and cnt.R == sum(rc)
    assert sum(cnt.lodVisits) == cnt.S


def test_thread_count_does_not_change_the_result():
    ws, fr, W, H = scenes.scene_frame("proc256_t075_lod8")
    a = O.draw_segments(ws, fr, W, H, threads=1)
    b = O.draw_segments(ws, fr, W, H, threads=5)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2].as_dict() == b[2].as_dict()


def test_empty_frame_and_empty_world():
    """No rays -> nothing happens; a world with no voxels -> every ray is skybox."""
    from cpuvox_amd import host

    ws = scenes.load_world("proc256")
    fr = scenes.benchmark_frame(ws, 320, 200, 0.5)
    for s in fr.segments:
        s.RayCount = 0
    td, lr, cnt = O.draw_segments(ws, fr, 320, 200, clear=CLEAR)
    assert (td == CLEAR).all() and (lr == CLEAR).all() and cnt.R == 0
    empty = host.WorldSet.from_voxels((64, 64, 64), [], [], [], [])
    fr = scenes.make_frame(empty, 160, 120, (32, 40, 32), (30, 10, 0))
    td, lr, cnt = O.draw_segments(empty, fr, 160, 120, clear=CLEAR)
    written = np.concatenate([td[td != CLEAR], lr[lr != CLEAR]])
    assert written.size == cnt.P > 0 and (written == 0x191919FF).all() and cnt.E == 0 and cnt.C == 0


def test_single_voxel_world_is_drawn():
    from cpuvox_amd import host

    ws = host.WorldSet.from_voxels((64, 64, 64), [40], [10], [40], [0xFF8040FF])  # bytes A=FF R=40 G=80 B=FF
    fr = scenes.make_frame(ws, 320, 240, (20.5, 20.0, 20.5), (20.0, 45.0, 0.0))
    td, lr, cnt = O.draw_segments(ws, fr, 320, 240, clear=CLEAR)
    vals = set(np.unique(np.concatenate([td[td != CLEAR], lr[lr != CLEAR]])).tolist())
    assert vals == {0x191919FF, 0xFF8040FF}
