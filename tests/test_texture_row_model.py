"""CPU model of the certified cheap texture row (`tex_row_cheap`, cpuvox_amd/csrc/cvx_kernels.h) against the reference's arithmetic
(DrawSegmentRayJob.cs:524-531: two IEEE divisions per pixel).  The device test (tests/test_gpu_parity.py::test_cheap_texture_row_is_certified)
checks the shipped instructions with the hardware's own reciprocal; this one checks the ARGUMENT: the hardware reciprocal is only promised to be
within one ulp, so here it is the correctly rounded reciprocal pushed one ulp either way, in every combination -- and still no row that the bound
calls certain may differ from the exact one."""
import numpy as np
import pytest

F = np.float32


def _fma(a, b, c):
    # a * b is exact in float64 (two 24-bit significands); one rounding to float64 and one to float32 stand for the fused operation
    # (a double rounding can differ from the true fma by an ulp in rare cases: inside the error the bound assumes for every operation anyway)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def _rcp(x, ulps):
    r = (F(1.0) / x).astype(F)
    for _ in range(abs(ulps)):
        r = np.nextafter(r, F(np.inf) if ulps > 0 else F(-np.inf)).astype(F)
    return r


def _exact_row(y, bx, by, uvax, uvbx, uvay, uvby):
    with np.errstate(all="ignore"):
        l = ((y - bx) / (by - bx)).astype(F)
        wux = (uvax + (l * (uvbx - uvax).astype(F)).astype(F)).astype(F)
        wuy = (uvay + (l * (uvby - uvay).astype(F)).astype(F)).astype(F)
        u = (wuy / wux).astype(F)
    return u


def _cheap_row(y, bx, by, uvax, uvbx, uvay, uvby, k1, k2):
    with np.errstate(all="ignore"):
        d = (by - bx).astype(F)
        rd = np.where(np.abs(d) <= F(2.0 ** 100), _rcp(d, k1), F(np.nan)).astype(F)
        a1, a2 = (uvbx - uvax).astype(F), (uvby - uvay).astype(F)
        k = F(40.0 * 2.0 ** -24)
        a1s = (k * np.abs(a1)).astype(F)
        a2s = (k * np.abs(a2)).astype(F)
        bxs = _fma(np.full_like(y, k), np.abs(uvax), np.full_like(y, F(2.0 ** -140)))
        bys = (k * np.abs(uvay)).astype(F)
        n = (y - bx).astype(F)
        lq = (n * rd).astype(F)
        wx, wy = _fma(lq, a1, uvax), _fma(lq, a2, uvay)
        r = _rcp(wx, k2)
        uq = (wy * r).astype(F)
        sx, sy = _fma(np.abs(lq), a1s, bxs), _fma(np.abs(lq), a2s, bys)
        t = _fma(np.abs(uq), sx, (sy + sx).astype(F))
        bound = _fma(t, np.abs(r), _fma(np.abs(wx), np.full_like(y, F(2.0 ** -110)), (F(2.0 ** -21) * np.abs(uq)).astype(F)))
        certain = np.abs((uq - np.rint(uq)).astype(F)) > bound
    return uq, certain


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_certain_rows_equal_the_reference_rows_for_any_one_ulp_reciprocal(seed):
    rng = np.random.default_rng(seed)
    n = 400_000
    span = np.exp(rng.uniform(np.log(1e-3), np.log(3000.0), n))
    bx = rng.uniform(-200.0, 2300.0, n)
    by = bx + span
    y = np.rint(rng.uniform(bx - 1.0, by + 1.0)).clip(-2, 16385)
    zb = np.exp(rng.uniform(np.log(0.06), np.log(6000.0), n))
    zt = zb * np.exp(rng.uniform(-1.5, 1.5, n))
    if seed == 4:  # a level camera: both ends at (nearly) the same depth, uvB.x - uvA.x cancels
        zt = zb * (1.0 + rng.uniform(-1e-4, 1e-4, n) * rng.integers(0, 2, n))
    if seed == 5:  # ends far apart in depth
        zt = zb * np.exp(rng.uniform(-4.0, 4.0, n))
    ua = np.rint(np.exp(rng.uniform(0.0, np.log(600.0), n)))
    # a third of the samples: the run length chosen so that the row lands next to an integer for this pixel (where the floor is decided)
    t = np.clip((y - bx) / span, 0.0, 1.0)
    target = np.rint(rng.uniform(1, 60, n)) + rng.uniform(-1, 1, n) * np.exp(rng.uniform(np.log(1e-8), np.log(1e-3), n))
    wx = 1.0 / zb + t * (1.0 / zt - 1.0 / zb)
    near = rng.random(n) < 0.33
    ua = np.where(near & (t < 0.98), target * wx * zb / np.maximum(1.0 - t, 1e-9), ua)
    y, bx, by = y.astype(F), bx.astype(F), by.astype(F)
    uvax, uvbx = (F(1.0) / zb.astype(F)).astype(F), (F(1.0) / zt.astype(F)).astype(F)
    uvay, uvby = (ua.astype(F) / zb.astype(F)).astype(F), np.zeros(n, dtype=F)
    if seed in (2, 6):  # the swapped order (:496-499)
        uvax, uvbx, uvay, uvby = uvbx, uvax, uvby, uvay
    exact = _exact_row(y, bx, by, uvax, uvbx, uvay, uvby)
    certain_any = np.zeros(n, dtype=bool)
    for k1 in (-1, 0, 1):
        for k2 in (-1, 0, 1):
            uq, certain = _cheap_row(y, bx, by, uvax, uvbx, uvay, uvby, k1, k2)
            with np.errstate(all="ignore"):
                wrong = certain & (np.floor(uq) != np.floor(exact))
            assert not wrong.any(), f"reciprocals off by ({k1}, {k2}) ulp: {wrong.sum()} certain rows differ, first at {np.flatnonzero(wrong)[0]}"
            certain_any |= certain
    assert certain_any.mean() > 0.6  # (a third of the samples were put next to an integer on purpose)
