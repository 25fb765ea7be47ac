"""The hot-group phase's early rejection (rayrs_amd/csrc/device_path.h hot_triangle_intersect) is exact.

Triangle::intersect (geometry.rs:359-375) decides `d < 0. || u < 0. || v < 0. || u + v > 1.` on three quotients by one
denominator.  The kernel's hot-group phase makes the three divisions only if some lane of the wave is not SETTLED by two
facts about correctly rounded division (sign, and "at least 2"); this file restates the predicate with numpy -- whose `/`
is the IEEE division -- and checks on values from every corner of the format that settled implies rejected.  The GPU
tests compare the whole phase with the oracle bit for bit (tests/test_gpu_hot_group.py)."""
import numpy as np


def settled(n0, n1, n2, den):
    """device_path.h hot_triangle_intersect, the block before div3_by."""
    def hi(x):
        return (x.view(np.uint64) >> np.uint64(32)).astype(np.uint32)
    dh = hi(den)
    opp = [((hi(n) ^ dh) >> np.uint32(31)).astype(bool) for n in (n0, n1, n2)]
    aden, a0, a1, a2 = np.abs(den), np.abs(n0), np.abs(n1), np.abs(n2)
    den_le, den_ge = aden <= 2.0 ** 500, aden >= 2.0 ** -500
    big = [a >= 2.0 ** -500 for a in (a0, a1, a2)]
    fin1, fin2 = a1 <= 2.0 ** 500, a2 <= 2.0 ** 500
    with np.errstate(over="ignore", invalid="ignore"):
        two_den = aden * 2.0
    two1, two2 = a1 >= two_den, a2 >= two_den
    return (den_le & ((opp[0] & big[0]) | (opp[1] & big[1]) | (opp[2] & big[2]))) | \
           (den_le & den_ge & ((~opp[1] & two1 & fin2) | (~opp[2] & two2 & fin1)))


def rejected(n0, n1, n2, den):
    with np.errstate(all="ignore"):
        d, u, v = n0 / den, n1 / den, n2 / den
        return (d < 0.0) | (u < 0.0) | (v < 0.0) | (u + v > 1.0)


SPECIAL = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, -2.2250738585072014e-308,
                    1.7976931348623157e308, -1.7976931348623157e308, 1.0, -1.0, 2.0, -2.0, 0.5, 1.9999999999999998,
                    2.0 ** -500, -2.0 ** -500, 2.0 ** 500, -2.0 ** 500, np.nextafter(2.0 ** -500, 0), np.nextafter(2.0 ** 500, np.inf),
                    2.0 ** 501, 2.0 ** -501, 2.0 ** -1000, 2.0 ** 1000, 3.0, -3.0, 1e-310, -1e-310])


def check(n0, n1, n2, den):
    n0, n1, n2, den = (np.ascontiguousarray(x, dtype=np.float64) for x in (n0, n1, n2, den))
    s, r = settled(n0, n1, n2, den), rejected(n0, n1, n2, den)
    bad = s & ~r
    assert not bad.any(), (n0[bad][:3], n1[bad][:3], n2[bad][:3], den[bad][:3])
    return int(s.sum()), int(r.sum())


def test_settled_implies_rejected_on_every_combination_of_special_values():
    g = np.meshgrid(SPECIAL, SPECIAL, SPECIAL, SPECIAL, indexing="ij")
    s, r = check(*(x.ravel() for x in g))
    assert 0 < s < r


def test_settled_implies_rejected_on_random_bit_patterns_and_scaled_values():
    r = np.random.default_rng(1)
    n = 2_000_000
    bits = r.integers(0, 2 ** 64, (4, n), dtype=np.uint64)
    s, _ = check(*bits.view(np.float64))
    assert s > n // 4
    # magnitudes a renderer sees, with exponents spread over the format and exact ties (|n| = 2 |den|, n = 0)
    m = r.standard_normal((4, n)) * np.exp2(r.integers(-1074, 1023, (4, n)).astype(np.float64) * r.integers(0, 2, (4, n)))
    m[1, ::7] = 2.0 * m[3, ::7]
    m[2, ::11] = -2.0 * m[3, ::11]
    m[0, ::13] = 0.0
    m[3, ::17] = 0.0
    s, _ = check(*m)
    assert s > n // 2


def test_most_misses_of_a_distant_triangle_are_settled():
    """What the shortcut is for: a small triangle far from the ray has barycentric coordinates far outside [0, 2)."""
    r = np.random.default_rng(2)
    n = 200_000
    p1 = np.array([0.3, 1.2, -0.4])
    e1, e2 = np.array([0.02, 0.0, 0.01]), np.array([0.0, 0.015, 0.02])
    o = r.uniform(-20, 20, (n, 3))
    d = r.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tt = o - p1
    p, q = np.cross(d, e2), np.cross(tt, e1)
    den = p @ e1
    n0, n1, n2 = q @ e2, (p * tt).sum(1), (q * d).sum(1)
    s, rej = check(n0, n1, n2, den)
    assert rej > 0.999 * n and s > 0.99 * rej
