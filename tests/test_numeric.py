"""The numeric contract (include/rayrs_numeric.h) on the CPU side: the counter RNG
against an independent Python restatement + committed known-answer integers,
and the portable elementary functions against the platform libm (what the Rust
reference calls) within 2 ulp on the domains the path tracer uses."""
import math

import numpy as np
import pytest

import _oracle

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def path_key(seed, pixel, sample):
    h = mix64((seed + GOLDEN) & M64)
    h = mix64(h ^ ((pixel * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & M64))
    h = mix64(h ^ ((sample * 0xABC98388FB8FAC03 + 0x2545F4914F6CDD1D) & M64))
    return h


def draw_bits(seed, pixel, sample, draw):
    return mix64((path_key(seed, pixel, sample) + (draw + 1) * GOLDEN) & M64)


def test_rng_matches_independent_restatement():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        pixel = int(rng.integers(0, 2 ** 24))
        sample = int(rng.integers(0, 2 ** 14))
        draw = int(rng.integers(0, 300))
        assert _oracle.rng_bits(seed, pixel, sample, draw) == draw_bits(seed, pixel, sample, draw)


def test_rng_known_answers():
    """Committed integers (tests/golden/rng_known_answers.txt): the hash must never change."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "rng_known_answers.txt")
    rows = [line.split() for line in open(path) if line.strip() and not line.startswith("#")]
    assert len(rows) >= 16
    for seed, pixel, sample, draw, bits in rows:
        assert _oracle.rng_bits(int(seed, 16), int(pixel), int(sample), int(draw)) == int(bits, 16)


def test_uniform_is_53_bit_and_in_unit_interval():
    xs = np.array([(draw_bits(7, p, s, d) >> 11) * 2.0 ** -53 for p in range(40) for s in range(5) for d in range(5)])
    assert xs.min() >= 0.0 and xs.max() < 1.0
    assert abs(xs.mean() - 0.5) < 0.05
    assert np.all((xs * 2.0 ** 53) == np.floor(xs * 2.0 ** 53))


def _ulps(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a.view(np.int64) - b.view(np.int64))


CASES = [
    ("sin", 0, lambda r, n: r.uniform(0, 2 * math.pi, n), np.sin),
    ("cos", 1, lambda r, n: r.uniform(0, 2 * math.pi, n), np.cos),
    ("tan", 2, lambda r, n: r.uniform(0, math.pi, n), np.tan),
    ("tan_near_pole", 2, lambda r, n: math.pi / 2 + r.uniform(-1e-6, 1e-6, n), np.tan),
    ("log", 3, lambda r, n: 1.0 - r.uniform(0, 1, n), np.log),
    ("log_tiny", 3, lambda r, n: 2.0 ** -r.uniform(0, 53, n), np.log),
    ("exp", 4, lambda r, n: -r.uniform(0, 60, n), np.exp),
    ("exp_large_negative", 4, lambda r, n: -10 ** r.uniform(0, 2.85, n), np.exp),
    ("acos", 5, lambda r, n: r.uniform(-1, 1, n), np.arccos),
    ("acos_near_one", 5, lambda r, n: 1 - 10 ** -r.uniform(0, 16, n), np.arccos),
]


@pytest.mark.parametrize("name,fn,gen,ref", CASES, ids=[c[0] for c in CASES])
def test_portable_function_within_2_ulp_of_libm(name, fn, gen, ref):
    _oracle.set_math_mode(False)
    x = gen(np.random.default_rng(fn), 20000)
    got = _oracle.math_fn(fn, x)
    want = ref(x)
    ok = np.isfinite(want) & (want != 0)
    assert _ulps(got[ok], want[ok]).max() <= 2


def test_atan2_within_2_ulp_and_special_cases():
    _oracle.set_math_mode(False)
    r = np.random.default_rng(6)
    y, x = r.normal(size=20000), r.normal(size=20000)
    assert _ulps(_oracle.math_fn(6, y, x), np.arctan2(y, x)).max() <= 2
    for yy, xx in [(0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (0.0, 0.0)]:
        assert _oracle.math_fn(6, [yy], [xx])[0] == math.atan2(yy, xx)


def test_special_values():
    _oracle.set_math_mode(False)
    f = lambda fn, x: _oracle.math_fn(fn, [x])[0]
    assert f(5, 1.0) == 0.0 and f(5, -1.0) == math.pi and f(5, 0.0) == math.acos(0.0)
    assert math.isnan(f(5, 1.0000000000000002)) and math.isnan(f(5, float("nan")))
    assert f(3, 1.0) == 0.0 and f(3, 0.0) == -math.inf and math.isnan(f(3, -1.0))
    assert f(4, 0.0) == 1.0 and f(4, -800.0) == 0.0 and f(4, -745.2) == 0.0
    assert f(4, -740.0) == math.exp(-740.0)  # subnormal result, single rounding
    # tan(acos(0)) is finite, so the is_infinite() guards never fire
    assert f(2, f(5, 0.0)) == 1.633123935319537e16


def test_libm_mode_is_the_platform_libm():
    _oracle.set_math_mode(True)
    try:
        x = np.random.default_rng(1).uniform(0, 2 * math.pi, 1000)
        assert np.array_equal(_oracle.math_fn(0, x), np.array([math.sin(v) for v in x]))
    finally:
        _oracle.set_math_mode(False)
