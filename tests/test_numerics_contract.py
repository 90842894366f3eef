"""include/ptmi_math.h (the shared numerics contract) against float64 libm, and the
XORWOW skip-ahead machinery against brute-force stepping."""
import ctypes as C

import numpy as np

from oracle_binding import oracle_lib, rng_stream

F = np.float32


def _sincos(xs):
    L = oracle_lib()
    s = C.c_float(); c = C.c_float()
    out = np.zeros((len(xs), 2), F)
    for i, x in enumerate(xs):
        L.po_sincosf(float(x), C.byref(s), C.byref(c)); out[i] = (s.value, c.value)
    return out


def test_sincos_close_to_correctly_rounded():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), rng.uniform(-40, 40, 5000),
                         [0.0, np.pi / 2, np.pi, 2 * np.pi, 1e-30, 6.2831855]]).astype(F)
    got = _sincos(xs)
    ref = np.stack([np.sin(xs.astype(np.float64)), np.cos(xs.astype(np.float64))], 1)
    cr = ref.astype(F)                       # correctly rounded (float64 libm error << float ulp/2 boundary)
    assert (got == cr).mean() > 0.9999
    ulp = np.spacing(np.abs(cr)).astype(np.float64)
    assert (np.abs(got.astype(np.float64) - ref) <= 0.5000001 * ulp + 1e-45).all()


def test_powf_gamma():
    L = oracle_lib()
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(0, 1, 20000), 10.0 ** rng.uniform(-30, 0, 2000), [0.0, 1.0, 0.5]]).astype(F)
    g = F(1.0) / F(2.2)
    got = np.array([L.po_powf(float(x), float(g)) for x in xs], F)
    ref = np.power(xs.astype(np.float64), np.float64(g))
    assert (got == ref.astype(F)).mean() > 0.9999
    rel = np.abs(got.astype(np.float64) - ref) / np.maximum(ref, 1e-300)
    assert rel[xs > 0].max() < 6.1e-8
    assert L.po_powf(0.0, float(g)) == 0.0 and L.po_powf(1.0, float(g)) == 1.0


def test_expf_for_the_filter_weights():
    """ptmi_expf (grid_filter.h:35-37 gaussianWeight): correctly rounded on > 99.99 % of the range the filters use
    (x <= 0), exact limits: exp(0) = 1, underflow to 0 below -104, denormal results, +inf above 88.75, NaN through."""
    L = oracle_lib()
    rng = np.random.default_rng(3)
    xs = np.concatenate([-rng.uniform(0, 20, 20000), -10.0 ** rng.uniform(-30, 2, 5000), rng.uniform(0, 80, 2000),
                         [0.0, -0.0, -87.3, -88.0, -95.0, -103.0, -103.9]]).astype(F)
    got = np.array([L.po_expf(float(x)) for x in xs], F)
    ref = np.exp(xs.astype(np.float64))
    with np.errstate(over="ignore", under="ignore"):
        cr = ref.astype(F)
    assert (got == cr).mean() > 0.9999
    ulp = np.spacing(np.maximum(np.abs(cr), np.finfo(F).tiny)).astype(np.float64)
    assert (np.abs(got.astype(np.float64) - ref) <= 0.5000001 * ulp + 1e-45).all()
    assert L.po_expf(0.0) == 1.0 and L.po_expf(-104.5) == 0.0 and L.po_expf(-1e30) == 0.0
    assert 0.0 < L.po_expf(-100.0) < np.finfo(F).tiny                       # denormal, not flushed
    assert L.po_expf(89.0) == np.inf and np.isnan(L.po_expf(float("nan")))


def test_xorwow_matrix_powers_equal_direct_stepping():
    L = oracle_lib()
    L.po_rng_selftest.argtypes = [C.c_int, C.c_void_p]
    rng = np.random.default_rng(5)
    for log2n in (0, 1, 5, 12, 18):
        v = rng.integers(0, 2 ** 32, 5, dtype=np.uint64).astype(np.uint32)
        assert L.po_rng_selftest(log2n, v.ctypes.data) == 0


def test_curand_uniform_range_and_streams():
    u0, st0 = rng_stream(2023, 0, 4096)
    assert (u0 > 0).all() and (u0 <= 1).all() and abs(float(u0.mean()) - 0.5) < 0.02
    # subsequence 0 applies no jump: state = seed scramble, then plain xorwow steps
    s = np.uint32(2023) ^ np.uint32(0xaad26b49); t0 = np.uint32((1099087573 * int(s)) & 0xffffffff)
    t1 = np.uint32((2591861531 * 0xf7dcefdd) & 0xffffffff)
    v = [np.uint32((123456789 + int(t0)) & 0xffffffff), np.uint32(362436069) ^ t0,
         np.uint32((521288629 + int(t1)) & 0xffffffff), np.uint32(88675123) ^ t1, np.uint32((5783321 + int(t0)) & 0xffffffff)]
    d = (6615241 + int(t1) + int(t0)) & 0xffffffff
    outs = []
    for _ in range(4):
        t = int(v[0]) ^ (int(v[0]) >> 2)
        v = v[1:] + [np.uint32((int(v[4]) ^ ((int(v[4]) << 4) & 0xffffffff)) ^ (t ^ ((t << 1) & 0xffffffff)))]
        d = (d + 362437) & 0xffffffff
        x = (int(v[4]) + d) & 0xffffffff
        outs.append(F(F(x) * F(2.3283064e-10) + F(2.3283064e-10) / F(2.0)))
    assert (u0[:4] == np.array(outs, F)).all()
    # different pixels -> different streams
    u1, _ = rng_stream(2024, 1, 64)
    assert not (u0[:64] == u1).any()
