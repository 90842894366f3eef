"""The oracle's XORWOW against a third party's implementation of the same generator: rocRAND (ROCm 7.2, ROCRAND_VERSION
400200, /opt/rocm/include/rocrand/rocrand_xorwow.h), through oracle/rocrand_harness.cpp.

cuRAND itself (what the reference calls, integrator.h:63-64, 210, 279, 384-385) is not in the image.  rocRAND implements the
same engine - xorwow recurrence, Weyl sequence, output d + x[4], subsequences of 2^67 draws skipped with precomputed GF(2)
matrix powers - but seeds it with other scramble constants and converts to float with another offset.  So this pins, bit for
bit: the step, and the skip over n subsequences (the oracle builds T^(2^67 * 2^k) by 67 + k squarings, rocRAND ships
precomputed A^(2^67 * 4^k)).  It does NOT pin curand_init's seed scramble or curand_uniform's x * 2^-32 + 2^-33; those stay
restated from cuRAND's published header.  The device path is tied to the oracle by tests/test_gpu_parity.py
(test_rng_streams_match_oracle)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_binding import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_SO = os.path.join(ROOT, "oracle", "librocrand_pin.so")


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(PIN_SO):
        pytest.skip("oracle/librocrand_pin.so not built (make -C oracle)")
    R = C.CDLL(PIN_SO)
    L = oracle_lib()
    u32p = C.POINTER(C.c_uint32)
    R.rr_version.restype = C.c_int
    R.rr_next.argtypes = [u32p, u32p, C.c_int, u32p]
    R.rr_skip_subsequences.argtypes = [u32p, u32p, C.c_ulonglong]
    R.rr_skip.argtypes = [u32p, u32p, C.c_ulonglong]
    L.po_xorwow_next_raw.argtypes = [u32p, C.c_int, u32p]
    L.po_xorwow_skip_subsequences.argtypes = [u32p, C.c_uint64]
    L.po_rng_init.argtypes = [C.c_uint64, C.c_uint64, u32p]
    return R, L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def test_rocrand_version(libs):
    assert libs[0].rr_version() >= 300000


def test_step_matches_rocrand(libs):
    R, L = libs
    rng = np.random.default_rng(11)
    for trial in range(50):
        st = rng.integers(0, 2 ** 32, 6, dtype=np.uint64).astype(np.uint32)      # x[0..4], d
        n = 257
        x = st[:5].copy(); d = st[5:6].copy(); out_r = np.zeros(n, np.uint32)
        R.rr_next(_p(x), _p(d), n, _p(out_r))
        mine = st.copy(); out_o = np.zeros(n, np.uint32)
        L.po_xorwow_next_raw(_p(mine), n, _p(out_o))
        assert (out_r == out_o).all()
        assert (mine[:5] == x).all() and mine[5] == d[0]


def test_subsequence_skip_matches_rocrands_precomputed_matrices(libs):
    """T^(2^67 * n) for the n that pixel indices take (and bit patterns across all 32 bits)."""
    R, L = libs
    rng = np.random.default_rng(12)
    ns = [0, 1, 2, 3, 4, 5, 7, 255, 256, 1023, 1024 * 1024 - 1, 1024 * 1024, 4096 * 4096 - 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]
    ns += [int(v) for v in rng.integers(0, 2 ** 32, 40, dtype=np.uint64)]
    for n in ns:
        v = rng.integers(0, 2 ** 32, 5, dtype=np.uint64).astype(np.uint32)
        x = v.copy(); d = np.array([12345], np.uint32)
        R.rr_skip_subsequences(_p(x), _p(d), n)
        mine = v.copy()
        L.po_xorwow_skip_subsequences(_p(mine), n)
        assert (mine == x).all(), n
        assert d[0] == 12345                                                    # 2^67 is a multiple of 2^32: the Weyl value stays


def test_rng_init_is_the_scramble_followed_by_rocrands_skip(libs):
    """po_rng_init(seed, subsequence) = cuRAND's published seed scramble (restated, unpinned) + the pinned skip, and its
    draws are the pinned step."""
    R, L = libs
    for seed, sub in ((2023, 0), (2023 + 77, 77), (2023 + 1048575, 1048575), (12345 + 999, 8192 * 8192 - 1)):
        st0 = np.zeros(6, np.uint32); L.po_rng_init(seed, 0, _p(st0))
        st = np.zeros(6, np.uint32); L.po_rng_init(seed, sub, _p(st))
        x = st0[:5].copy(); d = st0[5:6].copy()
        R.rr_skip_subsequences(_p(x), _p(d), sub)
        assert (st[:5] == x).all() and st[5] == d[0]
        out_r = np.zeros(64, np.uint32); R.rr_next(_p(x), _p(d), 64, _p(out_r))
        out_o = np.zeros(64, np.uint32); L.po_xorwow_next_raw(_p(st), 64, _p(out_o))
        assert (out_r == out_o).all()


def test_rocrands_offset_skip_equals_direct_stepping(libs):
    """sanity of the harness itself: rocRAND's discard(n) against n single steps"""
    R, _ = libs
    rng = np.random.default_rng(13)
    for n in (1, 2, 3, 17, 1000, 65537):
        v = rng.integers(0, 2 ** 32, 6, dtype=np.uint64).astype(np.uint32)
        x = v[:5].copy(); d = v[5:6].copy()
        R.rr_skip(_p(x), _p(d), n)
        y = v[:5].copy(); e = v[5:6].copy(); out = np.zeros(n, np.uint32)
        R.rr_next(_p(y), _p(e), n, _p(out))
        assert (x == y).all() and d[0] == e[0]
