import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liboracle.so) -- the checker, never the thing under test."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def capi():
    import starkpack_winterfell_amd.capi as capi
    return capi


@pytest.fixture(scope="session")
def ctx(capi):
    """HIP context on device 0.  No fallback: fails loudly when the library or the GPU is missing."""
    c = capi.Context(0)
    yield c
    c.close()


F64_P = 2**64 - 2**32 + 1
F128_P = 2**128 - 45 * 2**40 + 1


def rand_f64(rng, n):
    """uniform Montgomery residues in [0, p) (Montgomery form is a bijection of [0,p))."""
    v = rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64, endpoint=True)
    bad = v >= np.uint64(F64_P)
    v[bad] -= np.uint64(F64_P)
    return v


def rand_f128(rng, n):
    """uniform-ish canonical elements of the 128-bit field as (n, 2) uint64 (lo, hi)."""
    v = rng.integers(0, 2**64 - 1, size=(n, 2), dtype=np.uint64, endpoint=True)
    # p = 2^128 - 45*2^40 + 1: hi == 2^64-1 and lo >= 0xFFFFD30000000001 is out of range; clear the top bit there
    bad = (v[:, 1] == np.uint64(2**64 - 1)) & (v[:, 0] >= np.uint64(0xFFFFD30000000001))
    v[bad, 1] = np.uint64(2**63)
    return v


def rand_cols(rng, field, n_cols, n_elems):
    if field == 1:
        return [rand_f64(rng, n_elems) for _ in range(n_cols)]
    return [rand_f128(rng, n_elems) for _ in range(n_cols)]
