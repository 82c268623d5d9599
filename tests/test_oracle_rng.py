"""The restated 48-bit LCG against the host libc's drand48 family (utils.h:84-137)."""
import ctypes as C
import ctypes.util

import pytest

from tests import oracle_binding as ob


def _libc():
    name = ctypes.util.find_library("c")
    if not name:
        pytest.skip("no libc to compare with")
    libc = C.CDLL(name)
    for f in ("srand48", "lrand48", "drand48"):
        if not hasattr(libc, f):
            pytest.skip("libc has no drand48 family")
    libc.drand48.restype = C.c_double
    libc.lrand48.restype = C.c_long
    libc.srand48.argtypes = [C.c_long]
    return libc


@pytest.mark.parametrize("seed", [1, 2, 12345, 2**31 - 1])
def test_against_libc(oracle, seed):
    libc = _libc()
    g = ob.Rand48()
    oracle.orc_srand48(C.byref(g), seed)
    libc.srand48(seed)
    for i in range(2000):
        if i % 3 == 0:
            assert oracle.orc_lrand48(C.byref(g)) == libc.lrand48()
        else:
            assert oracle.orc_drand48(C.byref(g)) == libc.drand48()


def test_jump_matches_stepping(oracle):
    for seed, n in ((1, 0), (1, 1), (7, 1000), (9, 8388608), (3, 8388608 + 12345)):
        a, b = ob.Rand48(), ob.Rand48()
        oracle.orc_srand48(C.byref(a), seed)
        oracle.orc_srand48(C.byref(b), seed)
        for _ in range(min(n, 20000)):
            oracle.orc_drand48(C.byref(a))
        if n > 20000:
            continue
        oracle.orc_rand48_jump(C.byref(b), n)
        assert a.x == b.x


def test_lazy_weight_equals_dense_init(oracle):
    """slot i of the dense initialisation (linear.cpp:117-120) == jump-ahead value"""
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=5)
    g = ob.Rand48()
    oracle.orc_srand48(C.byref(g), 5)
    tl_seed = oracle.orc_lrand48(C.byref(g))
    slots = [0, 1, 2, 1000, 8388607, 4194304, 123457]
    dense = e.weights(slots)
    for s, d in zip(slots, dense):
        assert oracle.orc_lazy_weight(tl_seed, 0, s, 0.0, 1.0) == d
