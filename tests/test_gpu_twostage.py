"""Two-stage tridiagonalisation (tests/gpu_probe/two_stage.hip, a probe library of its own: dense -> band by panel QR and compact-WY updates, band -> tridiagonal by
bulge chasing).  Not the product path (measured slower than the one-stage chain) but kept correct:
the eigenvalues of the band matrix and of the final tridiagonal matrix must be the input's."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hf(native_libs):
    import helfem_amd
    return helfem_amd



def _probe_lib():
    """tests/gpu_probe/libtwostage_probe.so (helfem_amd/build.py build_probe): the two-stage reduction is a probe, not product"""
    import ctypes, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(here, "tests", "gpu_probe", "libtwostage_probe.so")
    if not os.path.exists(path):
        from helfem_amd import build
        build.build_probe(verbose=False)
    return ctypes.CDLL(path)


def _sym(n, seed):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal((n, n))
    return np.asfortranarray(A + A.T)


@pytest.mark.parametrize("n,nrep", [(200, 1), (333, 2), (1000, 3)])
def test_band_reduction_preserves_the_spectrum(hf, n, nrep):
    L, ctx = _probe_lib(), hf.default_context()
    dp = ctypes.POINTER(ctypes.c_double)
    L.probe_band_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64, dp, ctypes.c_int, dp, ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_int), dp]
    A = _sym(n, n)
    AB = np.zeros(n * 64)
    bw, ldb, ms = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    assert L.probe_band_reduce(ctx.h, n, A.ctypes.data_as(dp), nrep, AB.ctypes.data_as(dp), ctypes.byref(bw), ctypes.byref(ldb),
                                   ctypes.byref(ms)) == 0, hf.lib().hfg_last_error()
    b, ld = bw.value, ldb.value
    AB = AB.reshape(n, ld)
    assert np.all(AB[:, b + 1:] == 0.0)
    B = np.zeros((n, n))
    for d in range(b + 1):
        v = AB[: n - d, d]
        B[np.arange(d, n), np.arange(0, n - d)] = v
        B[np.arange(0, n - d), np.arange(d, n)] = v
    w0, w1 = np.linalg.eigvalsh(A), np.linalg.eigvalsh(B)
    assert np.max(np.abs(w0 - w1)) < 1e-12 * n * np.max(np.abs(w0))


@pytest.mark.parametrize("n,nrep,G", [(200, 1, 4), (333, 2, 8), (1000, 3, 12)])
def test_bulge_chasing_preserves_the_spectrum(hf, n, nrep, G):
    import scipy.linalg as sl
    L, ctx = _probe_lib(), hf.default_context()
    dp = ctypes.POINTER(ctypes.c_double)
    L.probe_two_stage.argtypes = [ctypes.c_void_p, ctypes.c_int64, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp]
    A = _sym(n, 7 * n)
    d, e = np.zeros(n), np.zeros(n)
    m1, m2 = ctypes.c_double(), ctypes.c_double()
    assert L.probe_two_stage(ctx.h, n, A.ctypes.data_as(dp), nrep, G, 0, d.ctypes.data_as(dp), e.ctypes.data_as(dp), ctypes.byref(m1),
                                 ctypes.byref(m2)) == 0, hf.lib().hfg_last_error()
    w0 = np.linalg.eigvalsh(A)
    w1 = sl.eigvalsh_tridiagonal(d, e[:-1])
    assert np.max(np.abs(w0 - w1)) < 1e-12 * n * np.max(np.abs(w0))
