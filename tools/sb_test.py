"""Two-stage tridiagonalisation, stage 1 (dense -> band): eigenvalues of the band matrix against the input's, and timing."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helfem_amd as hf

import ctypes as _ct
from helfem_amd import build as _b
L = _ct.CDLL(_b.build_probe(verbose=False))  # tests/gpu_probe/libtwostage_probe.so
ctx = hf.default_context()
dp = ctypes.POINTER(ctypes.c_double)
L.probe_band_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64, dp, ctypes.c_int, dp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), dp]
for n, nrep in ((200, 1), (333, 2), (1380, 1), (1470, 3), (1400, 3)):
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n))
    A = np.asfortranarray(A + A.T)
    AB = np.zeros(n * 64)
    bw, ldb, ms = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    rc = L.probe_band_reduce(ctx.h, n, A.ctypes.data_as(dp), nrep, AB.ctypes.data_as(dp), ctypes.byref(bw), ctypes.byref(ldb), ctypes.byref(ms))
    if rc:
        print("FAILED", n, hf.lib().hfg_last_error())
        continue
    b, ld = bw.value, ldb.value
    AB = AB.reshape(n, ld)
    B = np.zeros((n, n))
    for d in range(b + 1):
        v = AB[: n - d, d]
        B[np.arange(d, n), np.arange(0, n - d)] = v
        B[np.arange(0, n - d), np.arange(d, n)] = v
    w0, w1 = np.linalg.eigvalsh(A), np.linalg.eigvalsh(B)
    print("n=%d x%d: stage 1 %.3f ms, max eigenvalue error %.2e (scale %.1f), bulge rows zero: %s" % (
        n, nrep, ms.value, np.max(np.abs(w0 - w1)), np.max(np.abs(w0)), bool(np.all(AB[:, b + 1:] == 0.0))), flush=True)

# both stages: eigenvalues of the tridiagonal matrix against the input's
import scipy.linalg as sl
L.probe_two_stage.argtypes = [ctypes.c_void_p, ctypes.c_int64, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp]
for n, nrep, G, delayed in ((200, 1, 4, 0), (333, 2, 8, 0), (1470, 3, 16, 0), (1470, 3, 16, 1), (1470, 3, 24, 0), (1400, 3, 12, 0)):
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n))
    A = np.asfortranarray(A + A.T)
    d, e = np.zeros(n), np.zeros(n)
    m1, m2 = ctypes.c_double(), ctypes.c_double()
    rc = L.probe_two_stage(ctx.h, n, A.ctypes.data_as(dp), nrep, G, delayed, d.ctypes.data_as(dp), e.ctypes.data_as(dp), ctypes.byref(m1), ctypes.byref(m2))
    if rc:
        print("FAILED", n, hf.lib().hfg_last_error(), flush=True)
        break
    w0 = np.linalg.eigvalsh(A)
    w1 = sl.eigvalsh_tridiagonal(d, e[:-1])
    print("n=%d x%d G=%d delayed=%d: stage 1 %.3f ms, stage 2 %.3f ms, max eigenvalue error %.2e (scale %.1f)" % (
        n, nrep, G, delayed, m1.value, m2.value, np.max(np.abs(w0 - w1)), np.max(np.abs(w0))), flush=True)
