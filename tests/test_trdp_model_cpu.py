"""The specification of the persistent tridiagonalisation (tools/trdp_model.py: one exchange per Householder column, products
formed with the UNNORMALISED column before the previous rank-2 update has been applied, the next row published raw) against
a textbook dsytd2 and against the spectrum -- the algebra hip/trdp.hip follows, checked without a GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import trdp_model as tm  # noqa: E402


@pytest.mark.parametrize("n,G", [(3, 1), (7, 2), (33, 5), (64, 8), (97, 11)])
def test_one_exchange_per_column_reproduces_dsytd2(n, G):
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n))
    A = A + A.T
    d0, e0, t0, V0 = tm.dsytd2_lower(A)
    d, e, tau, V = tm.persistent_model(A, G=G)
    sc = np.max(np.abs(A))
    assert np.max(np.abs(d - d0)) < 1e-12 * sc and np.max(np.abs(e[:n - 1] - e0[:n - 1])) < 1e-12 * sc
    assert np.max(np.abs(tau[:n - 2] - t0[:n - 2])) < 1e-12
    assert np.max(np.abs(np.tril(V, -1) - np.tril(V0, -1))) < 1e-11
    # and the tridiagonal matrix has the spectrum of A
    T = np.diag(d) + np.diag(e[:n - 1], 1) + np.diag(e[:n - 1], -1)
    assert np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A))) < 1e-11 * sc
