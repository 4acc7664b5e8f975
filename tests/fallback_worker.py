"""Worker of test_gpu_parity.py::test_fallback_variants: the alternative kernels kept behind environment switches
(read once per process) must give the same eigen-decomposition."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import helfem_amd as hf  # noqa: E402

rng = np.random.RandomState(11)
for n in (3, 4, 5, 70, 128, 129, 131, 190, 191, 192, 193, 195, 208, 300, 641, 1000):
    A = rng.uniform(-1, 1, (n, n))
    A = A + A.T
    E, C = hf.scf.eig_sym(A)
    Eref = np.linalg.eigvalsh(A)
    assert np.max(np.abs(E - Eref)) < 1e-11 * n, (n, np.max(np.abs(E - Eref)))
    assert np.max(np.abs(C.T @ C - np.eye(n))) < 1e-11 * n
    assert np.max(np.abs(A @ C - C * E)) < 1e-10 * n
print("ok")
