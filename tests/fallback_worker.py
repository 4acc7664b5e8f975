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
# generalized problem by symmetry blocks (the path with X: back-transformation folded into X or not, tile choices of the products)
sizes = [300, 641, 33]
N = sum(sizes)
F = np.zeros((N, N), order="F")
S = np.zeros((N, N), order="F")
blocks, off = [], 0
for n in sizes:
    B = rng.uniform(-1, 1, (n, n))
    F[off:off + n, off:off + n] = B + B.T
    G = rng.uniform(-1, 1, (n, n))
    S[off:off + n, off:off + n] = G @ G.T + n * np.eye(n)
    blocks.append(np.arange(off, off + n))
    off += n
X = hf.scf.form_Sinvh(S, False, blocks)
E, C = hf.scf.eig_gsym_sub(F, X, blocks)
import scipy.linalg
Er = np.sort(np.concatenate([scipy.linalg.eigh(F[np.ix_(b, b)], S[np.ix_(b, b)], eigvals_only=True) for b in blocks]))
assert np.max(np.abs(E - Er)) < 1e-10 * max(1.0, np.max(np.abs(Er))), np.max(np.abs(E - Er))
assert np.max(np.abs(C.T @ S @ C - np.eye(N))) < 1e-9
assert np.max(np.abs(F @ C - S @ C * E)) < 1e-8 * max(1.0, np.max(np.abs(Er)))
print("ok")
