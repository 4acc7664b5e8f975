"""Persistent tridiagonalisation (hip/trdp.hip) on the GPU: correctness against LAPACK and stage timing.

  python tools/trdp_check.py [sizes...]           single matrices through hfg_eig_sym
  HELFEM_TRD=chain python tools/trdp_check.py     the launch chain for comparison
  HELFEM_TRDP_STAMPS=1 ...                        phase durations per column (stderr)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import helfem_amd as hf  # noqa: E402


def check_single(n, seed=0):
    rng = np.random.RandomState(seed + n)
    A = rng.standard_normal((n, n))
    A = np.asfortranarray(A + A.T)
    ctx = hf.default_context()
    ctx.profile(True)
    ctx.profile_reset()
    E, C = hf.scf.eig_sym(A)
    ms, _ = ctx.profile_get("eig_tridiag")
    ctx.profile(False)
    Er = np.linalg.eigvalsh(A)
    sc = np.max(np.abs(Er))
    err = np.max(np.abs(E - Er)) / sc
    res = np.max(np.abs(A @ C - C * E)) / sc
    orth = np.max(np.abs(C.T @ C - np.eye(n)))
    print("n = %5d  eig_tridiag %8.3f ms (%.2f us/column)   |E - lapack| %.1e   residual %.1e   orthogonality %.1e"
          % (n, ms, ms * 1e3 / max(1, n - 2), err, res, orth), flush=True)
    assert err < 1e-12 and res < 1e-11 and orth < 1e-11, (err, res, orth)


def check_blocks(sizes, reps=3):
    N = sum(sizes)
    rng = np.random.RandomState(7)
    F = np.zeros((N, N), order="F")
    blocks = []
    off = 0
    for n in sizes:
        B = rng.standard_normal((n, n))
        F[off:off + n, off:off + n] = B + B.T
        blocks.append(np.arange(off, off + n))
        off += n
    X = np.asfortranarray(np.eye(N))
    ctx = hf.default_context()
    hf.scf.eig_gsym_sub(F, X, blocks)  # warm-up
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.time()
    for _ in range(reps):
        E, C = hf.scf.eig_gsym_sub(F, X, blocks)
    dt = (time.time() - t0) / reps
    ms, _ = ctx.profile_get("eig_tridiag")
    ctx.profile(False)
    Er = np.sort(np.concatenate([np.linalg.eigvalsh(F[np.ix_(b, b)]) for b in blocks]))
    sc = np.max(np.abs(Er))
    err = np.max(np.abs(E - Er)) / sc
    res = np.max(np.abs(F @ C - C * E)) / sc
    E2, C2 = hf.scf.eig_gsym_sub(F, X, blocks)
    same = np.array_equal(E, E2) and np.array_equal(C, C2)
    print("blocks %s  eig_tridiag %8.3f ms per solve (%.2f us/column of the largest)  host wall %.1f ms   |E - lapack| %.1e   residual %.1e   bitwise repeatable %s"
          % (sizes, ms / reps, ms / reps * 1e3 / (max(sizes) - 2), dt * 1e3, err, res, same), flush=True)
    assert err < 1e-12 and res < 1e-11 and same, (err, res, same)


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [259, 300, 700, 1000, 1400, 1536]
    print("HELFEM_TRD =", os.environ.get("HELFEM_TRD", "(default: persistent where it fits)"))
    for n in sizes:
        check_single(n)
    check_blocks([1380, 1470, 1380])
    check_blocks([700, 650])
    check_blocks([300] * 8)
    check_blocks([2100, 2001, 2001], reps=2)  # two groups: {2100} and {2001, 2001}
    check_blocks([1700, 300, 2500], reps=1)   # 2500 is beyond the register tiles: that block takes the launch chain
