"""Eigensolve of one large symmetric matrix (orders beyond the fused tridiagonalisation's 8064: the two-launch path),
stage times from the context's profiler and a residual check on the lowest and highest eigenpairs."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import helfem_amd as hf
ctx = hf.default_context()
for n in [int(a) for a in sys.argv[1:]] or [9000, 13000]:
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n)); A = np.asfortranarray(A + A.T)
    ctx.profile(True); ctx.profile_reset()
    t = time.time(); E, C = hf.scf.eig_sym(A, ctx); dt = time.time() - t
    st = {k: round(ctx.profile_get(k)[0], 1) for k in ("eig_tridiag", "eig_tridiag_solve", "eig_backtransform", "gemm")}
    ctx.profile(False)
    idx = np.r_[0:4, n - 4:n]
    res = np.max(np.abs(A @ C[:, idx] - C[:, idx] * E[idx])); orth = np.max(np.abs(C[:, idx].T @ C - np.eye(n)[idx]))
    print("n=%d: %.2f s host to host; stages ms %s; residual %.2e, orthogonality %.2e, trace error %.2e" % (
        n, dt, st, res, orth, abs(E.sum() - np.trace(A))), flush=True)
