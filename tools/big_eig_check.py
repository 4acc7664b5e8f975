import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import helfem_amd as hf
for n in (5000, 5200, 6102, 8200):
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n)); A = np.asfortranarray(A + A.T)
    t = time.time(); E, C = hf.scf.eig_sym(A); dt = time.time() - t
    w = np.linalg.eigvalsh(A)
    res = np.max(np.abs(A @ C - C * E)); orth = np.max(np.abs(C.T @ C - np.eye(n)))
    print("n=%d: %.2f s, max eigenvalue error %.2e, residual %.2e, orthogonality %.2e (scale %.1f)" % (n, dt, np.max(np.abs(E - w)), res, orth, np.max(np.abs(w))), flush=True)
