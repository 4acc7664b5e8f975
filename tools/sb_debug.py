"""Step-by-step comparison of stage 1 (tests/gpu_probe/two_stage.hip) with tools/two_stage_model.py: run with HELFEM_SB_NPANEL=1 and
HELFEM_SB_STEP=1..4 in the environment."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helfem_amd as hf
import two_stage_model as tm
import ctypes as _ct
from helfem_amd import build as _b
L = _ct.CDLL(_b.build_probe(verbose=False)); ctx = hf.default_context()
dp = ctypes.POINTER(ctypes.c_double)
L.probe_band_reduce_keep.argtypes = [ctypes.c_void_p, ctypes.c_int64, dp, dp]
L.probe_band_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, dp, ctypes.c_int64]
n, b = 200, 32
rng = np.random.RandomState(1)
A0 = rng.standard_normal((n, n)); A0 = np.asfortranarray(A0 + A0.T)
Aout = np.zeros((n, n), order="F")
assert L.probe_band_reduce_keep(ctx.h, n, A0.ctypes.data_as(dp), Aout.ctypes.data_as(dp)) == 0, hf.lib().hfg_last_error()
def fetch(which, count):
    out = np.zeros(count)
    assert L.probe_band_fetch(ctx.h, which, n, out.ctypes.data_as(dp), count) == 0
    return out
Vx = fetch(1, n * n).reshape((n, n), order="F")
T = fetch(2, 32 * 32).reshape((32, 32), order="F")
X = fetch(3, n * 32).reshape((n, 32), order="F")
Lm = fetch(4, n * 64).reshape((n, 64), order="F")
Rm = fetch(5, n * 64).reshape((n, 64), order="F")
# model, first panel by hand
A = A0.copy(); r0 = b; m = n - r0
P = A[r0:, :b].copy()
Vp = np.zeros((m, b)); tp = np.zeros(b); G = np.zeros((b, b))
for j in range(b):
    x = P[j:, j].copy(); alpha = x[0]; s = x[1:] @ P[j + 1:, :]; xn2 = s[j]
    nrm = np.sqrt(alpha * alpha + xn2); beta = -nrm if alpha >= 0 else nrm; t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta)
    w = P[j, :] + scale * s; vj = np.concatenate(([1.0], x[1:] * scale))
    for c in range(j + 1, b): P[j:, c] -= t * w[c] * vj
    G[:j, j] = w[:j]; P[j, j] = beta; P[j + 1:, j] = vj[1:]; tp[j] = t
for j in range(b): Vp[j, j] = 1.0; Vp[j + 1:, j] = P[j + 1:, j]
Tm = np.zeros((b, b))
for i in range(b):
    Tm[i, i] = tp[i]
    if i: Tm[:i, i] = -tp[i] * (Tm[:i, :i] @ G[:i, i])
print("V   err", np.max(np.abs(Vx[r0:, :b] - Vp)))
print("R   err", np.max(np.abs(np.triu(Aout[r0:r0 + b, :b]) - np.triu(P[:b, :]))))
print("T   err", np.max(np.abs(T - Tm)))
A22 = A0[r0:, r0:]
Xm = A22 @ Vp
print("X   err", np.max(np.abs(X[r0:, :] - Xm)), "scale", np.max(np.abs(Xm)))
Zm = Vp.T @ Xm; M2 = Tm.T @ Zm @ Tm; Y = Xm @ Tm; U = Y - Vp @ M2.T
print("Y   err", np.max(np.abs(Lm[r0:, :b] - Y)), " V in L", np.max(np.abs(Lm[r0:, b:] - Vp)), " V in R", np.max(np.abs(Rm[r0:, :b] - Vp)), " U err", np.max(np.abs(Rm[r0:, b:] - U)))
A22n = A22 - Y @ Vp.T - Vp @ U.T
print("A22 err", np.max(np.abs(Aout[r0:, r0:] - A22n)), "scale", np.max(np.abs(A22n)))
