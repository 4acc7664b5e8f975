"""NumPy model of the two-stage tridiagonalisation in tests/gpu_probe/two_stage.hip (stage 1: dense -> band by panel
QR + compact-WY two-sided updates; stage 2: band -> tridiagonal by bulge chasing, one column per sweep) and of the two
back-transformations.  Same index conventions and storage as the kernels; used to fix them before any GPU run and kept
as the readable statement of the algorithm.  Also checks the wavefront schedule of stage 2: tasks (s, k) with equal
3 s + k are independent.

    python tools/two_stage_model.py [n] [b]
"""
import sys
import numpy as np


def house(x):
    """LAPACK dlarfg: H = I - tau v v^T, v[0] = 1, H x = beta e1"""
    alpha = x[0]
    xn2 = float(np.dot(x[1:], x[1:]))
    if xn2 == 0.0:
        return np.concatenate(([1.0], np.zeros(len(x) - 1))), 0.0, alpha
    nrm = np.sqrt(alpha * alpha + xn2)
    beta = -nrm if alpha >= 0 else nrm
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def stage1(A, b):
    """A (n x n symmetric, full storage) -> band of half-width b in place; returns V (n x n explicit reflector
    columns: column c acts on rows c + b ..), tau (n), T blocks"""
    n = A.shape[0]
    V = np.zeros((n, n))
    tau = np.zeros(n)
    Ts = []
    j0 = 0
    while n - j0 - b >= 2:
        m = n - j0 - b
        r0 = j0 + b
        P = A[r0:, j0:j0 + b].copy()
        G = np.zeros((b, b))
        Vp = np.zeros((m, b))
        tp = np.zeros(b)
        for j in range(min(b, m)):
            x = P[j:, j].copy()
            alpha = x[0]
            s = x[1:] @ P[j + 1:, :]            # unnormalised dots with every column (what the kernel reduces)
            xn2 = s[j]
            if xn2 == 0.0:
                t, beta, scale = 0.0, alpha, 0.0
            else:
                nrm = np.sqrt(alpha * alpha + xn2)
                beta = -nrm if alpha >= 0 else nrm
                t = (beta - alpha) / beta
                scale = 1.0 / (alpha - beta)
            w = P[j, :] + scale * s               # v^T P[:, c]; for c < j this is the Gram entry v_c^T v_j
            vj = np.concatenate(([1.0], x[1:] * scale))
            for c in range(j + 1, b):
                P[j:, c] -= t * w[c] * vj
            G[:j, j] = w[:j]
            P[j, j] = beta
            P[j + 1:, j] = vj[1:]
            tp[j] = t
        for j in range(min(b, m)):
            Vp[j, j] = 1.0
            Vp[j + 1:, j] = P[j + 1:, j]
        # T (dlarft forward columnwise) from the Gram entries
        T = np.zeros((b, b))
        for i in range(b):
            T[i, i] = tp[i]
            if i:
                T[:i, i] = -tp[i] * (T[:i, :i] @ G[:i, i])
        R = np.triu(P[:b, :]) if m >= b else np.triu(P)
        A[r0:, j0:j0 + b] = 0.0
        A[r0:r0 + R.shape[0], j0:j0 + b] = R
        A[j0:j0 + b, r0:] = A[r0:, j0:j0 + b].T
        # two-sided update of the trailing matrix
        A22 = A[r0:, r0:]
        X = A22 @ Vp
        Zm = Vp.T @ X
        M2 = T.T @ Zm @ T
        Y = X @ T
        U = Y - Vp @ M2.T
        # A22 <- A22 - Y V^T - V U^T      (U = Y - V M2^T, so that the V M2 V^T term is included once)
        A22 -= Y @ Vp.T + Vp @ U.T
        V[r0:, j0:j0 + b] = Vp
        tau[j0:j0 + b] = tp
        Ts.append(T)
        j0 += b
    return V, tau, Ts


def to_band(A, b):
    n = A.shape[0]
    LDB = 2 * b
    AB = np.zeros((n, LDB))
    for j in range(n):
        for d in range(min(b + 1, n - j)):
            AB[j, d] = A[j + d, j]
    return AB


class Band:
    """lower band storage with room for the bulge: AB[j, d] = A[j + d, j], d < 2 b"""

    def __init__(self, AB, b):
        self.AB, self.b, self.n = AB, b, AB.shape[0]

    def get(self, rows, cols):
        out = np.zeros((len(rows), len(cols)))
        for a, i in enumerate(rows):
            for c, j in enumerate(cols):
                lo, hi = (i, j) if i >= j else (j, i)
                d = lo - hi
                if d < 2 * self.b:
                    out[a, c] = self.AB[hi, d]
        return out

    def put(self, rows, cols, M, tol_outside=1e-300):
        for a, i in enumerate(rows):
            for c, j in enumerate(cols):
                if i < j:
                    continue
                d = i - j
                if d < 2 * self.b:
                    self.AB[j, d] = M[a, c]
                else:
                    assert abs(M[a, c]) <= tol_outside, (i, j, M[a, c])


def task(band, s, k, VV, tau2):
    """bulge-chasing task k (1-based) of sweep s; returns False when the sweep has ended"""
    b, n = band.b, band.n
    r_first = s + (k - 1) * b + 1
    r_last = min(s + k * b, n - 1)
    if r_last - r_first + 1 < 2:
        return False
    R = list(range(r_first, r_last + 1))
    c0 = s if k == 1 else s + (k - 2) * b + 1
    x = band.get(R, [c0])[:, 0]
    v, t, beta = house(x.copy())
    VV[R, s] = v
    tau2[s, k - 1] = t
    xnew = np.zeros(len(R))
    xnew[0] = beta
    band.put(R, [c0], xnew[:, None])
    H = np.eye(len(R)) - t * np.outer(v, v)
    if k >= 2:
        Cp = list(range(c0 + 1, r_first))  # the other columns of the bulge block
        if Cp:
            Bk = band.get(R, Cp)
            band.put(R, Cp, H @ Bk)
    D = band.get(R, R)
    band.put(R, R, H @ D @ H)
    n_first = r_last + 1
    n_last = min(r_last + b, n - 1)
    if n_first <= n_last:
        Rn = list(range(n_first, n_last + 1))
        Bn = band.get(Rn, R)
        band.put(Rn, R, Bn @ H)
    return True


def stage2(AB, b, order="sweep"):
    n = AB.shape[0]
    band = Band(AB, b)
    VV = np.zeros((n, n))
    ntask = (n + b - 1) // b + 1
    tau2 = np.zeros((n, ntask))
    if order == "sweep":
        for s in range(n - 2):
            k = 1
            while task(band, s, k, VV, tau2):
                k += 1
    else:
        # wavefront order: all tasks with equal 3 s + k, in a scrambled order inside a front
        rng = np.random.default_rng(1)
        alive = {}
        tmax = 3 * (n - 3) + ntask + 1
        for t in range(1, tmax + 1):
            front = [(s, t - 3 * s) for s in range(0, n - 2) if 1 <= t - 3 * s <= ntask]
            rng.shuffle(front)
            for s, k in front:
                if alive.get(s, True):
                    alive[s] = task(band, s, k, VV, tau2)
    d = AB[:, 0].copy()
    e = np.concatenate((AB[:-1, 1], [0.0]))
    return d, e, VV, tau2


def backtransform2(Z, VV, tau2, b):
    n = Z.shape[0]
    for s in range(n - 3, -1, -1):
        k = 1
        while True:
            r_first = s + (k - 1) * b + 1
            r_last = min(s + k * b, n - 1)
            if r_last - r_first + 1 < 2:
                break
            v = VV[r_first:r_last + 1, s]
            t = tau2[s, k - 1]
            Z[r_first:r_last + 1, :] -= t * np.outer(v, v @ Z[r_first:r_last + 1, :])
            k += 1


def backtransform1(Z, V, Ts, b):
    n = Z.shape[0]
    for p in range(len(Ts) - 1, -1, -1):
        j0 = p * b
        Vp = V[:, j0:j0 + b]
        Z -= Vp @ (Ts[p] @ (Vp.T @ Z))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rng = np.random.default_rng(0)
    A0 = rng.standard_normal((n, n))
    A0 = A0 + A0.T
    A = A0.copy()
    V, tau, Ts = stage1(A, b)
    off = max(abs(A[i, j]) for i in range(n) for j in range(n) if abs(i - j) > b) if n > b + 1 else 0.0
    print("stage 1: max |A_ij| outside the band", off, " eig err", np.max(np.abs(np.linalg.eigvalsh(A) - np.linalg.eigvalsh(A0))))
    for order in ("sweep", "front"):
        AB = to_band(A, b)
        d, e, VV, tau2 = stage2(AB, b, order)
        T = np.diag(d) + np.diag(e[:-1], 1) + np.diag(e[:-1], -1)
        w, Z = np.linalg.eigh(T)
        print(order, ": stage 2 eig err", np.max(np.abs(w - np.linalg.eigvalsh(A0))))
        backtransform2(Z, VV, tau2, b)
        backtransform1(Z, V, Ts, b)
        print(order, ": residual", np.max(np.abs(A0 @ Z - Z * w)), " orthogonality", np.max(np.abs(Z.T @ Z - np.eye(n))))


if __name__ == "__main__":
    main()
