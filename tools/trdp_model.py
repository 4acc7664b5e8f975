"""Readable model of the persistent, register-resident tridiagonalisation (hip/trdp.hip).

The matrix is distributed by ROWS over G workgroups (full symmetric storage, every workgroup keeps its rows in
registers for the whole factorisation).  Per Householder column there is ONE exchange between the workgroups:

  published by workgroup k  : y[R_k] = A^(j-1)[R_k, j+1:] x_j   (its rows of the product with the UNNORMALISED column x_j,
                              formed with a matrix that still lacks the rank-2 update of column j-1),
                              dot_k = sum_{r in R_k} x_j[r] y[r]
  published by the owner of row j+1 : z = A^(j)[j+1, j+1:]       (that row fully updated; the kernel publishes the same
                              vector as COLUMN j+1 of the symmetric tile, every workgroup the entries of its own rows,
                              raw and one exchange ahead -- who stores it does not change the algebra checked here)

From (y, z, dots) and what it already has (x_j, v_{j-1}, w_{j-1} and three local sums) every workgroup derives
redundantly, with the same arithmetic and the same summation orders, beta, tau, v_j, w_j, d[j+1] and the next column
x_{j+1}; it then forms its rows of the next product, publishes them, and applies the rank-2 update of column j to its
registers while the others' data travel.

This file checks the algebra against a textbook dsytd2 (numpy) and is the specification the kernel follows.
"""
import numpy as np


def dsytd2_lower(A):
    """LAPACK dsytd2, lower: returns d, e, tau, V (reflectors below the subdiagonal, v[0] = 1 implicit)"""
    A = A.copy()
    n = A.shape[0]
    d, e, tau = np.zeros(n), np.zeros(n), np.zeros(n)
    for j in range(n - 2):
        x = A[j + 1:, j].copy()
        alpha = x[0]
        xn2 = float(x[1:] @ x[1:])
        if xn2 == 0.0:
            t, beta, s = 0.0, alpha, 0.0
        else:
            nrm = np.sqrt(alpha * alpha + xn2)
            beta = -nrm if alpha >= 0 else nrm
            t = (beta - alpha) / beta
            s = 1.0 / (alpha - beta)
        v = x * s
        v[0] = 1.0
        p = t * (A[j + 1:, j + 1:] @ v)
        w = p - 0.5 * t * (p @ v) * v
        A[j + 1:, j + 1:] -= np.outer(v, w) + np.outer(w, v)
        d[j], e[j], tau[j] = A[j, j], beta, t
        A[j + 2:, j] = v[1:]
    d[n - 2], e[n - 2], d[n - 1] = A[n - 2, n - 2], A[n - 1, n - 2], A[n - 1, n - 1]
    return d, e, tau, np.tril(A, -2)


def persistent_model(A0, G=5):
    """the exchange-per-column scheme; R holds every workgroup's rows (here simply the full matrix, updated one column late)"""
    n = A0.shape[0]
    R = A0.copy()  # "registers": A^(j-1) when the product with x_j is formed
    d, e, tau = np.zeros(n), np.zeros(n), np.zeros(n)
    Vout = np.zeros((n, n))
    m = (n + G - 1) // G
    owner = lambda r: r // m
    # ---- prologue: column 0 is read from the input by everybody ----
    x = np.zeros(n)
    x[1:] = A0[1:, 0]
    vp, wp = np.zeros(n), np.zeros(n)  # v_{j-1}, w_{j-1}
    c1 = c2 = 0.0  # w_{j-1}^T x_j, v_{j-1}^T x_j over rows >= j+1
    sig2 = float(x[2:] @ x[2:])
    d[0] = A0[0, 0]
    y = np.zeros(n)
    y[1:] = R[1:, 1:] @ x[1:]
    z = np.zeros(n)
    z[1:] = R[1, 1:]
    dots = [float(sum(x[r] * y[r] for r in range(max(1, k * m), min(n, (k + 1) * m)))) for k in range(G)]
    for j in range(n - 2):
        j1 = j + 1
        # ---- after the exchange: every workgroup, redundantly ----
        q = y - vp * c1 - wp * c2  # A^(j) x_j on rows >= j+1
        xtq = sum(dots) - 2.0 * c1 * c2
        alpha = x[j1]
        if sig2 == 0.0:
            t, beta, s = 0.0, alpha, 0.0
        else:
            nrm = np.sqrt(alpha * alpha + sig2)
            beta = -nrm if alpha >= 0 else nrm
            t = (beta - alpha) / beta
            s = 1.0 / (alpha - beta)
        z0, q0 = z[j1], q[j1]
        vtp = t * s * s * (xtq - 2.0 * beta * q0 + beta * beta * z0)
        a = -0.5 * t * vtp
        v = np.zeros(n)
        v[j1 + 1:] = s * x[j1 + 1:]
        v[j1] = 1.0
        p = np.zeros(n)
        p[j1:] = t * s * (q[j1:] - beta * z[j1:])
        w = np.zeros(n)
        w[j1:] = p[j1:] + a * v[j1:]
        w0 = w[j1]
        xn = np.zeros(n)
        xn[j1 + 1:] = z[j1 + 1:] - v[j1 + 1:] * w0 - w[j1 + 1:]
        d[j1] = z0 - 2.0 * w0
        e[j], tau[j] = beta, t
        Vout[j1 + 1:, j] = v[j1 + 1:]
        if j == n - 3:
            # last column: e[n-2] is the one element of the next column, d[n-1] from the last row's registers
            e[n - 2] = xn[n - 1]
            R[j1 - 0:, j1 - 0:] -= np.outer(vp[j1:], wp[j1:]) + np.outer(wp[j1:], vp[j1:])  # pending update j-1
            d[n - 1] = R[n - 1, n - 1] - 2.0 * v[n - 1] * w[n - 1]
            break
        c1n = float(w[j1 + 1:] @ xn[j1 + 1:])
        c2n = float(v[j1 + 1:] @ xn[j1 + 1:])
        sig2n = float(xn[j1 + 2:] @ xn[j1 + 2:])
        # ---- registers: the pending update of column j-1 has been applied while the exchange of column j was in flight ----
        R[j1:, j1:] -= np.outer(vp[j1:], wp[j1:]) + np.outer(wp[j1:], vp[j1:])  # R = A^(j)
        # owner of row j+2 publishes it fully updated (update j on the fly)
        zn = np.zeros(n)
        zn[j1 + 1:] = R[j1 + 1, j1 + 1:] - v[j1 + 1] * w[j1 + 1:] - w[j1 + 1] * v[j1 + 1:]
        yn = np.zeros(n)
        yn[j1 + 1:] = R[j1 + 1:, j1 + 1:] @ xn[j1 + 1:]  # product with the matrix that lacks update j
        dots = [float(sum(xn[r] * yn[r] for r in range(max(j1 + 1, k * m), min(n, (k + 1) * m)))) for k in range(G)]
        x, vp, wp, c1, c2, sig2, y, z = xn, v, w, c1n, c2n, sig2n, yn, zn
    return d, e, tau, Vout


if __name__ == "__main__":
    rng = np.random.RandomState(1)
    for n in (3, 4, 5, 9, 40, 131):
        A = rng.standard_normal((n, n))
        A = A + A.T
        d0, e0, t0, V0 = dsytd2_lower(A)
        d1, e1, t1, V1 = persistent_model(A, G=min(5, n))
        err = max(np.max(np.abs(d0 - d1)), np.max(np.abs(e0 - e1)), np.max(np.abs(t0 - t1)), np.max(np.abs(V0 - V1)))
        T = np.diag(d1) + np.diag(e1[:-1], -1) + np.diag(e1[:-1], 1)
        ev = np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A)))
        print("n = %4d   max |dsytd2 - model| = %.2e   eigenvalues %.2e" % (n, err, ev))
        assert err < 1e-11 * n and ev < 1e-11 * n
