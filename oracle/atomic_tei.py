"""ORACLE — TEST INFRASTRUCTURE ONLY.  NumPy / SciPy restatement of the radial two-electron TABLES of the reference's atomic
program, independent of everything under helfem_amd/ (it shares no code with helfem_amd/csrc/host/*.cpp; the special
functions come from scipy.special and from Gauss-Legendre quadrature, not from host/special.cpp):

  atomic::basis::TwoDBasis::compute_tei        src/atomic/TwoDBasis.cpp:666-739
  atomic::basis::TwoDBasis::compute_yukawa     src/atomic/TwoDBasis.cpp:741-778
  atomic::basis::TwoDBasis::compute_erfc       src/atomic/TwoDBasis.cpp:780-815
  RadialBasis::radial_integral                 libhelfem/src/RadialBasis.cpp:190-199
  RadialBasis::bessel_il_integral / kl         libhelfem/src/RadialBasis.cpp:201-209
  RadialBasis::twoe_integral / yukawa_integral libhelfem/src/RadialBasis.cpp:478-500
  RadialBasis::erfc_integral                   libhelfem/src/RadialBasis.cpp:502-558
  quadrature::twoe_inner_integral(_wrk)        libhelfem/src/quadrature.cpp:22-85
  quadrature::twoe_integral / yukawa_integral  libhelfem/src/quadrature.cpp:87-166
  quadrature::erfc_integral                    libhelfem/src/quadrature.cpp:168-222
  utils::bessel_il / bessel_kl                 libhelfem/src/utils.cpp:47-70 (i_L(x); (2/pi) sqrt(pi/2x) K_{L+1/2}(x))
  erfc_expn::Phi                               libhelfem/src/erfc_expn.cpp:181 -- NOT restated: Phi_L(X, x) is formed here from
                                               its definition, the Legendre projection of erfc(rho)/rho, rho^2 = X^2 + x^2 - 2 X x t

The quadrature rule, the Lobatto nodes, the Lagrange polynomials and utils::exchange_tei are those of oracle/diatomic_tei.py.
Only tests/ and the fixture generator (tests/golden/make_atomic_tei_golden.py) import this module.
"""
import math

import numpy as np
from scipy import special

from diatomic_tei import chebyshev, exchange_tei, lip_values, lobatto_nodes  # noqa: F401  (oracle/ is on sys.path)


def bessel_il(x, L):
    return special.spherical_in(L, x)


def bessel_kl(x, L):
    return special.spherical_kn(L, x) * (2.0 / math.pi)


_GL = np.polynomial.legendre.leggauss(96)


def erfc_phi(L, X, x):
    """Phi_L(X, x) with erfc(mu r12)/r12 = mu sum_L Phi_L(mu r, mu r') P_L(cos gamma): the Coulomb part in closed form,
    the smooth erf part by Gauss-Legendre quadrature over cos gamma"""
    X = np.asarray(X, dtype=float)[..., None]
    x = np.asarray(x, dtype=float)[..., None]
    t, w = _GL
    rho = np.sqrt(np.maximum(X * X + x * x - 2.0 * X * x * t, 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        f = np.where(rho > 1e-12, special.erf(rho) / rho, 2.0 / math.sqrt(math.pi))
    PL = special.eval_legendre(L, t)
    smooth = 0.5 * (2 * L + 1) * np.sum(w * PL * f, axis=-1)
    lo, hi = np.minimum(X[..., 0], x[..., 0]), np.maximum(X[..., 0], x[..., 0])
    return lo ** L / hi ** (L + 1) - smooth


class RadialSetup:
    """radial part of atomic::basis::TwoDBasis (point nucleus, primbas 4): the first element drops its first primitive
    (B(0) = 0), the last element its last (B(Rmax) = 0)"""

    def __init__(self, bval, nnodes, nquad):
        self.bval = np.asarray(bval, dtype=float)
        self.nel = len(self.bval) - 1
        self.x0 = lobatto_nodes(nnodes)
        self.xq, self.wq = chebyshev(nquad)

    def enabled(self, iel):
        n = len(self.x0)
        lo = 1 if iel == 0 else 0
        hi = n - 1 if iel == self.nel - 1 else n
        return np.arange(lo, hi)

    def bf(self, iel, x):
        return lip_values(self.x0, x)[:, self.enabled(iel)]

    def _r(self, iel, x):
        a, b = self.bval[iel], self.bval[iel + 1]
        return 0.5 * (a + b) + 0.5 * (b - a) * x, 0.5 * (b - a)

    def weighted(self, iel, f):
        """int B_i B_j f(r) dr over element iel"""
        r, rlen = self._r(iel, self.xq)
        bf = self.bf(iel, self.xq)
        return (bf * (self.wq * rlen * f(r))[:, None]).T @ bf

    def radial_integral(self, n, iel):
        return self.weighted(iel, lambda r: r ** float(n))

    def bessel_il_integral(self, L, lam, iel):
        return self.weighted(iel, lambda r: bessel_il(lam * r, L))

    def bessel_kl_integral(self, L, lam, iel):
        return self.weighted(iel, lambda r: bessel_kl(lam * r, L))

    def _inner(self, iel, fsmallbig, fbig):
        """twoe_inner_integral: inner[ip, k Nk + l] = int_rmin^r_ip B_k B_l(r') fsmallbig(r', r_ip) dr', every sub-interval
        with its own nquad points, the running sum rescaled by fbig(r_ip) / fbig(r_ip-1)"""
        a, b = self.bval[iel], self.bval[iel + 1]
        rmid0, rlen0 = 0.5 * (a + b), 0.5 * (b - a)
        r = rmid0 + rlen0 * self.xq
        nb = len(self.enabled(iel))
        inner = np.zeros((len(r), nb * nb))
        lo = a
        for ip in range(len(r)):
            hi = r[ip]
            mid, ln = 0.5 * (hi + lo), 0.5 * (hi - lo)
            rr = mid + ln * self.xq
            wp = self.wq * np.array([fsmallbig(ri, hi) for ri in rr]) * ln
            bf = self.bf(iel, (rr - rmid0) / rlen0)
            inner[ip] = ((bf * wp[:, None]).T @ bf).T.reshape(-1, order="F")  # vectorise(trans(wbf) * bf): column-major
            if ip:
                inner[ip] += inner[ip - 1] * (fbig(r[ip]) / fbig(r[ip - 1]))
            lo = hi
        return inner

    def _outer(self, iel, inner):
        _, rlen = self._r(iel, self.xq)
        bf = self.bf(iel, self.xq)
        nb = bf.shape[1]
        prod = np.zeros((len(self.xq), nb * nb))
        for fi in range(nb):
            for fj in range(nb):
                prod[:, fi * nb + fj] = bf[:, fi] * bf[:, fj] * self.wq * rlen
        ints = prod.T @ inner
        return ints + ints.T

    def twoe_integral(self, L, iel):
        return self._outer(iel, self._inner(iel, lambda r, R: (r / R) ** L / R, lambda r: r ** float(-L - 1)))

    def yukawa_integral(self, L, lam, iel):
        return self._outer(iel, self._inner(iel, lambda r, R: float(bessel_il(r * lam, L) * bessel_kl(R * lam, L)),
                                            lambda r: float(bessel_kl(r * lam, L))))

    def erfc_integral(self, L, mu, iel, kel):
        nq = len(self.xq)
        nint = nq if iel == kel else 1
        xi, wi = self.xq, self.wq
        ri, rleni = self._r(iel, xi)
        xk = np.zeros(nq * nint)
        wk = np.zeros(nq * nint)
        for ii in range(nint):
            s, e = ii * 2.0 / nint - 1.0, (ii + 1) * 2.0 / nint - 1.0
            xk[ii * nq:(ii + 1) * nq] = 0.5 * (e + s) + xi * 0.5 * (e - s)
            wk[ii * nq:(ii + 1) * nq] = wi * 0.5 * (e - s)
        rk, rlenk = self._r(kel, xk)
        Fn = erfc_phi(L, mu * ri[:, None] * np.ones((1, len(rk))), mu * np.ones((len(ri), 1)) * rk[None, :])
        bfi, bfk = self.bf(iel, xi), self.bf(kel, xk)
        ni, nk = bfi.shape[1], bfk.shape[1]
        pij = np.zeros((len(xi), ni * ni))
        for fi in range(ni):
            for fj in range(ni):
                pij[:, fi * ni + fj] = bfi[:, fi] * bfi[:, fj] * wi * rleni
        pkl = np.zeros((len(xk), nk * nk))
        for fi in range(nk):
            for fj in range(nk):
                pkl[:, fi * nk + fj] = bfk[:, fi] * bfk[:, fj] * wk * rlenk
        ints = pij.T @ Fn @ pkl
        if iel == kel:
            ints = 0.5 * (ints + ints.T)
        return ints


def compute_tables(setup, NL, lam, mu):
    """everything compute_tei(true), compute_yukawa(lam) and compute_erfc(mu) store, as dict name -> {(L, iel[, kel]): table}"""
    out = {k: {} for k in ("disjoint_L", "disjoint_m1L", "prim_tei", "prim_ktei", "disjoint_iL", "disjoint_kL", "yukawa_tei", "yukawa_ktei",
                           "erfc_tei", "erfc_ktei")}
    for L in range(NL):
        for iel in range(setup.nel):
            Ni = len(setup.enabled(iel))
            out["disjoint_L"][(L, iel)] = setup.radial_integral(L, iel)
            out["disjoint_m1L"][(L, iel)] = setup.radial_integral(-L - 1, iel)
            t = setup.twoe_integral(L, iel)
            out["prim_tei"][(L, iel)] = t
            out["prim_ktei"][(L, iel)] = exchange_tei(t, Ni, Ni, Ni, Ni)
            out["disjoint_iL"][(L, iel)] = setup.bessel_il_integral(L, lam, iel)
            out["disjoint_kL"][(L, iel)] = setup.bessel_kl_integral(L, lam, iel)
            y = setup.yukawa_integral(L, lam, iel)
            out["yukawa_tei"][(L, iel)] = y
            out["yukawa_ktei"][(L, iel)] = exchange_tei(y, Ni, Ni, Ni, Ni)
            for kel in range(setup.nel):
                Nk = len(setup.enabled(kel))
                e = setup.erfc_integral(L, mu, iel, kel)
                out["erfc_tei"][(L, iel, kel)] = e
                out["erfc_ktei"][(L, iel, kel)] = exchange_tei(e, Ni, Ni, Nk, Nk)
    return out
