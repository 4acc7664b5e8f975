"""ctypes access to the CPU oracle (oracle/liboracle.so) — the CHECKER, test infrastructure only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None
c_double_p = ctypes.POINTER(ctypes.c_double)
c_i64_p = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j8"])
        L = ctypes.CDLL(_PATH)
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_basis_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                       c_double_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_void_p)]
        L.orc_eval_fxc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                   c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double, ctypes.c_long,
                                   ctypes.c_long]
        L.orc_eval_fxc_shard.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double,
                                         ctypes.c_int, ctypes.c_int]
        L.orc_eval_fxc_pol_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                             c_double_p, ctypes.c_double, ctypes.c_long, ctypes.c_long]
        L.orc_exchange_blocks.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.orc_xc_polarized.argtypes = [ctypes.c_int, ctypes.c_int64, c_double_p, c_double_p, c_double_p, c_double_p,
                                       c_double_p, ctypes.c_double]
        for nm in ("orc_eval_fxc_pol", "orc_atomic_eval_fxc_pol"):
            getattr(L, nm).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                       c_double_p, ctypes.c_double]
        L.orc_xc_unpolarized.argtypes = [ctypes.c_int, ctypes.c_int64, c_double_p, c_double_p, c_double_p, c_double_p,
                                         c_double_p, ctypes.c_double]
        L.orc_scf_diatomic.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, c_double_p]
        L.orc_atomic_basis_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_void_p)]
        L.orc_atomic_eval_fxc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double]
        L.orc_atomic_onebody.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p]
        L.orc_atomic_radial_integral.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p]
        L.orc_atomic_prim_tei.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_i64_p]
        L.orc_scf_atomic.argtypes = [ctypes.c_int] * 7 + [ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                                          ctypes.c_char_p] + [ctypes.c_int] * 6 + [
                                                              ctypes.c_double, ctypes.c_int, c_double_p]
        for nm in ("orc_xc_polarized_mgga", "orc_xc_unpolarized_mgga"):
            getattr(L, nm).argtypes = [ctypes.c_int, ctypes.c_int64] + [c_double_p] * 7 + [ctypes.c_double]
        L.orc_model_potential.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_double, c_double_p]
        L.orc_atomic_compute_rs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        L.orc_atomic_rs_exchange.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
        for nm in ("orc_bessel_il", "orc_bessel_kl"):
            getattr(L, nm).argtypes = [ctypes.c_double, ctypes.c_int]
            getattr(L, nm).restype = ctypes.c_double
        L.orc_erfc_phi.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double]
        L.orc_erfc_phi.restype = ctypes.c_double
        L.orc_set_erfc_binomial_mode.argtypes = [ctypes.c_int]
        L.orc_set_erfc_binomial_mode.restype = None
        for name in ("orc_basis_destroy", "orc_basis_dims", "orc_compute_tei", "orc_coulomb", "orc_exchange",
                     "orc_grid_overlap", "orc_grid_kinetic"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())


def _p(a):
    return a.ctypes.data_as(c_double_p)


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _blocks(m_idx):
    ptr = np.zeros(len(m_idx) + 1, dtype=np.int64)
    for i, b in enumerate(m_idx):
        ptr[i + 1] = ptr[i] + len(b)
    idx = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.int64) for b in m_idx]))
    return ptr, idx


class OracleBasis(object):
    _fxc_pol = "orc_eval_fxc_pol"

    def __init__(self, Z1, Z2, Rhalf, nnodes, nquad, bval, lval, mval, lpad=10):
        bval = np.ascontiguousarray(bval, dtype=np.float64)
        lv = (ctypes.c_int * len(lval))(*lval)
        mv = (ctypes.c_int * len(mval))(*mval)
        h = ctypes.c_void_p()
        _check(lib().orc_basis_create(Z1, Z2, Rhalf, nnodes, nquad, _p(bval), len(bval), lv, mv, len(lval), lpad,
                                      ctypes.byref(h)))
        self.h = h
        dims = [ctypes.c_int64() for _ in range(5)]
        lib().orc_basis_dims(self.h, *[ctypes.byref(x) for x in dims])
        self.Nbf, self.Ndummy, self.Nrad, self.Nang, self.Nel = [x.value for x in dims]

    def __del__(self):
        try:
            lib().orc_basis_destroy(self.h)
        except Exception:
            pass

    def compute_tei(self, exchange=True):
        _check(lib().orc_compute_tei(self.h, 1 if exchange else 0))

    def coulomb(self, P):
        P = _f(P)
        J = np.zeros_like(P, order="F")
        _check(lib().orc_coulomb(self.h, _p(P), _p(J)))
        return J

    def coulomb_shard(self, P, rank, nranks):
        P = _f(P)
        J = np.zeros_like(P, order="F")
        _check(lib().orc_coulomb_shard(self.h, _p(P), _p(J), int(rank), int(nranks)))
        return J

    def eval_Fxc_shard(self, lang, mang, x_func, c_func, P, rank, nranks, thr=1e-12):
        P = _f(P)
        H = np.zeros_like(P, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib().orc_eval_fxc_shard(self.h, lang, mang, x_func, c_func, _p(P), _p(H), ctypes.byref(exc),
                                        ctypes.byref(nel), ctypes.byref(ekin), thr, int(rank), int(nranks)))
        return H, exc.value, nel.value, ekin.value

    def exchange(self, P):
        P = _f(P)
        K = np.zeros_like(P, order="F")
        _check(lib().orc_exchange(self.h, _p(P), _p(K)))
        return K

    def exchange_blocks(self, P, pairs):
        """only the output blocks (jang, kang) of `pairs` (the reference loops over them one by one, basis.cpp:1575-1579);
        the rest of the returned matrix is zero"""
        P = _f(P)
        K = np.zeros_like(P, order="F")
        js = (ctypes.c_int * len(pairs))(*[int(p[0]) for p in pairs])
        ks = (ctypes.c_int * len(pairs))(*[int(p[1]) for p in pairs])
        _check(lib().orc_exchange_blocks(self.h, _p(P), _p(K), len(pairs), js, ks))
        return K

    def eval_Fxc_points(self, lang, mang, x_func, c_func, P, points, thr=1e-12, Pb=None, threads=8):
        """the contributions of the listed radial points (indices into the E * nq list of dftgrid.cpp:779-801) summed;
        one point per call on `threads` host threads (ctypes releases the GIL).  Pb given: the unrestricted driver,
        returns (Ha, Hb, Exc, Nel, Ekin), else (H, Exc, Nel, Ekin)."""
        import threading
        work = list(points)
        lock = threading.Lock()
        res = []

        def run():
            while True:
                with lock:
                    if not work:
                        return
                    q = work.pop()
                if Pb is None:
                    r = self.eval_Fxc(lang, mang, x_func, c_func, P, thr, q_begin=q, q_end=q + 1)
                else:
                    r = self.eval_Fxc_pol(lang, mang, x_func, c_func, P, Pb, thr, q_begin=q, q_end=q + 1)
                with lock:
                    res.append((q, r))

        ths = [threading.Thread(target=run) for _ in range(max(1, min(threads, len(work))))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        res.sort(key=lambda x: x[0])  # fixed summation order
        nmat = 1 if Pb is None else 2
        out = [sum(r[1][k] for r in res) for k in range(nmat + 3)]
        return tuple(out)

    def eval_Fxc(self, lang, mang, x_func, c_func, P, thr=1e-12, q_begin=0, q_end=-1):
        P = _f(P)
        H = np.zeros_like(P, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib().orc_eval_fxc(self.h, lang, mang, x_func, c_func, _p(P), _p(H), ctypes.byref(exc),
                                  ctypes.byref(nel), ctypes.byref(ekin), thr, q_begin, q_end))
        return H, exc.value, nel.value, ekin.value

    def eval_Fxc_pol(self, lang, mang, x_func, c_func, Pa, Pb, thr=1e-12, q_begin=0, q_end=-1):
        Pa, Pb = _f(Pa), _f(Pb)
        Ha, Hb = np.zeros_like(Pa, order="F"), np.zeros_like(Pb, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib().orc_eval_fxc_pol_range(self.h, lang, mang, x_func, c_func, _p(Pa), _p(Pb), _p(Ha), _p(Hb),
                                            ctypes.byref(exc), ctypes.byref(nel), ctypes.byref(ekin), thr, q_begin, q_end))
        return Ha, Hb, exc.value, nel.value, ekin.value

    def grid_overlap(self, lang, mang):
        S = np.zeros((self.Nbf, self.Nbf), order="F")
        _check(lib().orc_grid_overlap(self.h, lang, mang, _p(S)))
        return S

    def grid_kinetic(self, lang, mang):
        T = np.zeros((self.Nbf, self.Nbf), order="F")
        _check(lib().orc_grid_kinetic(self.h, lang, mang, _p(T)))
        return T


class OracleAtomicBasis(object):
    _fxc_pol = "orc_atomic_eval_fxc_pol"

    def __init__(self, Z, nnodes, nquad, bval, lval, mval):
        bval = np.ascontiguousarray(bval, dtype=np.float64)
        lv = (ctypes.c_int * len(lval))(*lval)
        mv = (ctypes.c_int * len(mval))(*mval)
        h = ctypes.c_void_p()
        _check(lib().orc_atomic_basis_create(Z, nnodes, nquad, _p(bval), len(bval), lv, mv, len(lval), ctypes.byref(h)))
        self.h = h
        dims = [ctypes.c_int64() for _ in range(4)]
        lib().orc_atomic_basis_dims(self.h, *[ctypes.byref(x) for x in dims])
        self.Nbf, self.Nrad, self.Nang, self.Nel = [x.value for x in dims]

    def __del__(self):
        try:
            lib().orc_atomic_basis_destroy(self.h)
        except Exception:
            pass

    def onebody(self, which):
        M = np.zeros((self.Nbf, self.Nbf), order="F")
        _check(lib().orc_atomic_onebody(self.h, {"overlap": 0, "kinetic": 1, "nuclear": 2}[which], _p(M)))
        return M

    def radial_integral(self, n):
        M = np.zeros((self.Nrad, self.Nrad), order="F")
        _check(lib().orc_atomic_radial_integral(self.h, int(n), _p(M)))
        return M

    def prim_tei(self, L, iel, nmax=64):
        out = np.zeros(nmax ** 4)
        n = ctypes.c_int64()
        _check(lib().orc_atomic_prim_tei(self.h, int(L), int(iel), _p(out), ctypes.byref(n)))
        k = n.value
        return out[:k * k].reshape((k, k), order="F")

    def compute_tei(self, exchange=True):
        _check(lib().orc_atomic_compute_tei(self.h, 1 if exchange else 0))

    def coulomb(self, P):
        P = _f(P)
        J = np.zeros_like(P, order="F")
        _check(lib().orc_atomic_coulomb(self.h, _p(P), _p(J)))
        return J

    def exchange(self, P):
        P = _f(P)
        K = np.zeros_like(P, order="F")
        _check(lib().orc_atomic_exchange(self.h, _p(P), _p(K)))
        return K

    def compute_yukawa(self, lam):
        _check(lib().orc_atomic_compute_rs(self.h, 1, float(lam)))

    def compute_erfc(self, mu):
        _check(lib().orc_atomic_compute_rs(self.h, 2, float(mu)))

    def rs_exchange(self, P):
        P = _f(P)
        K = np.zeros_like(P, order="F")
        _check(lib().orc_atomic_rs_exchange(self.h, _p(P), _p(K)))
        return K

    def eval_Fxc(self, lang, mang, x_func, c_func, P, thr=1e-12):
        P = _f(P)
        H = np.zeros_like(P, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib().orc_atomic_eval_fxc(self.h, lang, mang, x_func, c_func, _p(P), _p(H), ctypes.byref(exc),
                                         ctypes.byref(nel), ctypes.byref(ekin), thr))
        return H, exc.value, nel.value, ekin.value

    def eval_Fxc_pol(self, lang, mang, x_func, c_func, Pa, Pb, thr=1e-12):
        Pa, Pb = _f(Pa), _f(Pb)
        Ha, Hb = np.zeros_like(Pa, order="F"), np.zeros_like(Pb, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(getattr(lib(), self._fxc_pol)(self.h, lang, mang, x_func, c_func, _p(Pa), _p(Pb), _p(Ha), _p(Hb),
                                             ctypes.byref(exc), ctypes.byref(nel), ctypes.byref(ekin), thr))
        return Ha, Hb, exc.value, nel.value, ekin.value


def model_potential(basis, p1, p2=None, lang=0, mang=0):
    """oracle counterpart of TwoDBasis.model_potential: basis is an OracleBasis (lang, mang = quadrature) or an
    OracleAtomicBasis; p = (kind, Z[, d[, H]])"""
    atomic = isinstance(basis, OracleAtomicBasis)
    a = tuple(p1) + (0.0, 0.0)
    b = tuple(p2 if p2 is not None else p1) + (0.0, 0.0)
    N = basis.Nbf
    H = np.zeros((N, N), order="F")
    _check(lib().orc_model_potential(basis.h, 1 if atomic else 0, int(lang), int(mang), int(a[0]), int(a[1]), float(a[2]),
                                     float(a[3]), int(b[0]), int(b[1]), float(b[2]), float(b[3]), _p(H)))
    return H


def scf_set_iguess(iguess):
    _check(lib().orc_scf_set_iguess(int(iguess)))


def bessel_il(x, L):
    return lib().orc_bessel_il(float(x), int(L))


def bessel_kl(x, L):
    return lib().orc_bessel_kl(float(x), int(L))


def erfc_phi(n, Xi, xi):
    return lib().orc_erfc_phi(int(n), float(Xi), float(xi))


def set_erfc_binomial_mode(mode):
    """test hook: 1 = the reference's binomial helper in the erfc short-range series (see host/special.h)"""
    lib().orc_set_erfc_binomial_mode(int(mode))


def xc_polarized_mgga(func_id, rho, sigma, tau, thr=1e-12):
    """rho (n,2), sigma (n,3), tau (n,2) -> exc (n), vrho (n,2), vsigma (n,3), vtau (n,2)"""
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    n = rho.shape[0]
    exc, vrho, vsigma, vtau = np.zeros(n), np.zeros((n, 2)), np.zeros((n, 3)), np.zeros((n, 2))
    _check(lib().orc_xc_polarized_mgga(func_id, n, _p(rho), _p(sigma), _p(tau), _p(exc), _p(vrho), _p(vsigma), _p(vtau), thr))
    return exc, vrho, vsigma, vtau


def xc_unpolarized_mgga(func_id, rho, sigma, tau, thr=1e-12):
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    n = rho.size
    exc, vrho, vsigma, vtau = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    _check(lib().orc_xc_unpolarized_mgga(func_id, n, _p(rho), _p(sigma), _p(tau), _p(exc), _p(vrho), _p(vsigma), _p(vtau), thr))
    return exc, vrho, vsigma, vtau


def xc_polarized(func_id, rho, sigma, thr=1e-12):
    """rho (n,2), sigma (n,3) -> exc (n), vrho (n,2), vsigma (n,3)"""
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    n = rho.shape[0]
    exc, vrho, vsigma = np.zeros(n), np.zeros((n, 2)), np.zeros((n, 3))
    _check(lib().orc_xc_polarized(func_id, n, _p(rho), _p(sigma), _p(exc), _p(vrho), _p(vsigma), thr))
    return exc, vrho, vsigma


def scf_set_occupations(occs=None, readocc=-1):
    L = lib()
    L.orc_scf_set_occupations.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    if occs is None:
        _check(L.orc_scf_set_occupations(0, 0, 0, None))
        return
    a = np.ascontiguousarray(np.asarray(occs, dtype=np.int32))
    _check(L.orc_scf_set_occupations(int(readocc), a.shape[0], a.shape[1], a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))


def scf_atomic(Z, lmax, mmax, nelem, nnodes, method, Q=0, nquad=0, Rmax=40.0, igrid=4, zexp=2.0, ldft=0, mdft=0,
               symmetry=1, maxit=50, convthr=1e-7, verbose=0, M=1, maverage=False, occs=None, readocc=-1):
    out = np.zeros(8)
    scf_set_occupations(occs, readocc)
    try:
        _check(lib().orc_scf_atomic(Z, Q, lmax, mmax, nelem, nnodes, nquad, Rmax, igrid, zexp, method.encode(), ldft, mdft,
                                    symmetry, M, 1 if maverage else 0, maxit, convthr, verbose, _p(out)))
    finally:
        scf_set_occupations(None)
    keys = ["Etot", "Ekin", "Epot", "Ecoul", "Exx", "Exc", "Enucr"]
    r = dict(zip(keys, out[:7]))
    r["iterations"] = int(out[7])
    r["converged"] = (out[7] - int(out[7])) > 0.25
    return r


def eig_sym(A):
    A = _f(A)
    n = A.shape[0]
    E = np.zeros(n)
    C = np.zeros((n, n), order="F")
    _check(lib().orc_eig_sym(ctypes.c_int64(n), _p(A), _p(E), _p(C)))
    return E, C


def eig_gsym(F, Sinvh):
    F, Sinvh = _f(F), _f(Sinvh)
    N, n = Sinvh.shape
    E = np.zeros(n)
    C = np.zeros((N, n), order="F")
    _check(lib().orc_eig_gsym(ctypes.c_int64(N), ctypes.c_int64(n), _p(F), _p(Sinvh), _p(E), _p(C)))
    return E, C


def eig_gsym_sub(F, Sinvh, m_idx):
    F, Sinvh = _f(F), _f(Sinvh)
    N = F.shape[0]
    ptr, idx = _blocks(m_idx)
    E = np.zeros(N)
    C = np.zeros((N, N), order="F")
    _check(lib().orc_eig_gsym_sub(ctypes.c_int64(N), _p(F), _p(Sinvh), len(m_idx), ptr.ctypes.data_as(c_i64_p),
                                  idx.ctypes.data_as(c_i64_p), _p(E), _p(C)))
    return E, C


def form_Sinvh(S, chol, m_idx):
    S = _f(S)
    N = S.shape[0]
    ptr, idx = _blocks(m_idx)
    X = np.zeros((N, N), order="F")
    _check(lib().orc_form_sinvh(ctypes.c_int64(N), _p(S), 1 if chol else 0, len(m_idx), ptr.ctypes.data_as(c_i64_p),
                                idx.ctypes.data_as(c_i64_p), _p(X)))
    return X


def form_density(C, nocc):
    C = _f(C)
    N, nc = C.shape
    P = np.zeros((N, N), order="F")
    _check(lib().orc_form_density(ctypes.c_int64(N), ctypes.c_int64(nc), _p(C), ctypes.c_int64(nocc), _p(P)))
    return P


def xc_unpolarized(func_id, rho, sigma, thr=1e-12):
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    n = rho.size
    exc, vrho, vsigma = np.zeros(n), np.zeros(n), np.zeros(n)
    _check(lib().orc_xc_unpolarized(func_id, n, _p(rho), _p(sigma), _p(exc), _p(vrho), _p(vsigma), thr))
    return exc, vrho, vsigma


def scf_diatomic(Z1, Z2, Rbond, lmmax, nelem, nnodes, method, nquad=0, Rmax=40.0, igrid=4, zexp=1.0, lpad=10, ldft=0,
                 mdft=0, symmetry=1, maxit=50, convthr=1e-7, verbose=0, M=1, occs=None, readocc=-1):
    out = np.zeros(8)
    lm = (ctypes.c_int * len(lmmax))(*lmmax)
    scf_set_occupations(occs, readocc)
    try:
        _check(lib().orc_scf_diatomic(Z1, Z2, Rbond, lm, len(lmmax), nelem, nnodes, nquad, Rmax, igrid, zexp, lpad,
                                      method.encode(), ldft, mdft, symmetry, M, maxit, convthr, verbose, _p(out)))
    finally:
        scf_set_occupations(None)
    keys = ["Etot", "Ekin", "Epot", "Ecoul", "Exx", "Exc", "Enucr"]
    r = dict(zip(keys, out[:7]))
    r["iterations"] = int(out[7])
    r["converged"] = (out[7] - int(out[7])) > 0.25
    return r
