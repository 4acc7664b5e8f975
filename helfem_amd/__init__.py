"""helfem_amd — MI355X-native (gfx950) implementation of HelFEM's SCF hot path.

Thin ctypes binding over the C ABI of include/helfem_gpu.h (helfem_amd/lib/libhelfem_amd.so).  The
class and method names mirror the reference's C++ interface for this path so that tests and drivers
read like the reference's own code:

    reference (C++)                                   here
    ------------------------------------------------  ---------------------------------------------
    diatomic::basis::TwoDBasis(Z1,Z2,Rhalf,poly,      TwoDBasis(Z1,Z2,Rhalf,nnodes,nquad,bval,lval,mval,lpad)
        n_quad,bval,lval,mval,lpad)   basis.cpp:307
    basis.overlap()/kinetic()/nuclear()               same
    basis.compute_tei(exchange)        basis.cpp:1166  same
    basis.coulomb(P) / exchange(P)     basis.cpp:1359  same (numpy in / numpy out)
    dftgrid::DFTGrid(&basis,ldft,mdft).eval_Fxc(...)  DFTGrid(basis,ldft,mdft).eval_Fxc(x_func,c_func,P,thr)
    scf::eig_gsym / eig_gsym_sub / form_density       scf.eig_gsym / scf.eig_gsym_sub / scf.form_density

There is no CPU fallback: every compute call needs a gfx950 device and raises RuntimeError otherwise.
Matrices are numpy float64 arrays in Fortran (column-major) order, i.e. arma::mat memory.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("HELFEM_AMD_LIB") or os.path.join(_HERE, "lib", "libhelfem_amd.so")  # HELFEM_AMD_LIB: A/B builds (tools/ab_build.py)
_lib = None

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_i64_p = ctypes.POINTER(ctypes.c_int64)


class hfg_diatomic_desc(ctypes.Structure):
    _fields_ = [("Z1", ctypes.c_int), ("Z2", ctypes.c_int), ("Rhalf", ctypes.c_double), ("primbas", ctypes.c_int),
                ("nnodes", ctypes.c_int), ("nquad", ctypes.c_int), ("bval", c_double_p), ("nbval", ctypes.c_int),
                ("lval", c_int_p), ("mval", c_int_p), ("nang", ctypes.c_int), ("lpad", ctypes.c_int)]


class hfg_model_pot(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("Z", ctypes.c_int), ("d", ctypes.c_double), ("H", ctypes.c_double)]


class hfg_atomic_desc(ctypes.Structure):
    _fields_ = [("Z", ctypes.c_int), ("primbas", ctypes.c_int), ("nnodes", ctypes.c_int), ("nquad", ctypes.c_int),
                ("bval", c_double_p), ("nbval", ctypes.c_int), ("lval", c_int_p), ("mval", c_int_p),
                ("nang", ctypes.c_int)]


def lib():
    """Load the native library (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("%s is missing: run `python -m helfem_amd.build` (or __graft_entry__.build()) first; "
                               "there is no Python/CPU fallback for the hot path" % _LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime.  If this library
        # (linked against /opt/rocm) is loaded first and torch afterwards, torch reports "No HIP GPUs are
        # available"; loading torch first makes both resolve to the same runtime (same SONAME).  torch is only
        # plumbing here (HBM buffers, streams, torch.distributed), so import it first when it is installed.
        if os.environ.get("HELFEM_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(_LIB_PATH)
        L.hfg_last_error.restype = ctypes.c_char_p
        L.hfg_version.restype = ctypes.c_char_p
        L.hfg_gaunt_coefficient.restype = ctypes.c_double
        L.hfg_gaunt_coefficient.argtypes = [ctypes.c_int] * 6
        L.hfg_modified_gaunt_coefficient.restype = ctypes.c_double
        L.hfg_modified_gaunt_coefficient.argtypes = [ctypes.c_int] * 6
        L.hfg_theta_lm.restype = ctypes.c_double
        L.hfg_theta_lm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.hfg_legendre_PQ.restype = None
        L.hfg_legendre_PQ.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, c_double_p]
        for nm in ("hfg_bessel_il", "hfg_bessel_kl"):
            getattr(L, nm).restype = ctypes.c_double
            getattr(L, nm).argtypes = [ctypes.c_double, ctypes.c_int]
        L.hfg_erfc_phi.restype = ctypes.c_double
        L.hfg_erfc_phi.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double]
        L.hfg_compute_rs_tei.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        L.hfg_model_potential.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(hfg_model_pot),
                                          ctypes.POINTER(hfg_model_pot), c_double_p]
        L.hfg_chebyshev_rule.restype = None
        L.hfg_chebyshev_rule.argtypes = [ctypes.c_int, c_double_p, c_double_p]
        L.hfg_lobatto_nodes.restype = None
        L.hfg_lobatto_nodes.argtypes = [ctypes.c_int, c_double_p]
        L.hfg_radial_grid.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p]
        L.hfg_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]
        L.hfg_xc_fock.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                  c_double_p, c_double_p, c_double_p, ctypes.c_double]
        L.hfg_xc_fock_pol.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double]
        L.hfg_xc_fock_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
        L.hfg_profile_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, c_double_p, c_i64_p]
        L.hfg_measure_kernel.argtypes = [ctypes.c_void_p, ctypes.c_char_p, c_double_p, c_i64_p]
        L.hfg_profile_names.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().hfg_last_error().decode())


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(c_double_p)


def device_count():
    return int(lib().hfg_device_count())


class Context(object):
    """hfg_ctx: one device + one stream."""

    def __init__(self, device=0, stream=None):
        """stream: None -> the context creates its own stream; an integer hipStream_t handle -> enqueue there, where
        the handle 0 (what torch.cuda.current_stream().cuda_stream returns for the default stream) means the null
        stream itself (HFG_NULL_STREAM), so that the kernels stay ordered with torch operations and collectives"""
        h = ctypes.c_void_p()
        if stream is None:
            sp = None
        elif int(stream) == 0:
            sp = ctypes.c_void_p(-1)  # HFG_NULL_STREAM
        else:
            sp = ctypes.c_void_p(int(stream))
        _check(lib().hfg_ctx_create(ctypes.byref(h), int(device), sp))
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            lib().hfg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().hfg_ctx_synchronize(self.h))

    def set_shard(self, rank, nranks):
        _check(lib().hfg_ctx_set_shard(self.h, int(rank), int(nranks)))

    def fix_sinvh(self, device_ptr):
        """declare the device matrix at this address constant (None withdraws): see hfg_ctx_fix_sinvh"""
        _check(lib().hfg_ctx_fix_sinvh(self.h, ctypes.c_void_p(device_ptr or 0)))

    def profile(self, on=True):
        _check(lib().hfg_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        _check(lib().hfg_profile_reset(self.h))

    def measure_kernel(self, name):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        _check(lib().hfg_measure_kernel(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def profile_names(self):
        buf = ctypes.create_string_buffer(1 << 16)
        _check(lib().hfg_profile_names(self.h, buf, len(buf)))
        return [x for x in buf.value.decode().split("\n") if x]

    def profile_get(self, name):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        _check(lib().hfg_profile_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def lm_to_l_m(lmmax):
    """diatomic::basis::lm_to_l_m (basis.cpp:287)."""
    lmmax = list(lmmax)
    cap = sum(2 * (l + 1) for l in lmmax) + 4
    lv = (ctypes.c_int * cap)()
    mv = (ctypes.c_int * cap)()
    n = ctypes.c_int(cap)
    arr = (ctypes.c_int * len(lmmax))(*lmmax)
    _check(lib().hfg_lm_list(arr, len(lmmax), lv, mv, ctypes.byref(n)))
    return list(lv[:n.value]), list(mv[:n.value])


def get_grid(mumax, nelem, igrid=4, zexp=1.0):
    """utils::get_grid (libhelfem/src/grid.cpp:18)."""
    b = np.zeros(nelem + 1)
    _check(lib().hfg_radial_grid(float(mumax), int(nelem), int(igrid), float(zexp), _p(b)))
    return b


class TwoDBasis(object):
    """diatomic::basis::TwoDBasis — setup on the host, coulomb/exchange on the GPU."""

    def __init__(self, Z1, Z2, Rhalf, nnodes, nquad, bval, lval, mval, lpad=10, ctx=None):
        self._bval = np.ascontiguousarray(bval, dtype=np.float64)
        self._lval = (ctypes.c_int * len(lval))(*lval)
        self._mval = (ctypes.c_int * len(mval))(*mval)
        d = hfg_diatomic_desc(int(Z1), int(Z2), float(Rhalf), 4, int(nnodes), int(nquad), _p(self._bval),
                              len(self._bval), self._lval, self._mval, len(lval), int(lpad))
        h = ctypes.c_void_p()
        _check(lib().hfg_diatomic_basis_create(ctypes.byref(d), ctypes.byref(h)))
        self.h = h
        self.ctx = ctx
        self.lval, self.mval = list(lval), list(mval)
        self._uploaded = None
        dims = [ctypes.c_int64() for _ in range(5)]
        _check(lib().hfg_basis_dims(self.h, *[ctypes.byref(x) for x in dims]))
        self._Nbf, self._Ndummy, self._Nrad, self._Nang, self._Nel = [x.value for x in dims]

    def __del__(self):
        try:
            if self.h:
                lib().hfg_basis_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def Nbf(self):
        return self._Nbf

    def Ndummy(self):
        return self._Ndummy

    def Nrad(self):
        return self._Nrad

    def Nang(self):
        return self._Nang

    def Nel(self):
        return self._Nel

    def _mat(self, fn):
        M = np.zeros((self._Nbf, self._Nbf), order="F")
        _check(fn(self.h, _p(M)))
        return M

    def overlap(self):
        return self._mat(lib().hfg_basis_overlap)

    def kinetic(self):
        return self._mat(lib().hfg_basis_kinetic)

    def nuclear(self):
        return self._mat(lib().hfg_basis_nuclear)

    def get_sym_idx(self, symm):
        n = ctypes.c_int()
        _check(lib().hfg_basis_sym_blocks(self.h, int(symm), ctypes.byref(n), None, None))
        ptr = np.zeros(n.value + 1, dtype=np.int64)
        idx = np.zeros(self._Nbf, dtype=np.int64)
        _check(lib().hfg_basis_sym_blocks(self.h, int(symm), ctypes.byref(n), ptr.ctypes.data_as(c_i64_p),
                                          idx.ctypes.data_as(c_i64_p)))
        return [idx[ptr[i]:ptr[i + 1]].copy() for i in range(n.value)]

    def compute_tei(self, exchange=True, device=False, ctx=None):
        """TwoDBasis::compute_tei.  device=True builds the in-element tables on the GPU (hfg_compute_tei_dev): the
        1 GB of primitive integrals at Nbf~4000 then never exists on the host"""
        if device:
            ctx = ctx or self.ctx or default_context()
            self.ctx = ctx
            _check(lib().hfg_compute_tei_dev(ctx.h, self.h, 1 if exchange else 0))
        else:
            _check(lib().hfg_compute_tei(self.h, 1 if exchange else 0))
        self._uploaded = None

    def overlap_with(self, other):
        """TwoDBasis::overlap(const TwoDBasis &rh), basis.cpp:713: <self | other>"""
        out = np.zeros((self.Nbf(), other.Nbf()), order="F")
        f = lib().hfg_basis_interbasis_overlap
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_double_p]
        _check(f(self.h, other.h, _p(out)))
        return out

    def lm_map(self):
        """the sorted (L, |M|) channels of the constructor (basis.cpp:333-375)"""
        n = ctypes.c_int(4096)
        L = (ctypes.c_int * 4096)()
        M = (ctypes.c_int * 4096)()
        _check(lib().hfg_basis_lm_map(self.h, L, M, ctypes.byref(n)))
        return [(L[i], M[i]) for i in range(n.value)]

    PRIM_TABLES = ("tei00", "tei02", "tei20", "tei22", "ktei00", "ktei02", "ktei20", "ktei22", "P0", "P2", "Q0", "Q2")

    def prim_table(self, name, ilm, iel):
        """one table of compute_tei (prim_tei**, prim_ktei**, disjoint_**) in the reference's shape; device-built
        tables are copied back from HBM"""
        which = self.PRIM_TABLES.index(name)
        r, c = ctypes.c_int64(), ctypes.c_int64()
        ctxh = self.ctx.h if self.ctx is not None else None
        f = lib().hfg_basis_get_prim
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                      ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _check(f(ctxh, self.h, which, int(ilm), int(iel), None, ctypes.byref(r), ctypes.byref(c)))
        out = np.zeros((r.value, c.value), order="F")
        _check(f(ctxh, self.h, which, int(ilm), int(iel), _p(out), ctypes.byref(r), ctypes.byref(c)))
        return out

    def upload(self, ldft=0, mdft=0, ctx=None):
        """tables -> HBM (also sets up the XC grid of DFTGrid(basis, ldft, mdft))"""
        ctx = ctx or self.ctx or default_context()
        self.ctx = ctx
        _check(lib().hfg_basis_upload(ctx.h, self.h, int(ldft), int(mdft)))
        self._uploaded = (ldft, mdft)

    def _ensure(self):
        if self._uploaded is None:
            self.upload()
        return self.ctx

    def coulomb(self, P):
        ctx = self._ensure()
        P = _f(P)
        J = np.zeros_like(P, order="F")
        _check(lib().hfg_coulomb(ctx.h, self.h, _p(P), _p(J)))
        return J

    def exchange(self, P):
        ctx = self._ensure()
        P = _f(P)
        K = np.zeros_like(P, order="F")
        _check(lib().hfg_exchange(ctx.h, self.h, _p(P), _p(K)))
        return K

    def model_potential(self, p1, p2=None):
        """TwoDGrid::model_potential(p1, p2) / atomic TwoDBasis::model_potential(p1): p = (kind, Z[, d[, H]]) with kind
        0 point nucleus, 1 GSZ (screening length d), 3 Thomas-Fermi.  The diatomic form needs upload(ldft, mdft)."""
        ctx = self._ensure()

        def mk(p):
            p = tuple(p) + (0.0, 0.0)
            return hfg_model_pot(int(p[0]), int(p[1]), float(p[2]), float(p[3]))
        a, b = mk(p1), mk(p2 if p2 is not None else p1)
        N = self.Nbf()
        H = np.zeros((N, N), order="F")
        _check(lib().hfg_model_potential(ctx.h, self.h, ctypes.byref(a), ctypes.byref(b), _p(H)))
        return H


def angular_basis(lmax, mmax):
    """atomic::basis::angular_basis (src/atomic/basis.cpp:174)."""
    cap = (lmax + 1) * (2 * mmax + 1) + 4
    lv = (ctypes.c_int * cap)()
    mv = (ctypes.c_int * cap)()
    n = ctypes.c_int(cap)
    _check(lib().hfg_angular_basis(int(lmax), int(mmax), lv, mv, ctypes.byref(n)))
    return list(lv[:n.value]), list(mv[:n.value])


class AtomicTwoDBasis(TwoDBasis):
    """atomic::basis::TwoDBasis (point nucleus) — setup on the host, coulomb/exchange on the GPU."""

    def __init__(self, Z, nnodes, nquad, bval, lval, mval, ctx=None):
        self._bval = np.ascontiguousarray(bval, dtype=np.float64)
        self._lval = (ctypes.c_int * len(lval))(*lval)
        self._mval = (ctypes.c_int * len(mval))(*mval)
        d = hfg_atomic_desc(int(Z), 4, int(nnodes), int(nquad), _p(self._bval), len(self._bval), self._lval,
                            self._mval, len(lval))
        h = ctypes.c_void_p()
        _check(lib().hfg_atomic_basis_create(ctypes.byref(d), ctypes.byref(h)))
        self.h = h
        self.ctx = ctx
        self.lval, self.mval = list(lval), list(mval)
        self._uploaded = None
        dims = [ctypes.c_int64() for _ in range(5)]
        _check(lib().hfg_basis_dims(self.h, *[ctypes.byref(x) for x in dims]))
        self._Nbf, self._Ndummy, self._Nrad, self._Nang, self._Nel = [x.value for x in dims]

    ATOMIC_TABLES = {"prim_tei": 0, "prim_ktei": 4, "disjoint_L": 8, "disjoint_m1L": 10, "disjoint_iL": 12, "disjoint_kL": 13,
                     "rs_tei": 14, "rs_ktei": 15}

    def atomic_table(self, name, L, iel, kel=None):
        """one table of compute_tei / compute_yukawa / compute_erfc in the reference's shape (erfc: element pair iel, kel)"""
        which = self.ATOMIC_TABLES[name]
        idx = int(iel) if kel is None else int(iel) * self._Nel + int(kel)
        r, c = ctypes.c_int64(), ctypes.c_int64()
        ctxh = self.ctx.h if self.ctx is not None else None
        f = lib().hfg_basis_get_prim
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                      ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _check(f(ctxh, self.h, which, int(L), idx, None, ctypes.byref(r), ctypes.byref(c)))
        out = np.zeros((r.value, c.value), order="F")
        _check(f(ctxh, self.h, which, int(L), idx, _p(out), ctypes.byref(r), ctypes.byref(c)))
        return out

    def compute_yukawa(self, lam):
        """TwoDBasis::compute_yukawa (src/atomic/TwoDBasis.cpp:741): tables of exp(-lambda r12)/r12 (host)"""
        _check(lib().hfg_compute_rs_tei(self.h, 1, float(lam)))
        self._uploaded = None

    def compute_erfc(self, mu):
        """TwoDBasis::compute_erfc (src/atomic/TwoDBasis.cpp:780): tables of erfc(mu r12)/r12 (host)"""
        _check(lib().hfg_compute_rs_tei(self.h, 2, float(mu)))
        self._uploaded = None

    def rs_exchange(self, P):
        """TwoDBasis::rs_exchange (src/atomic/TwoDBasis.cpp:1142) on the GPU"""
        ctx = self._ensure()
        P = _f(P)
        K = np.zeros_like(P, order="F")
        _check(lib().hfg_rs_exchange(ctx.h, self.h, _p(P), _p(K)))
        return K


class DFTGrid(object):
    """diatomic::dftgrid::DFTGrid (dftgrid.cpp:760)."""

    def __init__(self, basis, ldft, mdft):
        self.basis, self.ldft, self.mdft = basis, int(ldft), int(mdft)

    def eval_Fxc(self, x_func, c_func, P, thr=1e-12):
        """returns (H, Exc, Nel, Ekin) — DFTGrid::eval_Fxc, restricted (dftgrid.cpp:769)."""
        b = self.basis
        if b._uploaded != (self.ldft, self.mdft):
            b.upload(self.ldft, self.mdft)
        P = _f(P)
        H = np.zeros_like(P, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib().hfg_xc_fock(b.ctx.h, b.h, int(x_func), int(c_func), _p(P), _p(H), ctypes.byref(exc),
                                 ctypes.byref(nel), ctypes.byref(ekin), float(thr)))
        return H, exc.value, nel.value, ekin.value

    def eval_Fxc_pol(self, x_func, c_func, Pa, Pb, thr=1e-12):
        """returns (Ha, Hb, Exc, Nel, Ekin) — DFTGrid::eval_Fxc, unrestricted (dftgrid.cpp:812)."""
        b = self.basis
        if b._uploaded != (self.ldft, self.mdft):
            b.upload(self.ldft, self.mdft)
        Pa, Pb = _f(Pa), _f(Pb)
        Ha, Hb = np.zeros_like(Pa, order="F"), np.zeros_like(Pb, order="F")
        exc, nel, ekin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib().hfg_xc_fock_pol(b.ctx.h, b.h, int(x_func), int(c_func), _p(Pa), _p(Pb), _p(Ha), _p(Hb),
                                     ctypes.byref(exc), ctypes.byref(nel), ctypes.byref(ekin), float(thr)))
        return Ha, Hb, exc.value, nel.value, ekin.value


    def eval_Fxc_dev(self, x_func, c_func, P, Pb=None, thr=1e-12):
        """the device-buffer entry points hfg_xc_fock_dev / hfg_xc_fock_pol_dev, which honour the context's shard
        (Context.set_shard: only the radial points Q % n == rank contribute, dftgrid.cpp:779): the partial matrices
        and sums one rank of a multi-GPU run produces.  torch tensors serve as the HBM buffers.
        Returns (H, Exc, Nel, Ekin) or, with Pb, (Ha, Hb, Exc, Nel, Ekin)."""
        import torch
        b = self.basis
        if b._uploaded != (self.ldft, self.mdft):
            b.upload(self.ldft, self.mdft)
        dev = torch.device("cuda", b.ctx.device)
        N = b.Nbf()

        def up(M):
            return torch.from_numpy(np.asfortranarray(M, dtype=np.float64).ravel(order="F").copy()).to(dev)

        def ptr(t):
            return ctypes.c_void_p(t.data_ptr())
        scal = torch.zeros(3, dtype=torch.float64, device=dev)
        dP = up(P)
        dH = torch.zeros(N * N, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        L = lib()
        if Pb is None:
            L.hfg_xc_fock_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
            _check(L.hfg_xc_fock_dev(b.ctx.h, b.h, int(x_func), int(c_func), ptr(dP), ptr(dH), ptr(scal), float(thr)))
            b.ctx.synchronize()
            sc = scal.cpu().numpy()
            return dH.cpu().numpy().reshape((N, N), order="F"), sc[0], sc[1], sc[2]
        dPb = up(Pb)
        dHb = torch.zeros(N * N, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        L.hfg_xc_fock_pol_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_double]
        _check(L.hfg_xc_fock_pol_dev(b.ctx.h, b.h, int(x_func), int(c_func), ptr(dP), ptr(dPb), ptr(dH), ptr(dHb), ptr(scal),
                                     float(thr)))
        b.ctx.synchronize()
        sc = scal.cpu().numpy()
        return (dH.cpu().numpy().reshape((N, N), order="F"), dHb.cpu().numpy().reshape((N, N), order="F"), sc[0], sc[1],
                sc[2])


class scf(object):
    """namespace helfem::scf (src/general/scf_helpers.h)."""

    @staticmethod
    def _blocks(m_idx):
        ptr = np.zeros(len(m_idx) + 1, dtype=np.int64)
        for i, b in enumerate(m_idx):
            ptr[i + 1] = ptr[i] + len(b)
        idx = np.concatenate([np.asarray(b, dtype=np.int64) for b in m_idx]) if len(m_idx) else np.zeros(0, np.int64)
        return ptr, np.ascontiguousarray(idx)

    @staticmethod
    def eig_gsym(F, Sinvh, ctx=None):
        ctx = ctx or default_context()
        F, Sinvh = _f(F), _f(Sinvh)
        N, n = Sinvh.shape
        E = np.zeros(n)
        C = np.zeros((N, n), order="F")
        _check(lib().hfg_eig_gsym(ctx.h, ctypes.c_int64(N), ctypes.c_int64(n), _p(F), _p(Sinvh), _p(E), _p(C)))
        return E, C

    @staticmethod
    def eig_gsym_sub(F, Sinvh, m_idx, ctx=None):
        ctx = ctx or default_context()
        F, Sinvh = _f(F), _f(Sinvh)
        N = F.shape[0]
        ptr, idx = scf._blocks(m_idx)
        E = np.zeros(N)
        C = np.zeros((N, N), order="F")
        _check(lib().hfg_eig_gsym_sub(ctx.h, ctypes.c_int64(N), _p(F), _p(Sinvh), len(m_idx),
                                      ptr.ctypes.data_as(c_i64_p), idx.ctypes.data_as(c_i64_p), _p(E), _p(C)))
        return E, C

    @staticmethod
    def eig_gsym_sub_pair(Fa, Fb, Sinvh, m_idx, ctx=None):
        """the alpha and beta eig_gsym_sub calls of an unrestricted iteration as one batch (hfg_eig_gsym_sub_pair)"""
        ctx = ctx or default_context()
        Fa, Fb, Sinvh = _f(Fa), _f(Fb), _f(Sinvh)
        N = Fa.shape[0]
        ptr, idx = scf._blocks(m_idx)
        Ea, Eb = np.zeros(N), np.zeros(N)
        Ca, Cb = np.zeros((N, N), order="F"), np.zeros((N, N), order="F")
        f = lib().hfg_eig_gsym_sub_pair
        f.argtypes = [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p, ctypes.c_int, c_i64_p, c_i64_p, c_double_p,
                      c_double_p, c_double_p, c_double_p]
        _check(f(ctx.h, N, _p(Fa), _p(Fb), _p(Sinvh), len(m_idx), ptr.ctypes.data_as(c_i64_p), idx.ctypes.data_as(c_i64_p), _p(Ea), _p(Ca),
                 _p(Eb), _p(Cb)))
        return Ea, Ca, Eb, Cb

    @staticmethod
    def eig_sym(A, ctx=None):
        ctx = ctx or default_context()
        A = _f(A)
        n = A.shape[0]
        E = np.zeros(n)
        C = np.zeros((n, n), order="F")
        _check(lib().hfg_eig_sym(ctx.h, ctypes.c_int64(n), _p(A), _p(E), _p(C)))
        return E, C

    @staticmethod
    def form_Sinvh(S, chol, m_idx, ctx=None):
        ctx = ctx or default_context()
        S = _f(S)
        N = S.shape[0]
        ptr, idx = scf._blocks(m_idx)
        X = np.zeros((N, N), order="F")
        _check(lib().hfg_form_sinvh(ctx.h, ctypes.c_int64(N), _p(S), 1 if chol else 0, len(m_idx),
                                    ptr.ctypes.data_as(c_i64_p), idx.ctypes.data_as(c_i64_p), _p(X)))
        return X

    @staticmethod
    def form_density(C, nocc, ctx=None):
        ctx = ctx or default_context()
        C = _f(C)
        N, nc = C.shape
        P = np.zeros((N, N), order="F")
        _check(lib().hfg_form_density(ctx.h, ctypes.c_int64(N), ctypes.c_int64(nc), _p(C), ctypes.c_int64(nocc), _p(P)))
        return P

    @staticmethod
    def gemm(A, B, transA=False, transB=False, ctx=None):
        ctx = ctx or default_context()
        A, B = _f(A), _f(B)
        m = A.shape[1] if transA else A.shape[0]
        k = A.shape[0] if transA else A.shape[1]
        n = B.shape[0] if transB else B.shape[1]
        C = np.zeros((m, n), order="F")
        _check(lib().hfg_gemm(ctx.h, int(transA), int(transB), ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(k),
                              _p(A), ctypes.c_int64(A.shape[0]), _p(B), ctypes.c_int64(B.shape[0]), _p(C),
                              ctypes.c_int64(m)))
        return C


# ---- host-side special functions (pinned against the reference's known-answer tests) ----
def gaunt_coefficient(L, M, l, m, lp, mp):
    return lib().hfg_gaunt_coefficient(L, M, l, m, lp, mp)


def modified_gaunt_coefficient(lj, mj, L, M, li, mi):
    return lib().hfg_modified_gaunt_coefficient(lj, mj, L, M, li, mi)


def legendre_PQ(Lmax, Mmax, xi):
    """(P, Q) arrays [L, M] — layout of the Fortran wrapper calc_Plm_arr / calc_Qlm_arr."""
    P = np.zeros((Mmax + 1, Lmax + 1))
    Q = np.zeros((Mmax + 1, Lmax + 1))
    lib().hfg_legendre_PQ(int(Lmax), int(Mmax), float(xi), _p(P), _p(Q))
    return P.T.copy(), Q.T.copy()


def bessel_il(x, L):
    return lib().hfg_bessel_il(float(x), int(L))


def bessel_kl(x, L):
    return lib().hfg_bessel_kl(float(x), int(L))


def erfc_phi(n, Xi, xi):
    return lib().hfg_erfc_phi(int(n), float(Xi), float(xi))


def theta_lm(l, m, cth):
    return lib().hfg_theta_lm(int(l), int(m), float(cth))


def chebyshev(n):
    x, w = np.zeros(n), np.zeros(n)
    lib().hfg_chebyshev_rule(int(n), _p(x), _p(w))
    return x, w


def lobatto_nodes(n):
    x = np.zeros(n)
    lib().hfg_lobatto_nodes(int(n), _p(x))
    return x


def scf_set_iguess(iguess):
    """--iguess of the drivers for the following scf_* calls of this thread: 0 core Hamiltonian, 3 Thomas-Fermi"""
    _check(lib().hfg_scf_set_iguess(int(iguess)))


def scf_diatomic(Z1, Z2, Rbond, lmmax, nelem, nnodes, method, nquad=0, Rmax=40.0, igrid=4, zexp=1.0, lpad=10, ldft=0,
                 mdft=0, symmetry=1, maxit=50, convthr=1e-7, verbose=0, ctx=None, M=1, occs=None, readocc=-1):
    """Restricted closed-shell (M=1) or unrestricted (M=2S+1>1) diatomic SCF with every per-iteration step on the GPU
    (the loop of src/diatomic/main.cpp:780-995; flags as in main.cpp:89-133)."""
    ctx = ctx or default_context()
    L = lib()
    L.hfg_scf_diatomic.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_int_p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                   ctypes.c_double, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, c_double_p]
    out = np.zeros(12)
    lm = (ctypes.c_int * len(lmmax))(*lmmax)
    scf_set_occupations(occs, readocc)
    try:
        _check(L.hfg_scf_diatomic(ctx.h, Z1, Z2, Rbond, lm, len(lmmax), nelem, nnodes, nquad, Rmax, igrid, zexp, lpad,
                                  method.encode(), ldft, mdft, symmetry, M, maxit, convthr, verbose, _p(out)))
    finally:
        scf_set_occupations(None)
    keys = ["Etot", "Ekin", "Epot", "Ecoul", "Exx", "Exc", "Enucr"]
    r = dict(zip(keys, out[:7]))
    r["iterations"] = int(out[7])
    r["converged"] = (out[7] - int(out[7])) > 0.25
    r["tJ"], r["tK"], r["tXC"], r["tdiag"] = out[8:12]
    return r


def scf_set_occupations(occs=None, readocc=-1):
    """--readocc for the following scf_diatomic / scf_atomic calls of this thread: occs = rows of occs.dat (nalpha, nbeta, m
    [, parity | l, m]); None switches it off"""
    L = lib()
    L.hfg_scf_set_occupations.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    if occs is None:
        _check(L.hfg_scf_set_occupations(0, 0, 0, None))
        return
    a = np.ascontiguousarray(np.asarray(occs, dtype=np.int32))
    _check(L.hfg_scf_set_occupations(int(readocc), a.shape[0], a.shape[1], a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))


def scf_atomic(Z, lmax, mmax, nelem, nnodes, method, Q=0, nquad=0, Rmax=40.0, igrid=4, zexp=2.0, ldft=0, mdft=0,
               symmetry=1, maxit=50, convthr=1e-7, verbose=0, ctx=None, M=1, maverage=False, occs=None, readocc=-1):
    """Restricted closed-shell (M=1), unrestricted (M=2S+1>1) or restricted open-shell (M<0) atomic SCF with every per-iteration step on the GPU
    (the loop of src/atomic/main.cpp:760-1005; flags as in main.cpp:66-100)."""
    ctx = ctx or default_context()
    L = lib()
    L.hfg_scf_atomic.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_double, ctypes.c_int,
                                ctypes.c_double, ctypes.c_char_p] + [ctypes.c_int] * 6 + [ctypes.c_double,
                                ctypes.c_int, c_double_p]
    out = np.zeros(12)
    scf_set_occupations(occs, readocc)
    try:
        _check(L.hfg_scf_atomic(ctx.h, Z, Q, lmax, mmax, nelem, nnodes, nquad, Rmax, igrid, zexp, method.encode(), ldft,
                                mdft, symmetry, M, 1 if maverage else 0, maxit, convthr, verbose, _p(out)))
    finally:
        scf_set_occupations(None)
    keys = ["Etot", "Ekin", "Epot", "Ecoul", "Exx", "Exc", "Enucr"]
    r = dict(zip(keys, out[:7]))
    r["iterations"] = int(out[7])
    r["converged"] = (out[7] - int(out[7])) > 0.25
    r["tJ"], r["tK"], r["tXC"], r["tdiag"] = out[8:12]
    return r


class DeviceSCFStep(object):
    """One SCF iteration's hot path, device resident (torch tensors are only used as HBM buffers):

        P --(J + XC, compact, sharded)--> all-reduce --> F = sym(H0+J+XC) --(eig blocks, sharded)-->
        all-reduce --> (E, C) --> P' = C_occ C_occ^T

    i.e. main.cpp:784-788, 808-900 and 934-971 of the reference's diatomic driver without DIIS.
    """

    def __init__(self, basis, x_func, c_func, ldft, mdft, nocc, symmetry=1, device=0, rank=0, nranks=1,
                 dens_thr=1e-12, kfrac=0.0, device_tei=False, fock_shard="auto"):
        import torch
        self.torch = torch
        self.basis = basis
        self.x_func, self.c_func, self.nocc, self.thr = int(x_func), int(c_func), int(nocc), float(dens_thr)
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.ctx = Context(device, stream=torch.cuda.current_stream(self.dev).cuda_stream)
        self.ctx.set_shard(rank, nranks)
        self.rank, self.nranks = int(rank), int(nranks)
        # J + XC: "always" shards the build over the ranks ((L,|M|) channels, radial points) and all-reduces the compact
        # buffer; "never" lets every rank build all of it.  "auto" = never: after sum factorisation the whole J + XC build
        # takes 1.1 ms at Nbf = 4230 and 2.9 ms at 6102, less than an all-reduce of its 33 / 69 MB compact buffer over
        # xGMI.  The exchange build (tens of ms) and the eigensolve's blocks are always sharded.
        self.fock_shard = os.environ.get("HELFEM_FOCK_SHARD", fock_shard)  # the environment wins (tests, A/B runs)
        if device_tei:  # in-element tables (with the exchange-ordered ones when K is wanted) built on this context's device
            basis.compute_tei(float(kfrac) != 0.0, device=True, ctx=self.ctx)
        basis.upload(ldft, mdft, ctx=self.ctx)
        L = lib()
        L.hfg_fock_compact_size.restype = ctypes.c_int64
        L.hfg_eig_block_buf_size.restype = ctypes.c_int64
        L.hfg_fock_compact_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
        N = basis.Nbf()
        self.N = N
        self.blocks = basis.get_sym_idx(symmetry)
        self.blk_ptr, self.blk_idx = scf._blocks(self.blocks)
        f64 = dict(dtype=torch.float64, device=self.dev)
        blockid = np.zeros(N, dtype=np.int32)
        for ib, b in enumerate(self.blocks):
            blockid[b] = ib
        self.blockid = torch.from_numpy(blockid).to(self.dev)
        self.Fc = torch.zeros(int(L.hfg_fock_compact_size(basis.h)), **f64)
        self.scal = torch.zeros(3, **f64)
        self.F = torch.zeros(N * N, **f64)
        self.E = torch.zeros(N, **f64)
        self.C = torch.zeros(N * N, **f64)
        self.P = torch.zeros(N * N, **f64)
        nb = int(L.hfg_eig_block_buf_size(len(self.blocks), self.blk_ptr.ctypes.data_as(c_i64_p)))
        self.blockbuf = torch.zeros(nb, **f64)
        self.H0 = None
        self.Sinvh = None
        # hybrid functionals: F += kfrac K[Pa] (main.cpp:820-877); K is dense, its shards (output shells j % n == rank) sum
        self.kfrac = float(kfrac)
        self.have_C = False
        if self.kfrac != 0.0:
            self.K = torch.zeros(N * N, **f64)
            self.Pa = torch.zeros(N * N, **f64)
            L.hfg_exchange_occ_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_void_p]
            L.hfg_exchange_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def _ptr(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def set_matrices(self, H0, Sinvh):
        t = self.torch
        self.H0 = t.from_numpy(np.asfortranarray(H0).ravel(order="F").copy()).to(self.dev)
        self.ctx.fix_sinvh(None)
        self.Sinvh = t.from_numpy(np.asfortranarray(Sinvh).ravel(order="F").copy()).to(self.dev)
        self.ctx.fix_sinvh(self.Sinvh.data_ptr())  # constant until the next set_matrices

    def set_density(self, P, C=None):
        """total density P = 2 C_occ C_occ^T; C (N x >= nocc, optional): the orbitals it was formed from, which the exact
        exchange takes as the factors of P/2 (as inside the SCF loop, where the driver has just formed P from them)"""
        self.P.copy_(self.torch.from_numpy(np.asfortranarray(P).ravel(order="F").copy()))
        self.have_C = C is not None
        if C is not None:
            self.C[:self.N * self.nocc].copy_(self.torch.from_numpy(np.asfortranarray(C[:, :self.nocc]).ravel(order="F").copy()))

    def exchange(self):
        """K[Pa], Pa = P/2 (closed shell; basis.exchange(Pa) of main.cpp:822) into self.K"""
        self.torch.mul(self.P, 0.5, out=self.Pa)
        if self.have_C:
            _check(lib().hfg_exchange_occ_dev(self.ctx.h, self.basis.h, self._ptr(self.Pa), self._ptr(self.C),
                                              ctypes.c_int64(self.nocc), self._ptr(self.K)))
        else:
            _check(lib().hfg_exchange_dev(self.ctx.h, self.basis.h, self._ptr(self.Pa), self._ptr(self.K)))

    def fock_partial(self):
        _check(lib().hfg_fock_compact_dev(self.ctx.h, self.basis.h, self.x_func, self.c_func, self._ptr(self.P),
                                          self._ptr(self.Fc), self._ptr(self.scal), self.thr))

    def fock_finish(self):
        _check(lib().hfg_fock_finish_dev(self.ctx.h, self.basis.h, self._ptr(self.Fc), self._ptr(self.H0),
                                         self._ptr(self.blockid), self._ptr(self.F)))

    def eig_partial(self):
        _check(lib().hfg_eig_blocks_dev(self.ctx.h, ctypes.c_int64(self.N), self._ptr(self.F), self._ptr(self.Sinvh),
                                        len(self.blocks), self.blk_ptr.ctypes.data_as(c_i64_p),
                                        self.blk_idx.ctypes.data_as(c_i64_p), self._ptr(self.blockbuf)))

    def eig_finish(self):
        _check(lib().hfg_eig_assemble_dev(self.ctx.h, ctypes.c_int64(self.N), len(self.blocks),
                                          self.blk_ptr.ctypes.data_as(c_i64_p), self.blk_idx.ctypes.data_as(c_i64_p),
                                          self._ptr(self.blockbuf), self._ptr(self.E), self._ptr(self.C)))

    def density(self):
        _check(lib().hfg_form_density_dev(self.ctx.h, ctypes.c_int64(self.N), ctypes.c_int64(self.N), self._ptr(self.C),
                                          ctypes.c_int64(self.nocc), self._ptr(self.P)))

    def step(self, allreduce=None, exchange_blocks=None):
        """one iteration; allreduce(tensor) sums a tensor over ranks in place (None: single GPU);
        exchange_blocks(buffer, nblk) completes the per-block eigenvector slots on every rank (parallel.
        broadcast_block_slots_: one broadcast per block from its owner; None: the sum all-reduce, which the zero
        padding of the slots owned elsewhere also makes correct)"""
        shard_fock = self.nranks > 1 and self.fock_shard == "always"
        if self.nranks > 1 and not shard_fock:
            self.ctx.set_shard(0, 1)  # every rank builds the whole J + XC: no collective
        self.fock_partial()
        if self.nranks > 1 and not shard_fock:
            self.ctx.set_shard(self.rank, self.nranks)
        if allreduce is not None and (shard_fock or self.nranks == 1):
            allreduce(self.Fc)
            allreduce(self.scal)
        self.fock_finish()
        if self.kfrac != 0.0:
            self.exchange()
            if allreduce is not None:
                allreduce(self.K)
            self.F.add_(self.K, alpha=self.kfrac)  # K of a block-diagonal density is block diagonal: no mask needed
        self.eig_partial()
        if exchange_blocks is not None:
            exchange_blocks(self.blockbuf, len(self.blocks))
        elif allreduce is not None:
            allreduce(self.blockbuf)
        self.eig_finish()
        self.density()
        self.have_C = True  # self.C now holds the orbitals self.P was formed from

    def numpy(self, t, shape):
        return t.cpu().numpy().reshape(shape, order="F")
