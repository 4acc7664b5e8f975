"""ADIIS + CDIIS weights of the Fock extrapolation (helfem_amd/csrc/host/diis.cpp, the restatement of
/root/reference/src/general/diis.cpp:214-290, 297-372, 492-600) through the C ABI (host-side arithmetic, no GPU):
against the defining optimisation problems solved here with numpy/scipy."""
import ctypes

import numpy as np
import pytest


def weights(hf, B, T, E, maxerr, eps=1e-2, thr=1e-3, mode=0):
    n = len(E)
    L = hf.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    L.hfg_diis_weights.argtypes = [ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, dp,
                                   ctypes.POINTER(ctypes.c_int)]
    B = np.ascontiguousarray(B, dtype=float)
    T = np.ascontiguousarray(T, dtype=float)
    E = np.ascontiguousarray(E, dtype=float)
    w = np.zeros(n)
    dropped = ctypes.c_int(0)
    rc = L.hfg_diis_weights(n, B.ctypes.data_as(dp), T.ctypes.data_as(dp), E.ctypes.data_as(dp), float(maxerr), eps, thr, mode,
                            w.ctypes.data_as(dp), ctypes.byref(dropped))
    if rc != 0:
        raise RuntimeError(L.hfg_last_error().decode())
    return w, dropped.value


@pytest.fixture(scope="module")
def hf(native_libs):
    import helfem_amd
    return helfem_amd


def history(n, N=12, seed=0):
    """a synthetic SCF history: symmetric F_i, P_i and antisymmetric-like error vectors"""
    rng = np.random.RandomState(seed)
    Fs = [(lambda a: a + a.T)(rng.standard_normal((N, N))) for _ in range(n)]
    Ps = [(lambda c: c @ c.T)(rng.standard_normal((N, 3))) for _ in range(n)]
    errs = [rng.standard_normal(N * N) * 10.0 ** (-i) for i in range(n)]
    B = np.array([[e1 @ e2 for e2 in errs] for e1 in errs])
    T = np.array([[np.trace(P @ F) for F in Fs] for P in Ps])
    E = -np.arange(n, dtype=float)
    return B, T, E


def quadratic_history(n, N=10, seed=0):
    """densities P_i around a minimiser P* = mean(P_i) of the model energy |P - P*|^2 / 2, Fock matrices F_i = P_i - P*:
    the ADIIS functional is then minimised by equal weights (an interior point of the simplex)"""
    rng = np.random.RandomState(seed)
    Ps = [(lambda a: a + a.T)(rng.standard_normal((N, N))) for _ in range(n)]
    Pstar = sum(Ps) / n
    Fs = [P - Pstar for P in Ps]
    errs = [rng.standard_normal(N * N) for _ in range(n)]
    B = np.array([[e1 @ e2 for e2 in errs] for e1 in errs])
    T = np.array([[np.trace(P @ F) for F in Fs] for P in Ps])
    return B, T, -np.arange(n, dtype=float)


def test_adiis_finds_the_interior_minimum_of_a_quadratic_model(hf):
    for n in (2, 3, 5):
        B, T, E = quadratic_history(n, seed=n)
        w, dropped = weights(hf, B, T, E, 1.0, mode=2)
        assert dropped == 0 and np.max(np.abs(w - 1.0 / n)) < 1e-6, (n, w)


def test_cdiis_weights_solve_the_pulay_equations(hf):
    for n in (1, 2, 4, 6):
        B, T, E = history(n, seed=n)
        w, dropped = weights(hf, B, T, E, 1e-5, mode=1)
        assert dropped == 0 and abs(w.sum() - 1.0) < 1e-12
        # B w = lambda 1: B w is a constant vector
        r = B @ w
        assert np.max(np.abs(r - r.mean())) < 1e-9 * np.max(np.abs(B)), (n, r)


def test_cdiis_alone_refuses_large_errors(hf):
    B, T, E = history(3)
    with pytest.raises(RuntimeError, match="DIIS error too large"):
        weights(hf, B, T, E, 0.5, mode=1)


def adiis_energy(c, T):
    n = len(c) - 1
    PiF = T[:, n] - T[n, n]
    PiFj = T - T[:, [n]] - T[[n], :] + T[n, n]
    return 2.0 * c @ PiF + c @ PiFj @ c


def test_adiis_weights_minimise_the_adiis_functional_on_the_simplex(hf):
    from scipy.optimize import minimize
    for n in (2, 3, 5):
        B, T, E = history(n, seed=10 + n)
        w, dropped = weights(hf, B, T, E, 1.0, mode=2)
        # entries the extrapolation dropped (weight of the newest matrix below sqrt(eps): solve_F cuts the oldest) carry zero
        assert abs(w.sum() - 1.0) < 1e-12 and np.all(w >= 0.0) and np.all(w[:dropped] == 0.0)
        m = n - dropped
        Ts, ws = T[dropped:, dropped:], w[dropped:]
        if m == 1:
            continue
        best = min((minimize(adiis_energy, x0, args=(Ts,), method="SLSQP", bounds=[(0, 1)] * m,
                             constraints=[dict(type="eq", fun=lambda c: c.sum() - 1.0)], options=dict(ftol=1e-14, maxiter=500))
                    for x0 in [np.ones(m) / m] + [np.eye(m)[k] * 0.9 + 0.1 / m for k in range(m)]), key=lambda r: r.fun)
        # the reference's parametrisation c = x^2 / x.x with L-BFGS descends from the equal-weights start: never worse than
        # that start, and here it reaches the constrained minimum
        assert adiis_energy(ws, Ts) <= adiis_energy(np.ones(m) / m, Ts) + 1e-12
        assert adiis_energy(ws, Ts) <= best.fun + 1e-6 * (1.0 + abs(best.fun)), (n, adiis_energy(ws, Ts), best.fun)


def test_mixing_rule_between_diisthr_and_diiseps(hf):
    """w = diisw w_cdiis + (1 - diisw) w_adiis with diisw = clamp(1 - (err - thr)/(eps - thr)) (diis.cpp:236-270)"""
    B, T, E = quadratic_history(4, seed=3)
    wc, _ = weights(hf, B, T, E, 1e-6, mode=1)
    wa, dr = weights(hf, B, T, E, 1e-6, mode=2)
    assert dr == 0
    eps, thr = 1e-2, 1e-3
    for err, share in ((5e-4, 1.0), (thr, 1.0), (0.5 * (eps + thr), 0.5), (eps, 0.0), (0.3, 0.0)):
        w, dropped = weights(hf, B, T, E, err, eps, thr, mode=0)
        assert dropped == 0
        assert np.max(np.abs(w - (share * wc + (1.0 - share) * wa))) < 1e-12, (err, w)


def test_energy_rise_switches_cdiis_off(hf):
    """cool-off: an energy increase of more than 0.1 between the last two entries disables CDIIS (diis.cpp:248-256); with
    a DIIS error below diisthr the ADIIS share is zero too, the weight of the newest matrix vanishes and the history is
    cut down until one entry is left (solve_F, diis.cpp:394-403)"""
    B, T, E = history(3, seed=5)
    E = np.array([-3.0, -3.5, -3.0])
    w, dropped = weights(hf, B, T, E, 1e-5, mode=0)
    assert dropped == 2 and w[2] == 1.0
