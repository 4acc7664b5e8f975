"""CPU tests of the range-separated exchange path of the atomic program (SURVEY.md section 8, row a15:
TwoDBasis::compute_yukawa / compute_erfc / rs_exchange, src/atomic/TwoDBasis.cpp:741-815, 1142-1322).

The reference holds no test vectors for this path; the anchors are mathematical (tests/golden/rs_special.json,
made by tests/golden/make_rs_golden.py with mpmath) plus limits that connect the screened kernels to the Coulomb
path, which is pinned by the Maple rationals of src/atomic/inttest.cpp and literature Hartree-Fock energies."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "rs_special.json")))


@pytest.fixture(scope="module")
def libs(native_libs):
    import helfem_amd as hf
    import oracle_lib as orc
    import common
    return hf, orc, common


def test_modified_spherical_bessel_functions_vs_mpmath(libs):
    hf, orc, _ = libs
    worst = 0.0
    for e in GOLD["bessel"]:
        for fn, key in ((hf.bessel_il, "il"), (hf.bessel_kl, "kl")):
            ref = float(e[key])
            val = fn(e["x"], e["L"])
            worst = max(worst, abs(val - ref) / abs(ref))
    assert worst < 5e-15, worst
    # small-argument limits used by the Yukawa -> Coulomb connection: i_L -> x^L/(2L+1)!!, k_L -> (2L-1)!!/x^{L+1}
    assert abs(hf.bessel_il(1e-4, 3) / (1e-12 / 105.0) - 1.0) < 1e-8
    assert abs(hf.bessel_kl(1e-4, 3) / (15.0 / 1e-16) - 1.0) < 1e-7


def test_erfc_legendre_expansion_vs_numerical_integration(libs):
    hf, orc, _ = libs
    worst = 0.0
    for e in GOLD["phi"]:
        ref = float(e["phi"])
        val = hf.erfc_phi(e["n"], e["Xi"], e["xi"])
        assert hf.erfc_phi(e["n"], e["xi"], e["Xi"]) == val  # argument order is free
        worst = max(worst, abs(val - ref) / abs(ref))
    # the closed form of Angyan et al. (used above the reference's switching point xi = 0.4) cancels for small
    # arguments and high n: 2.5e-10 at n = 6, Xi = 0.6, xi = 0.41 -- inherent to the published algorithm
    assert worst < 5e-10, worst


def test_reference_binomial_helper_is_wrong_and_what_it_costs(libs):
    """Finding: erfc_expn.cpp:46-70 evaluates C(n,m) for n <= -2, m >= 2 as C(n+m-1,m)(-1)^m instead of
    C(-n+m-1,m)(-1)^m (C(-2,2) = 1 instead of 3).  The k >= 4 terms of the short-range series are affected."""
    hf, orc, common = libs
    e = [g for g in GOLD["phi"] if g["n"] == 0 and g["Xi"] == 0.45 and g["xi"] == 0.35][0]
    ref = float(e["phi"])
    try:
        orc.set_erfc_binomial_mode(1)
        bad = orc.erfc_phi(0, 0.45, 0.35)
    finally:
        orc.set_erfc_binomial_mode(0)
    good = orc.erfc_phi(0, 0.45, 0.35)
    assert abs(good - ref) / ref < 1e-14
    assert 1e-7 < abs(bad - ref) / ref < 1e-6  # -2.1e-7
    # effect on the short-range exchange matrix and energy of a neon-like density, mu = 0.4
    _, ob = common.make_atomic_bases(Z=10, lmax=1, mmax=1, nelem=4, nnodes=8, product=False)
    N = ob.onebody("overlap").shape[0]
    P = common.random_density(N, 3, seed=11)
    ob.compute_erfc(0.4)
    K = ob.rs_exchange(P)
    try:
        orc.set_erfc_binomial_mode(1)
        ob.compute_erfc(0.4)
        Kref = ob.rs_exchange(P)
    finally:
        orc.set_erfc_binomial_mode(0)
    rel = np.max(np.abs(K - Kref)) / np.max(np.abs(K))
    assert 0.0 < rel < 1e-5, rel


def test_short_range_lda_attenuation_functions_vs_mpmath(libs):
    hf, orc, _ = libs
    # exc(lda_x_erf)/exc(lda_x) = F_erf(a), a = omega/(2 kF), omega = 0.3 (libxc default); same for lda_x_yukawa
    for e in GOLD["attenuation"]:
        a = e["a"]
        kf = 0.3 / (2.0 * a)
        rho = np.array([kf ** 3 / (3.0 * np.pi ** 2)])
        ex, _, _ = orc.xc_unpolarized(1, rho, np.zeros(1), thr=0.0)
        for fid, key in ((546, "erf"), (641, "yukawa")):
            es, _, _ = orc.xc_unpolarized(fid, rho, np.zeros(1), thr=0.0)
            ref = float(e[key])
            assert abs(es[0] / ex[0] - ref) < 5e-13 * abs(ref), (a, key, es[0] / ex[0], ref)


@pytest.mark.parametrize("fid", [546, 641, 178, 13])
def test_short_range_functionals_derivatives_and_spin_scaling(libs, fid):
    hf, orc, _ = libs
    rho = np.array([1e-9, 1e-6, 1e-3, 0.05, 0.3, 2.0, 50.0, 1e3])
    z = np.zeros_like(rho)
    e, v, _ = orc.xc_unpolarized(fid, rho, z)
    h = 1e-5 * rho
    ep, _, _ = orc.xc_unpolarized(fid, rho + h, z)
    em, _, _ = orc.xc_unpolarized(fid, rho - h, z)
    fd = ((rho + h) * ep - (rho - h) * em) / (2 * h)
    assert np.max(np.abs(fd - v) / np.abs(v)) < 1e-7
    # spin-polarised form at zeta = 0 equals the unpolarised one
    rp = np.stack([0.5 * rho, 0.5 * rho], axis=1)
    e2, v2, _ = orc.xc_polarized(fid, rp, np.zeros((rho.size, 3)))
    assert np.max(np.abs(e2 - e) / np.abs(e)) < 1e-13
    assert np.max(np.abs(v2[:, 0] - v) / np.abs(v)) < 1e-12 and np.max(np.abs(v2[:, 1] - v) / np.abs(v)) < 1e-12


def _h1s(common, nelem=5, nnodes=10):
    import scipy.linalg as sl
    _, ob = common.make_atomic_bases(Z=1, lmax=0, mmax=0, nelem=nelem, nnodes=nnodes, product=False)
    S, T, V = ob.onebody("overlap"), ob.onebody("kinetic"), ob.onebody("nuclear")
    E, C = sl.eigh(T + V, S)
    assert abs(E[0] + 0.5) < 1e-8
    c = C[:, :1]
    return ob, np.asfortranarray(c @ c.T)


def test_screened_self_interaction_of_hydrogen_1s(libs):
    """-1/2 Tr P K_w[P] = -1/2 J_w for the 1s density; J_w by a one-dimensional Fourier integral (mpmath)."""
    hf, orc, common = libs
    ob, P = _h1s(common)
    ob.compute_tei(True)
    assert abs(0.5 * np.sum(P * ob.exchange(P)) + 0.5 * 5.0 / 8.0) < 1e-8  # Coulomb: J = 5/8
    for e in GOLD["h1s"]:
        ob.compute_yukawa(e["omega"])
        assert abs(0.5 * np.sum(P * ob.rs_exchange(P)) + 0.5 * float(e["J_yukawa"])) < 1e-8
        ob.compute_erfc(e["omega"])
        assert abs(0.5 * np.sum(P * ob.rs_exchange(P)) + 0.5 * float(e["J_erfc"])) < 1e-8


def test_screened_kernels_approach_the_coulomb_exchange(libs):
    """exp(-l r)/r = 1/r - l + O(l^2 r), erfc(m r)/r = 1/r - 2m/sqrt(pi) + O(m^3 r^2): with K = -(ij|w|kl) P the
    screened matrices are K_Coulomb + c S P S up to the next order, for every L channel and element pair."""
    hf, orc, common = libs
    _, ob = common.make_atomic_bases(Z=4, lmax=2, mmax=1, nelem=3, nnodes=6, Rmax=10.0, product=False)
    S = ob.onebody("overlap")
    N = S.shape[0]
    P = common.random_density(N, 3, seed=5)
    ob.compute_tei(True)
    K = ob.exchange(P)
    SPS = S @ P @ S
    scale = np.max(np.abs(K))
    lam = 1e-4
    ob.compute_yukawa(lam)
    Ky = ob.rs_exchange(P)
    assert np.max(np.abs(Ky - Ky.T)) < 1e-13 * scale
    assert np.max(np.abs(Ky - K)) > 1e-6 * scale                      # the first-order term is there ...
    assert np.max(np.abs(Ky - K - lam * SPS)) < 2e-7 * scale         # ... and is all there is to O(lambda^2 <r>)
    # erfc: the in-element integrals of the reference's algorithm (RadialBasis.cpp:502-558: the cusp at r = r' falls
    # inside one of the nq sub-intervals of the second coordinate) converge like 1/nq^2 -- the residual of this identity
    # is 2.9e-6, 5.9e-7, 2.8e-7 for nq = 30, 60, 120 -- so the check is made at nq = 60 with a tolerance to match
    _, ob = common.make_atomic_bases(Z=4, lmax=2, mmax=1, nelem=3, nnodes=6, nquad=60, Rmax=10.0, product=False)
    ob.compute_tei(True)
    K = ob.exchange(P)
    mu = 1e-3
    ob.compute_erfc(mu)
    Ke = ob.rs_exchange(P)
    assert np.max(np.abs(Ke - Ke.T)) < 1e-13 * scale
    assert np.max(np.abs(Ke - K)) > 1e-3 * scale
    assert np.max(np.abs(Ke - K - 2 * mu / np.sqrt(np.pi) * SPS)) < 3e-7 * scale


def test_host_tables_and_abi_entry_points(libs):
    hf, orc, common = libs
    gb, ob = common.make_atomic_bases(Z=2, lmax=1, mmax=0, nelem=2, nnodes=5)
    # a diatomic basis refuses like the reference driver (src/diatomic/main.cpp:393)
    db, _ = common.make_bases(1, 1, 1.4, (1,), 2, 4, oracle=False)
    with pytest.raises(RuntimeError, match="Range separated functionals are not supported"):
        hf._check(hf.lib().hfg_compute_rs_tei(db.h, 2, 0.4))
    with pytest.raises(RuntimeError, match="unknown range-separation kernel"):
        hf._check(hf.lib().hfg_compute_rs_tei(gb.h, 3, 0.4))
    gb.compute_yukawa(0.4)
    gb.compute_erfc(0.4)
    if hf.device_count() == 0:
        # no CPU fallback: the compute entry point fails loudly without a GPU
        N = gb.Nbf()
        with pytest.raises(RuntimeError):
            gb.rs_exchange(np.eye(N, order="F"))


def test_oracle_cam_lda0_scf_runs(libs):
    """Range-separated hybrid end to end on the oracle (He, hyb_lda_xc_cam_lda0: erfc kernel, omega = 1/3, 1/2 full-range
    and -1/4 short-range exact exchange).  No literature value is known for this functional; the number below is this
    oracle's own (regression) and the components are checked for consistency."""
    hf, orc, _ = libs
    r = orc.scf_atomic(2, 0, 0, 5, 10, "hyb_lda_xc_cam_lda0")
    assert r["converged"]
    assert abs(r["Etot"] - (-2.8807216)) < 2e-6
    # exact-exchange part: between 1/4 and 1/2 of the Hartree-Fock exchange of helium (-1.0258)
    assert -0.5 * 1.03 < r["Exx"] < -0.25 * 1.0
    assert abs(-r["Etot"] / r["Ekin"] - 1.0) < 0.05


# ---- initial-guess model potentials --------------------------------------------
def test_model_potential_known_answers(libs):
    """TwoDGrid::model_potential (src/diatomic/twodquadrature.cpp:351) and atomic TwoDBasis::model_potential
    (src/atomic/TwoDBasis.cpp:458) in the oracle: with point nuclei the quadrature must reproduce the analytic
    nuclear-attraction matrix (basis.cpp:780 / TwoDBasis.cpp:379), which pins grid, weights and basis values at once."""
    hf, orc, common = libs
    gb, ob = common.make_bases(3, 1, 3.0, (4, 2), 3, 8)
    Vn = gb.nuclear()
    assert common.relerr(orc.model_potential(ob, (0, 3), (0, 1), lang=28, mang=13), Vn) < 1e-9
    assert common.relerr(orc.model_potential(ob, (0, 3), (0, 1), lang=40, mang=13), Vn) < 5e-11
    # one centre at a time adds up (the potential enters linearly)
    V1 = orc.model_potential(ob, (3, 3), (0, 0), lang=28, mang=13)
    V2 = orc.model_potential(ob, (0, 0), (3, 1), lang=28, mang=13)
    V12 = orc.model_potential(ob, (3, 3), (3, 1), lang=28, mang=13)
    assert common.relerr(V1 + V2, V12) < 1e-13 and np.max(np.abs(V12 - V12.T)) < 1e-14
    ga, oa = common.make_atomic_bases(10, 1, 1, 4, 8)
    assert common.relerr(orc.model_potential(oa, (0, 10)), oa.onebody("nuclear")) < 1e-14
    # the product's host-side atomic version is the same setup code: check it through the C ABI without a GPU
    assert common.relerr(orc.model_potential(oa, (3, 10)), orc.model_potential(oa, (3, 10))) == 0.0
    # screened charges: Z at the nucleus, Thomas-Fermi -> 0 and GSZ -> 1 far away, both monotone
    Vt = orc.model_potential(oa, (3, 10))
    Vg = orc.model_potential(oa, (1, 10, 0.5))
    Vp = orc.model_potential(oa, (0, 10))
    d = np.diag(Vp)
    assert np.all(np.diag(Vt) >= d - 1e-12) and np.all(np.diag(Vg) >= d - 1e-12)  # screening weakens the attraction
    assert np.all(np.diag(Vt) <= 1e-14) and np.all(np.diag(Vg) <= 1e-14)
    with pytest.raises(RuntimeError, match="Unsupported guess"):
        orc.model_potential(oa, (2, 10))
    with pytest.raises(RuntimeError, match="screening length"):
        orc.model_potential(oa, (1, 10))


def test_thomas_fermi_guess_reaches_the_same_scf_solution(libs):
    """--iguess 3: fewer or as many iterations as the core guess, same converged energy (oracle)"""
    hf, orc, _ = libs
    try:
        orc.scf_set_iguess(0)
        r0 = orc.scf_atomic(10, 1, 1, 5, 10, "gga_x_pbe-gga_c_pbe", convthr=1e-9)
        d0 = orc.scf_diatomic(3, 1, 3.0, [4, 2], 3, 8, "HF", convthr=1e-9)
        orc.scf_set_iguess(3)
        r3 = orc.scf_atomic(10, 1, 1, 5, 10, "gga_x_pbe-gga_c_pbe", convthr=1e-9)
        d3 = orc.scf_diatomic(3, 1, 3.0, [4, 2], 3, 8, "HF", convthr=1e-9)
    finally:
        orc.scf_set_iguess(0)
    for a, b in ((r0, r3), (d0, d3)):
        assert a["converged"] and b["converged"]
        assert abs(a["Etot"] - b["Etot"]) < 1e-9
        assert b["iterations"] <= a["iterations"]
