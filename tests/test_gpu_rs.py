"""GPU parity of the range-separated exchange path of the atomic program, through the C ABI:
hfg_compute_rs_tei + hfg_basis_upload + hfg_rs_exchange against the oracle's loop-for-loop restatement of
atomic::basis::TwoDBasis::rs_exchange (src/atomic/TwoDBasis.cpp:1142-1322), the short-range LDA exchange functionals
on the XC grid, and range-separated hybrid SCF energies."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hf(native_libs):
    import helfem_amd
    if helfem_amd.device_count() < 1:
        pytest.fail("no HIP device: the GPU parity tests need a real MI355X")
    return helfem_amd


RS_CASES = {
    # name: (Z, lmax, mmax, nelem, nnodes)
    "s_only": (2, 0, 0, 3, 6),
    "sp": (10, 1, 1, 3, 5),
    "spd_m1": (18, 2, 1, 2, 6),
    "spdf_full_one_element": (4, 3, 3, 1, 5),
    "p_many_elements": (18, 1, 1, 7, 4),
}


@pytest.fixture(scope="module", params=sorted(RS_CASES))
def rcase(request, hf):
    import common
    Z, lmax, mmax, nelem, nnodes = RS_CASES[request.param]
    gb, ob = common.make_atomic_bases(Z, lmax, mmax, nelem, nnodes)
    gb.compute_tei(True)
    ob.compute_tei(True)
    return request.param, gb, ob


def _densities(gb):
    import common
    N = gb.Nbf()
    yield "general", common.random_density(N, 3, seed=11)
    yield "m_blocked", common.random_density(N, 2, seed=12, blocks=gb.get_sym_idx(1))
    yield "indefinite", common.random_density(N, 3, seed=31) - common.random_density(N, 2, seed=32)
    rng = np.random.RandomState(5)
    P = rng.uniform(-1, 1, size=(N, N))
    yield "full_rank", np.asfortranarray(P + P.T)  # not a low-rank density: general kernels


@pytest.mark.parametrize("kind,omega", [("yukawa", 0.4), ("yukawa", 2.0), ("erfc", 0.4), ("erfc", 1.5)])
def test_rs_exchange_parity(rcase, kind, omega):
    import common
    name, gb, ob = rcase
    if kind == "yukawa":
        gb.compute_yukawa(omega)
        ob.compute_yukawa(omega)
    else:
        gb.compute_erfc(omega)
        ob.compute_erfc(omega)
    for tag, P in _densities(gb):
        K = gb.rs_exchange(P)
        Ko = ob.rs_exchange(P)
        assert common.relerr(K, Ko) < 1e-12, (name, kind, omega, tag, common.relerr(K, Ko))
        # the Coulomb-kernel tables of the same handle are untouched by the second table set
        if tag == "general":
            assert common.relerr(gb.exchange(P), ob.exchange(P)) < 1e-12, (name, kind, "exchange after rs_exchange")
    assert np.all(gb.rs_exchange(np.zeros((gb.Nbf(), gb.Nbf()), order="F")) == 0.0)


def test_rs_exchange_general_kernels_parity(rcase, monkeypatch):
    """HELFEM_EXCHANGE=general: the kernels that take any symmetric P, for the factorised (Yukawa) tables too"""
    import common
    name, gb, ob = rcase
    monkeypatch.setenv("HELFEM_EXCHANGE", "general")
    gb.compute_yukawa(0.7)
    ob.compute_yukawa(0.7)
    tag, P = list(_densities(gb))[1]
    assert common.relerr(gb.rs_exchange(P), ob.rs_exchange(P)) < 1e-12, name
    # and for the pair tables of the erfc kernel, whose fast path is a different set of kernels (element-pair products)
    gb.compute_erfc(0.7)
    ob.compute_erfc(0.7)
    Kgen = gb.rs_exchange(P)
    assert common.relerr(Kgen, ob.rs_exchange(P)) < 1e-12, name
    monkeypatch.delenv("HELFEM_EXCHANGE")
    assert common.relerr(gb.rs_exchange(P), Kgen) < 1e-12, name


def test_rs_exchange_requires_tables(hf):
    import common
    gb, _ = common.make_atomic_bases(2, 0, 0, 2, 5, oracle=False)
    gb.compute_tei(True)
    N = gb.Nbf()
    with pytest.raises(RuntimeError, match="Primitive teis have not been computed"):
        gb.rs_exchange(np.eye(N, order="F"))


@pytest.mark.parametrize("funcs", [(546, 13), (641, 0), (178, 0), (0, 13)])
def test_short_range_lda_functionals_on_the_grid(hf, funcs):
    import common
    gb, ob = common.make_atomic_bases(10, 1, 1, 3, 6)
    gb.compute_tei(False)
    ob.compute_tei(False)
    ldft, mdft = 14, 9
    gb.upload(ldft, mdft)
    grid = hf.DFTGrid(gb, ldft, mdft)
    x, c = funcs
    N = gb.Nbf()
    blocks = gb.get_sym_idx(1)
    P = common.random_density(N, 2, seed=12, blocks=blocks)
    H, Exc, Nel, _ = grid.eval_Fxc(x, c, P)
    Ho, Exco, Nelo, _ = ob.eval_Fxc(ldft, mdft, x, c, P)
    assert abs(Exc - Exco) < 1e-11 * max(1.0, abs(Exco)), (funcs, Exc, Exco)
    assert common.relerr(H, Ho) < 1e-10, (funcs, common.relerr(H, Ho))
    Pa = common.random_density(N, 2, seed=3, blocks=blocks)
    Pb = common.random_density(N, 1, seed=4, blocks=blocks)
    Ha, Hb, Exc, Nel, _ = grid.eval_Fxc_pol(x, c, Pa, Pb)
    Hao, Hbo, Exco, Nelo, _ = ob.eval_Fxc_pol(ldft, mdft, x, c, Pa, Pb)
    assert abs(Exc - Exco) < 1e-11 * max(1.0, abs(Exco)), (funcs, "pol", Exc, Exco)
    assert common.relerr(Ha, Hao) < 1e-10 and common.relerr(Hb, Hbo) < 1e-10, (funcs, "pol")


RS_SCF_CASES = [
    ("He_CAM-LDA0", dict(Z=2, lmax=0, mmax=0, nelem=5, nnodes=10, method="hyb_lda_xc_cam_lda0")),
    ("Li_CAM-LDA0_unrestricted", dict(Z=3, lmax=0, mmax=0, nelem=5, nnodes=10, method="hyb_lda_xc_cam_lda0", M=2)),
    ("Ne_CAM-LDA0", dict(Z=10, lmax=1, mmax=1, nelem=5, nnodes=10, method="hyb_lda_xc_cam_lda0")),
]


@pytest.mark.parametrize("name,kw", RS_SCF_CASES, ids=[c[0] for c in RS_SCF_CASES])
def test_range_separated_hybrid_scf_energy_parity(hf, name, kw, monkeypatch):
    """converged energies, device-resident loop and host-pointer loop, against the oracle: 1e-8 Eh (BASELINE's bar)"""
    import oracle_lib as orc
    o = orc.scf_atomic(convthr=1e-9, maxit=60, **kw)
    g = hf.scf_atomic(convthr=1e-9, maxit=60, **kw)
    assert o["converged"] and g["converged"]
    for k in ("Etot", "Ekin", "Epot", "Ecoul", "Exx", "Exc"):
        assert abs(g[k] - o[k]) < 1e-8 * max(1.0, abs(o[k])), (name, k, g[k], o[k])
    monkeypatch.setenv("HELFEM_SCF", "host")
    h = hf.scf_atomic(convthr=1e-9, maxit=60, **kw)
    assert h["converged"] and abs(h["Etot"] - o["Etot"]) < 1e-8 and abs(h["Exx"] - o["Exx"]) < 1e-8


def test_diatomic_driver_rejects_range_separation(hf):
    """src/diatomic/main.cpp:393-394"""
    with pytest.raises(RuntimeError, match="Range separated functionals are not supported"):
        hf.scf_diatomic(1, 1, 1.4, [2], 2, 5, "hyb_lda_xc_cam_lda0")


# ---- initial-guess model potentials --------------------------------------------
@pytest.mark.parametrize("pots", [((0, 3), (0, 1)), ((3, 3), (3, 1)), ((1, 3, 0.56), (1, 1, 1.0)), ((3, 3), (0, 0))])
def test_model_potential_parity_diatomic(hf, pots):
    """hfg_model_potential against the oracle's TwoDGrid::model_potential, and with point nuclei against the analytic
    nuclear-attraction matrix"""
    import common
    import oracle_lib as orc
    gb, ob = common.make_bases(3, 1, 3.0, (4, 2), 3, 8)
    gb.compute_tei(False)
    ldft, mdft = 28, 13
    gb.upload(ldft, mdft)
    V = gb.model_potential(*pots)
    Vo = orc.model_potential(ob, pots[0], pots[1], lang=ldft, mang=mdft)
    assert common.relerr(V, Vo) < 1e-11, (pots, common.relerr(V, Vo))
    if pots[0][0] == 0 and pots[1][0] == 0:
        assert common.relerr(V, gb.nuclear()) < 1e-9


def test_model_potential_atomic_and_errors(hf):
    import common
    import oracle_lib as orc
    ga, oa = common.make_atomic_bases(10, 1, 1, 4, 8)
    ga.compute_tei(False)
    assert common.relerr(ga.model_potential((3, 10)), orc.model_potential(oa, (3, 10))) < 1e-14
    assert common.relerr(ga.model_potential((0, 10)), ga.nuclear()) < 1e-14
    with pytest.raises(RuntimeError, match="Unsupported guess"):
        hf.scf_set_iguess(2)
    gb, _ = common.make_bases(1, 1, 1.4, (2,), 2, 5, oracle=False)
    gb.compute_tei(False)
    gb.upload(0, 0)
    with pytest.raises(RuntimeError, match="XC grid tables were not uploaded"):
        gb.model_potential((3, 1), (3, 1))


def test_thomas_fermi_guess_scf_parity(hf, monkeypatch):
    """--iguess 3 through the device-resident loop and the host-pointer loop against the oracle (same guess there)"""
    import oracle_lib as orc
    cases = [("atomic", dict(Z=10, lmax=1, mmax=1, nelem=5, nnodes=10, method="gga_x_pbe-gga_c_pbe")),
             ("diatomic", dict(Z1=3, Z2=1, Rbond=3.0, lmmax=[4, 2], nelem=3, nnodes=8, method="HF")),
             ("diatomic", dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[4, 3], nelem=3, nnodes=8, method="lda_x-lda_c_vwn"))]
    try:
        hf.scf_set_iguess(3)
        orc.scf_set_iguess(3)
        for prog, kw in cases:
            gfn, ofn = (hf.scf_atomic, orc.scf_atomic) if prog == "atomic" else (hf.scf_diatomic, orc.scf_diatomic)
            o = ofn(convthr=1e-9, maxit=60, **kw)
            g = gfn(convthr=1e-9, maxit=60, **kw)
            assert o["converged"] and g["converged"] and g["iterations"] == o["iterations"], (kw, g["iterations"], o["iterations"])
            assert abs(g["Etot"] - o["Etot"]) < 1e-8 * max(1.0, abs(o["Etot"]) / 10), (kw, g["Etot"], o["Etot"])
            monkeypatch.setenv("HELFEM_SCF", "host")
            h = gfn(convthr=1e-9, maxit=60, **kw)
            monkeypatch.delenv("HELFEM_SCF")
            assert h["converged"] and abs(h["Etot"] - o["Etot"]) < 1e-8 * max(1.0, abs(o["Etot"]) / 10)
    finally:
        hf.scf_set_iguess(0)
        orc.scf_set_iguess(0)
