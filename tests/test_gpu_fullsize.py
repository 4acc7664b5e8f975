"""The hot path at BASELINE.json's full size (configs[3] sizing of bench.py: N2, PBE, 5 elements x 15 nodes,
lmmax = [20,20] -> Nbf = 4230, symmetry blocks 1380/1470/1380, XC grid 92 x 13; configs[4] sizing: LiF, Nbf = 6102).

Two kinds of checks:
  * DIRECT comparison with the oracle / LAPACK wherever that costs seconds (the `*_against_the_oracle` tests below):
    the whole Coulomb matrix, the eigenvalues of the Fock matrix of one SCF step against LAPACK per symmetry block, the XC
    matrix and sums of a radial shard (Context.set_shard(r, 25): 15 of the 375 radial points -- the oracle's dense
    algorithm needs 2.4 s per point), restricted and spin-polarised, and a sample of output blocks of the exchange
    matrix (the reference builds K block by block, basis.cpp:1575-1579; the oracle takes a block list);
  * size-independent properties of each stage for what the oracle cannot reach in seconds: orthonormality and
    residuals of the eigensolve, Tr PS, linearity and symmetry of J and K, the sign of the exchange energy, the electron
    count of the XC quadrature and the consistency of the XC matrix with the derivative of the XC energy."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full(native_libs):
    import helfem_amd as hf
    if hf.device_count() < 1:
        pytest.fail("no HIP device: the full-size checks need a real MI355X")
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["n2_pbe_nbf4230"])
    basis, bval, lval, mval, ldft, mdft = bench.build_basis(hf, w)
    basis.compute_tei(True, device=True)  # in-element tables built on the GPU: the 1 GB never exists on the host
    basis.upload(ldft, mdft)
    N = basis.Nbf()
    assert N == 4230
    S, T, V = basis.overlap(), basis.kinetic(), basis.nuclear()
    blocks = basis.get_sym_idx(1)
    assert sorted(len(b) for b in blocks) == [1380, 1380, 1470]
    X = hf.scf.form_Sinvh(S, False, blocks)
    H0 = np.asfortranarray(T + V)
    E, C = hf.scf.eig_gsym_sub(H0, X, blocks)
    return dict(hf=hf, basis=basis, ldft=ldft, mdft=mdft, N=N, S=S, H0=H0, X=X, blocks=blocks, E=E, C=C, w=w, bval=bval, lval=lval,
                mval=mval)


def _oracle_basis(fx, exchange):
    """the CPU checker's basis of a bench workload (tests/oracle_lib.py; tables on all host threads)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import oracle_lib as orc
    os.environ.setdefault("HELFEM_NUM_THREADS", str(bench.usable_cores()))
    w = fx["w"]
    ob = orc.OracleBasis(w["Z1"], w["Z2"], 0.5 * w["Rbond"], w["nnodes"], 5 * w["nnodes"], fx["bval"], fx["lval"], fx["mval"], 10)
    ob.compute_tei(exchange)
    return ob


def _sample_pairs(basis):
    """output blocks (jang, kang) of K covering sigma-sigma, sigma-pi, pi-pi', pi-pi with equal m, low and high l"""
    l, m = basis.lval, basis.mval
    def first(ll, mm):
        return next(i for i in range(len(l)) if l[i] == ll and m[i] == mm)
    lmax_s = max(l[i] for i in range(len(l)) if m[i] == 0)
    lmax_p = max(l[i] for i in range(len(l)) if m[i] == 1)
    # (in the homonuclear workload blocks between shells of different inversion parity vanish for the sampled density --
    # sigma_g + pi_u orbitals -- so the pairs keep l_j + l_k even within one m and odd between sigma and pi)
    s0, s1, s2, s3, sh = first(0, 0), first(1, 0), first(2, 0), first(3, 0), first(lmax_s, 0)
    p1, p3, pm1, pm3, ph = first(1, 1), first(3, 1), first(1, -1), first(3, -1), first(lmax_p, 1)
    return [(s0, s0), (s0, s2), (s1, s1), (s1, s3), (s0, p1), (s2, p1), (p1, p1), (p1, p3), (p1, pm1), (pm1, pm3), (s1, sh),
            (pm1, ph)]


def _block_of(basis, jang):
    """pure (boundary-cleaned) indices of angular shell jang (basis.cpp:482: m != 0 shells drop their first radial function)"""
    Nrad = basis.Nrad()
    off = 0
    for a in range(jang):
        off += Nrad - (1 if basis.mval[a] != 0 else 0)
    return np.arange(off, off + Nrad - (1 if basis.mval[jang] != 0 else 0))


def _step_fock(ctx_fixture):
    """the Fock matrix of one bench step from the core-Hamiltonian density: F = H0 + J + XC (PBE), block-masked"""
    hf, basis, C, N = ctx_fixture["hf"], ctx_fixture["basis"], ctx_fixture["C"], ctx_fixture["N"]
    nocc = ctx_fixture["w"]["nocc"]
    P = np.asfortranarray(2.0 * hf.scf.form_density(C, nocc))
    J = basis.coulomb(P)
    H = hf.DFTGrid(basis, ctx_fixture["ldft"], ctx_fixture["mdft"]).eval_Fxc(101, 130, P)[0]
    F = np.zeros((N, N), order="F")
    Fd = ctx_fixture["H0"] + J + H
    for b in ctx_fixture["blocks"]:
        F[np.ix_(b, b)] = Fd[np.ix_(b, b)]
    return P, J, F


def _check_coulomb_eig_xc_k(fx, shard_rank):
    """the four direct comparisons at one full-size workload"""
    hf, basis, N, blocks, X = fx["hf"], fx["basis"], fx["N"], fx["blocks"], fx["X"]
    w = fx["w"]
    ob = _oracle_basis(fx, True)
    assert ob.Nbf == N
    P, J, F = _step_fock(fx)
    # (i) Coulomb: the whole matrix (TwoDBasis::coulomb, basis.cpp:1359)
    Jo = ob.coulomb(P)
    assert np.max(np.abs(J - Jo)) < 1e-12 * np.max(np.abs(Jo))
    # (ii) eigenvalues of the step's Fock matrix per symmetry block against LAPACK (scf::eig_gsym_sub, scf_helpers.cpp:142)
    E, C = hf.scf.eig_gsym_sub(F, X, blocks)
    try:
        import torch
        eigh = lambda A: torch.linalg.eigvalsh(torch.from_numpy(np.ascontiguousarray(A))).numpy()
    except ImportError:
        eigh = np.linalg.eigvalsh
    Eref = []
    for b in blocks:
        cols = np.where(np.max(np.abs(X[b, :]), axis=0) > 0)[0]
        Xb = X[np.ix_(b, cols)]
        Eref.append(eigh(Xb.T @ F[np.ix_(b, b)] @ Xb))
    Eref = np.sort(np.concatenate(Eref))
    assert np.max(np.abs(E - Eref)) < 1e-10 * np.max(np.abs(Eref))
    # (iii) XC on a radial shard: 15 of the 375 points (Q % 25 == r), restricted and spin-polarised
    # (DFTGrid::eval_Fxc, dftgrid.cpp:769 / :812)
    NQ = w["nelem"] * 5 * w["nnodes"]
    pts = [q for q in range(NQ) if q % 25 == shard_rank]
    grid = hf.DFTGrid(basis, fx["ldft"], fx["mdft"])
    basis.ctx.set_shard(shard_rank, 25)
    try:
        H, Exc, Nel, _ = grid.eval_Fxc_dev(101, 130, P)
        Pa = np.asfortranarray(hf.scf.form_density(fx["C"], w["nocc"]))
        Pb = np.asfortranarray(hf.scf.form_density(fx["C"], w["nocc"] - 1))  # a doublet-like spin density
        Ha, Hb, Excp, Nelp, _ = grid.eval_Fxc_dev(101, 130, Pa, Pb)
    finally:
        basis.ctx.set_shard(0, 1)
    import bench
    nth = bench.usable_cores()
    Ho, Exco, Nelo, _ = ob.eval_Fxc_points(fx["ldft"], fx["mdft"], 101, 130, P, pts, threads=nth)
    sc = np.max(np.abs(Ho))
    assert sc > 1e-3  # the shard really carries part of the matrix
    assert np.max(np.abs(H - Ho)) < 1e-10 * sc, np.max(np.abs(H - Ho)) / sc
    assert abs(Exc - Exco) < 1e-11 * abs(Exco) and abs(Nel - Nelo) < 1e-11 * abs(Nelo)
    Hao, Hbo, Excpo, Nelpo, _ = ob.eval_Fxc_points(fx["ldft"], fx["mdft"], 101, 130, Pa, pts, Pb=Pb, threads=nth)
    sc = max(np.max(np.abs(Hao)), np.max(np.abs(Hbo)))
    assert np.max(np.abs(Ha - Hao)) < 1e-10 * sc and np.max(np.abs(Hb - Hbo)) < 1e-10 * sc
    assert np.max(np.abs(Hao - Hbo)) > 1e-6 * sc  # the two spins really differ
    assert abs(Excp - Excpo) < 1e-11 * abs(Excpo) and abs(Nelp - Nelpo) < 1e-11 * abs(Nelpo)
    # (iv) exchange: a sample of output blocks (TwoDBasis::exchange, basis.cpp:1532; the reference loops over (jang, kang))
    # density: the closed shell plus one orbital that mixes the three m blocks, so that the sigma-pi and pi-pi' blocks of K
    # do not vanish by symmetry (K_jk needs m_j - m_i = m_k - m_l with P_il != 0)
    u = np.zeros(N)
    for b in blocks:
        cols = np.where(np.max(np.abs(fx["C"][b, :]), axis=0) > 1e-8)[0]
        u += fx["C"][:, cols[0]]
    Ph = np.asfortranarray(0.5 * P + 0.3 * np.outer(u, u))
    K = basis.exchange(Ph)
    pairs = _sample_pairs(basis)
    Ko = ob.exchange_blocks(Ph, pairs)
    sk = np.max(np.abs(K))
    mags = [float(np.max(np.abs(Ko[np.ix_(_block_of(basis, j), _block_of(basis, k))])) / sk) for (j, k) in pairs]
    assert sum(1 for m in mags if m > 1e-9) >= 8, mags  # enough of the sampled blocks carry weight (high-l blocks are tiny)
    for (j, k) in pairs:
        rj, rk = _block_of(basis, j), _block_of(basis, k)
        blk, blko = K[np.ix_(rj, rk)], Ko[np.ix_(rj, rk)]
        assert np.max(np.abs(blk - blko)) < 1e-11 * sk, ((j, k), np.max(np.abs(blk - blko)) / sk)


def test_fullsize_stages_against_the_oracle(full):
    """Nbf = 4230 (BASELINE configs[3]): J, eigenvalues, an XC radial shard (restricted + polarised) and nine blocks of K
    compared directly with the CPU checker / LAPACK"""
    _check_coulomb_eig_xc_k(full, shard_rank=3)


def test_fullsize_half_inverse_and_eigensolve(full):
    S, X, H0, E, C, N = full["S"], full["X"], full["H0"], full["E"], full["C"], full["N"]
    assert np.max(np.abs(X.T @ S @ X - np.eye(N))) < 1e-9                      # main.cpp:475-479 run-time identity
    assert np.all(np.diff(E) >= 0.0)                                           # global sort of eig_gsym_sub
    SC = S @ C
    assert np.max(np.abs(C.T @ SC - np.eye(N))) < 1e-9                         # C^T S C = 1
    scale = np.max(np.abs(E))
    assert np.max(np.abs(H0 @ C - SC * E)) < 1e-9 * scale                      # F C = S C E, every column
    # block structure: an orbital lives in one symmetry block
    for b in full["blocks"]:
        other = np.setdiff1d(np.arange(N), b)
        cols = np.where(np.max(np.abs(C[b, :]), axis=0) > 1e-8)[0]
        assert len(cols) == len(b) and np.max(np.abs(C[np.ix_(other, cols)])) < 1e-10
    # the two lowest core-Hamiltonian levels of a homonuclear diatomic are the near-degenerate gerade / ungerade 1s
    # combinations, E ~ -Z^2/2 - Z/R to first order
    assert abs(E[0] - E[1]) < 5e-3 and abs(E[0] - (-24.5 - 7.0 / 2.068)) < 0.05


def test_fullsize_density_coulomb_exchange(full):
    hf, basis, S, C, N = full["hf"], full["basis"], full["S"], full["C"], full["N"]
    nocc = full["w"]["nocc"]
    P = hf.scf.form_density(C, nocc)
    assert abs(np.sum(P * S) - nocc) < 1e-9                                    # Tr P S = N_occ (main.cpp:794)
    assert np.max(np.abs(P @ S @ P - P)) < 1e-9                                # idempotence in the S metric
    P2 = hf.scf.form_density(np.asfortranarray(C[:, 3:]), 5)
    J1, J2 = basis.coulomb(P), basis.coulomb(P2)
    J12 = basis.coulomb(np.asfortranarray(P + 2.0 * P2))
    sc = np.max(np.abs(J12))
    assert np.max(np.abs(J12 - J1 - 2.0 * J2)) < 1e-12 * sc                    # linearity
    assert np.max(np.abs(J1 - J1.T)) < 1e-12 * sc
    assert np.sum(P * J1) > 0.0                                                # Coulomb self-energy
    assert abs(np.sum(P * J2) - np.sum(P2 * J1)) < 1e-11 * abs(np.sum(P * J2))  # (P|P2) = (P2|P)
    K1, K2 = basis.exchange(P), basis.exchange(P2)
    K12 = basis.exchange(np.asfortranarray(P + 2.0 * P2))
    sk = np.max(np.abs(K12))
    assert np.max(np.abs(K12 - K1 - 2.0 * K2)) < 1e-11 * sk
    assert np.max(np.abs(K1 - K1.T)) < 1e-11 * sk
    assert np.sum(P * K1) < 0.0                                                # exchange lowers the energy
    assert abs(np.sum(P * K2) - np.sum(P2 * K1)) < 1e-10 * abs(np.sum(P * K2))
    # one orbital: exchange cancels the Coulomb self-interaction exactly, K[p] p = -J[p] p on that orbital
    c0 = np.asfortranarray(C[:, :1])
    p0 = hf.scf.form_density(c0, 1)
    assert abs(np.sum(p0 * basis.coulomb(p0)) + np.sum(p0 * basis.exchange(p0))) < 1e-10 * np.sum(p0 * basis.coulomb(p0))


def test_fullsize_exchange_with_more_than_64_factors(full):
    """a density of 7 + 60 = 67 factors: the fast path factorises the residual matrix a second time; the result must be the
    sum of the two parts' exchange matrices (linearity: each part alone needs one factor group) and symmetric"""
    hf, basis, C, N = full["hf"], full["basis"], full["C"], full["N"]
    Pa = hf.scf.form_density(C, 7)
    Pb = hf.scf.form_density(np.asfortranarray(C[:, 40:100]), 60)
    K = basis.exchange(np.asfortranarray(Pa + 0.01 * Pb))
    Ka, Kb = basis.exchange(Pa), basis.exchange(np.asfortranarray(0.01 * Pb))
    sk = np.max(np.abs(K))
    assert np.max(np.abs(K - Ka - Kb)) < 1e-11 * sk
    assert np.max(np.abs(K - K.T)) < 1e-11 * sk


def test_fullsize_xc_quadrature(full):
    hf, basis, S, C, N = full["hf"], full["basis"], full["S"], full["C"], full["N"]
    grid = hf.DFTGrid(basis, full["ldft"], full["mdft"])
    nocc = full["w"]["nocc"]
    P = np.asfortranarray(2.0 * hf.scf.form_density(C, nocc))                  # restricted: total density
    H, Exc, Nel, _ = grid.eval_Fxc(101, 130, P)
    assert abs(Nel - 2.0 * nocc) < 1e-7 * nocc                                 # integrated density (main.cpp:860)
    assert np.max(np.abs(H - H.T)) < 1e-11 * np.max(np.abs(H))
    assert Exc < 0.0
    # H is the derivative of Exc: (Exc[P + h D] - Exc[P - h D]) / 2h = Tr H D for a density-like direction D
    D = np.asfortranarray(hf.scf.form_density(np.asfortranarray(C[:, nocc - 1:nocc + 2]), 3))
    h = 1e-4
    Ep = grid.eval_Fxc(101, 130, np.asfortranarray(P + h * D))[1]
    Em = grid.eval_Fxc(101, 130, np.asfortranarray(P - h * D))[1]
    fd, an = (Ep - Em) / (2 * h), float(np.sum(H * D))
    assert abs(fd - an) < 1e-6 * abs(an), (fd, an)
    # spin-polarised evaluation with equal spin densities is the restricted one.  Exchange alone: to rounding (the
    # per-channel density screen acts on rho/2 in both); with correlation the polarised call raises a channel below the
    # threshold to it (as libxc does), which moves far-field elements by parts in 1e9.
    half = np.asfortranarray(0.5 * P)
    Hx = grid.eval_Fxc(101, 0, P)[0]
    Hxa, Hxb, _, _, _ = grid.eval_Fxc_pol(101, 0, half, half)
    assert np.max(np.abs(Hxa - Hx)) < 1e-13 * np.max(np.abs(Hx)) and np.max(np.abs(Hxa - Hxb)) < 1e-14
    Ha, Hb, Exc2, Nel2, _ = grid.eval_Fxc_pol(101, 130, half, half)
    assert abs(Exc2 - Exc) < 1e-12 * abs(Exc) and np.max(np.abs(Ha - H)) < 1e-7 * np.max(np.abs(H))
    # model-potential quadrature at full size: point nuclei reproduce the analytic nuclear attraction
    Vq = basis.model_potential((0, 7), (0, 7))
    Vn = basis.nuclear()
    assert np.max(np.abs(Vq - Vn)) < 1e-8 * np.max(np.abs(Vn))


def test_fullsize_xc_without_density_threshold(full):
    """--dftthr 0: the far field of the 40 a.u. grid holds densities below 1e-30; every functional stays finite there
    and the matrices do not move beyond what the screened tail carries."""
    hf, basis, C = full["hf"], full["basis"], full["C"]
    grid = hf.DFTGrid(basis, full["ldft"], full["mdft"])
    P = np.asfortranarray(2.0 * hf.scf.form_density(C, full["w"]["nocc"]))
    half = np.asfortranarray(0.5 * P)
    for x, c in ((101, 130), (1, 7), (202, 231)):
        H0, E0, N0, _ = grid.eval_Fxc(x, c, P, 1e-12)
        H, E, N, _ = grid.eval_Fxc(x, c, P, 0.0)
        assert np.all(np.isfinite(H)) and np.isfinite(E), (x, c)
        assert abs(E - E0) < 1e-9 * abs(E0) and abs(N - N0) < 1e-9 * N0
        Ha, Hb, Ep, _, _ = grid.eval_Fxc_pol(x, c, half, half, 0.0)
        assert np.all(np.isfinite(Ha)) and np.all(np.isfinite(Hb)) and abs(Ep - E) < 1e-11 * abs(E), (x, c)


# ---- BASELINE config 5 sizing: LiF, lmmax = [29, 29], 5 x 15 -> Nbf = 6102, symmetry blocks 2100 / 2001 / 2001 ----------
@pytest.fixture(scope="module")
def lif(native_libs):
    import helfem_amd as hf
    if hf.device_count() < 1:
        pytest.fail("no HIP device: the full-size checks need a real MI355X")
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["lif_pbe_nbf6102"])
    basis, bval, lval, mval, ldft, mdft = bench.build_basis(hf, w)
    basis.compute_tei(True, device=True)
    basis.upload(ldft, mdft)
    N = basis.Nbf()
    assert N == 6102
    S = basis.overlap()
    H0 = np.asfortranarray(basis.kinetic() + basis.nuclear())
    blocks = basis.get_sym_idx(1)
    assert sorted(len(b) for b in blocks) == [2001, 2001, 2100]
    X = hf.scf.form_Sinvh(S, False, blocks)
    E, C = hf.scf.eig_gsym_sub(H0, X, blocks)
    return dict(hf=hf, basis=basis, ldft=ldft, mdft=mdft, N=N, S=S, H0=H0, X=X, blocks=blocks, E=E, C=C, w=w, bval=bval, lval=lval,
                mval=mval)


def test_lif_nbf6102_eigensolve(lif):
    """blocks of 2100 / 2001 / 2001: the regime where the symmetric sweep of the tridiagonalisation covers most columns"""
    S, X, H0, E, C, N = lif["S"], lif["X"], lif["H0"], lif["E"], lif["C"], lif["N"]
    assert np.max(np.abs(X.T @ S @ X - np.eye(N))) < 1e-9
    assert np.all(np.diff(E) >= 0.0)
    SC = S @ C
    assert np.max(np.abs(C.T @ SC - np.eye(N))) < 1e-9
    assert np.max(np.abs(H0 @ C - SC * E)) < 1e-9 * np.max(np.abs(E))
    for b in lif["blocks"]:
        other = np.setdiff1d(np.arange(N), b)
        cols = np.where(np.max(np.abs(C[b, :]), axis=0) > 1e-8)[0]
        assert len(cols) == len(b) and np.max(np.abs(C[np.ix_(other, cols)])) < 1e-10
    # heteronuclear core Hamiltonian: the lowest level is the fluorine 1s, E ~ -Z^2/2 - Z'/R; the next ones are the
    # fluorine n = 2 levels (-Z^2/8 - Z'/R, split by the lithium field), the lithium 1s (-4.5 - 9/R = -7.5) lies above them
    R = lif["w"]["Rbond"]
    assert abs(E[0] - (-40.5 - 3.0 / R)) < 0.1 and abs(E[1] - (-81.0 / 8.0 - 3.0 / R)) < 0.3
    assert np.min(np.abs(E[:8] - (-4.5 - 9.0 / R))) < 0.1


def test_lif_nbf6102_stages_against_the_oracle(lif):
    """Nbf = 6102 (BASELINE configs[4] sizing): the same direct comparisons"""
    _check_coulomb_eig_xc_k(lif, shard_rank=11)


def test_lif_nbf6102_coulomb_exchange_xc(lif):
    hf, basis, S, C, N = lif["hf"], lif["basis"], lif["S"], lif["C"], lif["N"]
    nocc = lif["w"]["nocc"]
    P = hf.scf.form_density(C, nocc)
    assert abs(np.sum(P * S) - nocc) < 1e-9
    P2 = hf.scf.form_density(np.asfortranarray(C[:, 2:]), 4)
    J1, J2 = basis.coulomb(P), basis.coulomb(P2)
    J12 = basis.coulomb(np.asfortranarray(P + 2.0 * P2))
    sc = np.max(np.abs(J12))
    assert np.max(np.abs(J12 - J1 - 2.0 * J2)) < 1e-12 * sc and np.max(np.abs(J1 - J1.T)) < 1e-12 * sc
    assert abs(np.sum(P * J2) - np.sum(P2 * J1)) < 1e-11 * abs(np.sum(P * J2))
    K1, K2 = basis.exchange(P), basis.exchange(P2)
    K12 = basis.exchange(np.asfortranarray(P + 2.0 * P2))
    sk = np.max(np.abs(K12))
    assert np.max(np.abs(K12 - K1 - 2.0 * K2)) < 1e-11 * sk and np.max(np.abs(K1 - K1.T)) < 1e-11 * sk
    assert np.sum(P * K1) < 0.0
    c0 = np.asfortranarray(C[:, :1])
    p0 = hf.scf.form_density(c0, 1)
    assert abs(np.sum(p0 * basis.coulomb(p0)) + np.sum(p0 * basis.exchange(p0))) < 1e-10 * np.sum(p0 * basis.coulomb(p0))
    grid = hf.DFTGrid(basis, lif["ldft"], lif["mdft"])
    Pt = np.asfortranarray(2.0 * P)
    H, Exc, Nel, _ = grid.eval_Fxc(101, 130, Pt)
    assert abs(Nel - 2.0 * nocc) < 1e-7 * nocc and Exc < 0.0
    assert np.max(np.abs(H - H.T)) < 1e-11 * np.max(np.abs(H))
    D = np.asfortranarray(hf.scf.form_density(np.asfortranarray(C[:, nocc - 1:nocc + 2]), 3))
    h = 1e-4
    Ep = grid.eval_Fxc(101, 130, np.asfortranarray(Pt + h * D))[1]
    Em = grid.eval_Fxc(101, 130, np.asfortranarray(Pt - h * D))[1]
    fd, an = (Ep - Em) / (2 * h), float(np.sum(H * D))
    assert abs(fd - an) < 1e-6 * abs(an), (fd, an)


def test_symmetry0_single_4230_eigenproblem(full):
    """--symmetry 0 (scf::eig_gsym, scf_helpers.cpp:131): ONE unsymmetrised 4230-dimensional problem instead of the three m
    blocks; same levels as the blocked solve, and the residual / orthonormality of every column"""
    hf, S, H0, N = full["hf"], full["S"], full["H0"], full["N"]
    X0 = hf.scf.form_Sinvh(S, False, [np.arange(N)])
    assert np.max(np.abs(X0.T @ S @ X0 - np.eye(N))) < 1e-9
    E0, C0 = hf.scf.eig_gsym(H0, X0)
    assert np.all(np.diff(E0) >= 0.0)
    SC = S @ C0
    assert np.max(np.abs(C0.T @ SC - np.eye(N))) < 1e-9
    assert np.max(np.abs(H0 @ C0 - SC * E0)) < 1e-9 * np.max(np.abs(E0))
    assert np.max(np.abs(E0 - full["E"])) < 1e-9 * np.max(np.abs(E0))  # the m-blocked solve finds the same spectrum


def test_pair_eigensolve_at_full_size(full):
    """both spins' blocks (2 x 1380/1470/1380) in one batch: each spin's result satisfies its own eigen-equations, and the
    alpha result equals the single-matrix solve"""
    hf, S, H0, X, N, blocks = full["hf"], full["S"], full["H0"], full["X"], full["N"], full["blocks"]
    rng = np.random.RandomState(2)
    D = np.zeros((N, N))
    for b in blocks:  # a perturbation that keeps the block structure
        v = rng.standard_normal(len(b))
        D[np.ix_(b, b)] = 0.05 * np.outer(v, v)
    Fb = H0 + S @ D @ S
    Ea, Ca, Eb, Cb = hf.scf.eig_gsym_sub_pair(H0, Fb, X, blocks)
    assert np.max(np.abs(Ea - full["E"])) < 1e-10 * np.max(np.abs(Ea))
    for F, E, C in ((H0, Ea, Ca), (Fb, Eb, Cb)):
        SC = S @ C
        assert np.all(np.diff(E) >= 0.0)
        assert np.max(np.abs(C.T @ SC - np.eye(N))) < 1e-9
        assert np.max(np.abs(F @ C - SC * E)) < 1e-9 * np.max(np.abs(E))
    assert np.max(np.abs(Eb - Ea)) > 1e-6  # the second matrix really is another problem


def test_eigensolver_beyond_one_cu_lds():
    """n = 5200: the top merge of the divide & conquer stage no longer fits its work arrays into one CU's LDS (they move to
    a global buffer) and the fused tridiagonalisation runs with 41 slabs per dimension; residual and orthonormality"""
    import helfem_amd as hf
    n = 5200
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n))
    A = np.asfortranarray(A + A.T)
    E, C = hf.scf.eig_sym(A)
    assert np.all(np.diff(E) >= 0.0)
    scale = np.max(np.abs(E))
    assert np.max(np.abs(A @ C - C * E)) < 1e-11 * scale
    assert np.max(np.abs(C.T @ C - np.eye(n))) < 1e-11
    assert abs(np.sum(E) - np.trace(A)) < 1e-9 * scale * np.sqrt(n)


@pytest.mark.parametrize("kw,ref,tol", [
    (dict(Z=86, lmax=3, mmax=3, nelem=10, nnodes=15, method="lda_x-lda_c_vwn", symmetry=2), -21861.346869, 5e-6),  # NIST LDA reference data
    (dict(Z=54, lmax=2, mmax=2, nelem=10, nnodes=15, method="HF", symmetry=2), -7232.138364, 5e-6),                 # numerical HF limit
])
def test_heavy_atoms_against_reference_data(kw, ref, tol):
    """radon / LDA against the NIST atomic reference data and xenon / HF against the numerical Hartree-Fock limit: the device
    path alone (the CPU oracle would need minutes here); f shells, 16 (l, m) symmetry blocks, 139 radial functions"""
    import helfem_amd as hf
    r = hf.scf_atomic(convthr=1e-8, maxit=60, **kw)
    assert r["converged"] and abs(r["Etot"] - ref) < tol, (r["Etot"], ref)
