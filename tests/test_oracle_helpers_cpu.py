"""The checker's partial evaluations, which the full-size GPU tests rely on (tests/test_gpu_fullsize.py), against its
complete ones on a small basis: a list of output blocks of the exchange matrix (the reference builds K block by block,
/root/reference/src/diatomic/basis.cpp:1575-1579) and the XC matrix summed radial point by radial point
(/root/reference/src/diatomic/dftgrid.cpp:779-801, :822-848)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _block(gb, Nrad, j):
    off = 0
    for a in range(j):
        off += Nrad - (1 if gb.mval[a] != 0 else 0)
    return np.arange(off, off + Nrad - (1 if gb.mval[j] != 0 else 0))


def test_exchange_block_list_and_pointwise_xc(native_libs):
    import common
    gb, ob = common.make_bases(7, 7, 2.068, (3, 2), 2, 5)
    ob.compute_tei(True)
    N = ob.Nbf
    rng = np.random.RandomState(0)
    C = rng.uniform(-1, 1, (N, 3))
    P = np.asfortranarray(C @ C.T)
    K = ob.exchange(P)
    pairs = [(0, 0), (0, 1), (1, 5), (4, 6), (5, 5)]
    Kb = ob.exchange_blocks(P, pairs)
    mask = np.zeros((N, N), bool)
    for j, k in pairs:
        mask[np.ix_(_block(gb, ob.Nrad, j), _block(gb, ob.Nrad, k))] = True
    assert np.array_equal(Kb[mask], K[mask]) and np.max(np.abs(Kb[~mask])) == 0.0
    assert np.max(np.abs(K[mask])) > 0.0
    NQ = 2 * 25
    H, Exc, Nel, _ = ob.eval_Fxc(24, 13, 101, 130, P)
    H2, Exc2, Nel2, _ = ob.eval_Fxc_points(24, 13, 101, 130, P, range(NQ), threads=4)
    assert np.max(np.abs(H - H2)) <= 1e-14 * np.max(np.abs(H))
    assert abs(Exc - Exc2) < 1e-13 * abs(Exc) and abs(Nel - Nel2) < 1e-13 * abs(Nel)
    # a shard is the sum of its points
    pts = [q for q in range(NQ) if q % 5 == 2]
    H3, Exc3, _, _ = ob.eval_Fxc_shard(24, 13, 101, 130, P, 2, 5)
    H4, Exc4, _, _ = ob.eval_Fxc_points(24, 13, 101, 130, P, pts, threads=2)
    assert np.max(np.abs(H3 - H4)) <= 1e-14 * np.max(np.abs(H)) and abs(Exc3 - Exc4) < 1e-13 * abs(Exc)
    Pb = np.asfortranarray(C[:, :2] @ C[:, :2].T)
    Ha, Hb, E3, N3, _ = ob.eval_Fxc_pol(24, 13, 101, 130, P, Pb)
    Ha2, Hb2, E4, N4, _ = ob.eval_Fxc_points(24, 13, 101, 130, P, range(NQ), Pb=Pb, threads=4)
    assert np.max(np.abs(Ha - Ha2)) <= 1e-14 * np.max(np.abs(Ha)) and np.max(np.abs(Hb - Hb2)) <= 1e-14 * np.max(np.abs(Hb))
    assert abs(E3 - E4) < 1e-13 * abs(E3) and abs(N3 - N4) < 1e-13 * abs(N3)
