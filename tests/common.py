"""Shared builders for the parity tests: the same basis in the product (GPU) and in the oracle (CPU)."""
import numpy as np

import helfem_amd as hf
import oracle_lib as orc


def make_bases(Z1=1, Z2=1, Rbond=1.4, lmmax=(4,), nelem=2, nnodes=6, nquad=0, Rmax=40.0, igrid=4, zexp=1.0, lpad=10,
               product=True, oracle=True):
    lval, mval = hf.lm_to_l_m(list(lmmax))
    Rhalf = 0.5 * Rbond
    mumax = np.arccosh(Rmax / Rhalf)
    bval = hf.get_grid(float(np.log(Rmax / Rhalf + np.sqrt((Rmax / Rhalf) ** 2 - 1.0))), nelem, igrid, zexp)
    assert abs(bval[-1] - mumax) < 1e-12
    if nquad == 0:
        nquad = 5 * nnodes
    gb = hf.TwoDBasis(Z1, Z2, Rhalf, nnodes, nquad, bval, lval, mval, lpad) if product else None
    ob = orc.OracleBasis(Z1, Z2, Rhalf, nnodes, nquad, bval, lval, mval, lpad) if oracle else None
    return gb, ob


def make_atomic_bases(Z=2, lmax=1, mmax=1, nelem=3, nnodes=6, nquad=0, Rmax=40.0, igrid=4, zexp=2.0, product=True,
                      oracle=True):
    """the basis of the atomic program (src/atomic/main.cpp:245-273): r grid on [0,Rmax], shells from lmax/mmax"""
    lval, mval = hf.angular_basis(lmax, mmax)
    bval = hf.get_grid(Rmax, nelem, igrid, zexp)
    if nquad == 0:
        nquad = 5 * nnodes
    gb = hf.AtomicTwoDBasis(Z, nnodes, nquad, bval, lval, mval) if product else None
    ob = orc.OracleAtomicBasis(Z, nnodes, nquad, bval, lval, mval) if oracle else None
    return gb, ob


def random_density(N, nocc=3, seed=1, blocks=None):
    """P = C C^T with seeded random orthonormal-ish C; block-diagonal over `blocks` if given"""
    rng = np.random.RandomState(seed)
    if blocks is None:
        C = rng.uniform(-1, 1, size=(N, nocc))
    else:
        C = np.zeros((N, nocc * len(blocks)))
        for ib, idx in enumerate(blocks):
            C[np.ix_(idx, range(ib * nocc, (ib + 1) * nocc))] = rng.uniform(-1, 1, size=(len(idx), nocc))
    return np.asfortranarray(C @ C.T)


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
