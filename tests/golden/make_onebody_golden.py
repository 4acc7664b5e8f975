"""Golden one-electron matrices of the diatomic basis (TwoDBasis::overlap / kinetic / nuclear,
/root/reference/src/diatomic/basis.cpp:677-817), computed by the NumPy restatement oracle/diatomic_onebody.py, which shares
no code with the product.  Heteronuclear, sigma + pi shells, two elements (so that the cos(theta) coupling of V, the shared
boundary function and the dropped last primitive are all exercised).

    python tests/golden/make_onebody_golden.py        ->  tests/golden/diatomic_onebody.npz (inputs and expected matrices)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import diatomic_onebody as ob  # noqa: E402
import diatomic_tei as dt  # noqa: E402

CASE = dict(Z1=3, Z2=9, Rbond=2.955, lmmax=[2, 1], nelem=2, nnodes=5, nquad=25, Rmax=40.0, zexp=1.0)


def main():
    c = CASE
    Rh = 0.5 * c["Rbond"]
    bval = dt.get_grid_exp(float(np.arccosh(c["Rmax"] / Rh)), c["nelem"], c["zexp"])
    lval, mval = dt.lm_to_l_m(c["lmmax"])
    S, T, V = ob.one_electron(c["Z1"], c["Z2"], Rh, bval, c["nnodes"], c["nquad"], lval, mval)
    out = dict(S=S, T=T, V=V, bval=bval, lval=np.array(lval), mval=np.array(mval))
    for k, v in c.items():
        out["case/" + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "diatomic_onebody.npz"), **out)
    print("wrote diatomic_onebody.npz:", S.shape)


if __name__ == "__main__":
    main()
