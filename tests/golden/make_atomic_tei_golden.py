"""Generates tests/golden/atomic_tei.npz: the radial two-electron tables of the reference's atomic program (compute_tei,
compute_yukawa, compute_erfc) from the independent NumPy / SciPy restatement oracle/atomic_tei.py.

  python tests/golden/make_atomic_tei_golden.py

Case: exponential grid (igrid 4, zexp 2) to Rmax = 8 with 3 elements, 6 Lobatto nodes, 30 quadrature points, L = 0 .. 4,
Yukawa lambda = 0.4, erfc mu = 0.4."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import atomic_tei as at  # noqa: E402
from diatomic_tei import get_grid_exp  # noqa: E402

CASE = dict(Rmax=8.0, nelem=3, zexp=2.0, nnodes=6, nquad=30, NL=5, lam=0.4, mu=0.4)

if __name__ == "__main__":
    bval = get_grid_exp(CASE["Rmax"], CASE["nelem"], CASE["zexp"])
    setup = at.RadialSetup(bval, CASE["nnodes"], CASE["nquad"])
    T = at.compute_tables(setup, CASE["NL"], CASE["lam"], CASE["mu"])
    out = {"bval": bval}
    for k, v in CASE.items():
        out["case_" + k] = np.array(v)
    for name, d in T.items():
        for key, m in d.items():
            out[name + "_" + "_".join(str(q) for q in key)] = m
    np.savez_compressed(os.path.join(HERE, "atomic_tei.npz"), **out)
    print("wrote", len(out), "arrays")
