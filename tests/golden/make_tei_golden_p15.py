"""High-order golden samples of the primitive two-electron tables (TwoDBasis::compute_tei, basis.cpp:1166-1302 with
quadrature::twoe_integral, quadrature.cpp:22-123) at the ELEMENT ORDER OF THE BENCH WORKLOAD: 15-node LIPs, 75-point
quadrature, the first two radial elements of the N2 grid (Rh = 1.034, 5 elements to mu_max), channels up to L = 40.
Computed by oracle/diatomic_tei.py (NumPy, no product code) with mpmath Legendre functions at 40 digits ("exact") and with
the reference's own Fortran library ("ref").  A full table is 225 x 225 per (k,l), channel and element: the fixture keeps
400 seeded entries of each plus the full disjoint 15 x 15 tables.

    python tests/golden/make_tei_golden_p15.py [exact|ref|both]     ->  tests/golden/diatomic_tei_p15.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import diatomic_tei as dt  # noqa: E402

CASE = dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[19, 19], nelem_grid=5, nelem=2, nnodes=15, nquad=75, Rmax=40.0, zexp=1.0, lpad=10)
CHANNELS = [(0, 0), (3, 1), (17, 2), (31, 0), (40, 2)]
NSAMPLE = 400


def sample_indices(n, seed):
    rng = np.random.RandomState(seed)
    return rng.randint(0, n, NSAMPLE), rng.randint(0, n, NSAMPLE)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    c = CASE
    Rh = 0.5 * c["Rbond"]
    bval = dt.get_grid_exp(float(np.arccosh(c["Rmax"] / Rh)), c["nelem_grid"], c["zexp"])[:c["nelem"] + 1]
    lval, mval = dt.lm_to_l_m(c["lmmax"])
    lm, Lmax, Mmax = dt.lm_map_of(lval, mval)
    path = os.path.join(HERE, "diatomic_tei_p15.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in (["exact", "ref"] if which == "both" else [which]):
        if name == "ref":
            leg = dt.reference_legendre_provider(os.path.join(ROOT, "oracle", "_ref", "libref_legendre.so"), Lmax, Mmax, c["lpad"])
        else:
            leg = dt.mpmath_legendre_provider(40)
        st = dt.Setup(bval, c["nnodes"], c["nquad"], leg)
        for (L, M) in CHANNELS:
            ilm = lm.index((L, M))
            for iel in range(st.nel):
                t0 = time.time()
                for tag, k in (("P0", ("P", 0)), ("P2", ("P", 2)), ("Q0", ("Q", 0)), ("Q2", ("Q", 2))):
                    out["%s/%s/%d/%d" % (name, tag, ilm, iel)] = st.disjoint(k[0], k[1], iel, L, M)
                for tag, (k, l) in (("00", (0, 0)), ("02", (0, 2)), ("20", (2, 0)), ("22", (2, 2))):
                    t = st.twoe_integral(k, l, iel, L, M)
                    ii, jj = sample_indices(t.shape[0], 1000 * ilm + 10 * iel + k + l // 2)
                    out["%s/tei%s/%d/%d" % (name, tag, ilm, iel)] = t[ii, jj]
                    out["%s/tei%s_norm/%d/%d" % (name, tag, ilm, iel)] = np.array(np.linalg.norm(t))
                print(name, (L, M), "element", iel, "%.1f s" % (time.time() - t0), flush=True)
    out["bval"] = bval
    out["lval"] = np.array(lval)
    out["mval"] = np.array(mval)
    out["lm_map"] = np.array(lm)
    out["channels"] = np.array(CHANNELS)
    out["nsample"] = np.array(NSAMPLE)
    for k, v in c.items():
        out["case/" + k] = np.array(v)
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
