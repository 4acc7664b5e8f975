"""Golden primitive-integral tables of the diatomic basis (TwoDBasis::compute_tei, src/diatomic/basis.cpp:1166-1302 with
quadrature::twoe_integral, src/diatomic/quadrature.cpp:22-123), computed by the NumPy restatement oracle/diatomic_tei.py
-- which shares no code with the product -- with P_L^M / Q_L^M values

  set "ref":   from the reference's own Fortran Legendre library (oracle/_ref/libref_legendre.so, built from
               /root/reference/src/legendre/*.f90 by oracle/build_ref.sh), called as LegendreTable::compute calls it;
  set "exact": from mpmath at 40 digits (the reference library's Q_L^M loses accuracy towards xi -> 1).

Run in the build container (the reference tree and oracle/_ref exist only there):
    python tests/golden/make_tei_golden.py
Output: tests/golden/diatomic_tei.npz -- inputs (basis descriptor) and expected tables only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import diatomic_tei as dt  # noqa: E402

# the case: two radial elements, four-node LIPs (so the last element has 3 primitives), sigma + pi shells
CASE = dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[2, 1], nelem=2, nnodes=4, nquad=20, Rmax=40.0, zexp=1.0, lpad=10)


def build(provider_name):
    c = CASE
    Rh = 0.5 * c["Rbond"]
    mumax = float(np.arccosh(c["Rmax"] / Rh))
    bval = dt.get_grid_exp(mumax, c["nelem"], c["zexp"])
    lval, mval = dt.lm_to_l_m(c["lmmax"])
    lm, Lmax, Mmax = dt.lm_map_of(lval, mval)
    if provider_name == "ref":
        leg = dt.reference_legendre_provider(os.path.join(ROOT, "oracle", "_ref", "libref_legendre.so"), Lmax, Mmax, c["lpad"])
    else:
        leg = dt.mpmath_legendre_provider(40)
    st = dt.Setup(bval, c["nnodes"], c["nquad"], leg)
    tabs = dt.compute_tei(st, lm, exchange=True)
    return bval, lval, mval, lm, tabs


def main():
    out = {}
    for name in ("ref", "exact"):
        bval, lval, mval, lm, tabs = build(name)
        for key, per_lm in tabs.items():
            for ilm, per_el in enumerate(per_lm):
                for iel, m in enumerate(per_el):
                    out["%s/%s/%d/%d" % (name, key, ilm, iel)] = np.asfortranarray(m)
        print(name, "done:", len(lm), "channels")
    out["bval"] = bval
    out["lval"] = np.array(lval)
    out["mval"] = np.array(mval)
    out["lm_map"] = np.array(lm)
    for k, v in CASE.items():
        out["case/" + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "diatomic_tei.npz"), **out)
    print("wrote", os.path.join(HERE, "diatomic_tei.npz"))


if __name__ == "__main__":
    main()
