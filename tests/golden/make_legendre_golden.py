"""Golden P_L^M(xi), Q_L^M(xi) tables from the reference's own Fortran Legendre library, compiled from the
reference sources into oracle/_ref/libref_legendre.so by oracle/build_ref.sh (run in the build container).
The calls are the ones LegendreTable::compute makes (src/general/legendretable.cpp:74-75):
calc_Plm_arr / calc_Qlm_arr(array, Lpad, Lpad, xi), of which the (0..Lmax, 0..Mmax) corner is kept.
Output: tests/golden/legendre_reference.json (inputs and outputs only)."""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_legendre.so"))
dp = ctypes.POINTER(ctypes.c_double)
for f in (lib.calc_Plm_arr, lib.calc_Qlm_arr):
    f.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_double]


def table(fn, lpad, xi):
    a = np.zeros((lpad + 1, lpad + 1))
    os.chdir("/tmp")  # the Fortran library writes fort.9 into the cwd
    fn(a.ctypes.data_as(dp), lpad, lpad, xi)
    return a.T  # [L, M]


cases = []
Lmax, Mmax, lpad = 12, 3, 10
for mu in [1e-3, 0.05, 0.3, 0.5, 1.0, 2.4, 4.0]:
    xi = float(np.cosh(mu))
    P = table(lib.calc_Plm_arr, Lmax + lpad, xi)
    Q = table(lib.calc_Qlm_arr, Lmax + lpad, xi)
    Pl = [[P[L, M] if L >= M else 0.0 for M in range(Mmax + 1)] for L in range(Lmax + 1)]
    Ql = [[Q[L, M] if L >= M else 0.0 for M in range(Mmax + 1)] for L in range(Lmax + 1)]
    cases.append({"mu": mu, "xi": xi, "P": Pl, "Q": Ql})
json.dump({"source": "src/legendre/*.f90 via Legendre_Wrapper.f90:135,173 (lpad=10 as in main.cpp:99)",
           "Lmax": Lmax, "Mmax": Mmax, "lpad": lpad, "cases": cases},
          open(os.path.join(HERE, "legendre_reference.json"), "w"))
print("wrote", len(cases), "cases")
