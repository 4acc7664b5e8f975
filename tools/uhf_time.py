"""Unrestricted iteration at the bench basis (triplet N2, PBE, Nbf = 4230): time of the eigensolve step of the last iteration with
both spins' blocks in one batch (default) and one spin after the other (HELFEM_EIG_PAIR=0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helfem_amd as hf

kw = dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[20, 20], nelem=5, nnodes=15, method="gga_x_pbe-gga_c_pbe", M=3, maxit=int(sys.argv[1]) if len(sys.argv) > 1 else 4,
          convthr=1e-12)
try:
    r = hf.scf_diatomic(**kw)
except RuntimeError as e:  # not converged in so few iterations is fine here
    print("note:", str(e)[:80])
    r = None
if r:
    print("HELFEM_EIG_PAIR=%s: eigensolve %.2f ms, J %.2f ms, XC %.2f ms, Etot %.10f after %d iterations" % (
        os.environ.get("HELFEM_EIG_PAIR", "1"), 1e3 * r["tdiag"], 1e3 * r["tJ"], 1e3 * r["tXC"], r["Etot"], r["iterations"]))
