"""End-to-end atomic SCF at BASELINE configs[1] (Ar, PBE, 20 radial elements x 15 nodes, lmax = mmax = 1: Nbf = 1116,
per-(l,m) symmetry blocks of 279) and configs[0] (He, LDA, 5 x 8) on the device-resident driver:
python tools/atomic_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helfem_amd as hf  # noqa: E402

hf.scf_atomic(Z=2, lmax=0, mmax=0, nelem=2, nnodes=6, method="lda_x-lda_c_vwn")  # warm up the runtime
for name, kw in (("He LDA 5x8", dict(Z=2, lmax=0, mmax=0, nelem=5, nnodes=8, method="lda_x-lda_c_vwn")),
                 ("Ar PBE 20x15", dict(Z=18, lmax=1, mmax=1, nelem=20, nnodes=15, method="gga_x_pbe-gga_c_pbe")),
                 ("Ar PBE0 20x15", dict(Z=18, lmax=1, mmax=1, nelem=20, nnodes=15, method="hyb_gga_xc_pbeh")),
                 ("Ar CAM-LDA0 20x15", dict(Z=18, lmax=1, mmax=1, nelem=20, nnodes=15, method="hyb_lda_xc_cam_lda0"))):
    for sym in (2, 1):
        t = time.time()
        r = hf.scf_atomic(symmetry=sym, convthr=1e-7, maxit=60, **kw)
        dt = time.time() - t
        print("%-18s symmetry %d: Etot %.9f, %2d iterations%s, %.3f s total = %.1f ms/iteration including setup "
              "(last iteration: J %.2f K %.2f XC %.2f eig %.2f ms)" % (
                  name, sym, r["Etot"], r["iterations"], "" if r["converged"] else " (NOT converged)", dt,
                  1e3 * dt / max(1, r["iterations"]), 1e3 * r["tJ"], 1e3 * r["tK"], 1e3 * r["tXC"], 1e3 * r["tdiag"]), flush=True)
