import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helfem_amd as hf
kw = dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[12, 12], nelem=4, nnodes=12, method="gga_x_pbe-gga_c_pbe", convthr=1e-7, maxit=40)
hf.scf_diatomic(Z1=1, Z2=1, Rbond=1.4, lmmax=[2], nelem=2, nnodes=6, method="HF")  # warm up the runtime
t = time.time(); r = hf.scf_diatomic(**kw); dt = time.time() - t
print("driver=%s: Etot %.9f, %d iterations, %.2f s total, %.1f ms/iteration (last-iteration timers J %.4f XC %.4f diag %.4f)" % (
    os.environ.get("HELFEM_SCF", "device"), r["Etot"], r["iterations"], dt, 1e3 * dt / r["iterations"], r["tJ"], r["tXC"], r["tdiag"]))
