"""Set-up times at the bench basis (N2, Nbf = 4230) without torch in the process (as the executables run): context,
first kernel launch (code object load), in-element tables on the device, first and second upload, one-electron matrices."""
import os, sys, time
os.environ["HELFEM_NO_TORCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t = time.time(); import helfem_amd as hf; hf.lib(); print("library load %.3f s" % (time.time() - t))
import bench
w = bench.WORKLOADS["n2_pbe_nbf4230"]
t = time.time(); ctx = hf.default_context(); print("context (HIP initialisation) %.3f s" % (time.time() - t))
A = np.asfortranarray(np.eye(64))
t = time.time(); hf.gemm(A, A, ctx=ctx) if hasattr(hf, "gemm") else hf.scf.eig_sym(A, ctx=ctx); print("first kernel launch %.3f s" % (time.time() - t))
t = time.time(); basis, bval, lval, mval, ldft, mdft = bench.build_basis(hf, w); print("basis %.3f s" % (time.time() - t))
for rep in range(2):
    t = time.time(); basis.compute_tei(False, device=True); t1 = time.time() - t
    t = time.time(); basis.upload(ldft, mdft); t2 = time.time() - t
    print("rep %d: compute_tei(device) %.3f s, upload %.3f s" % (rep, t1, t2))
t = time.time(); S = basis.overlap(); print("overlap %.3f s" % (time.time() - t))
t = time.time(); T = basis.kinetic(); V = basis.nuclear(); print("kinetic + nuclear %.3f s" % (time.time() - t))
blocks = basis.get_sym_idx(1)
t = time.time(); X = hf.scf.form_Sinvh(S, False, blocks, ctx=ctx); print("Sinvh (host pointers) %.3f s" % (time.time() - t))
