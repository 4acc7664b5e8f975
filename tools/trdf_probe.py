import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helfem_amd as hf
ctx = hf.default_context()
rng = np.random.RandomState(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
A = rng.uniform(-1, 1, (n, n)); A = A + A.T
hf.scf.eig_sym(A, ctx=ctx)
for _ in range(3):
    ms, k = ctx.measure_kernel("k_trdf")
print("C=%s DBG=%s: %.2f us per launch (%d launches)" % (os.environ.get("HELFEM_TRDF_C"), os.environ.get("HELFEM_TRDF_DBG"), 1e3 * ms / k, k))
