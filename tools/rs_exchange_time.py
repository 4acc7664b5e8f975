"""Times exchange / rs_exchange (Yukawa, erfc) of the atomic program at BASELINE config 2's basis (Ar, 20 x 15 nodes,
lmax = mmax = 1, Nbf = 1116) for a 9-orbital density: python tools/rs_exchange_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helfem_amd as hf  # noqa: E402
import common  # noqa: E402

gb, _ = common.make_atomic_bases(18, 1, 1, 20, 15, oracle=False)
t = time.time()
gb.compute_tei(True)
print("compute_tei %.2f s" % (time.time() - t))
N = gb.Nbf()
P = common.random_density(N, 3, seed=1, blocks=gb.get_sym_idx(1))
for kind, omega in (("coulomb", 0.0), ("yukawa", 0.4), ("erfc", 0.4)):
    t = time.time()
    if kind == "yukawa":
        gb.compute_yukawa(omega)
    elif kind == "erfc":
        gb.compute_erfc(omega)
    t_tab = time.time() - t
    fn = gb.exchange if kind == "coulomb" else gb.rs_exchange
    fn(P)
    gb.ctx.synchronize()
    t = time.time()
    for _ in range(5):
        K = fn(P)
    gb.ctx.synchronize()
    print("%-8s tables %.2f s, K build %.2f ms (host-pointer call, Nbf = %d)" % (kind, t_tab, (time.time() - t) / 5 * 1e3, N))
