"""Worker of tests/test_gpu_parity.py::test_exchange_rb_kernels_at_large_element_order: exact exchange against the oracle
at 15, 16 and 17 nodes per element -- the matrix-core RB kernel with full 16 x 16 tiles (p = 16) and with one padding row
(p = 15), the one-pair vector kernel the fast path falls back to beyond 16 nodes, and (HELFEM_EXL_RB = 1 / 4, read once per
process) the two vector kernels as checkers on the padded element tables; restricted to small angular bases so that the
oracle's A^4 loops stay in seconds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import helfem_amd as hf  # noqa: F401
    import common
    worst = 0.0
    for nnodes in (15, 16, 17):
        gb, ob = common.make_bases(3, 1, 3.015, (2, 1), 2, nnodes)
        gb.compute_tei(True)
        ob.compute_tei(True)
        gb.upload(0, 0)
        N = gb.Nbf()
        for P in (common.random_density(N, 4, seed=21, blocks=gb.get_sym_idx(1)),
                  common.random_density(N, 3, seed=22) - common.random_density(N, 2, seed=23)):
            worst = max(worst, common.relerr(gb.exchange(P), ob.exchange(P)))
    gb, ob = common.make_atomic_bases(10, 2, 1, 2, 15)
    gb.compute_tei(True)
    ob.compute_tei(True)
    gb.upload(0, 0)
    P = common.random_density(gb.Nbf(), 5, seed=24)
    worst = max(worst, common.relerr(gb.exchange(P), ob.exchange(P)))
    print("worst relative deviation %.3e" % worst)
    assert worst < 1e-11, worst


if __name__ == "__main__":
    main()
