"""bench.py's host-side helpers (no GPU): the std::mt19937_64 restatement behind `--density seeded`, the algorithmic
work behind `stage_rooflines`, the cgroup-aware core count."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_mt19937_64_known_answers():
    """C++11 [rand.predef]: the 10000th consecutive invocation of a default-constructed std::mt19937_64 (seed 5489)
    produces 9981545732273789042; the first output of that seed is 14514284786278117030"""
    import bench
    g = bench.MT19937_64(5489)
    first = g.next()
    assert first == 14514284786278117030
    v = first
    for _ in range(9999):
        v = g.next()
    assert v == 9981545732273789042


def test_stage_rooflines_algorithmic_work():
    """Coulomb bytes 2 (Nd^2 + 2 N_LM R^2) 8 B + 4 N_lm E p^4 8 B; eigensolve products 2 n^3 + lower tiles + 2 n^3"""
    import bench

    class Stub(object):
        def lm_map(self):
            return [(0, 0), (1, 0), (1, 1), (2, 1)]  # N_lm = 4, N_LM = 2 + 2 * 2 = 6

        def Nrad(self):
            return 10

        def Nang(self):
            return 3

    fams = {"coulomb": {"ms_per_step": 1.0}, "eig_products": {"ms_per_step": 2.0}}
    out = bench.stage_rooflines(Stub(), {"nelem": 2, "nnodes": 3}, [256, 100], fams)
    c = [o for o in out if o["bound"] == "hbm"][0]
    assert c["algorithmic_bytes"] == 2.0 * (30 * 30 + 2.0 * 6 * 100) * 8 + 4.0 * 4 * 2 * 81 * 8
    assert abs(c["achieved"] - c["algorithmic_bytes"] / 1e-3 / 1e9) < 1e-9 and abs(c["frac"] - c["achieved"] / 8000.0) < 1e-15
    g = [o for o in out if o["bound"] == "mfma"][0]
    # n = 256: two full tile rows, lower tiles 3 x 128^2; n = 100: one tile of 100 x 100
    fl = (2.0 * 256 ** 3 + 2.0 * 256 * 3 * 128 * 128 + 2.0 * 256 ** 3) + (2.0 * 100 ** 3 + 2.0 * 100 * 100 * 100 + 2.0 * 100 ** 3)
    assert g["flops"] == fl and abs(g["achieved"] - fl / 2e-3 / 1e12) < 1e-12


def test_usable_cores_is_positive_and_bounded():
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
