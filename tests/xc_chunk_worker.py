"""Worker of tests/test_gpu_parity.py::test_xc_kernels_with_chunked_angular_tables: run with HELFEM_XC_LDS_LIMIT set to a few
kilobytes, so that the XC kernels take their chunked paths (theta points / rows through LDS in pieces) on a small basis;
compares restricted, spin-polarised and meta-GGA Fock matrices and the model potential with the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import helfem_amd as hf
    import common
    assert os.environ.get("HELFEM_XC_LDS_LIMIT"), "run through the test"
    gb, ob = common.make_bases(3, 9, 2.955, (4, 3, 2), 2, 5)
    gb.compute_tei(False)
    ob.compute_tei(False)
    ldft, mdft = 4 * 4 + 12, 4 * 3 + 5
    gb.upload(ldft, mdft)
    N = gb.Nbf()
    blocks = gb.get_sym_idx(1)
    Pa = common.random_density(N, 3, seed=5, blocks=blocks)
    Pb = common.random_density(N, 2, seed=6, blocks=blocks)
    grid = hf.DFTGrid(gb, ldft, mdft)
    worst = 0.0
    for xf, cf in ((101, 130), (1, 7), (202, 231)):
        H, Exc, Nel, _ = grid.eval_Fxc(xf, cf, Pa + Pb)
        Ho, Exco, Nelo, _ = ob.eval_Fxc(ldft, mdft, xf, cf, Pa + Pb)
        worst = max(worst, common.relerr(H, Ho), abs(Exc - Exco) / abs(Exco))
        Ha, Hb, Excp, _, _ = grid.eval_Fxc_pol(xf, cf, Pa, Pb)
        Hao, Hbo, Excpo, _, _ = ob.eval_Fxc_pol(ldft, mdft, xf, cf, Pa, Pb)
        worst = max(worst, common.relerr(Ha, Hao), common.relerr(Hb, Hbo), abs(Excp - Excpo) / abs(Excpo))
    print("worst relative deviation %.3e" % worst)
    assert worst < 1e-9, worst


if __name__ == "__main__":
    main()
