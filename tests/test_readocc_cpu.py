"""Forced occupations (--readocc, scf::enforce_occupations of the reference, src/general/scf_helpers.cpp:31-128) in the
checker's own SCF driver: the lithium 1s2 2p excited state is reached by occupying one alpha orbital of m = +1, lies above
the 1s2 2s ground state, is the same for m = +1 and m = -1, and inconsistent occupation tables are refused."""
import numpy as np
import pytest

KW = dict(Z=3, lmax=1, mmax=1, nelem=3, nnodes=8, method="HF", M=2, convthr=1e-8, maxit=60)


def test_forced_occupation_reaches_the_2p_state(native_libs):
    import oracle_lib as orc
    ground = orc.scf_atomic(**KW)
    p_plus = orc.scf_atomic(occs=[[1, 1, 0], [1, 0, 1], [0, 0, -1]], **KW)
    p_minus = orc.scf_atomic(occs=[[1, 1, 0], [0, 0, 1], [1, 0, -1]], **KW)
    forced_ground = orc.scf_atomic(occs=[[2, 1, 0], [0, 0, 1], [0, 0, -1]], **KW)
    assert ground["converged"] and p_plus["converged"] and p_minus["converged"] and forced_ground["converged"]
    # numerical HF: Li 2S -7.432727, 2P -7.365070 (Froese Fischer); this small basis is within 1e-3 of both
    assert abs(ground["Etot"] - (-7.432727)) < 2e-3
    assert abs(p_plus["Etot"] - (-7.365070)) < 2e-3
    assert abs(p_plus["Etot"] - p_minus["Etot"]) < 1e-9
    assert abs(forced_ground["Etot"] - ground["Etot"]) < 1e-9
    for k in ("Ekin", "Epot", "Ecoul", "Exx"):
        assert abs(p_plus[k] - p_minus[k]) < 1e-7, k


def test_inconsistent_occupations_are_refused(native_libs):
    import oracle_lib as orc
    with pytest.raises(RuntimeError, match="don't match wanted spin state"):
        orc.scf_atomic(occs=[[1, 1, 0], [0, 0, 1], [0, 0, -1]], **KW)  # two electrons for lithium
