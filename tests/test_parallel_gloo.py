"""N>1 path on CPU: two and four gloo ranks, the same sharding rules and the same collectives as the multi-GPU run
(helfem_amd/parallel.py), with the CPU oracle standing in for the kernels.  Checks that summing the shards'
zero-padded partial results with ONE all-reduce reproduces the unsharded Fock matrix / energies, and that the
block-distributed eigensolve + owner broadcasts + global sort reproduces scf::eig_gsym_sub."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import common
    import oracle_lib as orc
    from helfem_amd import parallel
    r, _, w = parallel.init(backend="gloo")
    assert (r, w) == (rank, world)
    gb, ob = common.make_bases(7, 7, 2.068, (3, 2), 2, 5)
    ob.compute_tei(False)
    N = ob.Nbf
    blocks = gb.get_sym_idx(1)
    P = common.random_density(N, 2, seed=21, blocks=blocks)
    ldft, mdft = 24, 13
    # ---- Fock build: this rank's (L,|M|) channels of J and radial points of XC, then ONE all-reduce ----
    J = ob.coulomb_shard(P, rank, world)
    H, Exc, Nel, _ = ob.eval_Fxc_shard(ldft, mdft, 101, 130, P, rank, world)
    buf = torch.from_numpy(np.concatenate([(J + H).ravel(order="F"), [Exc, Nel]]))
    parallel.allreduce_sum_(buf)
    F_part = buf[:-2].numpy().reshape((N, N), order="F")
    # ---- hybrid step: the exchange build shards over OUTPUT shells (jang % world == rank, basis.cpp:1578); the partial
    # matrices have disjoint supports and sum to K ----
    ob.compute_tei(True)
    A = len(gb.lval)
    mine = [(j, k) for j in parallel.owned_units(A, rank, world) for k in range(A)]
    Kp = ob.exchange_blocks(0.5 * P, mine) if mine else np.zeros((N, N))
    kbuf = torch.from_numpy(np.asfortranarray(Kp).ravel(order="F").copy())
    parallel.allreduce_sum_(kbuf)
    K_sum = kbuf.numpy().reshape((N, N), order="F")
    # ---- eigensolve: block ib on rank ib % world, zero-padded slots, ONE all-reduce, global sort ----
    S = gb.overlap()
    H0 = gb.kinetic() + gb.nuclear()
    X = orc.form_Sinvh(S, False, blocks)
    F = np.zeros_like(H0)
    for b in blocks:
        F[np.ix_(b, b)] = (H0 + F_part)[np.ix_(b, b)]
    nmax = max(len(b) for b in blocks)
    slot = nmax * nmax + nmax
    bb = np.zeros(len(blocks) * slot)
    coff = np.cumsum([0] + [len(b) for b in blocks])
    for ib in parallel.owned_units(len(blocks), rank, world):
        idx = blocks[ib]
        n = len(idx)
        Eb, Cb = orc.eig_gsym(F[np.ix_(idx, idx)], X[np.ix_(idx, range(coff[ib], coff[ib] + n))])
        bb[ib * slot:ib * slot + n * n] = Cb.ravel(order="F")
        bb[ib * slot + nmax * nmax:ib * slot + nmax * nmax + n] = Eb
    tb = torch.from_numpy(bb)
    # every block has ONE owner: broadcast from the owners (ranks beyond the number of blocks own nothing and only receive)
    parallel.broadcast_block_slots_(tb, len(blocks))
    Eall = np.concatenate([bb[ib * slot + nmax * nmax:ib * slot + nmax * nmax + len(blocks[ib])] for ib in range(len(blocks))])
    order = np.argsort(Eall, kind="stable")
    C = np.zeros((N, N))
    for ib, idx in enumerate(blocks):
        n = len(idx)
        C[np.ix_(idx, range(coff[ib], coff[ib] + n))] = bb[ib * slot:ib * slot + n * n].reshape((n, n), order="F")
    C = C[:, order]
    E = Eall[order]
    t = parallel.max_over_ranks(float(rank))
    if rank == 0:
        np.savez(out, F_part=F_part, Exc=buf[-2].item(), Nel=buf[-1].item(), E=E, C=C, F=F, tmax=t, K_sum=K_sum)
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_matches_unsharded(tmp_path, native_libs, world):
    """world 4 with 3 symmetry blocks: rank 3 owns no block and no eigenvector slot (idle-rank logic of the exchange)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    res = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    import oracle_lib as orc
    gb, ob = common.make_bases(7, 7, 2.068, (3, 2), 2, 5)
    ob.compute_tei(False)
    blocks = gb.get_sym_idx(1)
    P = common.random_density(ob.Nbf, 2, seed=21, blocks=blocks)
    J = ob.coulomb(P)
    H, Exc, Nel, _ = ob.eval_Fxc(24, 13, 101, 130, P)
    assert np.max(np.abs(res["F_part"] - (J + H))) < 1e-12 * np.max(np.abs(J + H))
    assert abs(res["Exc"] - Exc) < 1e-12 * max(1.0, abs(Exc)) and abs(res["Nel"] - Nel) < 1e-12 * max(1.0, abs(Nel))
    assert res["tmax"] == world - 1.0  # MAX over ranks of the rank id
    ob.compute_tei(True)
    K = ob.exchange(0.5 * P)
    assert np.max(np.abs(res["K_sum"] - K)) <= 1e-14 * np.max(np.abs(K))  # the shards of the exchange build sum to K
    S = gb.overlap()
    X = orc.form_Sinvh(S, False, blocks)
    Eo, Co = orc.eig_gsym_sub(res["F"], X, blocks)
    assert np.max(np.abs(res["E"] - Eo)) < 1e-11 * max(1.0, np.max(np.abs(Eo)))
    C = res["C"]
    assert np.max(np.abs(C.T @ S @ C - np.eye(len(Eo)))) < 1e-9
    assert np.max(np.abs(res["F"] @ C - S @ C * res["E"])) < 1e-8 * max(1.0, np.max(np.abs(Eo)))


def test_ownership_rules():
    sys.path.insert(0, ROOT)
    from helfem_amd import parallel
    for n in (1, 2, 3, 8):
        units = list(range(11))
        got = sorted(u for r in range(n) for u in parallel.owned_units(11, r, n))
        assert got == units
