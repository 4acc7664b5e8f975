"""One-process-per-GPU plumbing for the sharded SCF iteration (torch.distributed over RCCL).

The hot path shards naturally:
  * Coulomb: (L,|M|) channels are independent (basis.cpp:1414)      -> rank r owns channels ilm % nranks == r
  * XC: radial quadrature points are independent (dftgrid.cpp:779)  -> rank r owns points Q % nranks == r
  * eigensolve: symmetry blocks are independent (scf_helpers.cpp:148) -> rank r owns blocks ib % nranks == r
  * exact exchange: output shells are independent (basis.cpp:1578)   -> rank r owns shells jang % nranks == r
Each rank produces partial Fock contributions in a zero-padded buffer; one sum all-reduce of the *compact*
(block-banded) J + XC buffer, and one of the dense exchange matrix of hybrid runs, completes them on every rank.
DeviceSCFStep(fock_shard="auto") does not shard J + XC at all: after sum factorisation that build (1-3 ms) is cheaper than
the all-reduce of its buffer, so every rank builds it whole; the exchange build and the eigensolve's blocks stay sharded.  The eigenvector blocks have ONE owner each, so they are not
summed: every block slot is broadcast from its owner (all broadcasts in flight together) -- half the bytes of a sum
all-reduce of the zero-padded buffer and no additions; ranks beyond the number of blocks own nothing and only receive.
No other data-path collective exists.  The same functions run on CPU tensors with the gloo backend,
which is how the N>1 logic is tested without GPUs (tests/test_parallel_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    # HELFEM_DIST_FORCE=1: initialise the process group for a single rank too, so that the RCCL code path (group set-up,
    # all-reduce of the step's device buffers, barrier) can be exercised on a one-GPU box
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # HELFEM_DIST_BACKEND=gloo lets several ranks rehearse the N>1 path on ONE GPU (RCCL refuses two ranks
            # on the same device); the default on a GPU node is RCCL ("nccl")
            backend = os.environ.get("HELFEM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def forced():
    return os.environ.get("HELFEM_DIST_FORCE") == "1"


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def owner(unit, nranks):
    """round-robin ownership used by the kernels (fock.hip, eig.hip)"""
    return unit % nranks


def owned_units(nunits, rank, nranks):
    return [u for u in range(nunits) if owner(u, nranks) == rank]


def allreduce_sum_(t):
    """in-place sum all-reduce (no-op for a single process)"""
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_block_slots_(buf, nblk, nranks=None):
    """buf: nblk equal slots, slot ib complete on rank owner(ib) -> complete on every rank.  One broadcast per block, all
    issued before the first wait (different roots: they travel concurrently over different xGMI links)."""
    if not _active():
        return buf
    world = dist.get_world_size() if nranks is None else nranks
    slot = buf.numel() // nblk
    work = [dist.broadcast(buf[ib * slot:(ib + 1) * slot], src=owner(ib, world), async_op=True) for ib in range(nblk)]
    for w in work:
        w.wait()
    return buf


def barrier():
    if _active():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
