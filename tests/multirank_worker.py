"""Worker of tests/test_gpu_multirank.py: N ranks (gloo) rehearse the sharded SCF step on ONE GPU, without host
synchronisation between the kernels and the collectives, and write the eigenvalues of three consecutive steps."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import torch
    import helfem_amd as hf
    from helfem_amd import parallel
    import bench
    rank, local_rank, world = parallel.init(backend="gloo" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    w = bench.WORKLOADS["n2_pbe_small"]
    basis, bval, lval, mval, ldft, mdft = bench.build_basis(hf, w)
    basis.compute_tei(False)
    N = basis.Nbf()
    step = hf.DeviceSCFStep(basis, w["x"], w["c"], ldft, mdft, w["nocc"], symmetry=1, device=0, rank=rank, nranks=world)
    ctx = step.ctx
    S = basis.overlap()
    H0 = basis.kinetic() + basis.nuclear()
    Sinvh = hf.scf.form_Sinvh(S, False, step.blocks, ctx=ctx)
    step.set_matrices(H0, Sinvh)
    E0, C0 = hf.scf.eig_gsym_sub(H0, Sinvh, step.blocks, ctx=ctx)
    P0 = 2.0 * hf.scf.form_density(C0, w["nocc"], ctx=ctx)
    allred = parallel.allreduce_sum_ if world > 1 else None
    xblocks = parallel.broadcast_block_slots_ if world > 1 else None
    res = []
    for it in range(3):
        step.set_density(P0)
        step.step(allred, xblocks)
        step.P.mul_(2.0)  # a torch operation on the same stream, no synchronisation in between
        res.append(dict(E=step.E.cpu().numpy()[:12].tolist(), exc=float(step.scal[0].item()),
                        trPS=float((step.P.view(N, N).cpu().numpy() * S).sum())))
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
