import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import helfem_amd as hf
from helfem_amd import parallel
import bench
rank, local_rank, world = parallel.init()
w = bench.WORKLOADS[os.environ.get("WL", "n2_pbe_small")]
basis, bval, lval, mval, ldft, mdft = bench.build_basis(hf, w)
basis.compute_tei(False)
N = basis.Nbf()
step = hf.DeviceSCFStep(basis, w["x"], w["c"], ldft, mdft, w["nocc"], symmetry=1, device=0, rank=rank, nranks=world)
ctx = step.ctx
S = basis.overlap(); H0 = basis.kinetic() + basis.nuclear()
blocks = step.blocks
Sinvh = hf.scf.form_Sinvh(S, False, blocks, ctx=ctx)
step.set_matrices(H0, Sinvh)
E0, C0 = hf.scf.eig_gsym_sub(H0, Sinvh, blocks, ctx=ctx)
P0 = 2.0 * hf.scf.form_density(C0, w["nocc"], ctx=ctx)
step.set_density(P0)
step.fock_partial(); parallel.allreduce_sum_(step.Fc); parallel.allreduce_sum_(step.scal); step.fock_finish()
step.eig_partial()
torch.cuda.synchronize()
nb = len(blocks); nmax = max(len(b) for b in blocks); slot = nmax * nmax + nmax
bb = step.blockbuf.view(nb, slot)
print(rank, "before", [float(bb[i].abs().sum()) for i in range(nb)], flush=True)
parallel.allreduce_sum_(step.blockbuf)
torch.cuda.synchronize()
print(rank, "after ", [float(bb[i].abs().sum()) for i in range(nb)], flush=True)
step.eig_finish(); torch.cuda.synchronize()
print(rank, "E[:5]", step.E[:5].cpu().numpy(), flush=True)
allred = parallel.allreduce_sum_ if world > 1 else None
for it in range(3):
    step.set_density(P0)
    step.step(allred)
    step.P.mul_(2.0)
    torch.cuda.synchronize()
    print(rank, "iter", it, "E[:3]", step.E[:3].cpu().numpy(), "sumE", float(step.E.sum()), "slots", [float(bb[i].abs().sum()) for i in range(nb)], flush=True)
