import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import helfem_amd as hf
import bench
w=dict(bench.WORKLOADS[sys.argv[1]])
basis,bval,lval,mval,ldft,mdft=bench.build_basis(hf,w)
t=time.time(); basis.compute_tei(True); print("tei",time.time()-t)
basis.upload(0,0)
N=basis.Nbf()
import common
P=common.random_density(N,7,seed=3,blocks=basis.get_sym_idx(1))
ctx=basis.ctx
for i in range(2):
    t=time.time(); K=basis.exchange(P); print("exchange host-to-host s",time.time()-t)
ctx.profile(True); ctx.profile_reset()
K=basis.exchange(P)
print("exchange dev ms", ctx.profile_get("exchange"))
t=time.time(); J=basis.coulomb(P); print("coulomb s",time.time()-t)
# a density of rank 21 + 50 = 71 > 64 factors: two factor groups on the fast path (round 1: general kernels, 1.4 s)
if len(sys.argv) > 2 and sys.argv[2] == "groups":
    P2 = P + 0.01 * common.random_density(N, 50, seed=9)
    K2 = basis.exchange(P2)
    ctx.profile_reset()
    K2 = basis.exchange(P2)
    print("exchange dev ms, 71 factors", ctx.profile_get("exchange"))
    # linearity as the size-independent check
    K3 = basis.exchange(0.01 * common.random_density(N, 50, seed=9))
    print("linearity", np.max(np.abs(K2 - K - K3)) / np.max(np.abs(K2)))
