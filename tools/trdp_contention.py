"""How much does the persistent tridiagonalisation lose when another stream keeps the rest of the chip busy?
(decides whether work that needs only finished reflectors may run beside the later phases)
  python tools/trdp_contention.py [matmul order]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import helfem_amd as hf  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sizes = [1380, 1470, 1380]
N = sum(sizes)
rng = np.random.RandomState(7)
F = np.zeros((N, N), order="F")
blocks, off = [], 0
for n in sizes:
    B = rng.standard_normal((n, n))
    F[off:off + n, off:off + n] = B + B.T
    blocks.append(np.arange(off, off + n))
    off += n
X = np.asfortranarray(np.eye(N))
ctx = hf.default_context()
hf.scf.eig_gsym_sub(F, X, blocks)


def solve(reps=3):
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        hf.scf.eig_gsym_sub(F, X, blocks)
    ms, _ = ctx.profile_get("eig_tridiag")
    ctx.profile(False)
    return ms / reps


print("alone: eig_tridiag %.3f ms" % solve())
side = torch.cuda.Stream()
A = torch.randn(m, m, dtype=torch.float64, device="cuda")
stop = False
count = [0]


def load():
    with torch.cuda.stream(side):
        while not stop:
            for _ in range(20):
                torch.mm(A, A)
                count[0] += 1
            side.synchronize()


t = threading.Thread(target=load)
t.start()
time.sleep(0.2)
c0, t0 = count[0], time.time()
r = solve()
dt = time.time() - t0
done = count[0] - c0
stop = True
t.join()
print("beside fp64 matmuls of order %d on another stream (%.1f TFLOP/s sustained there): eig_tridiag %.3f ms" % (m, done * 2.0 * m ** 3 / dt * 1e-12, r))
