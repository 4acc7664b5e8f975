"""FP64 GEMM tile kernel alone: hfg_gemm_dev on device-resident operands, timed with the library's HIP-event brackets.
   python tools/gemm_bench.py            (HELFEM_MFMA=16x16x4 for the other matrix instruction)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import helfem_amd as hf

ctx = hf.default_context()
L = hf.lib()
shapes = [(1400, 1400, 1400, 0, 0), (1400, 1400, 1400, 1, 0), (4230, 4230, 4230, 0, 0), (2816, 2816, 2816, 0, 0), (1400, 1400, 64, 0, 1),
          (225, 3000, 900, 0, 0), (6102, 6102, 6102, 0, 0), (8192, 8192, 8192, 0, 0)]
for (m, n, k, tA, tB) in shapes:
    A = torch.randn((k, m) if not tA else (m, k), dtype=torch.float64, device="cuda")  # column-major m x k == row-major k x m
    B = torch.randn((n, k) if not tB else (k, n), dtype=torch.float64, device="cuda")
    C = torch.zeros((n, m), dtype=torch.float64, device="cuda")
    lda = m if not tA else k
    ldb = k if not tB else n
    def run():
        rc = L.hfg_gemm_dev(ctx.h, tA, tB, ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(k), ctypes.c_void_p(A.data_ptr()),
                            ctypes.c_int64(lda), ctypes.c_void_p(B.data_ptr()), ctypes.c_int64(ldb), ctypes.c_void_p(C.data_ptr()),
                            ctypes.c_int64(m))
        assert rc == 0, L.hfg_last_error()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ctx.profile(True)
    ctx.profile_reset()
    reps = 10
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms, nl = ctx.profile_get("gemm")
    ctx.profile(False)
    # check against torch
    Aop = A.T if not tA else A
    Bop = B.T if not tB else B
    ref = (Aop @ Bop).T
    err = float((C - ref).abs().max() / ref.abs().max())
    print("m=%d n=%d k=%d tA=%d tB=%d: %.3f ms  %.1f TFLOP/s  relerr %.1e  [%s]" % (m, n, k, tA, tB, ms / reps, 2.0 * m * n * k / (ms / reps) * 1e-9,
                                                                              err, os.environ.get("HELFEM_MFMA", "4x4x4_4b")))
