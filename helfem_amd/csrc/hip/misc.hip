// Small helpers around the eigensolver: S^{-1/2} per symmetry block (reference utils::invh,
// libhelfem/src/utils.cpp:160-183 and TwoDBasis::Sinvh, src/diatomic/basis.cpp:627-652) and
// scf::form_density (src/general/scf_helpers.cpp:22-29).
#include "common.h"

namespace hfg {

void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);
void eig_sym_dev(hfg_ctx *ctx, int n, const double *dA, double *dE, double *dC);

// Sn(i,j) = S(rows[i],rows[j]) / sqrt(S_ii S_jj)
__global__ void k_gather_normalized(const double *__restrict__ S, int N, const int64_t *__restrict__ rows, int n,
                                    double *__restrict__ Sn) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  size_t ri = rows[i], rj = rows[j];
  double di = S[ri * N + ri], dj = S[rj * N + rj];
  Sn[(size_t)j * n + i] = S[rj * N + ri] / sqrt(di * dj);
}

// W(:,j) = V(:,j) * lambda_j^{-1/2}
__global__ void k_scale_cols_invsqrt(const double *__restrict__ V, const double *__restrict__ lam, int n,
                                     double *__restrict__ W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  W[(size_t)j * n + i] = V[(size_t)j * n + i] / sqrt(lam[j]);
}

// Sinvh(rows[i], coff+j) = X(i,j) / sqrt(S(rows[i],rows[i]))
__global__ void k_scatter_sinvh(const double *__restrict__ X, const double *__restrict__ S, int N,
                                const int64_t *__restrict__ rows, int n, int coff, double *__restrict__ Sinvh) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  size_t ri = rows[i];
  Sinvh[(size_t)(coff + j) * N + ri] = X[(size_t)j * n + i] / sqrt(S[ri * N + ri]);
}

void form_sinvh_dev(hfg_ctx *ctx, int N, const double *dS, bool chol, int nblk, const int64_t *blk_ptr,
                    const int64_t *blk_idx, double *dSinvh) {
  if (chol)
    throw std::logic_error("form_sinvh: the Cholesky variant (--diag 0) is not implemented on the device yet\n");
  hipStream_t s = ctx->stream;
  DevBuf<double> &idxbuf = ctx->ws[2];
  idxbuf.resize((size_t)N + 16);
  int64_t *drows = (int64_t *)idxbuf.p;
  HFG_HIP_CHECK(hipMemcpyAsync(drows, blk_idx, sizeof(int64_t) * blk_ptr[nblk], hipMemcpyHostToDevice, s));
  HFG_HIP_CHECK(hipMemsetAsync(dSinvh, 0, sizeof(double) * (size_t)N * N, s));
  int coff = 0;
  for (int ib = 0; ib < nblk; ib++) {
    int n = (int)(blk_ptr[ib + 1] - blk_ptr[ib]);
    if (n == 0) continue;
    DevBuf<double> &Sn = ctx->ws[3], &V = ctx->ws[4], &W = ctx->ws[5], &lam = ctx->ws[6], &X = ctx->ws[7];
    Sn.resize((size_t)n * n);
    V.resize((size_t)n * n);
    W.resize((size_t)n * n);
    X.resize((size_t)n * n);
    lam.resize(n);
    dim3 grid((n + 255) / 256, n);
    hipLaunchKernelGGL(k_gather_normalized, grid, dim3(256), 0, s, dS, N, drows + blk_ptr[ib], n, Sn.p);
    eig_sym_dev(ctx, n, Sn.p, lam.p, V.p);
    hipLaunchKernelGGL(k_scale_cols_invsqrt, grid, dim3(256), 0, s, V.p, lam.p, n, W.p);
    gemm_dev(ctx, false, true, n, n, n, 1.0, W.p, n, V.p, n, 0.0, X.p, n);
    hipLaunchKernelGGL(k_scatter_sinvh, grid, dim3(256), 0, s, X.p, dS, N, drows + blk_ptr[ib], n, coff, dSinvh);
    coff += n;
  }
  HFG_HIP_CHECK(hipGetLastError());
}

void form_density_dev(hfg_ctx *ctx, int N, int ncols, const double *dC, int nocc, double *dP) {
  if (ncols < nocc) throw std::logic_error("Not enough orbitals!\n");
  ProfScope ps(ctx, "density");
  if (nocc == 0) {
    HFG_HIP_CHECK(hipMemsetAsync(dP, 0, sizeof(double) * (size_t)N * N, ctx->stream));
    return;
  }
  gemm_dev(ctx, false, true, N, N, nocc, 1.0, dC, N, dC, N, 0.0, dP, N);
}

}  // namespace hfg
