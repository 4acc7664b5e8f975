// Small helpers around the eigensolver: S^{-1/2} per symmetry block (reference utils::invh,
// libhelfem/src/utils.cpp:160-183 and TwoDBasis::Sinvh, src/diatomic/basis.cpp:627-652) and
// scf::form_density (src/general/scf_helpers.cpp:22-29).
#include "common.h"
#include <algorithm>

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

// ---- Cholesky variant (utils::invh with chol = true, utils.cpp:168: Sinvh = inv(chol(S))) ----------------------
constexpr int CH_KB = 64;

// diagonal block: A_kk (kb x kb, lower part used) -> L_kk in place, and Linv = L_kk^{-1} (lower triangular, ld CH_KB)
__global__ __launch_bounds__(64) void k_potrf_inv(double *__restrict__ A, int lda, int kb, double *__restrict__ Linv,
                                                  int *__restrict__ status) {
  __shared__ double L[CH_KB][CH_KB + 1];
  __shared__ double X[CH_KB][CH_KB + 1];
  const int t = threadIdx.x;
  for (int c = 0; c < kb; c++) L[t][c] = (t < kb && c <= t) ? A[(size_t)c * lda + t] : 0.0;
  for (int c = 0; c < CH_KB; c++) X[t][c] = 0.0;
  __syncthreads();
  for (int j = 0; j < kb; j++) {
    // column j: L(j:,j) = (A(j:,j) - L(j:,0:j) L(j,0:j)^T) / sqrt(diag)
    double v = 0.0;
    if (t >= j && t < kb) {
      v = L[t][j];
      for (int k = 0; k < j; k++) v -= L[t][k] * L[j][k];
    }
    __syncthreads();
    if (t == j) {
      if (!(v > 0.0)) *status = 1;  // not positive definite ("Cholesky failed")
      L[j][j] = sqrt(v > 0.0 ? v : 1.0);
    }
    __syncthreads();
    if (t > j && t < kb) L[t][j] = v / L[j][j];
    __syncthreads();
  }
  // inverse of the lower triangle, one column of X per thread: L X = I (forward substitution)
  if (t < kb) {
    for (int i = t; i < kb; i++) {
      double sacc = (i == t) ? 1.0 : 0.0;
      for (int k = t; k < i; k++) sacc -= L[i][k] * X[k][t];
      X[i][t] = sacc / L[i][i];
    }
  }
  __syncthreads();
  for (int c = 0; c < kb; c++) {
    if (t < kb && c <= t) A[(size_t)c * lda + t] = L[t][c];
    if (t < CH_KB) Linv[(size_t)c * CH_KB + t] = (t < kb) ? X[t][c] : 0.0;
  }
}

// Sinvh(rows[i], coff+j) = X(j,i) / sqrt(S(rows[i],rows[i]))  for j >= i (X = L^{-1}, so R^{-1} = X^T is upper)
__global__ void k_scatter_sinvh_T(const double *__restrict__ X, const double *__restrict__ S, int N,
                                  const int64_t *__restrict__ rows, int n, int coff, double *__restrict__ Sinvh) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  size_t ri = rows[i];
  Sinvh[(size_t)(coff + j) * N + ri] = (j >= i) ? X[(size_t)i * n + j] / sqrt(S[ri * N + ri]) : 0.0;
}

// X (n x n, lower) = L^{-1} for the normalised overlap block Sn (overwritten by its Cholesky factor L)
static void chol_inverse_block(hfg_ctx *ctx, int n, double *Sn, double *X, double *Dinv, double *T, int *dstatus) {
  hipStream_t s = ctx->stream;
  const int nb = (n + CH_KB - 1) / CH_KB;
  for (int kblk = 0; kblk < nb; kblk++) {
    const int k0 = kblk * CH_KB, kb = std::min(CH_KB, n - k0), rest = n - k0 - kb;
    double *Akk = Sn + (size_t)k0 * n + k0;
    double *Dk = Dinv + (size_t)kblk * CH_KB * CH_KB;
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(64), 0, s, Akk, n, kb, Dk, dstatus);
    if (rest > 0) {
      double *Aik = Sn + (size_t)k0 * n + k0 + kb;
      // L_ik = A_ik L_kk^{-T}
      gemm_dev(ctx, false, true, rest, kb, kb, 1.0, Aik, n, Dk, CH_KB, 0.0, T, rest);
      HFG_HIP_CHECK(hipMemcpy2DAsync(Aik, sizeof(double) * n, T, sizeof(double) * rest, sizeof(double) * rest, kb,
                                     hipMemcpyDeviceToDevice, s));
      // A22 -= L_ik L_ik^T
      double *A22 = Sn + (size_t)(k0 + kb) * n + k0 + kb;
      gemm_dev(ctx, false, true, rest, rest, kb, -1.0, Aik, n, Aik, n, 1.0, A22, n);
    }
  }
  // X = L^{-1}: X_ii = Dinv_i,  X(i, 0:i) = -Dinv_i (L(i, 0:i) X(0:i, 0:i))   block row by block row
  HFG_HIP_CHECK(hipMemsetAsync(X, 0, sizeof(double) * (size_t)n * n, s));
  for (int ib = 0; ib < nb; ib++) {
    const int i0 = ib * CH_KB, kb = std::min(CH_KB, n - i0);
    const double *Di = Dinv + (size_t)ib * CH_KB * CH_KB;
    HFG_HIP_CHECK(hipMemcpy2DAsync(X + (size_t)i0 * n + i0, sizeof(double) * n, Di, sizeof(double) * CH_KB,
                                   sizeof(double) * kb, kb, hipMemcpyDeviceToDevice, s));
    if (i0 > 0) {
      gemm_dev(ctx, false, false, kb, i0, i0, 1.0, Sn + i0, n, X, n, 0.0, T, kb);
      gemm_dev(ctx, false, false, kb, i0, kb, -1.0, Di, CH_KB, T, kb, 0.0, X + i0, n);
    }
  }
}

void form_sinvh_dev(hfg_ctx *ctx, int N, const double *dS, bool chol, int nblk, const int64_t *blk_ptr,
                    const int64_t *blk_idx, double *dSinvh) {
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
    if (chol) {
      // blocked right-looking Cholesky + blocked triangular inverse on the matrix cores; Sinvh = (L^{-1})^T
      DevBuf<double> &Dinv = ctx->ws[6];
      Dinv.resize((size_t)((n + CH_KB - 1) / CH_KB) * CH_KB * CH_KB + 8);
      int *dstatus = (int *)(Dinv.p + (size_t)((n + CH_KB - 1) / CH_KB) * CH_KB * CH_KB);
      HFG_HIP_CHECK(hipMemsetAsync(dstatus, 0, sizeof(int), s));
      chol_inverse_block(ctx, n, Sn.p, X.p, Dinv.p, W.p, dstatus);
      hipLaunchKernelGGL(k_scatter_sinvh_T, grid, dim3(256), 0, s, X.p, dS, N, drows + blk_ptr[ib], n, coff, dSinvh);
      int st = 0;
      HFG_HIP_CHECK(hipMemcpyAsync(&st, dstatus, sizeof(int), hipMemcpyDeviceToHost, s));
      HFG_HIP_CHECK(hipStreamSynchronize(s));
      if (st) throw std::logic_error("Cholesky decomposition of the overlap matrix failed\n");
      coff += n;
      continue;
    }
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
