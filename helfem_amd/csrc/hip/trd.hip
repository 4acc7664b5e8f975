// Blocked Householder tridiagonalisation of symmetric matrices on gfx950, batched over problems
// (LAPACK dsytrd/dlatrd, lower variant; the reduction stage of the dsyevd the reference reaches
// through arma::eig_sym, src/general/scf_helpers.cpp:135).
//
// Inside a panel of NB columns the trailing matrix is NOT updated; every column costs one sweep
// over the trailing matrix (y = A22 v, the bandwidth-bound half of the reduction: sum_k 8 (n-k)^2
// bytes) plus O(m NB) corrections with the panel's V and W, and the rank-2NB update
// A22 -= V W^T + W V^T is applied once per panel on the matrix cores.  Two launches per column:
//
//   k_trdb_gemv (row slab x column slab)   v from the already-updated column; partial A22 v; partial
//                                          v^T A22 v; partial V^T v, W^T v
//   k_trdb_w    (row slab)                 reduces the partials (redundantly, they are tiny), forms
//                                          w = tau (p - V(W^T v) - W(V^T v)) - (tau^2/2)(v^T p') v with
//                                          v^T p' = v^T A v - 2 (V^T v).(W^T v) (no extra global reduction),
//                                          and the next column A(:,i+1) - V W(i+1,:)^T - W V(i+1,:)^T
//                                          together with the partial norms its Householder vector needs.
// No atomics; all reductions have a fixed order, so the factorisation is bitwise reproducible.
#include "common.h"

namespace hfg {

void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);

constexpr int TB_MAXB = 8;
constexpr int TB_NB = 32;   // panel width
constexpr int TB_NCS = 8;   // column slabs of the trailing-matrix sweep

struct TrdBatch {
  int n[TB_MAXB];
  double *A[TB_MAXB];
  double *d[TB_MAXB], *e[TB_MAXB], *tau[TB_MAXB];
  double *V[TB_MAXB], *W[TB_MAXB];   // n x NB, ld n
  double *col[TB_MAXB];              // n: current (updated) column
  double *normp[TB_MAXB];            // partial sum of squares per 64-row slab of the current column
  double *pp[TB_MAXB];               // NCS x n partial A22 v
  double *dots[TB_MAXB];             // partial v^T A22 v per gemv workgroup
  double *cpart[TB_MAXB];            // nslab x 2NB partial V^T v, W^T v
};

__device__ inline void tb_householder(const double *__restrict__ x, int m, double xn2, double &tau, double &beta,
                                      double &scale) {
  double alpha = x[0];
  if (xn2 == 0.0) {
    tau = 0.0;
    beta = alpha;
    scale = 0.0;
  } else {
    double nrm = sqrt(alpha * alpha + xn2);
    beta = (alpha >= 0.0) ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
}

// load column i of A into col[], partial norms of rows > i+1 (first column of a panel / of the matrix)
__global__ __launch_bounds__(64) void k_trdb_loadcol(TrdBatch b, int i) {
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (i > n - 3) return;
  const int m = n - i - 1;
  const int nrs = (m + 63) / 64;
  const int rs = blockIdx.x;
  if (rs >= nrs) return;
  const int lane = threadIdx.x;
  const int lr = rs * 64 + lane;  // local row in x = A[i+1:n, i]
  double v = 0.0;
  if (lr < m) {
    v = b.A[blk][(size_t)i * n + i + 1 + lr];
    b.col[blk][i + 1 + lr] = v;
  }
  if (rs == 0 && lane == 0) {
    b.col[blk][i] = b.A[blk][(size_t)i * n + i];
    b.normp[blk][nrs] = 0.0;  // consumers sum one slot more (the slab count of k_trdb_w at the previous column)
  }
  double s = (lr >= 1 && lr < m) ? v * v : 0.0;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) b.normp[blk][rs] = s;
}

// column i (panel column c): v, tau; partial p = A22 v ; partial dots
__global__ __launch_bounds__(256) void k_trdb_gemv(TrdBatch b, int i, int c) {
  extern __shared__ double sh[];  // v[m], red[4*64]
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (i > n - 3) return;
  const int m = n - i - 1;
  const int nrs = (m + 63) / 64;
  const int rs = blockIdx.x / TB_NCS, cs = blockIdx.x % TB_NCS;
  if (rs >= nrs) return;
  double *vsh = sh;
  double *red = sh + n;
  const double *x = b.col[blk] + i + 1;
  double xn2 = 0.0;
  for (int k = 0; k < (m + 1 + 63) / 64; k++) xn2 += b.normp[blk][k];
  double tau, beta, scale;
  tb_householder(x, m, xn2, tau, beta, scale);
  for (int k = threadIdx.x; k < m; k += blockDim.x) vsh[k] = (k == 0) ? 1.0 : x[k] * scale;
  __syncthreads();
  double *A = b.A[blk];
  if (blockIdx.x == 0) {
    // panel bookkeeping by the first workgroup: V(:,c) = v, Householder vector stored in A for the back-transformation
    double *Vc = b.V[blk] + (size_t)c * n;
    for (int k = threadIdx.x; k < n; k += blockDim.x) Vc[k] = (k >= i + 1) ? vsh[k - i - 1] : 0.0;
    for (int k = 1 + threadIdx.x; k < m; k += blockDim.x) A[(size_t)i * n + i + 1 + k] = vsh[k];
    if (threadIdx.x == 0) {
      b.tau[blk][i] = tau;
      b.e[blk][i] = beta;
      b.d[blk][i] = b.col[blk][i];
    }
  }
  const int cchunk = (m + TB_NCS - 1) / TB_NCS;
  const int c0 = cs * cchunk, c1 = min(m, c0 + cchunk);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = rs * 64 + lane;
  double acc = 0.0;
  if (row < m) {
    const double *a = A + (size_t)(i + 1) * n + (i + 1) + row;
    for (int cc = c0 + wave; cc < c1; cc += 4) acc += a[(size_t)cc * n] * vsh[cc];
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0) {
    double p = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    if (row < m) b.pp[blk][(size_t)cs * n + row] = p;
    double dv = (row < m) ? p * vsh[row] : 0.0;
    for (int o = 32; o > 0; o >>= 1) dv += __shfl_down(dv, o, 64);
    if (lane == 0) b.dots[blk][blockIdx.x] = dv;
  }
  // partial V^T v and W^T v of this row slab (done once per row slab)
  if (cs == 0 && c > 0) {
    const double vr = (row < m) ? vsh[row] : 0.0;
    const int grow = i + 1 + row;
    for (int cc = wave; cc < 2 * c; cc += 4) {
      const double *M = (cc < c) ? b.V[blk] + (size_t)cc * n : b.W[blk] + (size_t)(cc - c) * n;
      double t = (row < m) ? M[grow] * vr : 0.0;
      for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
      if (lane == 0) b.cpart[blk][(size_t)rs * 2 * TB_NB + cc] = t;
    }
  }
}

// w for this row slab, and the next column (if it is still inside the panel)
__global__ __launch_bounds__(64) void k_trdb_w(TrdBatch b, int i, int c, int do_next) {
  __shared__ double Vv[TB_NB], Wv[TB_NB], vrow[TB_NB + 1], wrow[TB_NB + 1];
  __shared__ double scal[4];
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (i > n - 3) return;
  const int m = n - i - 1;
  const int nrs = (m + 63) / 64;
  const int rs = blockIdx.x;
  if (rs >= nrs) return;
  const int lane = threadIdx.x;
  const double tau = b.tau[blk][i];
  // reduce the tiny partials (every workgroup does it for itself; fixed order)
  if (lane < c) {
    double a = 0.0, w = 0.0;
    for (int k = 0; k < nrs; k++) {
      a += b.cpart[blk][(size_t)k * 2 * TB_NB + lane];
      w += b.cpart[blk][(size_t)k * 2 * TB_NB + c + lane];
    }
    Vv[lane] = a;
    Wv[lane] = w;
  }
  if (lane == 0) {
    double s = 0.0;
    for (int k = 0; k < nrs * TB_NCS; k++) s += b.dots[blk][k];
    scal[0] = s;  // v^T A22 v
  }
  __syncthreads();
  if (lane == 0) {
    double vp = scal[0];
    for (int cc = 0; cc < c; cc++) vp -= 2.0 * Vv[cc] * Wv[cc];
    scal[1] = -0.5 * tau * tau * vp;  // alpha
  }
  __syncthreads();
  const double alpha = scal[1];
  const double *Vb = b.V[blk], *Wb = b.W[blk];
  auto w_of_row = [&](int lr) {  // lr = local row (global row i+1+lr)
    const int g = i + 1 + lr;
    double p = 0.0;
    for (int cs = 0; cs < TB_NCS; cs++) p += b.pp[blk][(size_t)cs * n + lr];
    for (int cc = 0; cc < c; cc++) p -= Vb[(size_t)cc * n + g] * Wv[cc] + Wb[(size_t)cc * n + g] * Vv[cc];
    return tau * p + alpha * Vb[(size_t)c * n + g];
  };
  const int lr = rs * 64 + lane;
  double w = 0.0;
  if (lr < m) {
    w = w_of_row(lr);
    b.W[blk][(size_t)c * n + i + 1 + lr] = w;
  }
  if (rs == 0)  // rows above the active part of W(:,c) are zero
    for (int k = lane; k <= i; k += 64) b.W[blk][(size_t)c * n + k] = 0.0;
  if (!do_next || i + 1 > n - 3) return;
  // ---- next column i+1, rows i+1..n-1:  A(r,i+1) - sum_{cc<=c} V(r,cc) W(i+1,cc) + W(r,cc) V(i+1,cc) ----
  // row i+1 of the panel (local row 0); W(i+1,c) is recomputed here by every workgroup
  if (lane <= c) {
    vrow[lane] = Vb[(size_t)lane * n + i + 1];
    wrow[lane] = (lane < c) ? Wb[(size_t)lane * n + i + 1] : 0.0;
  }
  __syncthreads();
  if (lane == 0) wrow[c] = w_of_row(0);
  __syncthreads();
  double nv = 0.0;
  if (lr < m) {
    const int g = i + 1 + lr;
    double a = b.A[blk][(size_t)(i + 1) * n + g];
    for (int cc = 0; cc < c; cc++) a -= Vb[(size_t)cc * n + g] * wrow[cc] + Wb[(size_t)cc * n + g] * vrow[cc];
    a -= Vb[(size_t)c * n + g] * wrow[c] + w * vrow[c];
    b.col[blk][g] = a;
    nv = (lr >= 2) ? a * a : 0.0;  // x = col[i+2:], its tail x[1:] starts at local row 2
  }
  for (int o = 32; o > 0; o >>= 1) nv += __shfl_down(nv, o, 64);
  // slabs of the NEXT column are offset by one row: regroup so that normp[k] covers local rows of x
  // (x local index q = lr - 1); a 64-row slab of x spans two slabs of this kernel, so the partial sums are
  // stored per slab of THIS kernel and the consumer only needs their total.
  if (lane == 0) b.normp[blk][rs] = nv;
}

// d, e of the last 2x2 block (after the final trailing update)
__global__ void k_trdb_finish(TrdBatch b) {
  int blk = blockIdx.x;
  if (threadIdx.x != 0) return;
  int n = b.n[blk];
  double *A = b.A[blk];
  if (n >= 2) {
    b.d[blk][n - 2] = A[(size_t)(n - 2) * n + (n - 2)];
    b.e[blk][n - 2] = A[(size_t)(n - 2) * n + (n - 1)];
    b.tau[blk][n - 2] = 0.0;
  }
  b.d[blk][n - 1] = A[(size_t)(n - 1) * n + (n - 1)];
  b.e[blk][n - 1] = 0.0;
  if (n >= 1) b.tau[blk][n - 1] = 0.0;
}

struct TrdWork {
  DevBuf<double> V[TB_MAXB], W[TB_MAXB], col[TB_MAXB], normp[TB_MAXB], pp[TB_MAXB], dots[TB_MAXB], cpart[TB_MAXB];
};
static std::map<hfg_ctx *, TrdWork *> g_trd;
void trd_release(hfg_ctx *ctx) {
  auto it = g_trd.find(ctx);
  if (it != g_trd.end()) {
    delete it->second;
    g_trd.erase(it);
  }
}

/// A[blk] (n x n, ld n, full symmetric storage) -> d, e, tau and the Householder vectors below the subdiagonal
void tridiagonalize_batch(hfg_ctx *ctx, int nblk, const int *ns, double *const *A, double *const *d, double *const *e,
                          double *const *tau) {
  if (nblk > TB_MAXB) throw std::logic_error("tridiagonalize_batch: too many blocks");
  TrdWork *wp;
  auto it = g_trd.find(ctx);
  if (it == g_trd.end()) {
    wp = new TrdWork();
    g_trd[ctx] = wp;
  } else
    wp = it->second;
  TrdWork &w = *wp;
  TrdBatch b;
  int nmax = 0;
  for (int i = 0; i < nblk; i++) {
    int n = ns[i];
    nmax = std::max(nmax, n);
    int nslab = (n + 63) / 64 + 1;
    w.V[i].resize((size_t)n * TB_NB);
    w.W[i].resize((size_t)n * TB_NB);
    w.col[i].resize(n);
    w.normp[i].resize(nslab);
    w.pp[i].resize((size_t)TB_NCS * n);
    w.dots[i].resize((size_t)TB_NCS * nslab);
    w.cpart[i].resize((size_t)nslab * 2 * TB_NB);
    b.n[i] = n;
    b.A[i] = A[i];
    b.d[i] = d[i];
    b.e[i] = e[i];
    b.tau[i] = tau[i];
    b.V[i] = w.V[i].p;
    b.W[i] = w.W[i].p;
    b.col[i] = w.col[i].p;
    b.normp[i] = w.normp[i].p;
    b.pp[i] = w.pp[i].p;
    b.dots[i] = w.dots[i].p;
    b.cpart[i] = w.cpart[i].p;
  }
  hipStream_t s = ctx->stream;
  size_t shb = (size_t)(nmax + 4 * 64 + 8) * sizeof(double);
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_trdb_gemv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  for (int j0 = 0; j0 <= nmax - 3; j0 += TB_NB) {
    {
      int m = nmax - j0 - 1;
      hipLaunchKernelGGL(k_trdb_loadcol, dim3((m + 63) / 64, nblk), dim3(64), 0, s, b, j0);
    }
    const int jend = std::min(j0 + TB_NB, nmax - 2);  // columns j0 .. jend-1 (global, for the largest block)
    for (int i = j0; i < jend; i++) {
      const int c = i - j0;
      const int m = nmax - i - 1;
      const int nrs = (m + 63) / 64;
      hipLaunchKernelGGL(k_trdb_gemv, dim3(nrs * TB_NCS, nblk), dim3(256), shb, s, b, i, c);
      hipLaunchKernelGGL(k_trdb_w, dim3(nrs, nblk), dim3(64), 0, s, b, i, c, (i + 1 < jend) ? 1 : 0);
    }
    // trailing update per block: columns processed in this panel for block k: j0 .. min(j0+NB, n_k-2)-1
    for (int k = 0; k < nblk; k++) {
      int n = ns[k];
      int ncols = std::min(TB_NB, std::max(0, n - 2 - j0));
      if (ncols <= 0) continue;
      int j1 = j0 + ncols;
      int mt = n - j1;
      if (mt <= 0) continue;
      double *A22 = A[k] + (size_t)j1 * n + j1;
      gemm_dev(ctx, false, true, mt, mt, ncols, -1.0, w.V[k].p + j1, n, w.W[k].p + j1, n, 1.0, A22, n);
      gemm_dev(ctx, false, true, mt, mt, ncols, -1.0, w.W[k].p + j1, n, w.V[k].p + j1, n, 1.0, A22, n);
    }
  }
  hipLaunchKernelGGL(k_trdb_finish, dim3(nblk), dim3(64), 0, s, b);
  HFG_HIP_CHECK(hipGetLastError());
}

}  // namespace hfg
