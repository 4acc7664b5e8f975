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
__global__ __launch_bounds__(64) void k_trdb_loadcol(const TrdBatch *__restrict__ bp, int i) {
  const TrdBatch &b = *bp;
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

// column i (panel column c): v, tau; partial p = A22 v ; partial dots.
// Workgroup = 128 rows x one column slab; a lane owns two consecutive rows, the four waves split the
// slab's columns and meet in LDS.  The kernel is latency bound (one dependent chain per launch), so the
// loads of the trailing matrix, which do not depend on v, are issued BEFORE the Householder prologue and
// the slabs are kept small (32 columns) so that the whole matrix is in flight at once.
__global__ __launch_bounds__(256) void k_trdb_gemv(const TrdBatch *__restrict__ bp, int i, int c, int ncs) {
  const TrdBatch &b = *bp;
  extern __shared__ double sh[];  // v[m], red[4*128]
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (i > n - 3) return;
  const int m = n - i - 1;
  const int nrs = (m + 1 + 127) / 128;  // one spare row for the alignment shift
  const int rs = blockIdx.x / ncs, cs = blockIdx.x % ncs;
  if (rs >= nrs) return;
  double *vsh = sh;
  double *red = sh + n;
  __shared__ double nsum[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *A = b.A[blk];
  const int cchunk = (m + ncs - 1) / ncs;
  const int c0 = cs * cchunk, c1 = min(m, c0 + cchunk);
  // A lane owns the row pair (row, row+1).  For even n the pairs are shifted by delta so that every pair is a
  // 16-byte aligned double2 in every column (1 KiB per wave instruction at the full 16-B/lane rate).
  const bool vec2 = ((n & 1) == 0);
  const int delta = vec2 ? (int)((((size_t)(i + 1) * n + (i + 1))) & 1) : 0;
  const int row = rs * 128 + 2 * lane - delta;
  const double *a = A + (size_t)(i + 1) * n + (i + 1) + row;
  // ---- issue the first batch of matrix loads (16 columns x 2 rows per lane) ----
  constexpr int NU = 16;
  double r0[NU], r1[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    int cc = c0 + wave + 4 * u;
    bool ok = (cc < c1);
    r0[u] = 0.0;
    r1[u] = 0.0;
    if (ok) {
      if (vec2 && row >= 0 && row + 1 < m) {
        double2 t = *reinterpret_cast<const double2 *>(a + (size_t)cc * n);
        r0[u] = t.x;
        r1[u] = t.y;
      } else {
        if (row >= 0 && row < m) r0[u] = a[(size_t)cc * n];
        if (row + 1 >= 0 && row + 1 < m) r1[u] = a[(size_t)cc * n + 1];
      }
    }
  }
  // ---- Householder vector of the (already updated) column i ----
  const double *x = b.col[blk] + i + 1;
  {
    const int np = (m + 1 + 63) / 64;
    double t = 0.0;
    for (int k = tid; k < np; k += 256) t += b.normp[blk][k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if (lane == 0) nsum[wave] = t;
  }
  double xv[4];  // up to 1024 rows per pass; longer columns loop below
  const int npass = (m + 255) / 256;
  for (int q = 0; q < 4; q++) {
    int k = tid + 256 * q;
    xv[q] = (q < npass && k < m) ? x[k] : 0.0;
  }
  const double alpha0 = x[0];
  __syncthreads();
  const double xn2 = (nsum[0] + nsum[1]) + (nsum[2] + nsum[3]);
  double tau, beta, scale;
  {
    if (xn2 == 0.0) {
      tau = 0.0;
      beta = alpha0;
      scale = 0.0;
    } else {
      double nrm = sqrt(alpha0 * alpha0 + xn2);
      beta = (alpha0 >= 0.0) ? -nrm : nrm;
      tau = (beta - alpha0) / beta;
      scale = 1.0 / (alpha0 - beta);
    }
  }
  for (int q = 0; q < 4 && q < npass; q++) {
    int k = tid + 256 * q;
    if (k < m) vsh[k] = (k == 0) ? 1.0 : xv[q] * scale;
  }
  for (int k = tid + 1024; k < m; k += 256) vsh[k] = x[k] * scale;
  __syncthreads();
  if (blockIdx.x == 0) {
    // panel bookkeeping by the first workgroup: V(:,c) = v, Householder vector stored in A for the back-transformation
    double *Vc = b.V[blk] + (size_t)c * n;
    for (int k = tid; k < n; k += 256) Vc[k] = (k >= i + 1) ? vsh[k - i - 1] : 0.0;
    for (int k = 1 + tid; k < m; k += 256) A[(size_t)i * n + i + 1 + k] = vsh[k];
    if (tid == 0) {
      b.tau[blk][i] = tau;
      b.e[blk][i] = beta;
      b.d[blk][i] = b.col[blk][i];
    }
  }
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int u = 0; u < NU; u++) {
    int cc = c0 + wave + 4 * u;
    double vc = (cc < c1) ? vsh[cc] : 0.0;
    acc0 += r0[u] * vc;
    acc1 += r1[u] * vc;
  }
  for (int cc = c0 + wave + 4 * NU; cc < c1; cc += 4) {  // slabs wider than 64 columns (very large matrices only)
    double vc = vsh[cc];
    if (row >= 0 && row < m) acc0 += a[(size_t)cc * n] * vc;
    if (row + 1 >= 0 && row + 1 < m) acc1 += a[(size_t)cc * n + 1] * vc;
  }
  red[wave * 128 + 2 * lane] = acc0;
  red[wave * 128 + 2 * lane + 1] = acc1;
  __syncthreads();
  if (tid < 128) {
    const int r = rs * 128 + tid - delta;
    const bool okr = (r >= 0 && r < m);
    double p = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    if (okr) b.pp[blk][(size_t)cs * n + r] = p;
    double dv = okr ? p * vsh[r] : 0.0;
    for (int o = 32; o > 0; o >>= 1) dv += __shfl_down(dv, o, 64);
    if ((tid & 63) == 0) b.dots[blk][blockIdx.x * 2 + (tid >> 6)] = dv;
  }
  // partial V^T v and W^T v of this row slab (done once per row slab): each wave takes columns cc = wave, wave+4, ...
  if (cs == 0 && c > 0) {
    for (int cc = wave; cc < 2 * c; cc += 4) {
      const double *M = (cc < c) ? b.V[blk] + (size_t)cc * n : b.W[blk] + (size_t)(cc - c) * n;
      double t = 0.0;
      for (int h = 0; h < 2; h++) {
        int r = rs * 128 + h * 64 + lane;
        if (r < m) t += M[i + 1 + r] * vsh[r];
      }
      for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
      if (lane == 0) b.cpart[blk][(size_t)rs * 2 * TB_NB + cc] = t;
    }
  }
}

// w for a 64-row slab, and the next column (if it is still inside the panel).  256 threads: lane = row,
// wave q takes the panel columns cc = q, q+4, ...; the four partial sums meet in LDS.  All global loads
// are issued up front (one latency instead of five).
__global__ __launch_bounds__(256) void k_trdb_w(const TrdBatch *__restrict__ bp, int i, int c, int do_next, int ncs) {
  const TrdBatch &b = *bp;
  __shared__ double Vv[TB_NB], Wv[TB_NB], vrow[TB_NB + 1], wrow[TB_NB + 1];
  __shared__ double red[4 * 64], cred[4 * 64];
  __shared__ double scal[12];
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (i > n - 3) return;
  const int m = n - i - 1;
  const int nrs = (m + 63) / 64;          // slabs of this kernel
  const int nrg = (m + 1 + 127) / 128;    // row slabs of the gemv kernel
  const int rs = blockIdx.x;
  if (rs >= nrs) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *Vb = b.V[blk], *Wb = b.W[blk];
  const int lr = rs * 64 + lane;
  const int g = i + 1 + lr;
  const bool live = lr < m;
  const bool next = do_next && (i + 1 <= n - 3);
  // ---- up-front loads ----
  double vv[8], ww[8];  // V(g,cc), W(g,cc) for cc = wave + 4u
#pragma unroll
  for (int u = 0; u < 8; u++) {
    int cc = wave + 4 * u;
    bool ok = live && cc < c;
    vv[u] = ok ? Vb[(size_t)cc * n + g] : 0.0;
    ww[u] = ok ? Wb[(size_t)cc * n + g] : 0.0;
  }
  // partial sums are read with fully unrolled, predicated loads: a rolled loop of "load, wait, add" costs one
  // memory round trip per iteration on this latency-bound kernel
  double psum = 0.0;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    int cs = wave + 4 * u;
    if (live && cs < ncs) psum += b.pp[blk][(size_t)cs * n + lr];
  }
  const double vg = live ? Vb[(size_t)c * n + g] : 0.0;
  const double anext = (live && next) ? b.A[blk][(size_t)(i + 1) * n + g] : 0.0;
  const double tau = b.tau[blk][i];
  double p0 = 0.0, v0r = 0.0, w0r = 0.0;  // row i+1 (local row 0) quantities, spread over the threads
  if (next) {
    if (tid < ncs) p0 = b.pp[blk][(size_t)tid * n];
    if (tid <= c) v0r = Vb[(size_t)tid * n + i + 1];
    if (tid < c) w0r = Wb[(size_t)tid * n + i + 1];
  }
  // ---- reduce the partials of the gemv kernel (all workgroups redundantly, fixed order) ----
  {
    // cpart: thread (cc = lane, q = wave) sums the row slabs k = q, q+4, ...
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < 12; u++) {
      int k = wave + 4 * u;
      if (lane < 2 * c && k < nrg) a += b.cpart[blk][(size_t)k * 2 * TB_NB + lane];
    }
    for (int k = wave + 48; k < nrg; k += 4)
      if (lane < 2 * c) a += b.cpart[blk][(size_t)k * 2 * TB_NB + lane];
    cred[wave * 64 + lane] = a;
  }
  {
    const int nd = nrg * ncs * 2;
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int k = tid + 256 * u;
      if (k < nd) s += b.dots[blk][k];
    }
    for (int k = tid + 2048; k < nd; k += 256) s += b.dots[blk][k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) scal[4 + wave] = s;
  }
  __syncthreads();
  if (tid < 2 * c) {
    double a = (cred[tid] + cred[64 + tid]) + (cred[128 + tid] + cred[192 + tid]);
    if (tid < c) Vv[tid] = a;
    else Wv[tid - c] = a;
  }
  if (next && tid <= c) {
    vrow[tid] = v0r;
    wrow[tid] = w0r;
  }
  __syncthreads();
  if (tid == 0) {
    double vp = (scal[4] + scal[5]) + (scal[6] + scal[7]);  // v^T A22 v
    for (int cc = 0; cc < c; cc++) vp -= 2.0 * Vv[cc] * Wv[cc];
    scal[1] = -0.5 * tau * tau * vp;  // alpha
  }
  // p' = sum_cs pp - V (W^T v) - W (V^T v), split over the four waves
  double part = psum;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    int cc = wave + 4 * u;
    if (cc < c) part -= vv[u] * Wv[cc] + ww[u] * Vv[cc];
  }
  red[wave * 64 + lane] = part;
  if (next) {
    // row 0 of the trailing block (global row i+1), needed by every workgroup for the next column
    if (tid < c) p0 -= v0r * Wv[tid] + w0r * Vv[tid];
    for (int o = 32; o > 0; o >>= 1) p0 += __shfl_down(p0, o, 64);
    if (lane == 0) scal[8 + wave] = p0;
  }
  __syncthreads();
  const double alpha = scal[1];
  double w = 0.0;
  if (wave == 0 && live) {
    double p = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    w = tau * p + alpha * vg;
    b.W[blk][(size_t)c * n + g] = w;
  }
  if (rs == 0)  // rows above the active part of W(:,c) are zero
    for (int k = tid; k <= i; k += 256) b.W[blk][(size_t)c * n + k] = 0.0;
  if (!next) return;
  if (tid == 0) {
    double pz = (scal[8] + scal[9]) + (scal[10] + scal[11]);
    wrow[c] = tau * pz + alpha * 1.0;  // V(i+1,c) = 1
  }
  __syncthreads();
  if (wave == 0) red[lane] = w;
  __syncthreads();
  // ---- next column i+1, rows i+1..n-1:  A(r,i+1) - sum_{cc<=c} V(r,cc) W(i+1,cc) + W(r,cc) V(i+1,cc) ----
  double upd = 0.0;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    int cc = wave + 4 * u;
    if (cc < c) upd += vv[u] * wrow[cc] + ww[u] * vrow[cc];
  }
  if (wave == 0 && live) upd += vg * wrow[c] + red[lane] * vrow[c];
  __syncthreads();
  red[wave * 64 + lane] = upd;
  __syncthreads();
  if (wave == 0) {
    double nv = 0.0;
    if (live) {
      double a = anext - ((red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]));
      b.col[blk][g] = a;
      nv = (lr >= 2) ? a * a : 0.0;  // x = col[i+2:], its tail x[1:] starts at local row 2
    }
    for (int o = 32; o > 0; o >>= 1) nv += __shfl_down(nv, o, 64);
    if (lane == 0) b.normp[blk][rs] = nv;
  }
}

// d, e of the last 2x2 block (after the final trailing update)
__global__ void k_trdb_finish(const TrdBatch *__restrict__ bp) {
  const TrdBatch &b = *bp;
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
  DevBuf<TrdBatch> desc;
  std::vector<int> last_ns;  // sizes of the last batch (for the measurement replay)
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
    w.pp[i].resize((size_t)64 * n);
    w.dots[i].resize((size_t)2 * 64 * nslab);
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
  w.desc.resize(1);
  HFG_HIP_CHECK(hipMemcpyAsync(w.desc.p, &b, sizeof(TrdBatch), hipMemcpyHostToDevice, s));
  HFG_HIP_CHECK(hipStreamSynchronize(s));  // b lives on this stack frame
  const TrdBatch *db = w.desc.p;
  size_t shb = (size_t)(nmax + 4 * 128 + 8) * sizeof(double);
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_trdb_gemv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  for (int j0 = 0; j0 <= nmax - 3; j0 += TB_NB) {
    {
      int m = nmax - j0 - 1;
      hipLaunchKernelGGL(k_trdb_loadcol, dim3((m + 63) / 64, nblk), dim3(64), 0, s, db, j0);
    }
    const int jend = std::min(j0 + TB_NB, nmax - 2);  // columns j0 .. jend-1 (global, for the largest block)
    for (int i = j0; i < jend; i++) {
      const int c = i - j0;
      const int m = nmax - i - 1;
      const int nrs = (m + 63) / 64, nrg = (m + 1 + 127) / 128;
      // 64-column slabs (16 columns per wave, all issued before the prologue); at most 64 slabs
      int ncs = std::max(1, std::min(64, (m + 63) / 64));
      hipLaunchKernelGGL(k_trdb_gemv, dim3(nrg * ncs, nblk), dim3(256), shb, s, db, i, c, ncs);
      hipLaunchKernelGGL(k_trdb_w, dim3(nrs, nblk), dim3(256), 0, s, db, i, c, (i + 1 < jend) ? 1 : 0, ncs);
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
  hipLaunchKernelGGL(k_trdb_finish, dim3(nblk), dim3(64), 0, s, db);
  HFG_HIP_CHECK(hipGetLastError());
  w.last_ns.assign(ns, ns + nblk);
}

/// Measurement replay of the dominant kernel: the k_trdb_gemv launch of every Householder column of the last
/// batch, back to back on the context's stream between two HIP events (the companion k_trdb_w launches are left
/// out, the sweep's duration does not depend on the matrix values).  Returns total ms and the launch count.
void trd_measure_gemv(hfg_ctx *ctx, double *ms, int64_t *launches) {
  *ms = 0.0;
  *launches = 0;
  auto it = g_trd.find(ctx);
  if (it == g_trd.end() || it->second->last_ns.empty()) throw std::logic_error("no tridiagonalisation has run on this context");
  TrdWork &w = *it->second;
  const int nblk = (int)w.last_ns.size();
  int nmax = 0;
  for (int n : w.last_ns) nmax = std::max(nmax, n);
  hipStream_t s = ctx->stream;
  const TrdBatch *db = w.desc.p;
  size_t shb = (size_t)(nmax + 4 * 128 + 8) * sizeof(double);
  hipEvent_t e0, e1;
  HFG_HIP_CHECK(hipEventCreate(&e0));
  HFG_HIP_CHECK(hipEventCreate(&e1));
  HFG_HIP_CHECK(hipEventRecord(e0, s));
  int count = 0;
  for (int i = 0; i <= nmax - 3; i++) {
    const int c = i % TB_NB;
    const int m = nmax - i - 1;
    const int nrg = (m + 1 + 127) / 128;
    int ncs = std::max(1, std::min(64, (m + 63) / 64));
    hipLaunchKernelGGL(k_trdb_gemv, dim3(nrg * ncs, nblk), dim3(256), shb, s, db, i, c, ncs);
    count++;
  }
  HFG_HIP_CHECK(hipEventRecord(e1, s));
  HFG_HIP_CHECK(hipEventSynchronize(e1));
  float t = 0.f;
  HFG_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = t;
  *launches = count;
}

}  // namespace hfg
