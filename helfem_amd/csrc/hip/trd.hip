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
#include "wave.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstring>

namespace hfg {

void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);
void gemm_tasklist_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_acc_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN, bool tile64);
void tridiagonalize_persistent(hfg_ctx *ctx, int nblk, const int *ns, double *const *A, double *const *d, double *const *e,
                               double *const *tau, std::vector<char> &done);  // trdp.hip

constexpr int TB_MAXB = 8;
constexpr int TB_NB = 16;   // panel width (measured: 16 beats 8 and 32 at n ~ 1400 x 3 blocks; 32 again after the DPP work: 11.7 vs 7.6 us per column)
constexpr int TB_NCS = 8;   // column slabs of the trailing-matrix sweep
constexpr int TF_T = 128;    // tile edge of the fused kernel
constexpr int TF_MAXS = 64;  // max slabs per dimension: n <= TF_T * (TF_MAXS - 1) = 8064 (larger problems take the two-kernel path)

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
  // fused one-launch-per-column variant (k_trdf): everything below is double buffered by column parity
  double *fx[TB_MAXB];               // 2 x n: unnormalised current column x
  double *fpp[TB_MAXB];              // 2 x TF_MAXS x n: partial A22 x per column slab
  double *fdots[TB_MAXB];            // 2 x TF_MAXS^2: partial x^T A22 x per workgroup
  double *fxn2[TB_MAXB];             // 2 x TF_MAXS: partial |x[1:]|^2 per column slab
  double *fcp[TB_MAXB];              // 2 x TF_MAXS x 2NB: partial V^T x, W^T x per column slab
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
  s = wave_sum(s);
  if (lane == 0) b.normp[blk][rs] = s;
}

// column i (panel column c): v, tau; partial p = A22 v ; partial dots.
// Workgroup = 128 rows x one column slab; a lane owns two consecutive rows, the four waves split the
// slab's columns and meet in LDS.  The kernel is latency bound (one dependent chain per launch), so the
// loads of the trailing matrix, which do not depend on v, are issued BEFORE the Householder prologue and
// the slabs are kept small (32 columns) so that the whole matrix is in flight at once.
__global__ __launch_bounds__(256) void k_trdb_gemv(const TrdBatch *__restrict__ bp, int i, int c, int ncs) {
  const TrdBatch &b = *bp;
  // LDS: v on this workgroup's column chunk and on its rows only (the whole vector, 8 n bytes, left room for ONE
  // workgroup per CU beyond n ~ 8000 and was read from L2 by every workgroup: more bytes than the matrix itself),
  // vcol[cchunk], vrow[129], red[4*128]
  extern __shared__ double sh[];
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (i > n - 3) return;
  const int m = n - i - 1;
  const int nrs = (m + 1 + 127) / 128;  // one spare row for the alignment shift
  const int rs = blockIdx.x / ncs, cs = blockIdx.x % ncs;
  if (rs >= nrs) return;
  __shared__ double nsum[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *A = b.A[blk];
  const int cchunk = (m + ncs - 1) / ncs;
  const int c0 = cs * cchunk, c1 = min(m, c0 + cchunk);
  double *vcol = sh;
  double *vrow = sh + cchunk;
  double *red = vrow + 136;
  // A lane owns the row pair (row, row+1).  For even n the pairs are shifted by delta so that every pair is a
  // 16-byte aligned double2 in every column (1 KiB per wave instruction at the full 16-B/lane rate).
  const bool vec2 = ((n & 1) == 0);
  const int delta = vec2 ? (int)((((size_t)(i + 1) * n + (i + 1))) & 1) : 0;
  const int row = rs * 128 + 2 * lane - delta;
  const double *a = A + (size_t)(i + 1) * n + (i + 1) + row;
  // ---- matrix loads in batches of 16 columns x 2 rows per lane, all of a batch in flight at once ----
  constexpr int NU = 16;
  double r0[NU], r1[NU];
  const bool pair_ok = vec2 && row >= 0 && row + 1 < m;
  const bool ok0 = row >= 0 && row < m, ok1 = row + 1 >= 0 && row + 1 < m;
  auto load_batch = [&](int base) {
#pragma unroll
    for (int u = 0; u < NU; u++) {
      int cc = base + wave + 4 * u;
      bool ok = (cc < c1);
      r0[u] = 0.0;
      r1[u] = 0.0;
      if (ok) {
        if (pair_ok) {
          double2 t = *reinterpret_cast<const double2 *>(a + (size_t)cc * n);
          r0[u] = t.x;
          r1[u] = t.y;
        } else {
          if (ok0) r0[u] = a[(size_t)cc * n];
          if (ok1) r1[u] = a[(size_t)cc * n + 1];
        }
      }
    }
  };
  load_batch(c0);  // the first batch travels during the Householder prologue
  // ---- Householder vector of the (already updated) column i ----
  const double *x = b.col[blk] + i + 1;
  {
    const int np = (m + 1 + 63) / 64;
    double t = 0.0;
    for (int k = tid; k < np; k += 256) t += b.normp[blk][k];
    t = wave_sum(t);
    if (lane == 0) nsum[wave] = t;
  }
  // x on the column chunk (two values per thread up to 512 columns, the rest in the loop below) and on the rows
  double xc[2];
  for (int q = 0; q < 2; q++) {
    int k = c0 + tid + 256 * q;
    xc[q] = (k < c1) ? x[k] : 0.0;
  }
  const int rrow = rs * 128 - delta + tid;  // vrow[t] = v[rs 128 - delta + t], t < 129
  double xr = (tid < 129 && rrow >= 0 && rrow < m) ? x[rrow] : 0.0, xr2 = 0.0;
  if (tid == 0) {
    const int r2 = rs * 128 - delta + 128;
    xr2 = (r2 >= 0 && r2 < m) ? x[r2] : 0.0;
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
  for (int q = 0; q < 2; q++) {
    int k = c0 + tid + 256 * q;
    if (k < c1) vcol[k - c0] = (k == 0) ? 1.0 : xc[q] * scale;
  }
  for (int k = c0 + tid + 512; k < c1; k += 256) vcol[k - c0] = x[k] * scale;
  if (tid < 128) vrow[tid] = (rrow == 0) ? 1.0 : xr * scale;  // rows outside the block: 0 (xr = 0)
  if (tid == 0) vrow[128] = (rs * 128 - delta + 128 == 0) ? 1.0 : xr2 * scale;
  __syncthreads();
  if (blockIdx.x == 0) {
    // panel bookkeeping by the first workgroup: V(:,c) = v, Householder vector stored in A for the back-transformation
    double *Vc = b.V[blk] + (size_t)c * n;
    for (int k = tid; k < n; k += 256) Vc[k] = (k >= i + 1) ? ((k == i + 1) ? 1.0 : x[k - i - 1] * scale) : 0.0;
    for (int k = 1 + tid; k < m; k += 256) A[(size_t)i * n + i + 1 + k] = x[k] * scale;
    if (tid == 0) {
      b.tau[blk][i] = tau;
      b.e[blk][i] = beta;
      b.d[blk][i] = b.col[blk][i];
    }
  }
  double acc0 = 0.0, acc1 = 0.0;
  for (int base = c0; base < c1; base += 4 * NU) {  // slabs wider than 64 columns: very large matrices only
    if (base > c0) load_batch(base);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      int cc = base + wave + 4 * u;
      double vc = (cc < c1) ? vcol[cc - c0] : 0.0;
      acc0 += r0[u] * vc;
      acc1 += r1[u] * vc;
    }
  }
  red[wave * 128 + 2 * lane] = acc0;
  red[wave * 128 + 2 * lane + 1] = acc1;
  __syncthreads();
  if (tid < 128) {
    const int r = rs * 128 + tid - delta;
    const bool okr = (r >= 0 && r < m);
    double p = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    if (okr) b.pp[blk][(size_t)cs * n + r] = p;
    double dv = okr ? p * vrow[tid] : 0.0;
    dv = wave_sum(dv);
    if ((tid & 63) == 0) b.dots[blk][blockIdx.x * 2 + (tid >> 6)] = dv;
  }
  // partial V^T v and W^T v of this row slab (done once per row slab): each wave takes columns cc = wave, wave+4, ...
  if (cs == 0 && c > 0) {
    for (int cc = wave; cc < 2 * c; cc += 4) {
      const double *M = (cc < c) ? b.V[blk] + (size_t)cc * n : b.W[blk] + (size_t)(cc - c) * n;
      double t = 0.0;
      for (int h = 0; h < 2; h++) {
        int r = rs * 128 + h * 64 + lane;
        if (r < m) t += M[i + 1 + r] * vrow[h * 64 + lane + delta];
      }
      t = wave_sum(t);
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
    s = wave_sum(s);
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
    p0 = wave_sum(p0);
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
    nv = wave_sum(nv);
    if (lane == 0) b.normp[blk][rs] = nv;
  }
}

// -------------------------------------------------------------------------------------------------
// One launch per Householder column.
//
// The reflector of column j is v = (x - beta e1)/(alpha - beta) with x the (already updated) column and
// beta = -sign(alpha) |x|.  Everything dlatrd needs from the trailing matrix is linear in v, so the sweep can be
// done with the UNNORMALISED x before |x| is known:
//     q = P x,   P = A22 - V W^T - W V^T  (stale trailing matrix + panel correction),   z = P e1 = next true column
//     p = P v = scale (q - beta z),   v^T p = scale^2 (x^T q - 2 beta q_0 + beta^2 z_0),   scale = 1/(alpha - beta)
//     w = tau p - (tau^2/2)(v^T p) v,     next column x' = z - v w_0 - w  (rows below its diagonal), d' = z_0 - 2 w_0
// Kernel K_i therefore (A) reduces the partial sums column i-1 left behind and finishes that column for the rows
// it needs (its tile's row slab and column slab; redundantly per workgroup, all loads independent), which yields
// x_i there, and (B) sweeps its 128 x 128 tile of the trailing matrix with x_i, leaving partial q, x^T q, |x|^2,
// V^T x, W^T x for K_{i+1}.  No norm, no second launch, no atomics; fixed summation orders.
// -------------------------------------------------------------------------------------------------
// Pointers read from the descriptor are generic to the compiler (flat_load, counted on vmcnt AND lgkmcnt); every
// buffer here is global memory, so the kernel casts them to the global address space once (global_load).
typedef __attribute__((address_space(1))) double gdouble;
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) d2_t gdouble2;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define HFG_G(p) ((gdouble *)(p))
#define HFG_GC(p) ((const gdouble *)(p))

// TF_NTH threads per workgroup.  1024 (one workgroup per CU) is the production shape.  The 512-thread instantiation
// (two per CU, HELFEM_TRDF_NTH=512) exists for A/B runs: measured slower everywhere (18.0 vs 17.4 ms per
// factorisation at 3 x n ~ 1400), also for the early columns whose 3 x 121 tiles need two rounds of 1024-thread
// workgroups (16-18 us per column against 8-9 us once a launch fits the 256 CUs): those columns are bound by what a
// CU can issue, not by occupancy.
template <int TF_NTH>
__global__ __launch_bounds__(TF_NTH) void k_trdf(const TrdBatch *__restrict__ bp, int i, int c, int sweep) {
  constexpr int TF_NG = TF_NTH / 256;  // thread groups sharing a row's panel corrections
  constexpr int TF_NW = TF_NTH / 64;   // waves
  // 512 threads: the panel corrections of a row are split over two thread groups (cc = grp, grp+2, ...), the 128
  // columns of the tile over eight waves.  Load order matters more than thread count here: a workgroup pulls its
  // 128 KB tile through one CU in 2-3 us, so the small phase-A loads are issued first and the tile streams in
  // behind them while the reductions and the scalar algebra run.
  // workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the
  // 128 KB tile that is deliberately still in flight during phase A
#define TRDF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  const TrdBatch &b = *bp;
  const int blk = blockIdx.y;
  const unsigned long long tk0 = wall_clock64();
  const int n = b.n[blk];
  const int j = i - 1;
  const bool has_prev = (c > 0) && (j >= 0) && (j <= n - 3);
  const bool has_cur = (sweep & 1) && (i <= n - 3);
  const int dbg = (sweep >> 1) & 7;  // measurement-only switches of the replay (0 in the factorisation)
  // sweep & 16: symmetric sweep (the whole panel runs in this mode).  Rows and columns share one slab partition
  // (shifted by delta), only the tiles on and below the diagonal are launched, an off-diagonal tile (rs > cs) also
  // forms the transposed product T^T x_rows for its column slab and stores it in partial slot rs: every row still
  // receives exactly one partial per slot, so phase A of the next launch reads the same buffers.  Used while a full
  // grid would need more workgroups than the chip has CUs (two rounds of workgroups per column otherwise).
  const bool symm = (sweep & 16) != 0;
  // dbg & 4: workgroup 0 of block 0 records wall-clock stamps (100 MHz) of its phases into fdots' spare tail
#define TRDF_STAMP(k)                                                                                        \
  if ((dbg & 4) && blockIdx.x == 0 && blk == 0 && threadIdx.x == 0)                                          \
    ((gu64 *)(b.fcp[blk] + (size_t)2 * TF_MAXS * 2 * TB_NB))[(size_t)i * 8 + (k)] = wall_clock64() - tk0;
  if (!has_prev && !has_cur) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int grp = tid >> 8, rid = tid & 255;  // thread group, row slot
  const int m = n - i - 1;                    // rows below the diagonal of column i
  gdouble *A = HFG_G(b.A[blk]);
  gdouble *Vw = HFG_G(b.V[blk]), *Ww = HFG_G(b.W[blk]);
  gdouble *fxw = HFG_G(b.fx[blk]), *fppw = HFG_G(b.fpp[blk]), *fdotsw = HFG_G(b.fdots[blk]), *fxn2w = HFG_G(b.fxn2[blk]),
          *fcpw = HFG_G(b.fcp[blk]);
  gdouble *dw = HFG_G(b.d[blk]), *ew = HFG_G(b.e[blk]), *tauw = HFG_G(b.tau[blk]);
  const int par = i & 1, ppar = par ^ 1;
  const int cp = c - 1;  // panel column of the previous reflector; panel columns cc < cp precede it

  TRDF_STAMP(0)
  // ---- tile of this workgroup ----
  const bool vec2 = ((n & 1) == 0);
  const int delta = (has_cur && vec2) ? ((i + 1) & 1) : 0;  // shift that makes every row pair a 16-byte aligned double2
  int rs, cs, ncs, nrt;
  if (has_cur && symm) {
    nrt = ncs = (m + delta + TF_T - 1) / TF_T;
    rs = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while (rs * (rs + 1) / 2 > (int)blockIdx.x) rs--;
    while ((rs + 1) * (rs + 2) / 2 <= (int)blockIdx.x) rs++;
    cs = blockIdx.x - rs * (rs + 1) / 2;
    if (rs >= nrt) return;
  } else if (has_cur) {
    ncs = (m + TF_T - 1) / TF_T;
    nrt = (m + 1 + TF_T - 1) / TF_T;
    rs = blockIdx.x / ncs;
    cs = blockIdx.x % ncs;
    if (rs >= nrt) return;
  } else {  // finishing only: row slabs over g >= i
    ncs = 1;
    nrt = (m + 1 + TF_T - 1) / TF_T;
    rs = blockIdx.x;
    cs = 0;
    if (rs >= nrt) return;
  }
  const int gR0 = has_cur ? (i + 1 + TF_T * rs - delta) : (i + TF_T * rs);  // first row of the row slab
  const int cshift = (has_cur && symm) ? delta : 0;                         // symmetric mode: one partition for both
  const int gC0 = i + 1 + TF_T * cs - cshift;                               // first row/col of the column slab

  __shared__ double xR[TF_T], xC[TF_T], vC[TF_T], wC[TF_T];
  __shared__ double red[TF_NW * TF_T];
  __shared__ double qpart[TF_NG][2 * TF_T], zpart[TF_NG][2 * TF_T];
  __shared__ double sVtX[TB_NB], sWtX[TB_NB], sVi[TB_NB], sWi[TB_NB];
  __shared__ double sred[64];
  __shared__ double scal[8];  // 0 beta 1 tau 2 scale 3 pv 4 w_i0

  constexpr int NU = TF_T / TF_NW;  // columns per wave
  double r0[NU], r1[NU];
  const int row = gR0 - (i + 1) + 2 * lane;  // local row (relative to i+1) of this lane's pair
  // (B, early) the loads of the trailing-matrix tile are independent of everything else; they are issued BEHIND the
  // small phase-A loads of ALL waves (after the first barrier, see there) so that those return first
  auto issue_tile = [&]() {
    if (!has_cur) return;
    // 32-bit element offsets from the (uniform) base pointer: n^2 < 2^31 for every n the launcher admits, and the
    // saddr + voffset form of global_load needs two VALU instructions per load instead of 64-bit multiply-adds
    const unsigned a0 = (unsigned)(i + 1) * (unsigned)n + (unsigned)(i + 1) + (unsigned)row;  // row = -1 wraps by one
    const bool ok0 = row >= 0 && row < m, ok1 = row + 1 >= 0 && row + 1 < m;
    if (vec2) {
      // one 16-byte load per column for every lane that owns at least one valid row.  The pair may straddle the edge
      // of the trailing block (row -1 above it, row m below it): those 8 bytes are still inside the matrix buffer
      // (the callers allocate two spare doubles behind the last column) and are discarded.  A single load
      // instruction per register pair: mixing 8- and 16-byte variants makes the compiler drain vmcnt between them.
      // Unconditional loads with clamped indices (lanes below the block re-read a pair near its top, columns right of
      // it re-read the last column; both are masked later): loads inside divergent branches -- or selects between
      // two addresses, which the compiler turns into branches -- make its vmcnt bookkeeping fall back to vmcnt(0)
      // at the next use of ANY load.
      const int rowc = (row < m) ? row : (row & 1);  // same parity (alignment); row = -1 and row = m-1 stay as they are
      const unsigned a1 = (unsigned)(i + 1) * (unsigned)n + (unsigned)(i + 1) + (unsigned)rowc;
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int cc = max(0, min(TF_T * cs - cshift + wave * NU + u, m - 1));  // local column, clamped
        d2_t t = *(const gdouble2 *)(A + (a1 + (unsigned)cc * (unsigned)n));
        r0[u] = t.x;  // rows outside the block are masked after the FMAs, x is zero on the clamped columns
        r1[u] = t.y;
      }
    } else {
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int cc = TF_T * cs - cshift + wave * NU + u;
        const unsigned off = a0 + (unsigned)max(cc, 0) * (unsigned)n;
        r0[u] = (ok0 && cc >= 0 && cc < m) ? A[off] : 0.0;
        r1[u] = (ok1 && cc >= 0 && cc < m) ? A[off + 1u] : 0.0;
      }
    }
  };

  // ---- (A) finish column j = i-1 on the rows this workgroup needs ----
  // row slot rid < 128: row gR0 + rid of the row slab; rid >= 128: row gC0 + rid - 128 of the column slab
  const bool isR = rid < TF_T;
  const int g = isR ? gR0 + rid : gC0 + (rid - TF_T);
  const bool live = (g >= i + 1) && g < n && (isR || (has_cur && (rs != cs || (delta && !symm))));
  double xnew = 0.0, vg = 0.0, wg = 0.0;
  if (has_prev) {
    const int mp = m + 1;                         // rows of column j's reflector, g >= i
    const int pdelta = (symm && vec2) ? (i & 1) : 0;               // delta of K_{i-1} (symmetric mode)
    const int pncs = (mp + pdelta + TF_T - 1) / TF_T;              // column slabs (partial slots) of K_{i-1}
    const int pnrt = symm ? pncs : (mp + 1 + TF_T - 1) / TF_T;     // its row slabs
    const gdouble *ppv = fppw + (size_t)ppar * TF_MAXS * n;
    const gdouble *pdots = fdotsw + (size_t)ppar * TF_MAXS * TF_MAXS;
    const gdouble *pxn2 = fxn2w + (size_t)ppar * TF_MAXS;
    const gdouble *pcp = fcpw + (size_t)ppar * TF_MAXS * 2 * TB_NB;
    const gdouble *px = fxw + (size_t)ppar * n;
    const gdouble *Vb = Vw, *Wb = Ww;
    // per-row loads (all independent): group grp holds the panel columns cc = grp + 4u
    double aii = 0.0, alpha = 0.0;
    if (wave == 0) {  // wave-uniform (all 64 lanes read the same two words): a load under a divergent branch makes the
                      // compiler fall back to vmcnt(0) at the next use of any load
      aii = A[(size_t)i * n + i];
      alpha = px[i];
    }
    constexpr int HB = TB_NB / TF_NG;
    double vv[HB], ww[HB];
    double qraw = 0.0, xg = 0.0, ag = 0.0;
    // sums over the column slabs of K_{i-1}: predicated, fully unrolled loads (a rolled "load, wait, add" loop costs
    // one memory round trip per slab); slabs beyond 12 (n > 1536) take the rolled tail
    constexpr int PU = 12;
    double qp[PU];
    const unsigned ug = (unsigned)g, un = (unsigned)n;
#pragma unroll
    for (int k = 0; k < PU; k++) qp[k] = (live && grp == 0 && k < pncs) ? ppv[(unsigned)k * un + ug] : 0.0;
    if (live && grp == 0) {
      xg = px[ug];
      ag = A[(unsigned)i * un + ug];
    }
#pragma unroll
    for (int u = 0; u < HB; u++) {
      int cc = grp + TF_NG * u;
      bool ok = live && cc < cp && !(dbg & 2);
      vv[u] = ok ? Vb[(unsigned)cc * un + ug] : 0.0;
      ww[u] = ok ? Wb[(unsigned)cc * un + ug] : 0.0;
    }
    // scalar stage (every workgroup, redundantly): reductions of the partials of K_{i-1}
    {
      const int nd = symm ? pnrt * (pnrt + 1) / 2 : pnrt * pncs;  // tiles of K_{i-1}
      double s0 = (tid < nd) ? pdots[tid] : 0.0, s1 = (tid + TF_NTH < nd) ? pdots[tid + TF_NTH] : 0.0;  // x^T A22 x
      double t2 = (tid < pncs) ? pxn2[tid] : 0.0;                 // |x[1:]|^2
      double qi = (tid < pncs) ? ppv[(unsigned)tid * un + (unsigned)i] : 0.0;  // q_raw at row i
      double cpl = 0.0;  // this thread's share of the V^T x, W^T x partials / rows i of V, W (loads issued here)
      double cq[PU];
#pragma unroll
      for (int k2 = 0; k2 < PU; k2++) cq[k2] = 0.0;
      if (tid >= 256 && tid < 256 + 2 * TB_NB) {
        int t = tid - 256, cc = t % TB_NB;
        if (cc < cp) {
#pragma unroll
          for (int k2 = 0; k2 < PU; k2++)
            if (k2 < pncs) cq[k2] = pcp[(unsigned)(k2 * 2 * TB_NB + t)];
          for (int k2 = PU; k2 < pncs; k2++) cpl += pcp[(size_t)k2 * 2 * TB_NB + t];
        }
      } else if (tid >= 384 && tid < 384 + 2 * TB_NB) {
        int t = tid - 384, cc = t % TB_NB;
        if (cc < cp) cpl = (t < TB_NB) ? Vb[(unsigned)cc * un + (unsigned)i] : Wb[(unsigned)cc * un + (unsigned)i];
      }
      TRDF_STAMP(6)
      TRDF_STAMP(7)
      double s = s0 + s1;
      for (int k = tid + 2 * TF_NTH; k < nd; k += TF_NTH) s += pdots[k];
#pragma unroll
      for (int k2 = 0; k2 < PU; k2++) {
        cpl += cq[k2];
        qraw += qp[k2];
      }
      if (live && grp == 0)
        for (int k = PU; k < pncs; k++) qraw += ppv[(size_t)k * n + g];
      s = wave_sum(s);
      t2 = wave_sum(t2);
      qi = wave_sum(qi);
      if (lane == 0) {
        sred[wave] = s;
        if (wave == 0) {
          sred[40] = t2;
          sred[41] = qi;
        }
      }
      if (dbg & 4) {
        if (blockIdx.x == 0 && blk == 0 && threadIdx.x == 0)
          ((gu64 *)(b.fcp[blk] + (size_t)2 * TF_MAXS * 2 * TB_NB))[(size_t)i * 8 + 6] |= ((wall_clock64() - tk0) << 32);
      }
      // waves 4..7: V^T x, W^T x of the panel columns cc < cp and the rows i of V and W
      if (tid >= 256 && tid < 256 + 2 * TB_NB) {
        int t = tid - 256, cc = t % TB_NB;
        if (t < TB_NB) sVtX[cc] = cpl;
        else sWtX[cc] = cpl;
      } else if (tid >= 384 && tid < 384 + 2 * TB_NB) {
        int t = tid - 384, cc = t % TB_NB;
        if (t < TB_NB) sVi[cc] = cpl;
        else sWi[cc] = cpl;
      }
    }
    if ((dbg & 4) && blockIdx.x == 0 && blk == 0 && lane == 0)  // arrival of every wave at the first barrier
      ((gu64 *)(b.fcp[blk] + (size_t)2 * TF_MAXS * 2 * TB_NB))[(size_t)8 * n + (size_t)i * 16 + wave] = wall_clock64() - tk0;
    TRDF_LDS_BARRIER();
    // (B, early) the tile loads are issued HERE, behind the first barrier, not right behind this wave's phase-A loads:
    // the CU serves vector-memory instructions in issue order, and 8 KB of tile requests per wave in front of the
    // small phase-A loads of the waves after it delayed their arrival at the barrier (last group of waves: 3.4 -> 2.8 us
    // after entry; the tile is not needed before the row stage two barriers later and still arrives in time)
    issue_tile();
    TRDF_STAMP(1)
    if (wave == 0) {
      // lanes 0..31 hold one panel column each: three small dot products by shuffles, then lane 0 finishes
      double vi = (lane < TB_NB) ? sVi[lane & (TB_NB - 1)] : 0.0, wi = (lane < TB_NB) ? sWi[lane & (TB_NB - 1)] : 0.0;
      double vx = (lane < TB_NB) ? sVtX[lane & (TB_NB - 1)] : 0.0, wx = (lane < TB_NB) ? sWtX[lane & (TB_NB - 1)] : 0.0;
      double dq = vi * wx + wi * vx, dz = 2.0 * vi * wi, dx = 2.0 * vx * wx;
      dq = wave_sum(dq);
      dz = wave_sum(dz);
      dx = wave_sum(dx);
      if (lane == 0) {
        double xAx = 0.0;
        for (int k = 0; k < TF_NW; k++) xAx += sred[k];
        double xn2 = sred[40];
        double qi = sred[41] - dq;
        double zi = aii - dz, xq = xAx - dx;
        double tau, beta, scale;
        if (xn2 == 0.0) {
          tau = 0.0;
          beta = alpha;
          scale = 0.0;
        } else {
          double nrm = sqrt(alpha * alpha + xn2);
          beta = (alpha >= 0.0) ? -nrm : nrm;
          scale = 1.0 / (alpha - beta);
          tau = (beta - alpha) * (1.0 / beta);
        }
        double pvv = scale * scale * (xq - 2.0 * beta * qi + beta * beta * zi);
        double pi0 = scale * (qi - beta * zi);
        double wi0 = tau * pi0 - 0.5 * tau * tau * pvv;  // v_i = 1
        scal[0] = beta;
        scal[1] = tau;
        scal[2] = scale;
        scal[3] = pvv;
        scal[4] = wi0;
        if (blockIdx.x == 0) {
          ew[j] = beta;
          tauw[j] = tau;
          Vw[(size_t)cp * n + i] = 1.0;
          Vw[(size_t)(2 * TB_NB + cp) * n + i] = 1.0;
          Ww[(size_t)cp * n + i] = wi0;
          if (has_cur) dw[i] = zi - 2.0 * wi0;
        }
      }
    }
    // partial panel corrections of this group
    {
      double qc = 0.0, zc = 0.0;
#pragma unroll
      for (int u = 0; u < HB; u++) {
        int cc = grp + TF_NG * u;  // entries cc >= cp are zero
        qc += vv[u] * sWtX[cc] + ww[u] * sVtX[cc];
        zc += vv[u] * sWi[cc] + ww[u] * sVi[cc];
      }
      qpart[grp][rid] = qc;
      zpart[grp][rid] = zc;
    }
    TRDF_LDS_BARRIER();
    TRDF_STAMP(2)
    const double beta = scal[0], tau = scal[1], scale = scal[2], pvv = scal[3], wi0 = scal[4];
    if (grp == 0 && live) {
      double qs = 0.0, zs = 0.0;
#pragma unroll
      for (int gg = 0; gg < TF_NG; gg++) {
        qs += qpart[gg][rid];
        zs += zpart[gg][rid];
      }
      double q = qraw - qs, z = ag - zs;
      vg = xg * scale;
      double pr = scale * (q - beta * z);
      wg = tau * pr - 0.5 * tau * tau * pvv * vg;
      xnew = z - vg * wi0 - wg;
      // one writer per row: the workgroups of the first column slab
      if (isR && cs == 0) {
        Vw[(size_t)cp * n + g] = vg;
        Vw[(size_t)(2 * TB_NB + cp) * n + g] = vg;  // [V | W | V]: the rank-2NB update is one product
        Ww[(size_t)cp * n + g] = wg;
        A[(size_t)j * n + g] = vg;  // Householder vector for the back-transformation
        if (has_cur) fxw[(size_t)par * n + g] = xnew;
      }
    }
  } else if (has_cur) {
    // first column of a panel: the trailing matrix is up to date
    double xl = 0.0;
    if (grp == 0 && live) xl = A[(size_t)i * n + g];
    issue_tile();
    if (grp == 0 && live) {
      xnew = xl;
      if (isR && cs == 0) fxw[(size_t)par * n + g] = xnew;
    }
    if (blockIdx.x == 0 && tid == 0) dw[i] = A[(size_t)i * n + i];
  }
  if (!has_cur) return;

  // x_i on the row slab and the column slab (and v, w of the previous column on the column slab)
  const bool same = (rs == cs) && (!delta || symm);  // the two slabs are the same rows
  if (grp == 0) {
    if (isR) {
      xR[rid] = live ? xnew : 0.0;
      if (same) {
        xC[rid] = live ? xnew : 0.0;
        vC[rid] = live ? vg : 0.0;
        wC[rid] = live ? wg : 0.0;
      }
    } else if (!same) {
      xC[rid - TF_T] = live ? xnew : 0.0;
      vC[rid - TF_T] = live ? vg : 0.0;
      wC[rid - TF_T] = live ? wg : 0.0;
    }
  }
  TRDF_LDS_BARRIER();
  TRDF_STAMP(3)

  // ---- (B) sweep: partial q over this tile ----
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int u = 0; u < NU; u++) {
    double xc = xC[wave * NU + u];
    acc0 += r0[u] * xc;
    acc1 += r1[u] * xc;
  }
  {
    const bool ok0 = row >= 0 && row < m, ok1 = row + 1 >= 0 && row + 1 < m;
    red[wave * TF_T + 2 * lane] = ok0 ? acc0 : 0.0;
    red[wave * TF_T + 2 * lane + 1] = ok1 ? acc1 : 0.0;
    if (symm && rs > cs) {
      // transposed product of an off-diagonal tile: qT[col] = sum_rows T[row][col] x[row].  The wave's 64 lanes hold all
      // 128 rows of its NU columns: a halving butterfly (NU/2, NU/4, ... values exchanged with lane^32, ^16, ...) leaves
      // one column per lane group, the last log2(64/NU) steps add within the group.
      const double xr0 = ok0 ? xR[2 * lane] : 0.0, xr1 = ok1 ? xR[2 * lane + 1] : 0.0;
      double tv[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) tv[u] = r0[u] * xr0 + r1[u] * xr1;
      // lane bits 5.. select the column: after the NU-halving steps lane group (lane >> (6 - log2 NU)) holds column index
      constexpr int LG = (NU == 8) ? 3 : (NU == 16 ? 4 : 2);
      int width = NU, lmask = 32;
#pragma unroll
      for (int st = 0; st < LG; st++) {
        const int half = width / 2;
        const bool up = (lane & lmask) != 0;  // upper lanes keep the upper half of the values
#pragma unroll
        for (int u = 0; u < NU / 2; u++) {
          if (u < half) {
            // exchanges without the LDS crossbar: permlane swaps across the wave halves / rows, DPP inside a row
            if (lmask == 32) tv[u] = swap32_sum(tv[u], tv[u + half]);
            else if (lmask == 16) tv[u] = swap16_sum(tv[u], tv[u + half]);
            else {
              const double send = up ? tv[u] : tv[u + half];
              const double keep = up ? tv[u + half] : tv[u];
              tv[u] = keep + ((lmask == 8) ? dpp_f64<0x128>(send) : __shfl_xor(send, lmask, 64));  // row_ror:8 = lane ^ 8
            }
          }
        }
        width = half;
        lmask >>= 1;
      }
      // the remaining 6 - LG steps add within the lane group (any order)
      if (LG == 3) tv[0] = oct_sum(tv[0]);
      else if (LG == 4) tv[0] = quad_sum(tv[0]);
      else tv[0] = row16_sum(tv[0]);
      if ((lane & ((64 >> LG) - 1)) == 0) {
        const int u = lane >> (6 - LG);
        const int h = gC0 + wave * NU + u;
        if (h >= i + 1 && h < n) fppw[((size_t)par * TF_MAXS + rs) * n + h] = tv[0];
      }
    }
  }
  __syncthreads();
  TRDF_STAMP(4)
  double dv = 0.0;
  if (tid < TF_T) {
    const int gr = gR0 + tid;
    const bool okr = (gr >= i + 1 && gr < n);
    double pq = 0.0;
#pragma unroll
    for (int w = 0; w < TF_NW; w++) pq += red[w * TF_T + tid];
    if (okr) fppw[((size_t)par * TF_MAXS + cs) * n + gr] = pq;
    dv = okr ? pq * xR[tid] : 0.0;
    if (symm && rs > cs) dv *= 2.0;  // x_r^T T x_c + x_c^T T^T x_r
    dv = wave_sum(dv);
    if (lane == 0) sred[48 + wave] = dv;
  }
  // ---- per column slab (first row slab only): |x[1:]|^2, V^T x, W^T x over the slab ----
  const bool slab_owner = symm ? (rs == cs) : (rs == 0);  // the tile that forms the per-column-slab sums
  if (slab_owner) {
    if (tid >= TF_T && tid < 2 * TF_T) {
      int k = tid - TF_T;
      int h = gC0 + k;
      double xv = (h >= i + 2 && h < n) ? xC[k] : 0.0;
      double s2 = xv * xv;
      s2 = wave_sum(s2);
      if (lane == 0) sred[50 + (wave - 2)] = s2;
    }
    // panel columns cc < c (cc = c-1 is the reflector just finished: v, w of the slab are in LDS).  Wave q takes
    // q2 = q, q+8, ...; all global loads are issued before the first reduction.
    if (c > 0 && !(dbg & 1)) {
      constexpr int NQ = 2 * TB_NB / TF_NW;
      double m0[NQ], m1[NQ];
#pragma unroll
      for (int u = 0; u < NQ; u++) {
        int q2 = wave + TF_NW * u;
        m0[u] = 0.0;
        m1[u] = 0.0;
        if (q2 < 2 * c) {
          int cc = (q2 < c) ? q2 : q2 - c;
          bool isV = q2 < c;
          if (cc != c - 1) {
            const gdouble *M = (isV ? Vw : Ww) + (size_t)cc * n;
            int h0 = gC0 + lane, h1 = gC0 + 64 + lane;
            if (h0 < n) m0[u] = M[h0];
            if (h1 < n) m1[u] = M[h1];
          } else {
            m0[u] = isV ? vC[lane] : wC[lane];
            m1[u] = isV ? vC[64 + lane] : wC[64 + lane];
          }
        }
      }
      const double x0 = xC[lane], x1 = xC[64 + lane];
#pragma unroll
      for (int u = 0; u < NQ; u++) {
        int q2 = wave + TF_NW * u;
        if (q2 < 2 * c) {  // wave-uniform
          double t = m0[u] * x0 + m1[u] * x1;
          t = wave_sum(t);
          int cc = (q2 < c) ? q2 : q2 - c;
          if (lane == 0) fcpw[((size_t)par * TF_MAXS + cs) * 2 * TB_NB + ((q2 < c) ? cc : TB_NB + cc)] = t;
        }
      }
    }
  }
  __syncthreads();
  TRDF_STAMP(5)
  if (tid == 0) {
    fdotsw[(size_t)par * TF_MAXS * TF_MAXS + blockIdx.x] = sred[48] + sred[49];
    if (slab_owner) fxn2w[(size_t)par * TF_MAXS + cs] = sred[50] + sred[51];
  }
#undef TRDF_STAMP
#undef TRDF_LDS_BARRIER
}

// Tail of the factorisation: once the trailing matrix of every block fits one workgroup's LDS (order <= TT_MAX) the
// remaining columns are reduced by the unblocked algorithm (LAPACK dsytd2 algebra: v, p = tau A22 v,
// w = p - tau/2 (p^T v) v, A22 -= v w^T + w v^T) inside ONE launch per batch: a column then costs a few workgroup
// barriers (~1.5 us) instead of a dependent launch (~8 us).  Same reflector convention as k_trdf (beta = -sign(alpha)|x|,
// v_1 = 1 implicit, v stored below the subdiagonal of column j, d/e/tau), fixed summation orders.
constexpr int TT_MAX = 128;
constexpr int TT_LD = TT_MAX + 1;
constexpr int TT_CG = 8;  // column groups: thread (r, cg) owns the columns c = cg, cg + 8, ... of row r
__global__ __launch_bounds__(1024) void k_trd_tail(const TrdBatch *__restrict__ bp, int j0) {
  extern __shared__ double sm[];  // S[m][TT_LD] | v[TT_MAX] | w[TT_MAX] | part[TT_CG][TT_MAX] | red[32]
  const TrdBatch &b = *bp;
  const int blk = blockIdx.x;
  const int n = b.n[blk];
  const int m = n - j0;  // order of the trailing matrix (rows/cols j0 .. n-1)
  if (m < 3 || m > TT_MAX) return;
  double *S = sm, *v = sm + TT_MAX * TT_LD, *w = v + TT_MAX, *part = w + TT_MAX, *red = part + TT_CG * TT_MAX;
  gdouble *A = HFG_G(b.A[blk]);
  const int tid = threadIdx.x, r = tid & (TT_MAX - 1), cg = tid >> 7, wave = tid >> 6, lane = tid & 63;
  for (int c = cg; c < m; c += TT_CG)
    if (r < m) S[c * TT_LD + r] = A[(size_t)(j0 + c) * n + j0 + r];
  __syncthreads();
  for (int k = 0; k <= m - 3; k++) {
    const int j = j0 + k;
    const int k1 = k + 1;  // first row/col of the trailing block of this step
    // |x[1:]|^2 of the column below the subdiagonal element
    double sq = 0.0;
    if (tid < TT_MAX && tid > k1 && tid < m) {
      double xv = S[k * TT_LD + tid];
      sq = xv * xv;
    }
    sq = wave_sum(sq);
    if (lane == 0 && wave < 2) red[wave] = sq;
    __syncthreads();
    double tau, beta, scale;
    {
      const double xn2 = red[0] + red[1];
      const double alpha = S[k * TT_LD + k1];
      if (xn2 == 0.0) {
        tau = 0.0;
        beta = alpha;
        scale = 0.0;
      } else {
        const double nrm = sqrt(alpha * alpha + xn2);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
    }
    if (tid < TT_MAX) {
      double vv = 0.0;
      if (tid == k1) vv = 1.0;
      else if (tid > k1 && tid < m) vv = S[k * TT_LD + tid] * scale;
      v[tid] = vv;
      if (tid > k1 && tid < m) A[(size_t)j * n + j0 + tid] = vv;  // reflector for the back-transformation
    }
    if (tid == 0) {
      b.d[blk][j] = S[k * TT_LD + k];
      b.e[blk][j] = beta;
      b.tau[blk][j] = tau;
    }
    __syncthreads();
    // p = tau S22 v: partial sums over this thread's columns, then over the column groups
    {
      double acc = 0.0;
      if (r >= k1 && r < m)
        for (int c = k1 + ((cg - k1) % TT_CG + TT_CG) % TT_CG; c < m; c += TT_CG) acc += S[c * TT_LD + r] * v[c];
      part[cg * TT_MAX + r] = acc;
    }
    __syncthreads();
    double pv = 0.0;
    if (tid < TT_MAX) {
      double pr = 0.0;
      if (tid >= k1 && tid < m) {
#pragma unroll
        for (int g = 0; g < TT_CG; g++) pr += part[g * TT_MAX + tid];
        pr *= tau;
      }
      w[tid] = pr;  // p for now
      pv = pr * v[tid];
    }
    pv = wave_sum(pv);
    if (lane == 0 && wave < 2) red[2 + wave] = pv;
    __syncthreads();
    if (tid < TT_MAX) {
      const double a2 = -0.5 * tau * (red[2] + red[3]);
      w[tid] += a2 * v[tid];
    }
    __syncthreads();
    // S22 -= v w^T + w v^T
    if (r >= k1 && r < m) {
      const double vr = v[r], wr = w[r];
      for (int c = k1 + ((cg - k1) % TT_CG + TT_CG) % TT_CG; c < m; c += TT_CG) S[c * TT_LD + r] -= vr * w[c] + wr * v[c];
    }
    __syncthreads();
  }
  // the last 2 x 2 block goes back to the matrix for k_trdb_finish
  if (tid < 4) {
    const int rr = m - 2 + (tid & 1), cc = m - 2 + (tid >> 1);
    A[(size_t)(j0 + cc) * n + j0 + rr] = S[cc * TT_LD + rr];
  }
}

// Register-resident variant of the tail for orders up to TR_MAX = 192: the trailing matrix lives in the register file of
// ONE workgroup (768 threads; thread (rg, cg) holds the 4 x 12 elements of rows 4 rg .. 4 rg + 3 and columns
// cg, cg + 16, ...: 96 VGPRs), only vectors go through LDS.  A 4 x 12 register tile needs 12 + 24 LDS reads of v / p per
// column where a thread that owns a strided set of one row needs 144 (the LDS port, 2 clk per ds_read_b64 per wave, was
// the bound of that layout); the partial products of a row are summed over the 16 lanes of its column groups by DPP
// shuffles, so a column costs three workgroup barriers.  Same algebra, reflector convention and outputs as k_trd_tail.
constexpr int TR_MAX = 192;
constexpr int TR_RT = 4;                        // rows per thread
constexpr int TR_NCG = 16;                      // column groups
constexpr int TR_NU = TR_MAX / TR_NCG;          // columns per thread
constexpr int TR_NTH = (TR_MAX / TR_RT) * TR_NCG;  // 768
constexpr int TR_NW = TR_NTH / 64;
__global__ __launch_bounds__(TR_NTH) void k_trd_tail_reg(const TrdBatch *__restrict__ bp, int j0) {
  __shared__ double xs[TR_MAX], vs[TR_MAX], ps[TR_MAX], red[TR_NW], red2[TR_NW];
  const TrdBatch &b = *bp;
  const int blk = blockIdx.x;
  const int n = b.n[blk];
  const int m = n - j0;  // order of the trailing matrix (rows/cols j0 .. n-1)
  if (m < 3 || m > TR_MAX) return;
  gdouble *A = HFG_G(b.A[blk]);
  const int tid = threadIdx.x, cg = tid & (TR_NCG - 1), rg = tid >> 4, wave = tid >> 6, lane = tid & 63;
  const int r0 = TR_RT * rg;
  double a[TR_RT][TR_NU];
#pragma unroll
  for (int u = 0; u < TR_NU; u++) {
    const int c = cg + TR_NCG * u;
#pragma unroll
    for (int i = 0; i < TR_RT; i++) a[i][u] = (c < m && r0 + i < m) ? A[(size_t)(j0 + c) * n + j0 + r0 + i] : 0.0;
  }
  for (int k = 0; k <= m - 3; k++) {
    const int j = j0 + k;
    const int k1 = k + 1;  // first row/col of the trailing block of this step
    const int kc = k & (TR_NCG - 1), ku = k >> 4;
    // ---- column k from the registers of the threads that own it ----
    double xv[TR_RT] = {0.0, 0.0, 0.0, 0.0};
    double sq = 0.0;
    if (cg == kc) {
#pragma unroll
      for (int u = 0; u < TR_NU; u++)
        if (u == ku) {
#pragma unroll
          for (int i = 0; i < TR_RT; i++) xv[i] = a[i][u];
        }
#pragma unroll
      for (int i = 0; i < TR_RT; i++) {
        const int r = r0 + i;
        xs[r] = xv[i];
        if (r > k1 && r < m) sq += xv[i] * xv[i];
        if (r == k) b.d[blk][j] = xv[i];
      }
    }
    sq = wave_sum(sq);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    double tau, beta, scale;
    {
      double xn2 = 0.0;
#pragma unroll
      for (int q = 0; q < TR_NW; q++) xn2 += red[q];
      const double alpha = xs[k1];
      if (xn2 == 0.0) {
        tau = 0.0;
        beta = alpha;
        scale = 0.0;
      } else {
        const double nrm = sqrt(alpha * alpha + xn2);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
    }
    if (cg == kc) {
#pragma unroll
      for (int i = 0; i < TR_RT; i++) {
        const int r = r0 + i;
        double vv = 0.0;
        if (r == k1) vv = 1.0;
        else if (r > k1 && r < m) vv = xv[i] * scale;
        vs[r] = vv;
        if (r > k1 && r < m) A[(size_t)j * n + j0 + r] = vv;  // reflector for the back-transformation
      }
    }
    if (tid == 0) {
      b.e[blk][j] = beta;
      b.tau[blk][j] = tau;
    }
    __syncthreads();
    // ---- p = tau S22 v ----
    double acc[TR_RT] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < TR_NU; u++) {
      if (TR_NCG * u + TR_NCG - 1 < k1) continue;  // columns already reduced (workgroup-uniform)
      const double vc = vs[cg + TR_NCG * u];      // 0 for c < k1 and c >= m
#pragma unroll
      for (int i = 0; i < TR_RT; i++) acc[i] += a[i][u] * vc;
    }
    double vr[TR_RT], pr[TR_RT];
    double pv = 0.0;
#pragma unroll
    for (int i = 0; i < TR_RT; i++) {
      const double t = row16_sum(acc[i]);
      const int r = r0 + i;
      vr[i] = vs[r];
      pr[i] = (r >= k1 && r < m) ? tau * t : 0.0;
      pv += pr[i] * vr[i];
    }
    if (cg == 0) {
#pragma unroll
      for (int i = 0; i < TR_RT; i++) ps[r0 + i] = pr[i];
    } else {
      pv = 0.0;
    }
    pv = wave_sum(pv);
    if (lane == 0) red2[wave] = pv;
    __syncthreads();
    double a2 = 0.0;
#pragma unroll
    for (int q = 0; q < TR_NW; q++) a2 += red2[q];
    a2 *= -0.5 * tau;
    // ---- S22 -= v w^T + w v^T,  w = p + a2 v:  S_rc -= v_r p_c + (a2 v_r + w_r) v_c ----
    double cf[TR_RT];
#pragma unroll
    for (int i = 0; i < TR_RT; i++) cf[i] = 2.0 * a2 * vr[i] + pr[i];
#pragma unroll
    for (int u = 0; u < TR_NU; u++) {
      if (TR_NCG * u + TR_NCG - 1 < k1) continue;
      const double pc = ps[cg + TR_NCG * u], vc = vs[cg + TR_NCG * u];
#pragma unroll
      for (int i = 0; i < TR_RT; i++) a[i][u] -= vr[i] * pc + cf[i] * vc;
    }
  }
  // the last 2 x 2 block goes back to the matrix for k_trdb_finish
#pragma unroll
  for (int u = 0; u < TR_NU; u++) {
    const int c = cg + TR_NCG * u;
#pragma unroll
    for (int i = 0; i < TR_RT; i++) {
      const int r = r0 + i;
      if (c >= m - 2 && c < m && r >= m - 2 && r < m) A[(size_t)(j0 + c) * n + j0 + r] = a[i][u];
    }
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

/// order from which the remaining columns are reduced inside one launch: HELFEM_TRD_TAIL = 0 none, 1 the LDS-resident
/// kernel (128), 2 (default) the register-resident kernel (192)
static int trd_tail_order() {
  static const int mode = getenv("HELFEM_TRD_TAIL") ? atoi(getenv("HELFEM_TRD_TAIL")) : 2;
  return mode == 0 ? 0 : (mode == 1 ? TT_MAX : TR_MAX);
}

// Switch-over tile count of the symmetric sweep: panels whose full grid has more tiles than this run in symmetric mode.
// Default 0 = every panel: measured (bench workload, HELFEM_TRDF_SYM_MIN = 16 ... 300 and "always") the step time does
// not depend on it once the panels that exceed the CU count are symmetric, and sweeping one triangle everywhere halves
// the sweep's HBM traffic (PMC: 14.4 MB -> what the algorithm needs, one triangle plus the diagonal tiles).
static int trd_sym_min_tiles(hfg_ctx *) {
  static const int v = getenv("HELFEM_TRDF_SYM_MIN") ? atoi(getenv("HELFEM_TRDF_SYM_MIN")) : 0;
  return v;
}

struct TrdWork {
  DevBuf<double> V[TB_MAXB], col[TB_MAXB], normp[TB_MAXB], pp[TB_MAXB], dots[TB_MAXB], cpart[TB_MAXB];
  DevBuf<double> fx[TB_MAXB], fpp[TB_MAXB], fdots[TB_MAXB], fxn2[TB_MAXB], fcp[TB_MAXB];
  DevBuf<GemmTask> ptasks;  // panel updates of the fused variant, [panel][block]
  std::vector<GemmTask> h_ptasks;  // host copies of what ptasks / desc hold (upload_cached)
  std::vector<TrdBatch> h_desc;
  bool last_fused = false;
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

static thread_local bool g_in_chain_fallback = false;  // the chain below is running on what the persistent path left
/// A[blk] (n x n, ld n, full symmetric storage) -> d, e, tau and the Householder vectors below the subdiagonal
void tridiagonalize_batch(hfg_ctx *ctx, int nblk, const int *ns, double *const *A, double *const *d, double *const *e,
                          double *const *tau) {
  if (nblk > TB_MAXB) throw std::logic_error("tridiagonalize_batch: too many blocks");
  // one cooperative launch with the matrices resident in the register file when the batch fits the chip (trdp.hip);
  // the chain of launches below otherwise
  if (!g_in_chain_fallback) {
    std::vector<char> done;
    tridiagonalize_persistent(ctx, nblk, ns, A, d, e, tau, done);
    int ndone = 0;
    for (char c : done) ndone += c ? 1 : 0;
    if (ndone > 0) {
      auto itp = g_trd.find(ctx);
      if (itp != g_trd.end()) itp->second->last_ns.clear();
      if (ndone == nblk) return;
      // the matrices the persistent path left (order beyond its register tiles, a refused launch) go through the chain
      std::vector<int> ns2;
      std::vector<double *> A2, d2, e2, t2;
      for (int i = 0; i < nblk; i++)
        if (!done[i]) {
          ns2.push_back(ns[i]);
          A2.push_back(A[i]);
          d2.push_back(d[i]);
          e2.push_back(e[i]);
          t2.push_back(tau[i]);
        }
      g_in_chain_fallback = true;
      try {
        tridiagonalize_batch(ctx, (int)ns2.size(), ns2.data(), A2.data(), d2.data(), e2.data(), t2.data());
      } catch (...) {
        g_in_chain_fallback = false;
        throw;
      }
      g_in_chain_fallback = false;
      return;
    }
  }
  TrdWork *wp;
  auto it = g_trd.find(ctx);
  if (it == g_trd.end()) {
    wp = new TrdWork();
    g_trd[ctx] = wp;
  } else
    wp = it->second;
  TrdWork &w = *wp;
  TrdBatch b{};
  int nmax = 0;
  for (int i = 0; i < nblk; i++) {
    int n = ns[i];
    nmax = std::max(nmax, n);
    int nslab = (n + 63) / 64 + 1;
    w.V[i].resize((size_t)n * 3 * TB_NB);  // [V | W | V]
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
    b.W[i] = w.V[i].p + (size_t)n * TB_NB;
    b.col[i] = w.col[i].p;
    b.normp[i] = w.normp[i].p;
    b.pp[i] = w.pp[i].p;
    b.dots[i] = w.dots[i].p;
    b.cpart[i] = w.cpart[i].p;
    w.fx[i].resize((size_t)2 * n);
    w.fpp[i].resize((size_t)2 * TF_MAXS * n);
    w.fdots[i].resize((size_t)2 * TF_MAXS * TF_MAXS);
    w.fxn2[i].resize((size_t)2 * TF_MAXS);
    w.fcp[i].resize((size_t)2 * TF_MAXS * 2 * TB_NB + (size_t)24 * n + 64);  // + phase stamps of the measurement replay (8 per column, then 16 per-wave arrival times per column)
    b.fx[i] = w.fx[i].p;
    b.fpp[i] = w.fpp[i].p;
    b.fdots[i] = w.fdots[i].p;
    b.fxn2[i] = w.fxn2[i].p;
    b.fcp[i] = w.fcp[i].p;
  }
  hipStream_t s = ctx->stream;
  upload_cached(w.desc, w.h_desc, std::vector<TrdBatch>(1, b), s);
  const TrdBatch *db = w.desc.p;
  // k_trdb_gemv: v on a column chunk (at most 64 chunks per column) and on 129 rows, 4 x 128 partial sums
  size_t shb = (size_t)(std::max(64, (nmax + 63) / 64 + 1) + 136 + 4 * 128 + 8) * sizeof(double);
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_trdb_gemv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  // HELFEM_TRD=twokernel keeps the earlier two-launches-per-column variant (k_trdb_gemv + k_trdb_w)
  static const bool twokernel = (getenv("HELFEM_TRD") && !strcmp(getenv("HELFEM_TRD"), "twokernel"));
  const bool fused = !twokernel && nmax <= TF_T * (TF_MAXS - 1);
  w.last_fused = fused;
  // fused variant: the trailing updates A22 -= [V|W][W|V]^T of all blocks of a full panel are one task-list launch
  const int npanel = (nmax - 3) / TB_NB + 1;
  static const int force_sym0 = getenv("HELFEM_TRDF_SYM") ? atoi(getenv("HELFEM_TRDF_SYM")) : -1;
  static const bool band_off = getenv("HELFEM_TRD_BAND_UPDATE") && atoi(getenv("HELFEM_TRD_BAND_UPDATE")) == 0;
  static const bool acc128 = getenv("HELFEM_ACC_TILE") && atoi(getenv("HELFEM_ACC_TILE")) == 128;
  const bool band_update = !band_off && !acc128 && (force_sym0 >= 0 ? force_sym0 != 0 : trd_sym_min_tiles(ctx) == 0);
  if (fused) {
    std::vector<GemmTask> pt((size_t)npanel * nblk);
    for (int pi = 0; pi < npanel; pi++)
      for (int k = 0; k < nblk; k++) {
        GemmTask g;
        g.A = g.B = nullptr;
        g.C = nullptr;
        g.M = g.N = g.K = 0;
        g.lda = g.ldb = g.ldc = 1;
        const int n = ns[k], j0 = pi * TB_NB;
        const int ncols = std::min(TB_NB, std::max(0, n - 2 - j0));
        const int j1 = j0 + ncols, mt = n - j1;
        if (ncols == TB_NB && mt > 0) {
          g.A = w.V[k].p + j1;                          // [V | W], rows j1..
          g.B = w.V[k].p + (size_t)n * TB_NB + j1;      // [W | V]
          g.C = A[k] + (size_t)j1 * n + j1;
          g.M = g.N = mt;
          g.K = 2 * TB_NB;
          g.lda = g.ldb = g.ldc = n;
          g.tB = 1;
          g.alpha = -1.0;
          g.beta = 1.0;
          // every later sweep is symmetric: it reads the tiles on and below the diagonal of its own 128-partition, whose
          // origin moves by up to 17 within a panel -- at most 144 columns right of the diagonal, i.e. inside the band of
          // three 64-tile rows that the update keeps; so does the 192 x 192 block handed to the tail kernel
          g.sym = band_update ? 2 : 0;
        }
        pt[(size_t)pi * nblk + k] = g;
      }
    upload_cached(w.ptasks, w.h_ptasks, pt, s);
  }
  // LDS-resident tail (k_trd_tail) from the first panel boundary where every trailing matrix has order <= TT_MAX
  const int tail_max = trd_tail_order();
  int j_tail = nmax;  // no tail
  if (fused && tail_max > 0 && nmax >= 3) {
    j_tail = std::max(0, ((nmax - tail_max + TB_NB - 1) / TB_NB) * TB_NB);
    if (nmax - j_tail < 3) j_tail = nmax;
  }
  for (int j0 = 0; j0 <= nmax - 3 && j0 < j_tail; j0 += TB_NB) {
    if (fused) {
      const int jend = std::min(j0 + TB_NB, nmax - 2);
      // symmetric sweep (lower tiles only) for the panels whose full grid would not fit the chip in one round
      // (HELFEM_TRDF_SYM: 0 never, 1 always, default by size)
      static const int force_sym = getenv("HELFEM_TRDF_SYM") ? atoi(getenv("HELFEM_TRDF_SYM")) : -1;
      bool symm;
      {
        const int m0 = nmax - j0 - 1;
        const long full = (long)((m0 + 1 + TF_T - 1) / TF_T) * std::max(1, (m0 + TF_T - 1) / TF_T) * nblk;
        symm = (force_sym >= 0) ? (force_sym != 0) : (full > trd_sym_min_tiles(ctx));
      }
      for (int i = j0; i <= jend; i++) {
        const int sweep = ((i < jend) ? 1 : 0) | (symm ? 16 : 0);  // the last launch of the panel only finishes column jend-1
        const int m = nmax - i - 1;
        const int nrt = (m + 1 + TF_T - 1) / TF_T, ncs = std::max(1, (m + TF_T - 1) / TF_T);
        int grid = (sweep & 1) ? nrt * ncs : nrt;
        if (symm && (sweep & 1)) {
          const int delta = ((nmax & 1) == 0) ? ((i + 1) & 1) : 0;  // largest block; smaller blocks need no more tiles
          const int nt = (m + 1 + TF_T - 1) / TF_T;                 // >= ceil((m + delta)/T) for either parity of n
          grid = nt * (nt + 1) / 2;
          (void)delta;
        }
        static const int force_nth = getenv("HELFEM_TRDF_NTH") ? atoi(getenv("HELFEM_TRDF_NTH")) : 0;  // A/B runs
        const bool small_wg = (force_nth == 512);
        if (small_wg) hipLaunchKernelGGL(k_trdf<512>, dim3(grid, nblk), dim3(512), 0, s, db, i, i - j0, sweep);
        else hipLaunchKernelGGL(k_trdf<1024>, dim3(grid, nblk), dim3(1024), 0, s, db, i, i - j0, sweep);
      }
    } else {
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
    }
    // trailing update per block: columns processed in this panel for block k: j0 .. min(j0+NB, n_k-2)-1
    if (fused) {
      const int mt = nmax - j0 - TB_NB;
      if (mt > 0) {
        const GemmTask *pt = w.ptasks.p + (size_t)(j0 / TB_NB) * nblk;
        static const int acc_tile = getenv("HELFEM_ACC_TILE") ? atoi(getenv("HELFEM_ACC_TILE")) : 0;  // A/B runs: 64 or 128
        gemm_tasklist_acc_dev(ctx, pt, nblk, mt, mt, acc_tile != 128);
      }
    }
    for (int k = 0; k < nblk; k++) {
      int n = ns[k];
      int ncols = std::min(TB_NB, std::max(0, n - 2 - j0));
      if (ncols <= 0) continue;
      if (fused && ncols == TB_NB) continue;  // done above
      int j1 = j0 + ncols;
      int mt = n - j1;
      if (mt <= 0) continue;
      double *A22 = A[k] + (size_t)j1 * n + j1;
      gemm_dev(ctx, false, true, mt, mt, ncols, -1.0, w.V[k].p + j1, n, w.V[k].p + (size_t)n * TB_NB + j1, n, 1.0, A22, n);
      gemm_dev(ctx, false, true, mt, mt, ncols, -1.0, w.V[k].p + (size_t)n * TB_NB + j1, n, w.V[k].p + j1, n, 1.0, A22, n);
    }
  }
  if (j_tail < nmax && tail_max == TR_MAX) {
    hipLaunchKernelGGL(k_trd_tail_reg, dim3(nblk), dim3(TR_NTH), 0, s, db, j_tail);
  } else if (j_tail < nmax) {
    const size_t sht = (size_t)(TT_MAX * TT_LD + 2 * TT_MAX + TT_CG * TT_MAX + 32) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_trd_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sht));
      attr_set = true;
    }
    hipLaunchKernelGGL(k_trd_tail, dim3(nblk), dim3(1024), sht, s, db, j_tail);
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
  // HELFEM_TRD_GRAPH=1 (measurement only): the same launches captured into a hipGraph on a private stream and replayed
  // as ONE graph launch -- does the command processor chain dependent kernels faster than a stream of launches?
  static const bool use_graph = getenv("HELFEM_TRD_GRAPH") && atoi(getenv("HELFEM_TRD_GRAPH")) != 0;
  hipStream_t cap = nullptr;
  if (use_graph) {
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    HFG_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    HFG_HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    s = cap;
  }
  const TrdBatch *db = w.desc.p;
  // k_trdb_gemv: v on a column chunk (at most 64 chunks per column) and on 129 rows, 4 x 128 partial sums
  size_t shb = (size_t)(std::max(64, (nmax + 63) / 64 + 1) + 136 + 4 * 128 + 8) * sizeof(double);
  hipEvent_t e0, e1;
  HFG_HIP_CHECK(hipEventCreate(&e0));
  HFG_HIP_CHECK(hipEventCreate(&e1));
  if (!use_graph) HFG_HIP_CHECK(hipEventRecord(e0, s));
  int count = 0;
  int j_tail = nmax;  // the columns of the LDS-resident tail are not launches of this kernel
  if (w.last_fused && trd_tail_order() > 0 && nmax >= 3) {
    j_tail = std::max(0, ((nmax - trd_tail_order() + TB_NB - 1) / TB_NB) * TB_NB);
    if (nmax - j_tail < 3) j_tail = nmax;
  }
  for (int i = 0; i <= nmax - 3 && i < j_tail; i++) {
    const int c = i % TB_NB;
    const int m = nmax - i - 1;
    if (w.last_fused) {
      const int nrt = (m + 1 + TF_T - 1) / TF_T, ncs = std::max(1, (m + TF_T - 1) / TF_T);
      const int cfix = getenv("HELFEM_TRDF_C") ? atoi(getenv("HELFEM_TRDF_C")) : c;
      const int dbg = getenv("HELFEM_TRDF_DBG") ? atoi(getenv("HELFEM_TRDF_DBG")) : 0;
      static const int force_nth = getenv("HELFEM_TRDF_NTH") ? atoi(getenv("HELFEM_TRDF_NTH")) : 0;
      static const int force_sym = getenv("HELFEM_TRDF_SYM") ? atoi(getenv("HELFEM_TRDF_SYM")) : -1;
      const bool small_wg = (force_nth == 512);
      bool symm;
      {
        const int m0 = nmax - (i - c) - 1;  // the panel's first column decides, as in the factorisation
        const long full = (long)((m0 + 1 + TF_T - 1) / TF_T) * std::max(1, (m0 + TF_T - 1) / TF_T) * nblk;
        symm = (force_sym >= 0) ? (force_sym != 0) : (full > trd_sym_min_tiles(ctx));
      }
      const int grid = symm ? nrt * (nrt + 1) / 2 : nrt * ncs;
      const int sw = 1 | ((dbg & 7) << 1) | (symm ? 16 : 0);
      if (small_wg) hipLaunchKernelGGL(k_trdf<512>, dim3(grid, nblk), dim3(512), 0, s, db, i, cfix, sw);
      else hipLaunchKernelGGL(k_trdf<1024>, dim3(grid, nblk), dim3(1024), 0, s, db, i, cfix, sw);
    } else {
      const int nrg = (m + 1 + 127) / 128;
      int ncs = std::max(1, std::min(64, (m + 63) / 64));
      hipLaunchKernelGGL(k_trdb_gemv, dim3(nrg * ncs, nblk), dim3(256), shb, s, db, i, c, ncs);
    }
    count++;
  }
  if (use_graph) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HFG_HIP_CHECK(hipStreamEndCapture(cap, &graph));
    HFG_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HFG_HIP_CHECK(hipGraphLaunch(exec, cap));  // warm-up (uploads the graph)
    HFG_HIP_CHECK(hipStreamSynchronize(cap));
    HFG_HIP_CHECK(hipEventRecord(e0, cap));
    HFG_HIP_CHECK(hipGraphLaunch(exec, cap));
    HFG_HIP_CHECK(hipEventRecord(e1, cap));
    HFG_HIP_CHECK(hipEventSynchronize(e1));
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    (void)hipStreamDestroy(cap);
  } else {
    HFG_HIP_CHECK(hipEventRecord(e1, s));
    HFG_HIP_CHECK(hipEventSynchronize(e1));
  }
  float t = 0.f;
  HFG_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = t;
  *launches = count;
  if (w.last_fused && getenv("HELFEM_TRDF_DBG") && (atoi(getenv("HELFEM_TRDF_DBG")) & 4)) {
    const int n0 = w.last_ns[0];
    {
      std::vector<unsigned long long> wv((size_t)16 * n0);
      HFG_HIP_CHECK(hipMemcpy(wv.data(), w.fcp[0].p + (size_t)2 * TF_MAXS * 2 * TB_NB + (size_t)8 * n0, wv.size() * 8, hipMemcpyDeviceToHost));
      double aw[16] = {0};
      int cn = 0;
      for (int i = 40; i < n0 - 200; i++) {
        if (i % TB_NB == 0) continue;
        for (int k = 0; k < 16; k++) aw[k] += (double)wv[(size_t)i * 16 + k];
        cn++;
      }
      printf("  arrival at barrier 1 by wave (us):");
      for (int k = 0; k < 16; k++) printf(" %.2f", aw[k] / std::max(cn, 1) * 0.01);
      printf("\n");
    }
    std::vector<unsigned long long> st((size_t)8 * n0);
    HFG_HIP_CHECK(hipMemcpy(st.data(), w.fcp[0].p + (size_t)2 * TF_MAXS * 2 * TB_NB, st.size() * 8, hipMemcpyDeviceToHost));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int i = 40; i < n0 - 40; i++) {
      if (i % TB_NB == 0) continue;
      for (int k = 0; k < 6; k++) acc[k] += (double)st[(size_t)i * 8 + k];
      cnt++;
    }
    {
      double a6 = 0, a7 = 0, a8 = 0;
      int cn = 0;
      for (int i = 40; i < n0 - 40; i++) {
        if (i % TB_NB == 0) continue;
        a6 += (double)(st[(size_t)i * 8 + 6] & 0xffffffffull);
        a8 += (double)(st[(size_t)i * 8 + 6] >> 32);
        a7 += (double)st[(size_t)i * 8 + 7];
        cn++;
      }
      fprintf(stderr, "  phase A detail: loads issued %.2f | tile issued %.2f | own reductions done %.2f us\n", a6 / cn / 100.0,
              a7 / cn / 100.0, a8 / cn / 100.0);
    }
    fprintf(stderr, "k_trdf phase stamps (us since kernel entry, mean over %d columns): desc %.2f | loads+barrier1 %.2f | scalars+barrier2 %.2f | rows+barrier3 %.2f | tile+barrier4 %.2f | end %.2f\n",
            cnt, acc[0] / cnt / 100.0, acc[1] / cnt / 100.0, acc[2] / cnt / 100.0, acc[3] / cnt / 100.0, acc[4] / cnt / 100.0, acc[5] / cnt / 100.0);
  }
}

}  // namespace hfg
