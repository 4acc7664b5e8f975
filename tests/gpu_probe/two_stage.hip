// Two-stage tridiagonalisation of symmetric matrices on gfx950, batched over problems
// (the LAPACK dsyevd call behind scf::eig_gsym / eig_gsym_sub, /root/reference/src/general/scf_helpers.cpp:131-186).
//
// STATUS: both stages are correct (tests/test_gpu_twostage.py) and were MEASURED SLOWER than the one-stage chain of
// trd.hip at the sizes of this path (3 x 1470: stage 1 11.2 ms, stage 2 23.7 ms against 12.1 ms; DESIGN.md section 7 has the
// numbers and the reason -- a dependent step through device memory costs 3.5 us inside a kernel, 1.8 us as a kernel
// boundary).  The product path does not call this file; it is reachable through probe_band_reduce /
// probe_two_stage (libtwostage_probe.so) only and kept as the record of that measurement.  The back-transformation described below was
// therefore not written.
//
//   stage 1  dense -> band of half-width SB = 32: per panel of SB columns a Householder QR of the block below the band
//            (one workgroup per problem, the panel in registers), then the two-sided compact-WY update of the trailing
//            matrix as matrix products:  X = A22 V,  Z = V^T X,  Y = X T,  U = Y - V (T^T Z T)^T,
//            A22 <- A22 - [Y | V] [V | U]^T.   n / SB dependent steps instead of n.
//   stage 2  band -> tridiagonal by bulge chasing, one column per sweep (Schwarz / Lang); sweep s + 1 follows sweep s
//            three tasks behind.  One wavefront per sweep, a 32 x 32 block per task in its registers; consecutive sweeps
//            talk through HBM/L2 with agent-scope release / acquire counters, the three-task lag hides that latency.
//   back-transformation  Z <- Q1 (Q2 Z):  Q2 (the SB-long reflectors of stage 2) applied sweep by sweep to column slabs
//            of Z resident in LDS, Q1 through the compact-WY machinery of eig.hip.
// tools/two_stage_model.py is the NumPy statement of the same algorithm with the same index conventions.
#include "../../helfem_amd/csrc/hip/common.h"
#include "../../helfem_amd/csrc/hip/wave.h"
#include <cstdlib>
#include <cstring>

namespace hfg {

void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_acc_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN, bool tile64);

constexpr int SB = 32;             // half-bandwidth of the intermediate band matrix
constexpr int SB_LDB = 2 * SB;     // band storage: AB[j * SB_LDB + d] = A[j + d][j], d < 2 SB (room for the bulge)
constexpr int SB_MAXB = 8;
constexpr int SB_TMAX = 92;        // panel rows per thread (16 row groups): problems up to n = 16 * SB_TMAX + SB = 1504
constexpr int SB_KS = 4;           // split-K slabs of X = A22 V
constexpr int SB_RT = 64;          // row tile of the W kernels

struct SbBatch {
  int nblk;
  int n[SB_MAXB];
  double *A[SB_MAXB];     // n x n, ld n (full symmetric storage; on exit the band in its lower part)
  double *Vx[SB_MAXB];    // n x n: explicit stage-1 reflectors, column c acts on rows c + SB .. n-1 (unit entry at row c + SB)
  double *tau1[SB_MAXB];  // n
  double *T1[SB_MAXB];    // [panel][SB x SB] compact-WY factors, column-major
  double *Xs[SB_MAXB];    // [SB_KS][n x SB] split-K slabs of X, ld n
  double *X[SB_MAXB];     // n x SB, ld n
  double *Zp[SB_MAXB];    // [row tile][SB x SB] partial products V^T X
  double *Lm[SB_MAXB];    // n x 2 SB: [Y | V], ld n
  double *Rm[SB_MAXB];    // n x 2 SB: [V | U], ld n
  double *AB[SB_MAXB];    // n x SB_LDB band storage
  // stage 2
  double *VV[SB_MAXB];    // n x n: column s holds the reflectors of sweep s on rows s+1 .. n-1 (one per SB rows, each with its unit entry)
  double *tau2[SB_MAXB];  // [sweep][task]: SB2_KT per sweep
  int *prog[SB_MAXB];     // [sweep]: tasks completed (SB2_DONE when the sweep has ended); slot 0 of prog is sweep -1 (always done)
  double *d[SB_MAXB], *e[SB_MAXB];
  int *status;            // != 0: a wait ran out (the chain is broken), every wave leaves
};

// ---------------------------------------------------------------------------------------------------------------
// Stage 1, panel: Householder QR of P = A[r0:, j0:j0+SB] (m x SB, r0 = j0 + SB), one workgroup of 1024 threads per problem,
// the panel in registers.  Thread (c, g): column c = lane & 31, rows g + 32 t with g = 2 wave + (lane >> 5) (t < SB_TMAX):
// the 32 lanes of a half-wave hold the same rows of the 32 columns, so the column being eliminated is read from LDS with
// ONE address per half-wave (broadcast) -- with a column per half-wave instead every thread read the whole column and the
// kernel was bound by the LDS port (377 KB per column).  Per column j, three barriers:
//   A  the owner lanes have published the column x (double buffered),
//      every thread forms its share of x^T P[:, c] on the rows below j (the reflector is linear in the unnormalised x:
//      k_trdf's trick; the share of column j itself is |x|^2),
//   B  wave 0 sums the 32 shares per column, forms tau, beta and v^T P[:, c] = P[j][c] + scale x^T P[:, c] for all columns
//      (for c < j that is the Gram entry v_c^T v_j, which gives T = dlarft without another pass),
//   C  rank-1 update of the columns c > j, column j becomes (R[0:j+1, j]; v).
// ---------------------------------------------------------------------------------------------------------------
// The row loops are unrolled over register arrays; left alone, the scheduler issues EVERY LDS read of a column first and
// the arithmetic afterwards (one register pair per row in flight: the tall instantiations then spill).  Pinning the four
// values a chunk has just produced, with a memory clobber, keeps the next chunk's reads behind this chunk's arithmetic.
#define SB_CHUNK_FENCE() asm volatile("" ::: "memory")
#define SB_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")

// TT: rows per thread of this instantiation (the host picks the smallest one that holds the tallest panel of the batch);
// every loop over the rows is branch-free over TT -- rows beyond a problem's m are zeros in the registers and in LDS.
// (With run-time trip tests per row the compiler emitted a scalar branch and a full LDS wait per row.)
// 512 threads = 8 waves, two per SIMD: 256 registers per thread, so that a panel of 1472 x 32 (92 rows per thread) stays in
// registers (with 1024 threads and 128 registers the tall instantiations spilled 150 registers).
constexpr int SBP_NT = 512;            // threads of the panel kernel
constexpr int SBP_NG = SBP_NT / 32;    // row groups: thread (c, g) holds rows g + SBP_NG * t
template <int TT>
__global__ __launch_bounds__(SBP_NT) void k_sb_panel(const SbBatch *__restrict__ bp, int j0) {
  __shared__ double xs[2][TT * SBP_NG];
  __shared__ double part[SBP_NG][SB + 1];  // [row group][column]
  __shared__ double sG[SB][SB + 1];
  __shared__ double stau[SB], spj[SB], sf[SB];
  __shared__ double sscal[2];  // scale, beta
  __shared__ double sT[SB][SB + 1];
  const SbBatch &b = *bp;
  const int blk = blockIdx.x;
  const int n = b.n[blk];
  const int r0 = j0 + SB, m = n - r0;
  if (m < 2) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 31, g = 2 * wave + (lane >> 5);
  double *A = b.A[blk];
  double p[TT];
  {
    const double *col = A + (size_t)(j0 + c) * n + r0;
#pragma unroll
    for (int t = 0; t < TT; t++) {
      const int r = min(g + SBP_NG * t, m - 1);  // clamped: every load is unconditional, the value is dropped beyond m
      const double v = col[r];
      p[t] = (g + SBP_NG * t < m) ? v : 0.0;
      if ((t & 7) == 7) SB_CHUNK_FENCE();
    }
  }
  for (int t = tid; t < SB * (SB + 1); t += SBP_NT) (&sG[0][0])[t] = 0.0;
  for (int j = 0; j < SB; j++) {
    const int buf = j & 1;
    const double *x = xs[buf];
    if (c == j) {
#pragma unroll
      for (int t = 0; t < TT; t++) xs[buf][g + SBP_NG * t] = p[t];
    }
    if (g == (j & (SBP_NG - 1))) spj[c] = (j >= SBP_NG) ? p[1] : p[0];  // P[j][c]: row j = g + 16 t, t = j >> 4
    __syncthreads();                                                    // A
    // x^T P[:, c] on the rows below j: the rows 0 .. 31 (t = 0, 1) are the only ones that can lie at or above row j
    double s0 = (g > j) ? x[g] * p[0] : 0.0, s1 = (g + SBP_NG > j) ? x[g + SBP_NG] * p[1] : 0.0;
#pragma unroll
    for (int t = 2; t < TT; t += 2) {
      s0 += x[g + SBP_NG * t] * p[t];
      if (t + 1 < TT) s1 += x[g + SBP_NG * (t + 1)] * p[t + 1];
      if ((t & 7) == 6) SB_CHUNK_FENCE();
    }
    part[g][c] = s0 + s1;
    __syncthreads();  // B
    if (wave == 0 && lane < 32) {
      double S = 0.0;
#pragma unroll 4
      for (int gg = 0; gg < SBP_NG; gg++) S += part[gg][lane];
      const double q = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(S), j), __builtin_amdgcn_readlane(__double2loint(S), j));
      const double alpha = x[j];
      double tau, beta, scale;
      if (q == 0.0) {
        tau = 0.0;
        beta = alpha;
        scale = 0.0;
      } else {
        const double nrm = sqrt(alpha * alpha + q);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      const double wdot = spj[lane] + scale * S;  // v^T P[:, lane]
      sf[lane] = (lane > j) ? tau * wdot : 0.0;
      if (lane < j) sG[lane][j] = wdot;
      if (lane == j) {
        stau[j] = tau;
        sscal[0] = scale;
        sscal[1] = beta;
        b.tau1[blk][j0 + j] = tau;
      }
    }
    __syncthreads();  // C
    const double f = sf[c], scale = sscal[0];
    // one code path for all lanes: mul = scale and sub = 0 for the column itself (it becomes v), mul = 1 and
    // sub = f scale x for the others (f = 0 for the columns already done)
    const bool own = (c == j);
    const double mul = own ? scale : 1.0, fs = own ? 0.0 : f * scale;
    static_assert(TT % 4 == 0, "rows per thread in chunks of four");
    p[2] = p[2] * mul - fs * x[g + SBP_NG * 2];
    p[3] = p[3] * mul - fs * x[g + SBP_NG * 3];
#pragma unroll
    for (int t = 4; t < TT; t += 4) {
      p[t] = p[t] * mul - fs * x[g + SBP_NG * t];
      p[t + 1] = p[t + 1] * mul - fs * x[g + SBP_NG * (t + 1)];
      p[t + 2] = p[t + 2] * mul - fs * x[g + SBP_NG * (t + 2)];
      p[t + 3] = p[t + 3] * mul - fs * x[g + SBP_NG * (t + 3)];
      SB_PIN4(p[t], p[t + 1], p[t + 2], p[t + 3]);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int r = g + SBP_NG * t;
      const double xr = x[r];
      double pr = p[t];
      if (r > j) pr = pr * mul - fs * xr;
      else if (r == j) pr = own ? sscal[1] : pr - f;  // row j: R[j][j] = beta for the column itself, v_j = 1 for the others
      p[t] = pr;
    }
  }
  __syncthreads();
  // results: R into the panel of A (rows <= column inside the band), the explicit reflectors into Vx and into the
  // operand blocks [Y | V] and [V | U] of the trailing update
  double *Vx = b.Vx[blk], *Lm = b.Lm[blk], *Rm = b.Rm[blk];
  {
    double *v0 = Vx + (size_t)(j0 + c) * n + r0, *v1 = Lm + (size_t)(SB + c) * n + r0, *v2 = Rm + (size_t)c * n + r0;
    // rows 0 .. 31 (t = 0, 1) hold R on and above the diagonal and the unit diagonal of V; all other rows are plain v
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int r = g + SBP_NG * t;
      if (r < m) {
        const double val = p[t];
        if (r <= c) A[(size_t)(j0 + c) * n + r0 + r] = val;  // R (upper triangle of the first SB rows)
        const double v = (r > c) ? val : ((r == c) ? 1.0 : 0.0);
        v0[r] = v;
        v1[r] = v;
        v2[r] = v;
      }
    }
#pragma unroll
    for (int t = 2; t < TT; t++) {
      const int rr = g + SBP_NG * t;
      if (rr < m) {
        v0[rr] = p[t];
        v1[rr] = p[t];
        v2[rr] = p[t];
      }
      if ((t & 7) == 7) SB_CHUNK_FENCE();
    }
  }
  // T = dlarft(forward, columnwise) from the Gram entries: T(i,i) = tau_i, T(0:i, i) = -tau_i T(0:i,0:i) G(0:i, i);
  // lane = row of T, the rows in LDS
  if (wave == 0 && lane < SB) {
    for (int i = 0; i < SB; i++) sT[lane][i] = 0.0;
    for (int i = 0; i < SB; i++) {
      const double ti = stau[i];
      double acc = 0.0;
      for (int k = lane; k < i; k++) acc += sT[lane][k] * sG[k][i];  // T is upper triangular: T(lane, k) = 0 for k < lane
      if (lane < i) sT[lane][i] = -ti * acc;
      else if (lane == i) sT[lane][i] = ti;
    }
    double *Tp = b.T1[blk] + (size_t)(j0 / SB) * SB * SB;
    for (int i = 0; i < SB; i++) Tp[(size_t)i * SB + lane] = sT[lane][i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage 1, W kernels (row tiles of SB_RT rows): w1 sums the split-K slabs of X and forms the tile's share of Z = V^T X;
// w2 sums Z, forms M2 = T^T Z T and the operand blocks  Y = X T (into Lm[:, 0:SB]),  U = Y - V M2^T (into Rm[:, SB:2SB]).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sb_w1(const SbBatch *__restrict__ bp, int j0) {
  __shared__ double sX[SB_RT][SB + 1], sV[SB_RT][SB + 1];
  const SbBatch &b = *bp;
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  const int r0 = j0 + SB, m = n - r0;
  if (m < 2) return;
  const int tile = blockIdx.x;
  const int row0 = tile * SB_RT;
  if (row0 >= m) return;
  const int tid = threadIdx.x, rl = tid & 63, cg = tid >> 6;
  const int row = row0 + rl;
  const bool live = row < m;
  const double *Vx = b.Vx[blk];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = 8 * cg + u;
    double x = 0.0, v = 0.0;
    if (live) {
#pragma unroll
      for (int s = 0; s < SB_KS; s++) x += b.Xs[blk][((size_t)s * SB + c) * n + r0 + row];
      v = Vx[(size_t)(j0 + c) * n + r0 + row];
      b.X[blk][(size_t)c * n + r0 + row] = x;
    }
    sX[rl][c] = x;
    sV[rl][c] = v;
  }
  __syncthreads();
  // Zp[tile][c'][c] = sum_rows V[row][c'] X[row][c]; thread -> c' = tid & 31, c = (tid >> 5) * 4 .. + 3
  const int cp = tid & 31, c4 = (tid >> 5) * 4;
  double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll 8
  for (int r = 0; r < SB_RT; r++) {
    const double v = sV[r][cp];
    z0 += v * sX[r][c4];
    z1 += v * sX[r][c4 + 1];
    z2 += v * sX[r][c4 + 2];
    z3 += v * sX[r][c4 + 3];
  }
  double *Zp = b.Zp[blk] + (size_t)tile * SB * SB;
  Zp[(size_t)(c4)*SB + cp] = z0;
  Zp[(size_t)(c4 + 1) * SB + cp] = z1;
  Zp[(size_t)(c4 + 2) * SB + cp] = z2;
  Zp[(size_t)(c4 + 3) * SB + cp] = z3;
}

__global__ __launch_bounds__(256) void k_sb_w2(const SbBatch *__restrict__ bp, int j0) {
  __shared__ double sT[SB][SB + 1], sZ[SB][SB + 1], sW[SB][SB + 1], sM[SB][SB + 1];
  __shared__ double sX[SB_RT][SB + 1], sV[SB_RT][SB + 1];
  const SbBatch &b = *bp;
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  const int r0 = j0 + SB, m = n - r0;
  if (m < 2) return;
  const int row0 = blockIdx.x * SB_RT;
  if (row0 >= m) return;
  const int ntile = (m + SB_RT - 1) / SB_RT;
  const int tid = threadIdx.x;
  const double *Tp = b.T1[blk] + (size_t)(j0 / SB) * SB * SB;
  // Z = sum of the tiles' shares (fixed order), T
  for (int e = tid; e < SB * SB; e += 256) {
    double z = 0.0;
    for (int t = 0; t < ntile; t++) z += b.Zp[blk][(size_t)t * SB * SB + e];
    sZ[e % SB][e / SB] = z;           // column-major source: element (row e % SB, col e / SB)
    sT[e % SB][e / SB] = Tp[e];
  }
  const int rl = tid & 63, cg = tid >> 6;
  const int row = row0 + rl;
  const bool live = row < m;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = 8 * cg + u;
    sX[rl][c] = live ? b.X[blk][(size_t)c * n + r0 + row] : 0.0;
    sV[rl][c] = live ? b.Vx[blk][(size_t)(j0 + c) * n + r0 + row] : 0.0;
  }
  __syncthreads();
  // W = Z T, then M2 = T^T W   (32 x 32 each; thread -> row tid & 31, columns (tid >> 5) * 4 .. + 3)
  {
    const int i = tid & 31, c4 = (tid >> 5) * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < SB; k++) {
      const double z = sZ[i][k];
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] += z * sT[k][c4 + u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) sW[i][c4 + u] = a[u];
  }
  __syncthreads();
  {
    const int i = tid & 31, c4 = (tid >> 5) * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < SB; k++) {
      const double t = sT[k][i];  // T^T(i, k)
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] += t * sW[k][c4 + u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) sM[i][c4 + u] = a[u];
  }
  __syncthreads();
  // Y = X T,  U = Y - V M2^T
  if (live) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int c = 8 * cg + u;
      double y = 0.0, vm = 0.0;
      for (int k = 0; k < SB; k++) {
        y += sX[rl][k] * sT[k][c];
        vm += sV[rl][k] * sM[c][k];
      }
      b.Lm[blk][(size_t)c * n + r0 + row] = y;
      b.Rm[blk][(size_t)(SB + c) * n + r0 + row] = y - vm;
    }
  }
}

// band storage from the reduced matrix: AB[j][d] = A[j + d][j] for d <= SB, zero for the bulge rows
__global__ void k_sb_gather_band(const SbBatch *__restrict__ bp) {
  const SbBatch &b = *bp;
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), d = threadIdx.x & 63;
  if (j >= n) return;
  double v = 0.0;
  if (d <= SB && j + d < n) v = b.A[blk][(size_t)j * n + j + d];
  b.AB[blk][(size_t)j * SB_LDB + d] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Stage 2: band -> tridiagonal by bulge chasing (tools/two_stage_model.py: task()).  Sweep s eliminates column s below
// the subdiagonal with a reflector on rows s+1 .. s+SB and chases the bulge down the band: task k works on the rows
// R_k = s + (k-1) SB + 1 .. s + k SB -- the reflector that zeroes the first column of the bulge block B = A[R_k, R_{k-1}]
// is applied from the left to B, from both sides to D = A[R_k, R_k] and from the right to Bn = A[R_{k+1}, R_k], which
// becomes the next task's bulge block.  Task (s, k) needs (s-1, k+2): consecutive sweeps run three tasks apart, about
// n / (3 SB) of them at a time.
//
// One wavefront (a workgroup of 64) per sweep in flight; wave g owns the sweeps g, g + G, ...  A 32 x 32 block lives in
// 16 registers per lane ("row layout": lane = (row a = lane & 31, half h = lane >> 5), register u = column 16 h + u), so
// that a block's columns are contiguous 256-byte runs of the band storage.  Row sums (Bn v, D v) are in-lane sums plus one
// swap across the halves; column sums (v^T B and the strictly lower part of D under its diagonal) go through a padded
// LDS tile written in row layout and read down the columns.  Vectors needed "by register index" (v, w, q) are broadcast
// through LDS.  Sweeps hand data over through the band storage in HBM/L2: a wave releases (agent scope) after its stores
// and publishes its task count; the follower polls that count and acquires before it loads.  The publication runs one
// task behind the stores (they have drained by then), which costs one more task of lag but no stall per task.
// Every wait is bounded; a wait that runs out raises the batch's status word and every wave leaves.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SB2_DONE = 1 << 30;
constexpr int SB2_KT = 64;  // tau2 slots per sweep (>= n / SB + 2)

__device__ __forceinline__ void sb2_load(const double *__restrict__ AB, int n, int rbase, int cbase, int lane, double (&blk)[16]) {
  const int a = lane & 31, c16 = (lane >> 5) * 16, i = rbase + a;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int j = cbase + c16 + u, dd = i - j;
    const bool ok = (i < n) && (j < n) && (dd >= 0) && (dd < SB_LDB);
    const double v = AB[(size_t)min(j, n - 1) * SB_LDB + min(max(dd, 0), SB_LDB - 1)];
    blk[u] = ok ? v : 0.0;
  }
}
__device__ __forceinline__ void sb2_store(double *__restrict__ AB, int n, int rbase, int cbase, int lane, const double (&blk)[16]) {
  const int a = lane & 31, c16 = (lane >> 5) * 16, i = rbase + a;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int j = cbase + c16 + u, dd = i - j;
    if ((i < n) && (j < n) && (dd >= 0) && (dd < SB_LDB)) AB[(size_t)j * SB_LDB + dd] = blk[u];
  }
}
// 16 values vec[16 h .. 16 h + 15] of a 32-vector in LDS, for the lane's half h
__device__ __forceinline__ void sb2_bcast(const double *vec, int lane, double (&out)[16]) {
  const int c16 = (lane >> 5) * 16;
#pragma unroll
  for (int u = 0; u < 16; u++) out[u] = vec[c16 + u];
}
// column sums  sum_r w_r M[r][c]  of a block in row layout through the LDS tile; strict: only the rows r > c (the part of a
// lower-triangular block under its diagonal).  wb = the weights by register index (sb2_bcast of w).  Returned in BOTH
// halves of lane c = lane & 31.
__device__ __forceinline__ double sb2_colsum(double (*tile)[SB + 1], int lane, const double (&blk)[16], const double (&wb)[16], bool strict) {
  const int a = lane & 31, h = lane >> 5, c16 = h * 16;
#pragma unroll
  for (int u = 0; u < 16; u++) tile[a][c16 + u] = blk[u];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are done (one wave per workgroup: no barrier)
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int r = c16 + u;
    const double m = tile[r][a];
    s += ((!strict || r > a) ? m : 0.0) * wb[u];
  }
  __builtin_amdgcn_wave_barrier();
  return swap32_sum(s, s);
}

__global__ __launch_bounds__(64) void k_sb_chase(const SbBatch *__restrict__ bp, int G, int delayed) {
  __shared__ double tile[SB][SB + 1];
  __shared__ double vecv[SB], vecw[SB], vecq[SB];
  const SbBatch &b = *bp;
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  const int lane = threadIdx.x, a = lane & 31, h = lane >> 5;
  double *AB = b.AB[blk];
  int *prog = b.prog[blk] + 1;  // prog[-1] = sweep -1, always done
  volatile int *status = b.status;
  for (int s = blockIdx.x; s <= n - 3; s += G) {
    double B[16], D[16], Bn[16], vb[16];
#pragma unroll
    for (int u = 0; u < 16; u++) B[u] = 0.0;
    int published = 0;  // tasks of this sweep already published
    int k = 1;
    for (;; k++) {
      const int rf = s + (k - 1) * SB + 1;  // first row of R_k
      if (rf > n - 2) break;                // fewer than two rows: the sweep has ended
      const int c0 = (k == 1) ? s : rf - SB;
      // ---- wait for sweep s-1 to be k+2 tasks in (or done) ----
      {
        const int need = k + 2;
        int spins = 0;
        while (true) {
          const int got = __hip_atomic_load(&prog[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (got >= need) break;
          if (*status != 0) return;
          if (++spins > (1 << 22)) {
            if (lane == 0) atomicExch((int *)b.status, 1);
            return;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      // ---- operands ----
      if (k == 1) {
        // the column s itself: x = A[s+1 .. s+SB, s], carried as column 0 of B
        const int i = rf + a;
        const double x0 = (h == 0 && i < n) ? AB[(size_t)s * SB_LDB + 1 + a] : 0.0;
        B[0] = (h == 0) ? x0 : 0.0;
      }
      sb2_load(AB, n, rf, rf, lane, D);        // lower triangle (with the diagonal); zeros above
      sb2_load(AB, n, rf + SB, rf, lane, Bn);
      // ---- reflector from x = column 0 of B (lanes 0 .. 31, register 0) ----
      const double xa = (h == 0) ? B[0] : 0.0;
      const double xn2 = wave_sum((h == 0 && a >= 1) ? xa * xa : 0.0);
      const double alpha = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(xa)), __builtin_amdgcn_readfirstlane(__double2loint(xa)));
      double tau, beta, scale;
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
      if (h == 0) vecv[a] = (a == 0) ? 1.0 : xa * scale;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      const double va = vecv[a];  // v by row, in both halves
      sb2_bcast(vecv, lane, vb);  // v by register index
      // reflectors for the back-transformation
      if (h == 0 && rf + a < n) b.VV[blk][(size_t)s * n + rf + a] = va;
      if (lane == 0) b.tau2[blk][(size_t)s * SB2_KT + (k - 1)] = tau;
      // ---- left application to the bulge block: B <- (I - tau v v^T) B; its first column becomes (beta, 0, ...) ----
      if (k >= 2) {
        const double w = sb2_colsum(tile, lane, B, vb, false);  // v^T B by column, in lane c
        if (h == 0) vecw[a] = w;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double wb[16];
        sb2_bcast(vecw, lane, wb);
        const double tv = tau * va;
#pragma unroll
        for (int u = 0; u < 16; u++) B[u] -= tv * wb[u];
        if (h == 0) B[0] = (a == 0) ? beta : 0.0;
        sb2_store(AB, n, rf, c0, lane, B);
      } else if (h == 0) {
        const int i = rf + a;
        if (i < n) AB[(size_t)s * SB_LDB + 1 + a] = (a == 0) ? beta : 0.0;
      }
      // ---- two-sided application to the diagonal block (lower triangle in registers) ----
      {
        double rowp = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) rowp += D[u] * vb[u];
        rowp = swap32_sum(rowp, rowp);                               // sum_{c <= a} D[a][c] v_c
        const double colp = sb2_colsum(tile, lane, D, vb, true);      // sum_{r > a} D[r][a] v_r
        const double pa = tau * (rowp + colp);
        const double pv = wave_sum((h == 0) ? pa * va : 0.0);
        const double qa = pa - 0.5 * tau * pv * va;
        if (h == 0) vecq[a] = qa;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double qb[16];
        sb2_bcast(vecq, lane, qb);
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int c = h * 16 + u;
          if (c <= a) D[u] -= va * qb[u] + qa * vb[u];
        }
        sb2_store(AB, n, rf, rf, lane, D);
      }
      // ---- right application to the next block: Bn <- Bn (I - tau v v^T); it is the next task's bulge block ----
      {
        double y = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) y += Bn[u] * vb[u];
        y = swap32_sum(y, y);
        const double ty = tau * y;
#pragma unroll
        for (int u = 0; u < 16; u++) B[u] = Bn[u] - ty * vb[u];
      }
      // ---- publish ----
      if (!delayed) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&prog[s], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = k;
      } else if (k >= 2) {
        // the stores of task k-1 were issued a whole task ago
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&prog[s], k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = k - 1;
      }
    }
    // the pending block (rows of the last R_{k}: at most one row is left below the last reflector)
    {
      const int rf = s + (k - 1) * SB + 1;
      if (k >= 2 && rf <= n - 1) sb2_store(AB, n, rf, rf - SB, lane, B);
    }
    (void)published;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(&prog[s], SB2_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// d, e of the tridiagonal matrix from the band storage after the last sweep
__global__ void k_sb_finish(const SbBatch *__restrict__ bp) {
  const SbBatch &b = *bp;
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  b.d[blk][j] = b.AB[blk][(size_t)j * SB_LDB];
  b.e[blk][j] = (j + 1 < n) ? b.AB[blk][(size_t)j * SB_LDB + 1] : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
struct SbWork {
  DevBuf<double> Vx[SB_MAXB], tau1[SB_MAXB], T1[SB_MAXB], Xs[SB_MAXB], X[SB_MAXB], Zp[SB_MAXB], Lm[SB_MAXB], Rm[SB_MAXB], AB[SB_MAXB];
  DevBuf<double> VV[SB_MAXB], tau2[SB_MAXB];
  DevBuf<int> prog[SB_MAXB], status;
  SbBatch hb{};  // the batch as last described (stage 2 and the back-transformation read it)
  DevBuf<SbBatch> desc;
  std::vector<SbBatch> h_desc;
  DevBuf<GemmTask> xtasks, utasks;
  std::vector<GemmTask> h_xtasks, h_utasks;
};
static std::map<hfg_ctx *, SbWork *> g_sb;
void sb_release(hfg_ctx *ctx) {
  auto it = g_sb.find(ctx);
  if (it != g_sb.end()) {
    delete it->second;
    g_sb.erase(it);
  }
}
static SbWork &sb_work(hfg_ctx *ctx) {
  auto it = g_sb.find(ctx);
  if (it != g_sb.end()) return *it->second;
  SbWork *w = new SbWork();
  g_sb[ctx] = w;
  return *w;
}

bool sb_supported(int nblk, const int *ns) {
  if (nblk > SB_MAXB) return false;
  for (int i = 0; i < nblk; i++)
    if (ns[i] - SB > SBP_NG * SB_TMAX || ns[i] < 4 * SB) return false;
  return true;
}

/// stage 1 for a batch: A[blk] (n x n full symmetric, ld n) -> band form in place + band storage AB; reflectors in the
/// work area (read by the back-transformation)
void sb_reduce_to_band(hfg_ctx *ctx, int nblk, const int *ns, double *const *A) {
  if (!sb_supported(nblk, ns)) throw std::logic_error("sb_reduce_to_band: problem size outside the two-stage kernels' range");
  SbWork &w = sb_work(ctx);
  hipStream_t s = ctx->stream;
  SbBatch b{};
  b.nblk = nblk;
  int nmax = 0;
  for (int i = 0; i < nblk; i++) {
    const int n = ns[i];
    nmax = std::max(nmax, n);
    const int npanel = n / SB + 1, ntile = (n + SB_RT - 1) / SB_RT + 1;
    w.Vx[i].resize((size_t)n * n);
    w.tau1[i].resize(n + SB);
    w.T1[i].resize((size_t)npanel * SB * SB);
    w.Xs[i].resize((size_t)SB_KS * n * SB);
    w.X[i].resize((size_t)n * SB);
    w.Zp[i].resize((size_t)ntile * SB * SB);
    w.Lm[i].resize((size_t)n * 2 * SB);
    w.Rm[i].resize((size_t)n * 2 * SB);
    w.AB[i].resize((size_t)n * SB_LDB + SB_LDB);
    b.n[i] = n;
    b.A[i] = A[i];
    b.Vx[i] = w.Vx[i].p;
    b.tau1[i] = w.tau1[i].p;
    b.T1[i] = w.T1[i].p;
    b.Xs[i] = w.Xs[i].p;
    b.X[i] = w.X[i].p;
    b.Zp[i] = w.Zp[i].p;
    b.Lm[i] = w.Lm[i].p;
    b.Rm[i] = w.Rm[i].p;
    b.AB[i] = w.AB[i].p;
    HFG_HIP_CHECK(hipMemsetAsync(w.Vx[i].p, 0, sizeof(double) * (size_t)n * n, s));
    HFG_HIP_CHECK(hipMemsetAsync(w.tau1[i].p, 0, sizeof(double) * (n + SB), s));
  }
  upload_cached(w.desc, w.h_desc, std::vector<SbBatch>(1, b), s);
  const SbBatch *db = w.desc.p;
  // task lists of all panels: X slabs (SB_KS per block) and the rank-2SB update
  const int npan = (nmax - SB - 2) / SB + 1;  // panels with m = n - (p+1) SB >= 2 for the largest problem
  GemmTask none;
  none.A = none.B = nullptr;
  none.C = nullptr;
  none.M = none.N = none.K = 0;
  none.lda = none.ldb = none.ldc = 1;
  std::vector<GemmTask> xt((size_t)npan * SB_KS * nblk, none), ut((size_t)npan * nblk, none);
  for (int p = 0; p < npan; p++)
    for (int k = 0; k < nblk; k++) {
      const int n = ns[k], j0 = p * SB, r0 = j0 + SB, m = n - r0;
      if (m < 2) continue;
      const int chunk = (((m + SB_KS - 1) / SB_KS) + 15) / 16 * 16;
      for (int sl = 0; sl < SB_KS; sl++) {
        GemmTask g = none;
        const int k0 = sl * chunk, kk = std::max(0, std::min(chunk, m - k0));
        g.A = A[k] + (size_t)(r0 + (kk > 0 ? k0 : 0)) * n + r0;           // A22[:, k0:k0+kk]
        g.B = w.Vx[k].p + (size_t)j0 * n + r0 + (kk > 0 ? k0 : 0);        // V[k0:k0+kk, :]
        g.C = w.Xs[k].p + (size_t)sl * SB * n + r0;
        g.M = m;
        g.N = SB;
        g.K = kk;  // K = 0 writes zeros
        g.lda = g.ldb = g.ldc = n;
        xt[((size_t)p * SB_KS + sl) * nblk + k] = g;
      }
      GemmTask u = none;
      u.A = w.Lm[k].p + r0;
      u.B = w.Rm[k].p + r0;
      u.C = A[k] + (size_t)r0 * n + r0;
      u.M = u.N = m;
      u.K = 2 * SB;
      u.lda = u.ldb = u.ldc = n;
      u.tB = 1;
      u.alpha = -1.0;
      u.beta = 1.0;
      ut[(size_t)p * nblk + k] = u;
    }
  upload_cached(w.xtasks, w.h_xtasks, xt, s);
  upload_cached(w.utasks, w.h_utasks, ut, s);
  // diagnostics: HELFEM_SB_NPANEL stops after that many panels, HELFEM_SB_STEP inside the last one (1 panel QR only, 2 + X,
  // 3 + W kernels); sb_fetch_debug() then returns the intermediate arrays
  const int dbg_npan = getenv("HELFEM_SB_NPANEL") ? atoi(getenv("HELFEM_SB_NPANEL")) : -1;
  const int dbg_step = getenv("HELFEM_SB_STEP") ? atoi(getenv("HELFEM_SB_STEP")) : 4;
  w.hb = b;
  for (int p = 0; p < npan; p++) {
    if (dbg_npan >= 0 && p >= dbg_npan) break;
    const int last_step = (dbg_npan >= 0 && p == dbg_npan - 1) ? dbg_step : 4;
    const int j0 = p * SB, mmax = nmax - j0 - SB;
    {
      const int TTn = (mmax + SBP_NG - 1) / SBP_NG;
      if (TTn <= 16) hipLaunchKernelGGL(k_sb_panel<16>, dim3(nblk), dim3(SBP_NT), 0, s, db, j0);
      else if (TTn <= 32) hipLaunchKernelGGL(k_sb_panel<32>, dim3(nblk), dim3(SBP_NT), 0, s, db, j0);
      else if (TTn <= 48) hipLaunchKernelGGL(k_sb_panel<48>, dim3(nblk), dim3(SBP_NT), 0, s, db, j0);
      else if (TTn <= 64) hipLaunchKernelGGL(k_sb_panel<64>, dim3(nblk), dim3(SBP_NT), 0, s, db, j0);
      else if (TTn <= 80) hipLaunchKernelGGL(k_sb_panel<80>, dim3(nblk), dim3(SBP_NT), 0, s, db, j0);
      else hipLaunchKernelGGL(k_sb_panel<SB_TMAX>, dim3(nblk), dim3(SBP_NT), 0, s, db, j0);
    }
    if (last_step < 2) break;
    gemm_tasklist64_dev(ctx, w.xtasks.p + (size_t)p * SB_KS * nblk, SB_KS * nblk, mmax, SB);
    if (last_step < 3) break;
    const int ntile = (mmax + SB_RT - 1) / SB_RT;
    hipLaunchKernelGGL(k_sb_w1, dim3(ntile, nblk), dim3(256), 0, s, db, j0);
    hipLaunchKernelGGL(k_sb_w2, dim3(ntile, nblk), dim3(256), 0, s, db, j0);
    if (last_step < 4) break;
    gemm_tasklist_acc_dev(ctx, w.utasks.p + (size_t)p * nblk, nblk, mmax, mmax, mmax < 1024);
  }
  hipLaunchKernelGGL(k_sb_gather_band, dim3((nmax + 3) / 4, nblk), dim3(256), 0, s, db);
  HFG_HIP_CHECK(hipGetLastError());
}

/// stage 2 for the batch last reduced by sb_reduce_to_band: band storage -> tridiagonal (d, e device arrays of n each);
/// G sweeps in flight per block.  Throws when a wave's bounded wait ran out (status word).
void sb_chase(hfg_ctx *ctx, int nblk, const int *ns, double *const *d, double *const *e, int G, int delayed) {
  SbWork &w = sb_work(ctx);
  hipStream_t s = ctx->stream;
  SbBatch b = w.hb;
  if (b.nblk != nblk) throw std::logic_error("sb_chase: no band reduction of this batch precedes");
  int nmax = 0;
  w.status.resize(4);
  for (int i = 0; i < nblk; i++) {
    const int n = ns[i];
    nmax = std::max(nmax, n);
    if (n / SB + 3 > SB2_KT) throw std::logic_error("sb_chase: too many tasks per sweep");
    w.VV[i].resize((size_t)n * n);
    w.tau2[i].resize((size_t)n * SB2_KT);
    w.prog[i].resize(n + 2);
    b.VV[i] = w.VV[i].p;
    b.tau2[i] = w.tau2[i].p;
    b.prog[i] = w.prog[i].p;
    b.d[i] = d[i];
    b.e[i] = e[i];
    HFG_HIP_CHECK(hipMemsetAsync(w.prog[i].p, 0, sizeof(int) * (n + 2), s));
    const int done = SB2_DONE;
    HFG_HIP_CHECK(hipMemcpyAsync(w.prog[i].p, &done, sizeof(int), hipMemcpyHostToDevice, s));  // sweep -1
    HFG_HIP_CHECK(hipMemsetAsync(w.VV[i].p, 0, sizeof(double) * (size_t)n * n, s));
  }
  b.status = w.status.p;
  HFG_HIP_CHECK(hipMemsetAsync(w.status.p, 0, sizeof(int) * 4, s));
  w.hb = b;
  upload_cached(w.desc, w.h_desc, std::vector<SbBatch>(1, b), s);
  hipLaunchKernelGGL(k_sb_chase, dim3(G, nblk), dim3(64), 0, s, w.desc.p, G, delayed);
  hipLaunchKernelGGL(k_sb_finish, dim3((nmax + 255) / 256, nblk), dim3(256), 0, s, w.desc.p);
  HFG_HIP_CHECK(hipGetLastError());
}
int sb_chase_status(hfg_ctx *ctx) {
  SbWork &w = sb_work(ctx);
  int st = 0;
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  HFG_HIP_CHECK(hipMemcpy(&st, w.status.p, sizeof(int), hipMemcpyDeviceToHost));
  return st;
}

/// test access: band storage (n x SB_LDB, column j at [j * SB_LDB]) of the last reduction of this context, block blk
void sb_fetch_band(hfg_ctx *ctx, int blk, int n, double *hostAB) {
  SbWork &w = sb_work(ctx);
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  HFG_HIP_CHECK(hipMemcpy(hostAB, w.AB[blk].p, sizeof(double) * (size_t)n * SB_LDB, hipMemcpyDeviceToHost));
}
/// diagnostics: the work arrays of block 0 (which: 0 A, 1 Vx, 2 T1, 3 X, 4 Lm, 5 Rm), whole arrays
void sb_fetch_debug(hfg_ctx *ctx, int which, int n, double *host, size_t count) {
  SbWork &w = sb_work(ctx);
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  const double *src = which == 0 ? w.hb.A[0] : which == 1 ? w.Vx[0].p : which == 2 ? w.T1[0].p : which == 3 ? w.X[0].p : which == 4 ? w.Lm[0].p : w.Rm[0].p;
  (void)n;
  HFG_HIP_CHECK(hipMemcpy(host, src, sizeof(double) * count, hipMemcpyDeviceToHost));
}
int sb_bandwidth() { return SB; }
int sb_ldb() { return SB_LDB; }

}  // namespace hfg


// -------------------------------------------------------------------------------------------------------------------
// C entry points of this PROBE (tests/test_gpu_twostage.py, tools/sb_test.py, tools/sb_debug.py).  The two-stage
// reduction is not part of the product: it was built and measured in round 2 (stage 2's critical path of 3 n dependent
// tasks loses to the one-stage sweep at every order the product meets) and lives here, in a
// library of its own (tests/gpu_probe/libtwostage_probe.so, linked against libhelfem_amd.so for the GEMM task lists).
using namespace hfg;
#define HFG_TRY try {
#define HFG_CATCH                       \
  }                                     \
  catch (const std::exception &e) {     \
    hfg::set_error(e.what());           \
    return 2;                           \
  }                                     \
  return 0;
extern "C" {
// Diagnostic access to the first stage of the two-stage tridiagonalisation (this file): nrep copies of the symmetric
// matrix A (n x n) are reduced to band form in one batch; AB receives the band storage of the first copy
// (AB[j * ldb + d] = A_band[j + d][j]), ms the device time of the reduction.
int probe_band_reduce(hfg_ctx *ctx, int64_t n, const double *A, int nrep, double *AB, int *bandwidth, int *ldb, double *ms) {
  HFG_TRY
  if (nrep < 1 || nrep > 8) throw std::logic_error("probe_band_reduce: 1..8 copies\n");
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  std::vector<int> ns(nrep, (int)n);
  if (!sb_supported(nrep, ns.data())) throw std::logic_error("probe_band_reduce: size outside the kernels' range\n");
  std::vector<DevBuf<double> > dA(nrep);
  std::vector<double *> ptr(nrep);
  for (int k = 0; k < nrep; k++) {
    dA[k].resize((size_t)n * n + 2);
    HFG_HIP_CHECK(hipMemcpy(dA[k].p, A, sizeof(double) * n * n, hipMemcpyHostToDevice));
    ptr[k] = dA[k].p;
  }
  sb_reduce_to_band(ctx, nrep, ns.data(), ptr.data());  // warm-up (buffers, task lists)
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < nrep; k++) HFG_HIP_CHECK(hipMemcpy(dA[k].p, A, sizeof(double) * n * n, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  HFG_HIP_CHECK(hipEventCreate(&e0));
  HFG_HIP_CHECK(hipEventCreate(&e1));
  HFG_HIP_CHECK(hipEventRecord(e0, ctx->stream));
  sb_reduce_to_band(ctx, nrep, ns.data(), ptr.data());
  HFG_HIP_CHECK(hipEventRecord(e1, ctx->stream));
  HFG_HIP_CHECK(hipEventSynchronize(e1));
  float t = 0.f;
  HFG_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (ms) *ms = t;
  if (bandwidth) *bandwidth = sb_bandwidth();
  if (ldb) *ldb = sb_ldb();
  if (AB) sb_fetch_band(ctx, 0, (int)n, AB);
  HFG_CATCH
}
// both stages on nrep copies of A: d, e (n each) of copy 0, the times of the stages in ms (warm second run)
int probe_two_stage(hfg_ctx *ctx, int64_t n, const double *A, int nrep, int G, int delayed, double *d, double *e, double *ms1, double *ms2) {
  HFG_TRY
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  std::vector<int> ns(nrep, (int)n);
  if (!sb_supported(nrep, ns.data())) throw std::logic_error("probe_two_stage: size outside the kernels' range\n");
  std::vector<DevBuf<double>> dA(nrep), dd(nrep), de(nrep);
  std::vector<double *> ptr(nrep), pd(nrep), pe(nrep);
  for (int i = 0; i < nrep; i++) {
    dA[i].resize((size_t)n * n + 2);
    dd[i].resize(n);
    de[i].resize(n);
    ptr[i] = dA[i].p;
    pd[i] = dd[i].p;
    pe[i] = de[i].p;
  }
  hipEvent_t e0, e1, e2;
  HFG_HIP_CHECK(hipEventCreate(&e0));
  HFG_HIP_CHECK(hipEventCreate(&e1));
  HFG_HIP_CHECK(hipEventCreate(&e2));
  for (int rep = 0; rep < 2; rep++) {
    for (int i = 0; i < nrep; i++) HFG_HIP_CHECK(hipMemcpy(dA[i].p, A, sizeof(double) * n * n, hipMemcpyHostToDevice));
    HFG_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    sb_reduce_to_band(ctx, nrep, ns.data(), ptr.data());
    HFG_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    sb_chase(ctx, nrep, ns.data(), pd.data(), pe.data(), G, delayed);
    HFG_HIP_CHECK(hipEventRecord(e2, ctx->stream));
    HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (sb_chase_status(ctx) != 0) throw std::runtime_error("probe_two_stage: a bulge-chasing wave gave up waiting\n");
  }
  float t1 = 0, t2 = 0;
  HFG_HIP_CHECK(hipEventElapsedTime(&t1, e0, e1));
  HFG_HIP_CHECK(hipEventElapsedTime(&t2, e1, e2));
  *ms1 = t1;
  *ms2 = t2;
  HFG_HIP_CHECK(hipMemcpy(d, dd[0].p, sizeof(double) * n, hipMemcpyDeviceToHost));
  HFG_HIP_CHECK(hipMemcpy(e, de[0].p, sizeof(double) * n, hipMemcpyDeviceToHost));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipEventDestroy(e2);
  HFG_CATCH
}
// work arrays of the last probe_band_reduce (which: 0 A after the reduction -- only meaningful with nrep = 1 and while
// the call's buffers live, so this copies from the work area kept by the library --, 1 Vx, 2 T, 3 X, 4 [Y|V], 5 [V|U])
int probe_band_fetch(hfg_ctx *ctx, int which, int64_t n, double *out, int64_t count) {
  HFG_TRY
  if (which == 0) throw std::logic_error("probe_band_fetch: the reduced matrix is returned by probe_band_reduce_keep\n");
  sb_fetch_debug(ctx, which, (int)n, out, (size_t)count);
  HFG_CATCH
}
// the same reduction on ONE copy, returning the whole reduced matrix (n x n) as the kernels left it
int probe_band_reduce_keep(hfg_ctx *ctx, int64_t n, const double *A, double *Aout) {
  HFG_TRY
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  int ns = (int)n;
  if (!sb_supported(1, &ns)) throw std::logic_error("probe_band_reduce_keep: size outside the kernels' range\n");
  DevBuf<double> dA;
  dA.resize((size_t)n * n + 2);
  HFG_HIP_CHECK(hipMemcpy(dA.p, A, sizeof(double) * n * n, hipMemcpyHostToDevice));
  double *ptr = dA.p;
  sb_reduce_to_band(ctx, 1, &ns, &ptr);
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  HFG_HIP_CHECK(hipMemcpy(Aout, dA.p, sizeof(double) * n * n, hipMemcpyDeviceToHost));
  HFG_CATCH
}

int probe_release(hfg_ctx *ctx) {
  HFG_TRY
  hfg::sb_release(ctx);
  HFG_CATCH
}
}  // extern "C"
