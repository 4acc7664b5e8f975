// Persistent, register-resident Householder tridiagonalisation of symmetric matrices on gfx950 (LAPACK dsytd2 algebra,
// lower variant; the reduction stage of the dsyevd the reference reaches through arma::eig_sym,
// src/general/scf_helpers.cpp:135), batched over the symmetry blocks of scf::eig_gsym_sub (scf_helpers.cpp:142-186).
//
// ONE cooperative launch per batch.  The rows of every matrix are dealt out to workgroups (TP_NRG * R consecutive rows
// each, full symmetric storage) and stay in the REGISTER FILE of that workgroup's CU for the whole factorisation: the
// three trailing matrices of the bench workload (48 MB) fit the 128 MB of vector registers of the chip.  Per Householder
// column the workgroups of a matrix exchange one message through device memory (readable model and check of the algebra:
// tools/trdp_model.py):
//
//   every workgroup k   y[R_k]  = A^(j-1)[R_k, j+1:] x_j        its rows of the product with the UNNORMALISED column x_j,
//                                                               formed before the rank-2 update of column j-1 reaches
//                                                               its registers (that update runs while the data travel)
//                       dot_k   = sum_{r in R_k} x_j[r] y[r]
//   owner of row j+1    z       = A^(j)[j+1, j+1:]              that row fully updated
//
// From (y, z, dots) and x_j, v_{j-1}, w_{j-1} every workgroup derives beta, tau, v_j, w_j, d[j+1] and x_{j+1}
// redundantly, with identical arithmetic and summation orders (bitwise the same everywhere, reproducible run to run):
//   q = y - v_{j-1} c1 - w_{j-1} c2 (c1 = w_{j-1}.x_j, c2 = v_{j-1}.x_j),  x.q = sum dots - 2 c1 c2,
//   v = s (x - beta e1),  p = tau s (q - beta z),  v.p = tau s^2 (x.q - 2 beta q_0 + beta^2 z_0),
//   w = p - (tau/2)(v.p) v,  x_{j+1} = (z - w_0 v - w)[1:],  d[j+1] = z_0 - 2 w_0.
// No norm barrier, no second exchange, no atomics.
//
// Exchange words are 8-byte doubles stored write-through (sc1) and polled with sc1 loads; the data are their own flag: a
// ring of four slots per matrix is filled with a sentinel (all bits set, a NaN no arithmetic here produces) by one
// hipMemsetAsync before the launch, and every workgroup re-poisons the words IT will write two exchanges later (each
// word has one writer, which is also its poisoner: no write-after-read hazard between workgroups).  Every spin is
// bounded by a wall-clock limit; a workgroup that runs into it sets the status word and the launch drains.
// Co-residency of the grid is checked by hipLaunchCooperativeKernel; when it refuses, or the matrices do not fit the
// register file, tridiagonalize_batch (trd.hip) runs its chain of launches instead.
#include "common.h"
#include "wave.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

namespace hfg {

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int TP_MAXB = 8;
constexpr int TP_NT = 512;               // threads per workgroup: one workgroup per CU, two waves per SIMD, up to 256 VGPRs
                                         // (with 1024 threads and 128 VGPRs the 3 x 12 register tile spilled 96 of them)
constexpr int TP_NCG = 128;              // column groups: thread (rg, cg) holds columns cg, cg + 128, ...
constexpr int TP_NRG = TP_NT / TP_NCG;   // row groups
constexpr int TP_NW = TP_NT / 64;
constexpr int TP_SLOTS = 4;
constexpr int TP_MAXLAUNCH = 64;         // launches (groups x phases) of one batch
constexpr int TP_MAXG = 256;             // workgroups per matrix (landing area of the dot words)
constexpr unsigned long long TP_SENT = ~0ull;

struct TrdpDesc {
  int nblk;
  int wg0[TP_MAXB + 1];  // first workgroup of every matrix
  int n[TP_MAXB], G[TP_MAXB];
  int lda[TP_MAXB];    // leading dimension of A (a launch may work on the trailing part of a larger matrix)
  int jstop[TP_MAXB];  // columns this launch reduces before it writes the trailing matrix back and ends (>= n - 2: all)
  double *A[TP_MAXB], *d[TP_MAXB], *e[TP_MAXB], *tau[TP_MAXB];
  unsigned long long *xb[TP_MAXB];  // exchange ring: TP_SLOTS x (y: NP + 64 | z: NP | dots: TP_MAXG) words
  int *status;                      // 1: a spin ran into its limit (poisoned to -1 before the launch)
  unsigned long long *stamps;       // measurement only (nullptr in the product path): 4 wall-clock stamps per column
  long long spin_limit;             // wall-clock ticks (100 MHz)
  long long win_off;                // measurement only: offset (words) of the all-workgroup stamp window in stamps
};

__device__ __forceinline__ unsigned long long tp_load(const gu64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tp_store(gu64 *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long tp_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double tp_dbl(unsigned long long b) { return __longlong_as_double((long long)b); }
/// sum of the TP_NW wave partials held by lanes (lane % TP_NW), in every lane, same order everywhere
__device__ __forceinline__ double tp_wsum(double v) {
  if constexpr (TP_NW == 8) return oct_sum(v);
  else return row16_sum(v);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every wave would wait at
// every barrier for the write-through stores of the exchange (about a microsecond each) and for the prefetch of the next
// row z: everything the threads of a workgroup tell each other goes through LDS, and words of the exchange that are
// written twice (poison, then data) are written by the SAME thread both times, so no barrier has to order them.
#define TP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A/B switches (tools/ab_build.py builds variant libraries with -D...; compile time, because a run-time switch inside the
// loop costs registers).  Defaults = the measured best.
#ifndef TP_Z_BY_ROW
#define TP_Z_BY_ROW 0  // 1: the owner of row j+2 publishes the whole row (staged through LDS) instead of every workgroup its entries of column j+2
#endif
#ifndef TP_FLAG_FIRST
#define TP_FLAG_FIRST 0  // 1: a thread that misses a word of y watches its producer's dot word before it re-reads the word
#endif

// stamps of EVERY workgroup for a window of columns (measurement kernel only): wg x TP_WIN_N columns x 4 stamps behind the
// per-column stamps of the observed workgroup
constexpr int TP_WIN_0 = 400, TP_WIN_N = 64;
#define TP_WIN_STAMP(idx)                                                                                              \
  if constexpr (STAMPS) {                                                                                              \
    if (tid == 0 && (unsigned)(j1 - TP_WIN_0) < (unsigned)TP_WIN_N)                                                      \
      ((gu64 *)D.stamps)[D.win_off + ((size_t)blockIdx.x * TP_WIN_N + (j1 - TP_WIN_0)) * 4 + (idx)] = wall_clock64(); \
  }

template <int R, int U, bool STAMPS>
__global__ __launch_bounds__(TP_NT) void k_trdp(const TrdpDesc *__restrict__ dp) {
  constexpr int M = TP_NRG * R;   // rows per workgroup
  constexpr int NP = TP_NCG * U;  // padded order
  constexpr int EPT = (NP + TP_NT - 1) / TP_NT;  // vector elements per thread in the element-wise phase
  constexpr int NL = NP + 64;                    // LDS vectors: rows of the last workgroup may lie beyond NP
  static_assert(M <= 63, "the rows of a workgroup and its dot word are published by one wave");
  // Register budget: the tile takes 2 R U of the 256 VGPRs a thread of a 512-thread workgroup can have, and everything
  // else in the loop has to fit beside it.  What was measured while shaping this kernel (hipcc 7.2, -Rpass-analysis):
  // every conditional block inside the loop costs registers (four optional stamp stores: 45 spilled registers more on
  // the 6 x 12 tile), so the loop body below avoids predicates: clamped indices instead of bounds tests, whole rows
  // published including dead columns, dead rows updated with stale vectors (never read again).
  __shared__ double X[NL], VP[NL], WP[NL];
  __shared__ double YL[NL], ZL[NL];  // landing area of an exchange (each thread re-reads only what it stored itself)
  __shared__ double part[M][8];
  __shared__ double ZS[NP];  // staging of the row z (owner only)
  __shared__ double red[3][TP_NW];
  __shared__ double dots[TP_MAXG];
  __shared__ double zq0[2];
  __shared__ int abort_flag;
  const TrdpDesc &D = *dp;
  int blk = 0;
  while (blk + 1 < D.nblk && (int)blockIdx.x >= D.wg0[blk + 1]) blk++;
  const int k = (int)blockIdx.x - D.wg0[blk];
  const int n = D.n[blk], G = D.G[blk], lda = D.lda[blk], jstop = D.jstop[blk];
  gdouble *A = (gdouble *)D.A[blk];
  gdouble *dw = (gdouble *)D.d[blk], *ew = (gdouble *)D.e[blk], *tauw = (gdouble *)D.tau[blk];
  gu64 *xb = (gu64 *)D.xb[blk];
  // a slot: y by row (the rows of the last workgroup may reach beyond NP) | z by column | one dot word per workgroup
  constexpr int ZOFF = NP + 64, DOFF = 2 * NP + 64, slotw = DOFF + TP_MAXG;
  const long long spin_limit = D.spin_limit;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid & (TP_NCG - 1), rg = tid >> 7;
  const int row0 = k * M + rg * R;  // first row of this thread
  if constexpr (STAMPS) {
    if (tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      ((gu64 *)D.stamps)[D.win_off + (size_t)TP_MAXG * TP_WIN_N * 4 + blockIdx.x] = xcc & 15u;
    }
  }
  if (k * M >= n) return;

  // ---- registers: a[i][u] = A(row0 + i, cg + 128 u); the lower triangle is the input (LAPACK uplo = 'L') ----
  double a[R][U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int c = cg + TP_NCG * u;
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = row0 + i;
      double v = 0.0;
      if (r < n && c < n) v = (r >= c) ? A[(size_t)c * lda + r] : A[(size_t)r * lda + c];
      a[i][u] = v;
    }
    // a few column chunks at a time: with every load of the tile in flight their 64-bit addresses alone cost 2 R U registers.
    // Inline assembly ON PURPOSE: the compiler's wait-count bookkeeping does not see it, believes the tile still in flight
    // on entry to the main loop and puts an s_waitcnt vmcnt(0) in front of the first use of a tile register there -- at the
    // top of every pass, in front of the register update.  That wait retires the first attempt's loads and the publishing
    // wave's write-through stores before the update instead of in front of the poll behind it, and measures 4 % FASTER than
    // the precise bookkeeping the builtin gives (-DTP_BUILTIN_WAIT: 5.69 against 5.47 ms at 1380/1470/1380).
#ifdef TP_BUILTIN_WAIT
    if ((u & 3) == 3 || u == U - 1) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#else
    if ((u & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  for (int r = tid; r < NL; r += TP_NT) {
    X[r] = 0.0;
    VP[r] = 0.0;
    WP[r] = 0.0;
    YL[r] = 0.0;
    ZL[r] = 0.0;
  }
  if (tid == 0) abort_flag = 0;
  for (int r = tid; r < TP_MAXG; r += TP_NT) dots[r] = 0.0;
  __syncthreads();
  // ---- prologue: column 0 is the first x, nothing to correct ----
  for (int r = 1 + tid; r < n; r += TP_NT) X[r] = A[r];
  if (k == 0 && tid == 0) dw[0] = A[0];
  __syncthreads();

  // scalars of the column being reduced, carried from the end of one pass to the exchange of the next
  double c1 = 0.0, c2 = 0.0, beta = 0.0, tau = 0.0, scl = 0.0;

  // pass j: (j >= 0) consume exchange j -> v_j, w_j, x_{j+1};  then publish exchange j+1.  Pass -1 only publishes.
  // Critical path of a column: exchange lands -> barrier 1 -> scalars, element-wise -> barrier 2 -> rows of the product
  // -> barrier 3 -> publish.  Everything else (register update, the row z, partial sums and scalars of the next column)
  // runs between the publish and the next landing.
  unsigned long long zb[EPT], yb[EPT], dd = TP_SENT;  // words of the NEXT exchange, first fetched at the end of a pass
#pragma unroll
  for (int h = 0; h < EPT; h++) zb[h] = yb[h] = TP_SENT;
  const int jlast = min(n - 3, jstop - 1);  // last column this launch reduces
  int j = -1;
  for (;; j++) {
    const int j1 = j + 1, j2 = j + 2;
    if ((k + 1) * M - 1 < j1) return;  // no row of the trailing matrix left here
    // =============== rank-2 update of column j-1 on the registers ===============
    // (At the top of the loop body, in front of every use of the tile in this pass, so that the tile is updated in
    // place -- with the update behind its uses the compiler kept two copies of the tile.  VP = WP = 0 in the first two
    // passes; rows and columns already reduced are updated with stale values and never read again.)
    {
      double vr[R], wr[R];
#pragma unroll
      for (int i = 0; i < R; i++) {
        vr[i] = VP[row0 + i];
        wr[i] = WP[row0 + i];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double vc = VP[cg + TP_NCG * u], wc = WP[cg + TP_NCG * u];
#pragma unroll
        for (int i = 0; i < R; i++) a[i][u] = fma(-wr[i], vc, fma(-vr[i], wc, a[i][u]));  // two instructions per element
      }
    }
    // =============== row j+2 for exchange j+1, one exchange AHEAD of the products ===============
    // The registers now hold A^(j) (every update up to column j-1).  Row j+2 is published RAW; the consumers apply the
    // update of column j themselves (z = raw - v_j[j+2] w_j - w_j[j+2] v_j: they know v_j and w_j in full by then), so
    // this row never sits on the critical path of a column.  It is published as COLUMN j+2 of the symmetric tile, every
    // workgroup the entries of its own rows (2 R stores in four threads): with the owner of row j+2 publishing the whole
    // row, that workgroup -- the first one that still owns rows -- was the last to publish its y in nine columns of ten
    // (all-workgroup stamps, tools/trdp_window.py) and every other one waited for it.
#if !TP_Z_BY_ROW
    if (j < n - 3 && cg == (j2 & (TP_NCG - 1))) {
      // the tile is symmetric (up to the rounding of the two orders of its update): row j+2 is column j+2, and of that every
      // workgroup holds the entries of its own rows in the four threads (one per row group) of column group (j+2) % 128
      const int uz = j2 / TP_NCG;
      double zv[R];
#pragma unroll
      for (int i = 0; i < R; i++) zv[i] = 0.0;
#pragma unroll
      for (int u = 0; u < U; u++)
        if (u == uz) {
#pragma unroll
          for (int i = 0; i < R; i++) zv[i] = a[i][u];
        }
      gu64 *zcol = xb + (size_t)(j1 & (TP_SLOTS - 1)) * slotw + ZOFF;
#pragma unroll
      for (int i = 0; i < R; i++)
        if (row0 + i < n) tp_store(zcol + row0 + i, tp_bits(zv[i]));
    }
#else
    if (j < n - 3 && k == j2 / M) {
      // staged through LDS so that ALL threads store (three words each, the same thread that poisoned the word): the
      // workgroup that owns the row is also the one every other workgroup ends up waiting for
      if (rg == (j2 % M) / R) {
        // one of the R rows of this thread; a branch per row instead of a chain of selects per element (2 (R - 1) U
        // instructions in the workgroup every other one is waiting for)
        const int iz = (j2 % M) % R;
#pragma unroll
        for (int i = 0; i < R; i++)
          if (i == iz) {
#pragma unroll
            for (int u = 0; u < U; u++) ZS[cg + TP_NCG * u] = a[i][u];
          }
      }
      TP_LDS_BARRIER();
      gu64 *zrow = xb + (size_t)(j1 & (TP_SLOTS - 1)) * slotw + ZOFF;
#pragma unroll
      for (int h = 0; h < EPT; h++)
        if (tid + TP_NT * h < NP) tp_store(zrow + tid + TP_NT * h, tp_bits(ZS[tid + TP_NT * h]));
    }
#endif
    double pc1 = 0.0, pc2 = 0.0, ps = 0.0;  // partial sums for the next column
    if (j >= 0) {
      // =============== exchange j: wait for y (rows >= j+1), raw z (columns >= j+1), dots ===============
      gu64 *sb = xb + (size_t)(j & (TP_SLOTS - 1)) * slotw;
      const int kf = j1 / M;  // first workgroup that still owns rows
      const double vz = VP[j1], wz = WP[j1];  // v_{j-1}, w_{j-1} at the row that is now the pivot row
      // The words of y carry their own arrival (sentinel), but re-reading all of them until the last one has landed
      // costs the memory fabric 2.5 MB per attempt (212 workgroups x 12 KB: 0.5 us of its time); so after ONE attempt a
      // thread spins on the dot word of the workgroup that owns its row -- stored behind that workgroup's rows of y by the
      // same wave, a few cache lines per wave -- and fetches the row again when that word has landed.  The words of z were
      // published a whole exchange earlier and fetched at the end of the previous pass (zb); they are re-read only in
      // the unlikely case that they had not landed then.
      const int kd = min(kf + tid, G - 1);  // clamped: surplus threads poll the last word again
      if constexpr (STAMPS) {
        if (blk == 0 && k == G - 1 && tid == 0) ((gu64 *)D.stamps)[(size_t)j1 * 8 + 3] = wall_clock64();
      }
      TP_WIN_STAMP(3)
      // (the first attempt was issued at the end of the previous pass, behind the partial sums: its round trip, 0.45 us,
      // runs under the scalars of this column, the register update and the row z)
      {
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
          bool ok = dd != TP_SENT;
#pragma unroll
          for (int h = 0; h < EPT; h++) ok = ok && yb[h] != TP_SENT && zb[h] != TP_SENT;
          if constexpr (STAMPS) {
            if (blk == 0 && k == G - 1 && spins == 0) {
              // first check: how many lanes of each wave miss y / z / dot (wave 0 .. 7 -> stamps 4, packed by 8 bits x 3)
              bool my = false, mz = false;
#pragma unroll
              for (int h = 0; h < EPT; h++) {
                my = my || yb[h] == TP_SENT;
                mz = mz || zb[h] == TP_SENT;
              }
              const unsigned long long cy = __popcll(__ballot(my)), cz = __popcll(__ballot(mz)), cd = __popcll(__ballot(dd == TP_SENT));
              if (lane == 0) atomicAdd((unsigned long long *)D.stamps + (size_t)j1 * 8 + 4, cy | (cz << 16) | (cd << 32));
            }
          }
          if (ok) break;
          if (((++spins) & 15u) == 0u && (long long)(wall_clock64() - t0) > spin_limit) {
            abort_flag = 1;
            __hip_atomic_store((__attribute__((address_space(1))) int *)D.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
#pragma unroll
          for (int h = 0; h < EPT; h++) {
            const int r = min(j1 + tid + TP_NT * h, n - 1);
#if TP_FLAG_FIRST
            if (yb[h] == TP_SENT && tp_load(sb + DOFF + r / M) != TP_SENT) yb[h] = tp_load(sb + r);
#else
            if (yb[h] == TP_SENT) yb[h] = tp_load(sb + r);
#endif
            if (zb[h] == TP_SENT) zb[h] = tp_load(sb + ZOFF + r);
          }
          if (dd == TP_SENT) dd = tp_load(sb + DOFF + kd);
        }
      }
      if constexpr (STAMPS) {
        if (blk == 0 && k == G - 1 && tid == 0) ((gu64 *)D.stamps)[(size_t)j1 * 8 + 5] = wall_clock64();
      }
#pragma unroll
      for (int h = 0; h < EPT; h++) {
        const int r = min(j1 + tid + TP_NT * h, n - 1);
        YL[r] = tp_dbl(yb[h]);
        ZL[r] = tp_dbl(zb[h]);
      }
      if (tid == 0) {
        // element r = j+1: z_0 = raw - 2 v w there, q_0 = y_0 - v_{j-1}[j+1] c1 - w_{j-1}[j+1] c2, for everybody
        zq0[0] = tp_dbl(zb[0]) - 2.0 * vz * wz;
        zq0[1] = tp_dbl(yb[0]) - vz * c1 - wz * c2;
      }
      if (tid < TP_MAXG) dots[tid] = (kf + tid < G) ? tp_dbl(dd) : 0.0;
      TP_LDS_BARRIER();  // (1)
      if (abort_flag) return;
      if constexpr (STAMPS) {
        if (blk == 0 && k == G - 1 && tid == 0) ((gu64 *)D.stamps)[(size_t)j1 * 8 + 0] = wall_clock64();
      }
      TP_WIN_STAMP(0)
      // ---- scalars, redundantly in every wave (fixed orders: bitwise the same in every workgroup) ----
      double dsum = (dots[lane] + dots[lane + 64]) + (dots[lane + 128] + dots[lane + 192]);
      dsum = wave_sum(dsum);
      const double zz0 = zq0[0], q0 = zq0[1];
      const double xtq = dsum - 2.0 * c1 * c2;
      const double vtp = tau * scl * scl * (xtq - 2.0 * beta * q0 + beta * beta * zz0);
      const double aa = -0.5 * tau * vtp;
      const double ts = tau * scl;
      const double w0 = ts * (q0 - beta * zz0) + aa;
      // ---- element-wise: v_j, w_j, x_{j+1} (rolled: the register tile needs the room) ----
      const int own0 = k * M;
#pragma unroll
      for (int h = 0; h < EPT; h++) {  // (unrolled: the LDS round trips of the EPT elements overlap; 2 % of the column)
        const int r = j1 + tid + TP_NT * h;
        if (r >= n) continue;
        const double vpo = VP[r], wpo = WP[r];
        const double z = ZL[r] - vz * wpo - wz * vpo;  // row j+1 with the update of column j-1 applied
        const double q = YL[r] - vpo * c1 - wpo * c2;
        const double vn = (r == j1) ? 1.0 : scl * X[r];
        const double wn = ts * (q - beta * z) + aa * vn;
        double xn = 0.0;
        if (r > j1) {
          xn = z - vn * w0 - wn;
          pc1 += wn * xn;
          pc2 += vn * xn;
          if (r > j2) ps += xn * xn;
          if ((unsigned)(r - own0) < (unsigned)M) A[(size_t)j * lda + r] = vn;  // reflector for the back-transformation (LAPACK layout)
        }
        X[r] = xn;
        VP[r] = vn;
        WP[r] = wn;
      }
      if (k == kf && tid == 0) {
        dw[j1] = zz0 - 2.0 * w0;
        ew[j] = beta;
        tauw[j] = tau;
      }
    } else {
      // pass -1: |x_0[1:]|^2 of column 0
      for (int r = 2 + tid; r < n; r += TP_NT) ps += X[r] * X[r];
    }
    TP_LDS_BARRIER();  // (2): X = x_{j+1}, VP = v_j, WP = w_j
    if constexpr (STAMPS) {
      if (blk == 0 && k == G - 1 && tid == 0) ((gu64 *)D.stamps)[(size_t)j1 * 8 + 1] = wall_clock64();
    }
    TP_WIN_STAMP(1)
    if (j == jlast) break;  // the last column of the matrix (no successor to prepare) or of this phase

    // =============== publish exchange j+1: rows of the product with x_{j+1} ===============
    // (registers still lack the update of column j; X is zero on reduced columns)
    gu64 *sb1 = xb + (size_t)(j1 & (TP_SLOTS - 1)) * slotw;
    {
      double acc[R];
#pragma unroll
      for (int i = 0; i < R; i++) acc[i] = 0.0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double xc = X[cg + TP_NCG * u];
#pragma unroll
        for (int i = 0; i < R; i++) acc[i] += a[i][u] * xc;
      }
      // sums over the 16 lanes of a DPP row (four butterfly steps, the R chains interleave); the eight row sums of a
      // matrix row meet in LDS and the publishing lane adds them in a fixed order
#pragma unroll
      for (int i = 0; i < R; i++) acc[i] = row16_sum(acc[i]);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < R; i++) part[rg * R + i][(wave & 1) * 4 + (lane >> 4)] = acc[i];
      }
    }
    TP_LDS_BARRIER();  // (3)
    if (wave == 0) {
      const int r = k * M + lane;
      const bool live = lane < M && r >= j2 && r < n;
      double yv = 0.0, dc = 0.0;
      if (lane < M) {
        const double *pp = part[lane];
        yv = ((pp[0] + pp[1]) + (pp[2] + pp[3])) + ((pp[4] + pp[5]) + (pp[6] + pp[7]));
      }
      if (live) {
        tp_store(sb1 + r, tp_bits(yv));
        dc = X[r] * yv;
      }
      dc = wave_sum(dc);
      if (lane == 0) tp_store(sb1 + DOFF + k, tp_bits(dc));  // behind this wave's rows of y: doubles as their arrival flag
      // poison what this workgroup writes for exchange j+3 (its slot held exchange j-1, which everybody has consumed:
      // exchange j could only complete after every workgroup had published it, i.e. after it had read exchange j-1)
      gu64 *sb3 = xb + (size_t)((j + 3) & (TP_SLOTS - 1)) * slotw;
      if (lane < M) tp_store(sb3 + k * M + lane, TP_SENT);  // the same lanes that store the data: program order
      if (lane == 0) tp_store(sb3 + DOFF + k, TP_SENT);
    }
    if constexpr (STAMPS) {
      if (blk == 0 && k == G - 1 && tid == 0) ((gu64 *)D.stamps)[(size_t)j1 * 8 + 2] = wall_clock64();
    }
    TP_WIN_STAMP(2)
    // the row of exchange j+3 is published (raw, see above) at the top of pass j+2 by the owner of row j+4: that
    // workgroup poisons the words now
#if !TP_Z_BY_ROW
    if (j + 4 < n && cg == ((j + 4) & (TP_NCG - 1))) {  // the threads that will store these words at the top of pass j+2
      gu64 *sb3 = xb + (size_t)((j + 3) & (TP_SLOTS - 1)) * slotw;
#pragma unroll
      for (int i = 0; i < R; i++)
        if (row0 + i < n) tp_store(sb3 + ZOFF + row0 + i, TP_SENT);
    }
#else
    if (j + 4 < n && (j + 4) / M == k) {
      gu64 *sb3 = xb + (size_t)((j + 3) & (TP_SLOTS - 1)) * slotw;
      for (int c = tid; c < NP; c += TP_NT) tp_store(sb3 + ZOFF + c, TP_SENT);
    }
#endif
    // =============== off the critical path: z of exchange j+1 (published at the top of this pass), partial sums and
    // scalars of column j+1 ===============
#pragma unroll
    for (int h = 0; h < EPT; h++) zb[h] = tp_load(sb1 + ZOFF + min(j2 + tid + TP_NT * h, n - 1));
    pc1 = wave_sum(pc1);
    pc2 = wave_sum(pc2);
    ps = wave_sum(ps);
    if (lane == 0) {
      red[0][wave] = pc1;
      red[1][wave] = pc2;
      red[2][wave] = ps;
    }
    TP_LDS_BARRIER();  // (4)
    // first attempt at exchange j+1 (the workgroup every other one waits for finds its data there already)
#pragma unroll
    for (int h = 0; h < EPT; h++) yb[h] = tp_load(sb1 + min(j2 + tid + TP_NT * h, n - 1));
    dd = tp_load(sb1 + DOFF + min(j2 / M + tid, G - 1));
    {
      c1 = tp_wsum(red[0][lane & (TP_NW - 1)]);
      c2 = tp_wsum(red[1][lane & (TP_NW - 1)]);
      const double xn2 = tp_wsum(red[2][lane & (TP_NW - 1)]);
      const double alpha = X[j2];
      if (xn2 == 0.0) {
        tau = 0.0;
        beta = alpha;
        scl = 0.0;
      } else {
        const double nrm = sqrt(alpha * alpha + xn2);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        // one division for both quotients: 1 / (beta (alpha - beta)); the product cannot overflow or vanish for matrices
        // whose squared norms are representable (|beta| <= |alpha - beta| <= 2 |beta|)
        const double amb = alpha - beta;
        const double rq = 1.0 / (beta * amb);
        tau = -amb * amb * rq;  // (beta - alpha) / beta
        scl = beta * rq;        // 1 / (alpha - beta)
      }
    }
  }
  if (jlast != n - 3) {
    // ---- end of a phase (tridiagonalize_persistent): column j is reduced, v_j and w_j are in LDS; with their update the
    // registers hold the trailing matrix A^(j+1), whose lower triangle goes back to memory: the next launch takes it up
    // as a matrix of order n - j - 1, with the tile shape that fits THAT order ----
    // (the indices go through a value the compiler cannot see through: the 2 U address registers of these stores are
    // then computed HERE and not hoisted above the main loop, where they cost the 5 x 12 tile 33 registers)
    int opq;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
    const int cgo = cg + opq, row0o = row0 + opq;
    double vr[R], wr[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
      vr[i] = VP[row0o + i];
      wr[i] = WP[row0o + i];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int c = cgo + TP_NCG * u;
      const double vc = VP[c], wc = WP[c];
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = row0o + i;
        if (c > j && r >= c && r < n) A[(size_t)c * lda + r] = fma(-wr[i], vc, fma(-vr[i], wc, a[i][u]));
      }
    }
    return;
  }
  // ---- last column done: e[n-2] is the one element of x_{n-2}; d[n-1] from the registers of the last row, which lack the
  // update of column n-3 ----
  const int rl = n - 1;
  if (k == rl / M) {
    const int rgl = (rl % M) / R, il = (rl % M) % R, cgl = rl % TP_NCG, ul = rl / TP_NCG;
    if (rg == rgl && cg == cgl) {
      double av = 0.0;
#pragma unroll
      for (int i = 0; i < R; i++)
#pragma unroll
        for (int u = 0; u < U; u++)
          if (i == il && u == ul) av = a[i][u];
      dw[n - 1] = av - 2.0 * VP[rl] * WP[rl];
      ew[n - 2] = X[rl];
      ew[n - 1] = 0.0;
      tauw[n - 2] = 0.0;
      tauw[n - 1] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct TrdpWork {
  DevBuf<unsigned long long> ring;  // status block (16 words) + the exchange rings of all matrices
  DevBuf<unsigned long long> stamps;
  DevBuf<TrdpDesc> desc;
  std::vector<TrdpDesc> h_desc;
  unsigned long long *h_status = nullptr;  // pinned: the status words of the last batch's launches, copied back asynchronously
  bool pending_this_batch = false;
  bool pending = false;
  int ncu = 0;
  std::vector<int> last_ns;
  int last_R = 0, last_U = 0, last_grid = 0, last_nmax = 0;
};
static std::map<hfg_ctx *, TrdpWork *> g_trdp;

void trdp_release(hfg_ctx *ctx) {
  auto it = g_trdp.find(ctx);
  if (it != g_trdp.end()) {
    if (it->second->h_status) (void)hipHostFree(it->second->h_status);
    delete it->second;
    g_trdp.erase(it);
  }
}

/// throws when the last persistent launch of this context ended through a spin limit (call after a synchronisation)
void trdp_check_status(hfg_ctx *ctx) {
  auto it = g_trdp.find(ctx);
  if (it == g_trdp.end() || !it->second->pending) return;
  TrdpWork &w = *it->second;
  w.pending = false;
  bool bad = false;
  for (int q = 0; q < TP_MAXLAUNCH; q++) {
    bad = bad || (int)(w.h_status[q] & 0xffffffffull) == 1;
    w.h_status[q] = ~0ull;
  }
  if (bad) throw std::runtime_error("persistent tridiagonalisation: an exchange between workgroups ran into its time limit");
  // HELFEM_TRDP_STAMPS=1 (measurement builds of the kernel): phase durations of the last launch, printed once per launch
  if (w.stamps.p && !w.last_ns.empty()) {
    const int n0 = w.last_ns[0];
    std::vector<unsigned long long> st((size_t)8 * (n0 + 2));
    HFG_HIP_CHECK(hipMemcpy(st.data(), w.stamps.p, st.size() * 8, hipMemcpyDeviceToHost));
    // stamps of pass j (index j+1): 3 poll starts | 0 exchange landed | 1 element-wise done (barrier 2) | 2 published
    double wait = 0, post = 0, elem = 0, pub = 0, total = 0;
    int cnt = 0;
    for (int i = 2; i + 1 < n0 - 2; i++) {
      const double t0 = (double)st[(size_t)i * 8 + 0], t1 = (double)st[(size_t)i * 8 + 1], t2 = (double)st[(size_t)i * 8 + 2];
      const double t3 = (double)st[(size_t)i * 8 + 3];
      const double p2 = (double)st[(size_t)(i - 1) * 8 + 2], n0s = (double)st[(size_t)(i + 1) * 8 + 0];
      if (t0 == 0 || p2 == 0 || n0s == 0 || t3 == 0) continue;
      post += t3 - p2;   // published pass i-1 -> poll of exchange i starts: sums, scalars, register update, row z
      wait += t0 - t3;   // poll -> landed
      elem += t1 - t0;   // scalars + element-wise
      pub += t2 - t1;    // product, publish
      total += n0s - t0;
      cnt++;
    }
    if (getenv("HELFEM_TRDP_STAMPS") && atoi(getenv("HELFEM_TRDP_STAMPS")) >= 2 && n0 > 200) {
      fprintf(stderr, "k_trdp stamps, columns 100..160 (us): column | post-publish work | poll | element-wise | product+publish || thread 0 done polling at | lanes missing y, z, dot at the first check\n");
      for (int i = 100; i < 160; i++)
        fprintf(stderr, "  %4d  %.2f | %.2f | %.2f | %.2f | %.2f || %.2f | %llu %llu %llu\n", i, (double)(st[(size_t)(i + 1) * 8] - st[(size_t)i * 8]) * 0.01,
                (double)(st[(size_t)i * 8 + 3] - st[(size_t)(i - 1) * 8 + 2]) * 0.01, (double)(st[(size_t)i * 8] - st[(size_t)i * 8 + 3]) * 0.01,
                (double)(st[(size_t)i * 8 + 1] - st[(size_t)i * 8]) * 0.01, (double)(st[(size_t)i * 8 + 2] - st[(size_t)i * 8 + 1]) * 0.01, (double)(st[(size_t)i * 8 + 5] - st[(size_t)i * 8 + 3]) * 0.01,
                st[(size_t)i * 8 + 4] & 0xffff, (st[(size_t)i * 8 + 4] >> 16) & 0xffff, (st[(size_t)i * 8 + 4] >> 32) & 0xffff);
    }
    if (getenv("HELFEM_TRDP_STAMPS_FILE")) {
      // all workgroups, columns TP_WIN_0 .. TP_WIN_0 + TP_WIN_N: raw 100 MHz stamps (landed, element-wise done, published, poll start)
      std::vector<unsigned long long> win((size_t)TP_MAXG * TP_WIN_N * 4 + TP_MAXG);
      HFG_HIP_CHECK(hipMemcpy(win.data(), w.stamps.p + (size_t)8 * (w.last_nmax + 2), win.size() * 8, hipMemcpyDeviceToHost));
      if (FILE *f = fopen(getenv("HELFEM_TRDP_STAMPS_FILE"), "w")) {
        fprintf(f, "# grid %d R %d U %d first column %d columns %d; per line: workgroup column landed elementwise_done published poll_start\n", w.last_grid, w.last_R,
                w.last_U, TP_WIN_0, TP_WIN_N);
        for (int g = 0; g < w.last_grid; g++)
          for (int c = 0; c < TP_WIN_N; c++) {
            const unsigned long long *q = &win[((size_t)g * TP_WIN_N + c) * 4];
            fprintf(f, "%d %d %llu %llu %llu %llu %llu\n", g, TP_WIN_0 + c, q[0], q[1], q[2], q[3], win[(size_t)TP_MAXG * TP_WIN_N * 4 + g]);
          }
        fclose(f);
      }
    }
    if (cnt)
      fprintf(stderr, "k_trdp stamps (n = %d, R = %d, U = %d, grid %d; us per column over %d columns): column %.3f = sums+scalars+update %.3f | poll %.3f | scalars+element-wise %.3f | product+publish %.3f\n",
              n0, w.last_R, w.last_U, w.last_grid, cnt, total / cnt * 0.01, post / cnt * 0.01, wait / cnt * 0.01, elem / cnt * 0.01, pub / cnt * 0.01);
  }
}

typedef void (*trdp_kernel_t)(const TrdpDesc *);
template <int R, int U, bool STAMPS>
static trdp_kernel_t trdp_kernel() {
  return k_trdp<R, U, STAMPS>;
}
/// the tile shapes compiled (rows per thread x column chunks per thread); the stamping variants only where they are used
static trdp_kernel_t trdp_pick(int R, int U, bool stamps) {
#define TP_CASE(r, u) \
  if (R == r && U == u) return trdp_kernel<r, u, false>();
#define TP_CASE_S(r, u) \
  if (R == r && U == u) return stamps ? trdp_kernel<r, u, true>() : trdp_kernel<r, u, false>();
#ifdef TP_QUICK  // compile experiments
  TP_CASE_S(5, 12) TP_CASE(3, 8) TP_CASE(2, 4)
#else
  TP_CASE(1, 2) TP_CASE(2, 2) TP_CASE(3, 2) TP_CASE(4, 2)
  TP_CASE(1, 4) TP_CASE(2, 4) TP_CASE(3, 4) TP_CASE(4, 4) TP_CASE(5, 4) TP_CASE(6, 4)
  TP_CASE(1, 6) TP_CASE_S(2, 6) TP_CASE(3, 6) TP_CASE(4, 6) TP_CASE(5, 6) TP_CASE(6, 6)
  TP_CASE(1, 8) TP_CASE(2, 8) TP_CASE_S(3, 8) TP_CASE(4, 8) TP_CASE(5, 8) TP_CASE(6, 8)
  TP_CASE(1, 10) TP_CASE(2, 10) TP_CASE(3, 10) TP_CASE(4, 10) TP_CASE(5, 10) TP_CASE(6, 10)
  TP_CASE(1, 12) TP_CASE_S(2, 12) TP_CASE(3, 12) TP_CASE(4, 12) TP_CASE_S(5, 12) TP_CASE(6, 12)
  TP_CASE(1, 14) TP_CASE(2, 14) TP_CASE(3, 14) TP_CASE(4, 14)
  TP_CASE(2, 16) TP_CASE(3, 16) TP_CASE_S(4, 16)
  TP_CASE(2, 17) TP_CASE_S(3, 17) TP_CASE(4, 17)
#endif
#undef TP_CASE
#undef TP_CASE_S
  return nullptr;
}
static const int tp_rows_choices[] = {1, 2, 3, 4, 5, 6};
static const int tp_widths[] = {2, 4, 6, 8, 10, 12, 14, 16, 17};
/// column chunks per thread for matrices up to this order (0: beyond the register tiles)
static int tp_columns_for(int nmax) {
  static const int force = getenv("HELFEM_TRDP_U") ? atoi(getenv("HELFEM_TRDP_U")) : 0;  // measurement: a wider tile than needed
  if (force && nmax <= force * TP_NCG) return force;
  for (int u : tp_widths)
    if (nmax <= u * TP_NCG) return u;
  return 0;
}

/// true when a batch with these orders sends a LARGE matrix through the chain of launches (order beyond the register
/// tiles, or HELFEM_TRD selecting another variant): eig.hip then keeps the context's side stream out of the way
bool tridiagonalize_takes_chain(int nblk, const int *ns) {
  static const char *mode = getenv("HELFEM_TRD");
  const bool persistent = !(mode && strcmp(mode, "persistent") != 0);
  for (int i = 0; i < nblk; i++)
    if (ns[i] >= 1024 && (!persistent || tp_columns_for(ns[i]) == 0)) return true;
  return false;
}

/// Persistent path of tridiagonalize_batch.  The matrices are reduced in PHASES, one cooperative launch each, one after
/// the other on the stream: a launch takes the matrices with the largest trailing orders that fit the chip's register
/// file together (the three blocks 1380/1470/1380 of the bench workload from the start; of the blocks 2100/2001/2001 of
/// the LiF sizing first the largest alone, then pairs, and all three once they are down to 1536), reduces their leading
/// columns until the largest trailing matrix fits the next narrower tile, writes the trailing matrices back, and the
/// next launch takes them up as matrices of their own -- fewer column chunks AND fewer rows per thread, all workgroups
/// busy again.  (The time of a column follows the tile width the kernel is compiled for: 2.55 / 3.21 / 3.41 us per
/// column for the same 700-row matrices with 6 / 12 / 16 chunks; a phase boundary costs a launch and one pass over the
/// trailing matrix, some 20 us.)  done[i] says which matrices were factorised; the caller runs its chain of launches on
/// the others (order beyond the register tiles, the runtime refusing a cooperative launch, HELFEM_TRD selecting another
/// variant: nothing is touched then).  Same outputs as the chain: d, e, tau, reflectors below the subdiagonal.
void tridiagonalize_persistent(hfg_ctx *ctx, int nblk, const int *ns, double *const *A, double *const *d, double *const *e,
                               double *const *tau, std::vector<char> &done) {
  done.assign(nblk, 0);
  static const char *mode = getenv("HELFEM_TRD");
  if (mode && strcmp(mode, "persistent") != 0) return;
  if (nblk < 1 || nblk > TP_MAXB) return;
  static const int min_order = getenv("HELFEM_TRDP_MIN") ? atoi(getenv("HELFEM_TRDP_MIN")) : 256;
  TrdpWork *wp;
  auto it = g_trdp.find(ctx);
  if (it == g_trdp.end()) {
    wp = new TrdpWork();
    g_trdp[ctx] = wp;
    HFG_HIP_CHECK(hipDeviceGetAttribute(&wp->ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    HFG_HIP_CHECK(hipHostMalloc((void **)&wp->h_status, TP_MAXLAUNCH * sizeof(unsigned long long), hipHostMallocDefault));
    for (int i = 0; i < TP_MAXLAUNCH; i++) wp->h_status[i] = ~0ull;
  } else
    wp = it->second;
  TrdpWork &w = *wp;
  if (w.pending) {
    // the previous launches' status words were copied back on this stream; they are only READ when the stream says so
    if (hipStreamQuery(ctx->stream) == hipSuccess) trdp_check_status(ctx);
  }
  static const int forceR = getenv("HELFEM_TRDP_R") ? atoi(getenv("HELFEM_TRDP_R")) : 0;
  static const bool want_stamps = getenv("HELFEM_TRDP_STAMPS") && atoi(getenv("HELFEM_TRDP_STAMPS")) != 0;
  static const bool phases = !(getenv("HELFEM_TRDP_PHASES") && atoi(getenv("HELFEM_TRDP_PHASES")) == 0);
  static const int min_step = getenv("HELFEM_TRDP_STEP") ? std::max(1, atoi(getenv("HELFEM_TRDP_STEP"))) : 96;
  // ---- plan: largest matrices first; a matrix joins the current group while the group still fits ----
  struct Shape {
    int R = 0, U = 0, grid = 0;
  };
  auto fit = [&](const std::vector<int> &orders, Shape &g) -> bool {
    int nmax = 0;
    for (int o : orders) nmax = std::max(nmax, o);
    const int U = tp_columns_for(nmax);
    if (U == 0) return false;
    for (int r : tp_rows_choices) {
      if (forceR && r != forceR) continue;
      if (!trdp_pick(r, U, false)) continue;
      int grid = 0;
      bool ok = true;
      for (int o : orders) {
        const int G = (o + TP_NRG * r - 1) / (TP_NRG * r);
        ok = ok && G <= TP_MAXG;
        grid += G;
      }
      if (ok && grid <= w.ncu) {
        g.R = r;
        g.U = U;
        g.grid = grid;
        return true;
      }
    }
    return false;
  };
  std::vector<int> order;
  for (int i = 0; i < nblk; i++)
    if (ns[i] >= 3 && ns[i] >= min_order) order.push_back(i);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return ns[x] > ns[y]; });
  // ---- launches (descriptors identical from one SCF iteration to the next: uploaded once).  Before every launch the
  // matrices still to be reduced are looked at again, largest trailing order first, and the launch takes every one of
  // them that fits beside the ones already taken: the 2100-block of the LiF sizing runs alone until it is down to 1792,
  // the two 2001-blocks as a pair until they are, pairs of the three until all are at 1536 -- and from there the three
  // are reduced together, like the three blocks of the bench workload from the start (they used to be two groups that
  // ran one after the other to the end: 14.9 ms). ----
  struct Launch {
    Shape sh;
    size_t ring_words;
    std::vector<int> idx;  // matrices in it
    int nmax;
  };
  struct Act {
    int i, rem, off;
  };
  std::vector<Act> act;
  for (int i : order) {
    Shape t;
    if (fit(std::vector<int>(1, ns[i]), t)) act.push_back(Act{i, ns[i], 0});  // else: this matrix stays with the chain
  }
  if (act.empty()) return;
  std::vector<Launch> launches;
  std::vector<TrdpDesc> descs;
  size_t ring_words = 2 * TP_MAXLAUNCH;  // status words of the launches first (one 8-byte word each, 16-byte aligned regions behind)
  int nmax_all = 0;
  while (!act.empty()) {
    std::stable_sort(act.begin(), act.end(), [](const Act &x, const Act &y) { return x.rem > y.rem; });
    Launch L;
    std::vector<int> take, orders;  // positions in act
    for (size_t q = 0; q < act.size() && (int)take.size() < TP_MAXB; q++) {
      std::vector<int> trial = orders;
      trial.push_back(act[q].rem);
      Shape t;
      if (fit(trial, t)) {
        take.push_back((int)q);
        orders = trial;
        L.sh = t;
      }
    }
    if (take.empty()) return;  // cannot happen: every matrix fitted alone at its full order
    const int nmax = orders[0];
    // columns of this launch: until the largest trailing matrix fits the next narrower tile (at least min_step columns)
    int step = 0x3fffffff;
    if (phases && !want_stamps)
      for (int q = (int)(sizeof(tp_widths) / sizeof(int)) - 1; q >= 0; q--)
        if (tp_widths[q] < L.sh.U && nmax - tp_widths[q] * TP_NCG >= min_step) {
          step = nmax - tp_widths[q] * TP_NCG;
          break;
        }
    const int M = TP_NRG * L.sh.R, NP = TP_NCG * L.sh.U;
    TrdpDesc D = TrdpDesc{};
    D.nblk = (int)take.size();
    size_t words = 0;
    int wg = 0;
    std::vector<char> finished(act.size(), 0);
    for (int b = 0; b < D.nblk; b++) {
      Act &a = act[take[b]];
      const int i = a.i;
      const int G = (a.rem + M - 1) / M;
      const size_t o = (size_t)a.off;
      D.wg0[b] = wg;
      wg += G;
      D.n[b] = a.rem;
      D.G[b] = G;
      D.lda[b] = ns[i];
      D.A[b] = A[i] + o * (size_t)ns[i] + o;
      D.d[b] = d[i] + o;
      D.e[b] = e[i] + o;
      D.tau[b] = tau[i] + o;
      D.xb[b] = (unsigned long long *)words;  // offset for now, the base is added below
      words += (size_t)TP_SLOTS * ((size_t)2 * NP + 64 + TP_MAXG);
      if (a.rem - step >= 3) {
        D.jstop[b] = step;
        a.rem -= step;
        a.off += step;
      } else {
        D.jstop[b] = 0x3fffffff;  // runs to its end in this launch
        finished[take[b]] = 1;
      }
      L.idx.push_back(i);
    }
    for (int b = D.nblk; b <= TP_MAXB; b++) D.wg0[b] = wg;
    words = (words + 1) & ~(size_t)1;  // a multiple of 16 bytes for the poisoning memset
    L.ring_words = ring_words;  // (here: the launch's offset in the ring; every launch has its own region)
    L.nmax = nmax;
    ring_words += words;
    nmax_all = std::max(nmax_all, nmax);
    launches.push_back(L);
    descs.push_back(D);
    std::vector<Act> rest;
    for (size_t q = 0; q < act.size(); q++)
      if (!finished[q]) rest.push_back(act[q]);
    act = rest;
    if ((int)launches.size() > TP_MAXLAUNCH) return;
  }
  if ((int)launches.size() > TP_MAXLAUNCH) return;
  w.ring.resize(ring_words);  // one region per launch: ONE poisoning memset in front of the first launch, one copy of the status words behind the last
  const size_t win_off = (size_t)8 * (nmax_all + 2);
  if (want_stamps) w.stamps.resize(win_off + (size_t)TP_MAXG * TP_WIN_N * 4 + TP_MAXG);
  static const long long limit_ms = getenv("HELFEM_TRDP_LIMIT_MS") ? atoll(getenv("HELFEM_TRDP_LIMIT_MS")) : 200;
  for (size_t q = 0; q < descs.size(); q++) {
    TrdpDesc &D = descs[q];
    for (int b = 0; b < D.nblk; b++) D.xb[b] = w.ring.p + launches[q].ring_words + (size_t)D.xb[b];
    D.status = (int *)(w.ring.p + q);
    D.stamps = want_stamps ? w.stamps.p : nullptr;
    D.win_off = (long long)win_off;
    D.spin_limit = limit_ms * 100000ll;  // 100 MHz wall clock
  }
  hipStream_t s = ctx->stream;
  upload_cached(w.desc, w.h_desc, descs, s);
  w.last_ns.clear();
  std::vector<char> touched(nblk, 0), failed(nblk, 0);
  for (size_t q = 0; q < launches.size(); q++) {
    const Launch &L = launches[q];
    bool skip = false;
    for (int i : L.idx) skip = skip || failed[i];
    if (skip) continue;
    trdp_kernel_t kern = trdp_pick(L.sh.R, L.sh.U, want_stamps);
    if (want_stamps && q == 0) HFG_HIP_CHECK(hipMemsetAsync(w.stamps.p, 0, w.stamps.n * 8, s));
    if (!w.pending_this_batch) {
      HFG_HIP_CHECK(hipMemsetAsync(w.ring.p, 0xFF, ring_words * sizeof(unsigned long long), s));
      w.pending_this_batch = true;
    }
    const TrdpDesc *dptr = w.desc.p + q;
    void *args[] = {(void *)&dptr};
    hipError_t err;
    {
      // HIP events around this launch alone on the launch stream (bench.py: roofline), for the family and per tile shape
      static std::map<std::pair<int, int>, std::string> shape_names;
      std::string &nm = shape_names[std::make_pair(L.sh.R, L.sh.U)];
      if (nm.empty()) nm = "k_trdp<" + std::to_string(L.sh.R) + ", " + std::to_string(L.sh.U) + ">";
      ProfScope pk(ctx, "k_trdp");
      ProfScope ps(ctx, nm.c_str());
      static const bool coop = !(getenv("HELFEM_TRDP_COOP") && atoi(getenv("HELFEM_TRDP_COOP")) == 0);  // A/B: plain launch
      if (coop) err = hipLaunchCooperativeKernel((const void *)kern, dim3(L.sh.grid), dim3(TP_NT), args, 0, s);
      else err = hipLaunchKernel((const void *)kern, dim3(L.sh.grid), dim3(TP_NT), args, 0, s);
    }
    if (err != hipSuccess) {
      (void)hipGetLastError();  // refused (grid not co-resident on this device)
      bool first = true;
      for (int i : L.idx) first = first && !touched[i];
      if (!first)  // a later phase: the matrices are half reduced, there is nothing to fall back to
        throw std::runtime_error(std::string("persistent tridiagonalisation: cooperative launch refused in a later phase: ") + hipGetErrorString(err));
      static bool told = false;
      if (!told) {
        fprintf(stderr, "helfem_amd: cooperative launch of the persistent tridiagonalisation refused (%s): grid %d; using the launch chain\n",
                hipGetErrorString(err), L.sh.grid);
        told = true;
      }
      for (int i : L.idx) failed[i] = 1;  // the chain takes these matrices
      continue;
    }
    w.pending = true;
    for (int i : L.idx) touched[i] = 1;
    if (w.last_ns.empty()) {
      for (int i : L.idx) w.last_ns.push_back(ns[i]);
      w.last_R = L.sh.R;
      w.last_U = L.sh.U;
      w.last_grid = L.sh.grid;
      w.last_nmax = nmax_all;
    }
  }
  if (w.pending_this_batch) HFG_HIP_CHECK(hipMemcpyAsync(w.h_status, w.ring.p, TP_MAXLAUNCH * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  w.pending_this_batch = false;
  for (int i = 0; i < nblk; i++) done[i] = touched[i] && !failed[i];
}

/// replay of the last batch's launch on scratch copies is not possible (the kernel consumes its input); the bench
/// times the launch inside the eigensolve through the profiling events of the "eig_tridiag" scope instead.
void trdp_last_shape(hfg_ctx *ctx, int *R, int *U, int *grid) {
  *R = *U = *grid = 0;
  auto it = g_trdp.find(ctx);
  if (it == g_trdp.end()) return;
  *R = it->second->last_R;
  *U = it->second->last_U;
  *grid = it->second->last_grid;
}

}  // namespace hfg
