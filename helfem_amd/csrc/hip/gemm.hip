// FP64 GEMM on the gfx950 matrix cores:  C = alpha * op(A) * op(B) + beta * C, column-major.
//
// Used for the dense products of the generalized eigensolve (reference: scf::eig_gsym,
// src/general/scf_helpers.cpp:133 "Sinvh.t()*F*Sinvh" and :139 "C=Sinvh*C"), scf::form_density
// (:22-29) and the DIIS error matrix (src/general/diis.cpp:139-146), i.e. the products the
// reference hands to BLAS dgemm.
//
// v_mfma_f64_16x16x4_f64: one wave computes a 16x16 tile with K=4 per instruction.  Operand maps,
// verified on hardware with exact integer data (tests/gpu_probe/mfma_probe.hip):
//     a-operand: lane l holds X[l&15][l>>4]      b-operand: lane l holds Y[l>>4][l&15]
//     result reg r of lane l: D[(l>>4)+4r][l&15]
// The products are issued as D^T = B^T A^T (a-operand from the B tile, b-operand from the A tile)
// so that lane&15 runs along the rows of C, contiguous in column-major memory, and a result
// register stores 16 consecutive doubles (128 B) per 16-lane group.
//
// Matrix instruction.  Both FP64 forms are compiled: v_mfma_f64_16x16x4_f64 (MF = 1, the default) and the four-block
// v_mfma_f64_4x4x4_4b_f64 (MF = 0, HELFEM_MFMA=4x4x4).  Measured on gfx950 with this tile engine (tools/gemm_bench.py,
// profiles/r02_gemm_bench.txt): 54.4 / 51.5 TFLOP/s at m = n = k = 2816 and 47.0 / 46.2 at 4230 -- the same within the
// noise, 69 % of the 78.6 TFLOP/s data-sheet rate -- and 29.5 at 1400 (121 tiles of 128 x 128 on 256 CUs).  (A
// register-only loop of the 16x16x4 form with every instruction reading the SAME operand registers sustains only 36
// TFLOP/s, the 4x4x4 form 77: profiles/r02_fp64_rate.txt; that loop under-reports the 16x16x4 rate and was mistaken for
// the ceiling in round 1.)  Lane maps of the 4x4x4 form (tests/gpu_probe/mfma4x4_probe.hip,
// profiles/r02_mfma4x4_lane_maps.txt): a-operand lane 16 k + 4 blk + i holds A_blk[i][k], b-operand lane 16 k + 4 blk + j
// holds B_blk[k][j], result lane 16 i + 4 blk + j holds D_blk[i][j].  With the b-operand taken from the A tile exactly as
// for the 16x16x4 form (lane l: As[k0 + (l >> 4)][m0 + (l & 15)]) and the a-operand Bs[k0 + (l >> 4)][n0 + 4 c + (l & 3)],
// instruction c = 0..3 accumulates C[m0 + (l & 15)][n0 + 4 c + (l >> 4)]: four instructions fill the same four result
// registers, in the same layout, as one 16x16x4 instruction, so the epilogues are shared.
//
// Tiling: 256 threads = 4 waves in a 2x2 arrangement, block tile BM x BN (128x128 or 64x64,
// chosen by the launcher so that small problems still give >= 256 workgroups), BK = 16.
// LDS tiles are stored [k][m] with the row padded by 16 doubles: the four k-rows a wave reads at
// once then fall on disjoint 128-B bank groups (conflict-free ds_read_b64).
#include "common.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace hfg {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// one BM x BN tile (linear tile index id) of one product; As/Bs are the workgroup's LDS tiles
// ACC: the epilogue reads C (beta != 0).  All loads of the old tile are issued back to back with clamped indices
// before the first store: interleaved "load, scale, store" through the bounds branches serialises 64 memory round
// trips per thread, which made the rank-2NB trailing updates and the compact-WY updates of the eigensolver (K = 32
// or 64: pure streaming of C) run at 1.2-1.6 TB/s.
template <int BM, int BN, bool ACC = false, int MF = 1>
__device__ __forceinline__ void dgemm_tile(int id, int transA, int transB, int M, int N, int K, double alpha,
                                           const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb,
                                           double beta, double *__restrict__ C, int ldc, double (*As)[16][BM + 16],
                                           double (*Bs)[16][BN + 16], int sym = 0, int kbeg = 0, int kend = -1, int over = 0) {
  constexpr int BK = 16;
  constexpr int PAD = 16;
  constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
  constexpr int TM = WM / 16, TN = WN / 16;
  static_assert(PAD == 16, "LDS row padding is part of the tile types");

  // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a
  // contiguous strip of tiles (they share A row panels / B column panels in that XCD's L2)
  int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  const bool lower = !ACC && sym == 1;  // enumerate the lower tiles only
  int nwg = lower ? nbm * (nbm + 1) / 2 : nbm * nbn;
  if (!(over & 4)) {  // (over & 4: the caller has placed the tile itself, k_dgemm_tasklist_wl)
    int q = nwg / 8, r = nwg % 8, xcd = id % 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
  }
  int bm = (id % nbm) * BM, bn = (id / nbm) * BN;
  if (lower) {
    // symmetric product (BM == BN, M == N): the grid enumerates the tiles on and below the diagonal only (id counts
    // them row by row: id = i (i + 1) / 2 + j, j <= i); k_mirror_lower fills the upper triangle afterwards.  Skipping
    // the upper tiles of a full grid instead leaves their CU slots empty while other CUs still hold two tiles.
    int ti = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > id) ti--;
    while ((ti + 1) * (ti + 2) / 2 <= id) ti++;
    const int tj = id - ti * (ti + 1) / 2;
    if (ti >= nbm) return;
    bm = ti * BM;
    bn = tj * BN;
  }

  if constexpr (ACC) {
    // accumulating updates of a matrix of which only the lower triangle and a band above the diagonal are read afterwards
    // (sym == 2: the symmetric sweep of the tridiagonalisation, trd.hip): tiles more than three tile rows above the
    // diagonal are left alone
    if (sym == 2 && bm / BM + 3 < bn / BN) return;
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = (wave & 1) * WM, wn = (wave >> 1) * WN;
  const int l15 = lane & 15, l4 = lane >> 4;

  // accumulators: one double4 per 16 x 16 block for the 16x16x4 form; four separate doubles for the 4x4x4 form (as
  // elements of a double4 every instruction got a fresh destination register next to its source accumulator and the
  // 128-wide tile spilled 14 registers inside the K loop -- spill reloads share vmcnt with the operand prefetch)
  constexpr bool V4 = (MF == 1);
  double4_t accv[V4 ? TM : 1][V4 ? TN : 1];
  double accs[V4 ? 1 : TM][V4 ? 1 : TN][4];
  if constexpr (V4) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) accv[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  } else {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int c = 0; c < 4; c++) accs[i][j][c] = 0.0;
  }
  auto acc = [&](int i, int j, int r) -> double {
    if constexpr (V4) return accv[i][j][r];
    else return accs[i][j][r];
  };

  // ACC with 64 x 64 tiles: the 16 old values of C per thread are requested before the K loop, so their latency
  // hides behind the operand loads and the MFMAs (the product is a rank-32/64 update: C is all the traffic there is)
  constexpr bool PRE = ACC && (TM * TN <= 4);
  double cpre[PRE ? TM : 1][PRE ? TN : 1][4];
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int gm = min(bm + wm + i * 16 + l15, M - 1), gn = min(bn + wn + j * 16 + l4 + 4 * r, N - 1);
          cpre[i][j][r] = C[(size_t)gn * ldc + gm];
        }
  }

  // ---- operand staging: global -> registers -> LDS (k-major tiles As[k][m], Bs[k][n]), double buffered -------------
  // An operand is either contiguous along the tile's long dimension (A not transposed, B transposed: "kind C") or along
  // k (A transposed, B not transposed: "kind K").  A thread moves 16-byte vectors:
  //   kind C: vector v = (row pair 2 mv, column k), mv = v % (B?/2): 64 lanes read 1 KiB of one column, one
  //           ds_write_b128 per vector;
  //   kind K: vector v = (row m, k pair 2 kv), m = v % B?: lanes run along m, so that the two 8-byte LDS writes of a
  //           vector (rows 2 kv and 2 kv + 1 of the tile) are conflict free; the 16-byte global reads of a wave then hit
  //           64 different lines, each of which is used up by the thread's other vectors and by its neighbours.
  // Fast path (no bounds tests, 16-byte loads): the tile lies inside the operand for this k step, the leading dimension
  // is even and the base address 16-byte aligned; otherwise element-wise loads with bounds tests fill the same registers.
  constexpr int VA = BM * BK / 2 / 256, VB = BN * BK / 2 / 256;  // vectors per thread and tile
  const bool fastA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 1) == 0) && (bm + BM <= M || (over & 1));
  const bool fastB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((ldb & 1) == 0) && (bn + BN <= N || (over & 2));
  // per-thread origin inside a tile
  const int a_mv = tid % (BM / 2), a_kc = tid / (BM / 2);  // kind C: rows 2 a_mv, column a_kc + r * (512 / BM)
  const int a_m = tid % BM, a_kv = tid / BM;               // kind K: row a_m, k pair a_kv + r * (256 / BM)
  const int b_nv = tid % (BN / 2), b_kc = tid / (BN / 2);
  const int b_n = tid % BN, b_kv = tid / BN;
  const double *pA = transA ? A + (size_t)(bm + a_m) * lda + 2 * a_kv : A + (size_t)a_kc * lda + bm + 2 * a_mv;
  const double *pB = transB ? B + (size_t)b_kc * ldb + bn + 2 * b_nv : B + (size_t)(bn + b_n) * ldb + 2 * b_kv;
  d2_t va[VA], vb[VB];
  auto load_tiles = [&](int k0) {
    const bool fullk = (k0 + BK <= K);
    if (fastA && fullk) {
      if (!transA) {
#pragma unroll
        for (int r = 0; r < VA; r++) va[r] = *reinterpret_cast<const d2_t *>(pA + (size_t)(k0 + r * (512 / BM)) * lda);
      } else {
#pragma unroll
        for (int r = 0; r < VA; r++) va[r] = *reinterpret_cast<const d2_t *>(pA + k0 + 2 * r * (256 / BM));
      }
    } else {
#pragma unroll
      for (int r = 0; r < VA; r++) {
        int gm0, gk0, gm1, gk1;
        if (!transA) {
          gm0 = bm + 2 * a_mv;
          gm1 = gm0 + 1;
          gk0 = gk1 = k0 + a_kc + r * (512 / BM);
        } else {
          gm0 = gm1 = bm + a_m;
          gk0 = k0 + 2 * (a_kv + r * (256 / BM));
          gk1 = gk0 + 1;
        }
        d2_t v;
        v.x = (gm0 < M && gk0 < K) ? (transA ? A[(size_t)gm0 * lda + gk0] : A[(size_t)gk0 * lda + gm0]) : 0.0;
        v.y = (gm1 < M && gk1 < K) ? (transA ? A[(size_t)gm1 * lda + gk1] : A[(size_t)gk1 * lda + gm1]) : 0.0;
        va[r] = v;
      }
    }
    if (fastB && fullk) {
      if (transB) {
#pragma unroll
        for (int r = 0; r < VB; r++) vb[r] = *reinterpret_cast<const d2_t *>(pB + (size_t)(k0 + r * (512 / BN)) * ldb);
      } else {
#pragma unroll
        for (int r = 0; r < VB; r++) vb[r] = *reinterpret_cast<const d2_t *>(pB + k0 + 2 * r * (256 / BN));
      }
    } else {
#pragma unroll
      for (int r = 0; r < VB; r++) {
        int gn0, gk0, gn1, gk1;
        if (transB) {
          gn0 = bn + 2 * b_nv;
          gn1 = gn0 + 1;
          gk0 = gk1 = k0 + b_kc + r * (512 / BN);
        } else {
          gn0 = gn1 = bn + b_n;
          gk0 = k0 + 2 * (b_kv + r * (256 / BN));
          gk1 = gk0 + 1;
        }
        d2_t v;
        v.x = (gn0 < N && gk0 < K) ? (transB ? B[(size_t)gk0 * ldb + gn0] : B[(size_t)gn0 * ldb + gk0]) : 0.0;
        v.y = (gn1 < N && gk1 < K) ? (transB ? B[(size_t)gk1 * ldb + gn1] : B[(size_t)gn1 * ldb + gk1]) : 0.0;
        vb[r] = v;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    if (!transA) {
#pragma unroll
      for (int r = 0; r < VA; r++) *reinterpret_cast<d2_t *>(&As[buf][a_kc + r * (512 / BM)][2 * a_mv]) = va[r];
    } else {
#pragma unroll
      for (int r = 0; r < VA; r++) {
        As[buf][2 * (a_kv + r * (256 / BM))][a_m] = va[r].x;
        As[buf][2 * (a_kv + r * (256 / BM)) + 1][a_m] = va[r].y;
      }
    }
    if (transB) {
#pragma unroll
      for (int r = 0; r < VB; r++) *reinterpret_cast<d2_t *>(&Bs[buf][b_kc + r * (512 / BN)][2 * b_nv]) = vb[r];
    } else {
#pragma unroll
      for (int r = 0; r < VB; r++) {
        Bs[buf][2 * (b_kv + r * (256 / BN))][b_n] = vb[r].x;
        Bs[buf][2 * (b_kv + r * (256 / BN)) + 1][b_n] = vb[r].y;
      }
    }
  };

  // split K (kend >= 0): this workgroup forms the k range [kbeg, kend) of its tile and ADDS it to C (zeroed by the
  // caller) with FP64 atomics; with two halves per tile the sum has two addends, so its value does not depend on their order
  const bool splitk = kend >= 0;
  const int kstop = splitk ? min(kend, K) : K;
  load_tiles(kbeg);
  store_tiles(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kstop; k0 += BK) {
    const bool more = (k0 + BK < kstop);
    if (more) load_tiles(k0 + BK);  // the next step's operands travel while this step's MFMAs run
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double fa[TM];
#pragma unroll
      for (int i = 0; i < TM; i++) fa[i] = As[buf][kk + l4][wm + i * 16 + l15];
      if constexpr (MF == 1) {
        double fb[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) fb[j] = Bs[buf][kk + l4][wn + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            accv[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], accv[i][j], 0, 0, 0);
      } else {
        // column group by column group: only four a-operands are live at a time
#pragma unroll
        for (int j = 0; j < TN; j++) {
          double fb[4];
#pragma unroll
          for (int c = 0; c < 4; c++) fb[c] = Bs[buf][kk + l4][wn + j * 16 + 4 * c + (lane & 3)];
#pragma unroll
          for (int i = 0; i < TM; i++)
#pragma unroll
            for (int c = 0; c < 4; c++)
              accs[i][j][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fb[c], fa[i], accs[i][j][c], 0, 0, 0);
        }
      }
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();  // one barrier per step: buffer buf has been read by every wave, buffer buf ^ 1 is complete
    buf ^= 1;
  }
  // acc[i][j][r] = C[bm+wm+16i+l15][bn+wn+16j+l4+4r]
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int gm = bm + wm + i * 16 + l15, gn = bn + wn + j * 16 + l4 + 4 * r;
          if (gm < M && gn < N) C[(size_t)gn * ldc + gm] = alpha * acc(i, j, r) + beta * cpre[i][j][r];
        }
    return;
  } else if constexpr (ACC) {
    constexpr int HI = (TM > 2) ? TM / 2 : TM;  // rows of 16 handled per batch (32 or 16 values in flight per thread)
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += HI) {
      double cv[HI][TN][4];
#pragma unroll
      for (int i = 0; i < HI; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            int gm = min(bm + wm + (i0 + i) * 16 + l15, M - 1), gn = min(bn + wn + j * 16 + l4 + 4 * r, N - 1);
            cv[i][j][r] = C[(size_t)gn * ldc + gm];
          }
#pragma unroll
      for (int i = 0; i < HI; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            int gm = bm + wm + (i0 + i) * 16 + l15, gn = bn + wn + j * 16 + l4 + 4 * r;
            if (gm < M && gn < N) C[(size_t)gn * ldc + gm] = alpha * acc(i0 + i, j, r) + beta * cv[i][j][r];
          }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int gm = bm + wm + i * 16 + l15, gn = bn + wn + j * 16 + l4 + 4 * r;
        if (gm < M && gn < N) {
          size_t o = (size_t)gn * ldc + gm;
          double v = alpha * acc(i, j, r);
          if (splitk) {
            unsafeAtomicAdd(&C[o], v);
            continue;
          }
          if (beta != 0.0) v += beta * C[o];
          C[o] = v;
        }
      }
}

template <int BM, int BN, int MF = 1>
__global__ __launch_bounds__(256, 2) void k_dgemm(int transA, int transB, int M, int N, int K, double alpha,
                                               const double *__restrict__ A, int lda, const double *__restrict__ B,
                                               int ldb, double beta, double *__restrict__ C, int ldc) {
  __shared__ __attribute__((aligned(16))) double As[2][16][BM + 16];
  __shared__ __attribute__((aligned(16))) double Bs[2][16][BN + 16];
  dgemm_tile<BM, BN, false, MF>(blockIdx.x, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, As, Bs);
}

// the same tile engine over a device-side task list: C_t = A_t B_t, grid (max tiles, tasks)
template <int BM, int BN, bool ACC = false, int MF = 1>
__global__ __launch_bounds__(256, 2) void k_dgemm_tasklist(const GemmTask *__restrict__ tasks) {
  __shared__ __attribute__((aligned(16))) double As[2][16][BM + 16];
  __shared__ __attribute__((aligned(16))) double Bs[2][16][BN + 16];
  const GemmTask t = tasks[blockIdx.y];
  if (t.M <= 0 || t.N <= 0) return;
  const int sym = (!ACC && BM == BN && t.M == t.N && t.beta == 0.0) ? (t.sym == 1) : ((ACC && t.sym == 2) ? 2 : 0);
  const int nbm = (t.M + BM - 1) / BM;
  const int nt = (sym == 1) ? nbm * (nbm + 1) / 2 : nbm * ((t.N + BN - 1) / BN);
  if ((int)blockIdx.x >= nt) return;
  dgemm_tile<BM, BN, ACC, MF>(blockIdx.x, t.tA, t.tB, t.M, t.N, t.K, t.alpha, t.A, t.lda, t.B, t.ldb, t.beta, t.C, t.ldc,
                              As, Bs, sym, 0, -1, t.over);
}

// Task lists with an explicit workgroup list (task, tile), dealt out so that each XCD takes ONE contiguous eighth of the
// list: all tiles of a task then run on the same XCD and its A operand is fetched from HBM once instead of by all eight
// L2s (exchange_lr.hip: the 1.9 MB element table of a task was read 8 times -- 9 of the 15 GB the product fetched).
// SPLIT: the list holds two entries per tile, (task, 2 tile + half): each forms half of K and adds it to a zeroed C
// (k_dgemm_tasklist_split2's scheme).
template <int BM, int BN, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void k_dgemm_tasklist_wl(const GemmTask *__restrict__ tasks, const int2 *__restrict__ wl, int nwg) {
  __shared__ __attribute__((aligned(16))) double As[2][16][BM + 16];
  __shared__ __attribute__((aligned(16))) double Bs[2][16][BN + 16];
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int q8 = nwg / 8, r8 = nwg % 8;
  if (slot >= q8 + (xcd < r8 ? 1 : 0)) return;
  const int2 w = wl[(xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot];
  const GemmTask t = tasks[w.x];
  if constexpr (SPLIT) {
    const int half = w.y & 1;
    const int Kh = ((t.K / 2 + 15) / 16) * 16;
    const int kbeg = half ? Kh : 0, kend = half ? t.K : Kh;
    if (kbeg >= kend) return;
    dgemm_tile<BM, BN, false, 1>(w.y >> 1, t.tA, t.tB, t.M, t.N, t.K, t.alpha, t.A, t.lda, t.B, t.ldb, 0.0, t.C, t.ldc, As, Bs, 0, kbeg,
                                 kend, t.over | 4);
  } else {
    dgemm_tile<BM, BN, false, 1>(w.y, t.tA, t.tB, t.M, t.N, t.K, t.alpha, t.A, t.lda, t.B, t.ldb, t.beta, t.C, t.ldc, As, Bs, 0, 0, -1,
                                 t.over | 4);
  }
}

// Two workgroups per tile, each half of K (rounded to the k step): for batches whose tiles do not fill the chip evenly --
// 386 tiles of the eigensolve's products on 512 workgroup slots last two tile-times for 1.5 tile-times of average work,
// 772 half tiles fill them with three each; the 210 lower tiles of the symmetric product become 420 units that all
// run at once.  C must be zero on entry (beta == 0 tasks only).
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void k_dgemm_tasklist_split2(const GemmTask *__restrict__ tasks) {
  __shared__ __attribute__((aligned(16))) double As[2][16][BM + 16];
  __shared__ __attribute__((aligned(16))) double Bs[2][16][BN + 16];
  const GemmTask t = tasks[blockIdx.y];
  if (t.M <= 0 || t.N <= 0) return;
  const int sym = (BM == BN && t.M == t.N) ? (t.sym == 1) : 0;
  const int nbm = (t.M + BM - 1) / BM;
  const int nt = (sym == 1) ? nbm * (nbm + 1) / 2 : nbm * ((t.N + BN - 1) / BN);
  const int tile = blockIdx.x >> 1, half = blockIdx.x & 1;
  if (tile >= nt) return;
  const int Kh = ((t.K / 2 + 15) / 16) * 16;
  const int kbeg = half ? Kh : 0, kend = half ? t.K : Kh;
  if (kbeg >= kend) return;
  dgemm_tile<BM, BN, false, 1>(tile, t.tA, t.tB, t.M, t.N, t.K, t.alpha, t.A, t.lda, t.B, t.ldb, 0.0, t.C, t.ldc, As, Bs, sym, kbeg, kend);
}

/// HELFEM_MFMA=4x4x4 selects the v_mfma_f64_4x4x4_4b_f64 form of the tile engine (A/B runs; same speed, more LDS reads)
static bool mfma4() {
  static const bool v = (getenv("HELFEM_MFMA") && !strcmp(getenv("HELFEM_MFMA"), "4x4x4"));
  return v;
}

/// task lists whose products accumulate into C (beta != 0 in every active task): streaming epilogue
void gemm_tasklist_acc_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN, bool tile64) {
  if (ntasks <= 0 || maxM <= 0 || maxN <= 0) return;
  ProfScope ps(ctx, "gemm");
  if (tile64) {
    const int tiles = ((maxM + 63) / 64) * ((maxN + 63) / 64);
    if (mfma4()) hipLaunchKernelGGL((k_dgemm_tasklist<64, 64, true, 0>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
    else hipLaunchKernelGGL((k_dgemm_tasklist<64, 64, true>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  } else {
    const int tiles = ((maxM + 127) / 128) * ((maxN + 127) / 128);
    if (mfma4()) hipLaunchKernelGGL((k_dgemm_tasklist<128, 128, true, 0>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
    else hipLaunchKernelGGL((k_dgemm_tasklist<128, 128, true>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

// upper triangle := transpose of the lower one for the symmetric products of a task list (sym tasks only): 64 x 64
// blocks through LDS, coalesced on both sides
__global__ __launch_bounds__(256) void k_mirror_lower(const GemmTask *__restrict__ tasks) {
  __shared__ double tile[64][65];
  const GemmTask t = tasks[blockIdx.y];
  if (t.sym != 1 || t.M <= 0 || t.M != t.N) return;
  const int nb = (t.M + 63) / 64;
  // strictly-lower block (bi > bj) number blockIdx.x
  int bi = (int)((1.0 + sqrt(1.0 + 8.0 * (double)blockIdx.x)) * 0.5);
  while (bi * (bi - 1) / 2 > (int)blockIdx.x) bi--;
  while ((bi + 1) * bi / 2 <= (int)blockIdx.x) bi++;
  const int bj = blockIdx.x - bi * (bi - 1) / 2;
  if (bi >= nb) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) {
    int gm = bi * 64 + tx, gn = bj * 64 + c;
    tile[c][tx] = (gm < t.M && gn < t.N) ? t.C[(size_t)gn * t.ldc + gm] : 0.0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    // C(bj*64 + tx, bi*64 + c) = C(bi*64 + c, bj*64 + tx)
    int gm = bj * 64 + tx, gn = bi * 64 + c;
    if (gm < t.M && gn < t.N) t.C[(size_t)gn * t.ldc + gm] = tile[tx][c];
  }
}

/// after a task list with symmetric products (GemmTask::sym): fill the skipped upper triangles
void gemm_mirror_lower_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxN) {
  if (ntasks <= 0 || maxN <= 64) return;
  const int nb = (maxN + 63) / 64;
  hipLaunchKernelGGL(k_mirror_lower, dim3(nb * (nb - 1) / 2, ntasks), dim3(256), 0, ctx->stream, dtasks);
  HFG_HIP_CHECK(hipGetLastError());
}

/// launches the task list with 64 x 64 tiles (small products, more workgroups)
void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN) {
  if (ntasks <= 0 || maxM <= 0 || maxN <= 0) return;
  ProfScope ps(ctx, "gemm");
  const int tiles = ((maxM + 63) / 64) * ((maxN + 63) / 64);
  if (mfma4()) hipLaunchKernelGGL((k_dgemm_tasklist<64, 64, false, 0>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  else hipLaunchKernelGGL((k_dgemm_tasklist<64, 64>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  HFG_HIP_CHECK(hipGetLastError());
}

/// launches the task list with 128 x 128 tiles; max_mn = largest (M, N) over the tasks
/// the same with 128 x 64 tiles (no symmetric tasks): twice as many, half as large workgroups -- for batches whose
/// 128 x 128 tiles do not divide evenly over the CUs
void gemm_tasklist_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN) {
  if (ntasks <= 0 || maxM <= 0 || maxN <= 0) return;
  ProfScope ps(ctx, "gemm");
  const int tiles = ((maxM + 127) / 128) * ((maxN + 63) / 64);
  hipLaunchKernelGGL((k_dgemm_tasklist<128, 64>), dim3(tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  HFG_HIP_CHECK(hipGetLastError());
}

/// C = A B for task lists with beta == 0 whose outputs the caller has ZEROED: two workgroups per 128 x 128 tile
void gemm_tasklist_split2_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN) {
  if (ntasks <= 0 || maxM <= 0 || maxN <= 0) return;
  ProfScope ps(ctx, "gemm");
  const int tiles = ((maxM + 127) / 128) * ((maxN + 127) / 128);
  hipLaunchKernelGGL((k_dgemm_tasklist_split2<128, 128>), dim3(2 * tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  HFG_HIP_CHECK(hipGetLastError());
}

/// task list with a workgroup list (task, tile) in XCD order; rect: 128 x 64 tiles, else 128 x 128 (beta == 0 tasks, no sym)
/// tiles: 0 = 128 x 128, 1 = 128 x 64, 2 = 64 x 64 (the work list must have been enumerated with the same tile shape)
void gemm_tasklist_wl_dev(hfg_ctx *ctx, const GemmTask *dtasks, const int2 *dwl, int nwg, int tiles) {
  if (nwg <= 0) return;
  ProfScope ps(ctx, "gemm");
  const unsigned grid = 8u * (unsigned)((nwg + 7) / 8);
  if (tiles == 2) hipLaunchKernelGGL((k_dgemm_tasklist_wl<64, 64>), dim3(grid), dim3(256), 0, ctx->stream, dtasks, dwl, nwg);
  else if (tiles == 1) hipLaunchKernelGGL((k_dgemm_tasklist_wl<128, 64>), dim3(grid), dim3(256), 0, ctx->stream, dtasks, dwl, nwg);
  else hipLaunchKernelGGL((k_dgemm_tasklist_wl<128, 128>), dim3(grid), dim3(256), 0, ctx->stream, dtasks, dwl, nwg);
  HFG_HIP_CHECK(hipGetLastError());
}

/// split-K form of the above with 128 x 64 tiles: two list entries per tile, C zeroed by the caller
void gemm_tasklist_wl_split2_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, const int2 *dwl, int nwg) {
  if (nwg <= 0) return;
  ProfScope ps(ctx, "gemm");
  hipLaunchKernelGGL((k_dgemm_tasklist_wl<128, 64, true>), dim3(8u * (unsigned)((nwg + 7) / 8)), dim3(256), 0, ctx->stream, dtasks, dwl, nwg);
  HFG_HIP_CHECK(hipGetLastError());
}

/// the same with 128 x 64 tiles
void gemm_tasklist_split2_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN) {
  if (ntasks <= 0 || maxM <= 0 || maxN <= 0) return;
  ProfScope ps(ctx, "gemm");
  const int tiles = ((maxM + 127) / 128) * ((maxN + 63) / 64);
  hipLaunchKernelGGL((k_dgemm_tasklist_split2<128, 64>), dim3(2 * tiles, ntasks), dim3(256), 0, ctx->stream, dtasks);
  HFG_HIP_CHECK(hipGetLastError());
}

bool gemm_prefers_128(hfg_ctx *ctx, long tiles128);
void gemm_tasklist_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN) {
  if (ntasks <= 0 || maxM <= 0 || maxN <= 0) return;
  const int tiles = ((maxM + 127) / 128) * ((maxN + 127) / 128);
  if (ntasks <= 65535 && !gemm_prefers_128(ctx, (long)ntasks * tiles)) {  // (an upper bound of the tile count: tasks may be smaller)
    gemm_tasklist64_dev(ctx, dtasks, ntasks, maxM, maxN);
    return;
  }
  ProfScope ps(ctx, "gemm");
  for (int t0 = 0; t0 < ntasks; t0 += 65535) {
    int nt = std::min(65535, ntasks - t0);
    if (mfma4()) hipLaunchKernelGGL((k_dgemm_tasklist<128, 128, false, 0>), dim3(tiles, nt), dim3(256), 0, ctx->stream, dtasks + t0);
    else hipLaunchKernelGGL((k_dgemm_tasklist<128, 128>), dim3(tiles, nt), dim3(256), 0, ctx->stream, dtasks + t0);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

/// 128 x 128 tiles (two workgroups per CU, 59-63 TFLOP/s when they fill the chip) or 64 x 64 tiles (three per CU, 52-56)?
/// The large tiles only when their count fills whole rounds of the 2 x CU slots to 85 %: 484 tiles (n = 2816) 59.1
/// against 56.1 TFLOP/s, 2304 (n = 6102) 55.8 against 54.1, 4096 (n = 8192) 63.5 against 58.5 -- but 1156 (n = 4230, 2.26
/// rounds) 47.7 against 52.1, and the 386 tiles of the eigensolve's three blocks 34 against 37 (profiles/r03_gemm_bench.txt).
/// HELFEM_GEMM_TILE = 64 / 128 forces one.
bool gemm_prefers_128(hfg_ctx *ctx, long tiles128) {
  static const int force_tile = getenv("HELFEM_GEMM_TILE") ? atoi(getenv("HELFEM_GEMM_TILE")) : 0;
  if (force_tile) return force_tile == 128;
  static int slots = 0;
  if (!slots) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    slots = 2 * ncu;
  }
  const long rounds = (tiles128 + slots - 1) / slots;
  return tiles128 >= slots / 2 && (double)tiles128 >= 0.85 * (double)(rounds * slots);
}

void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc) {
  if (M <= 0 || N <= 0) return;
  ProfScope ps(ctx, "gemm");
  long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (gemm_prefers_128(ctx, big_tiles)) {
    if (mfma4())
      hipLaunchKernelGGL((k_dgemm<128, 128, 0>), dim3((unsigned)big_tiles), dim3(256), 0, ctx->stream, (int)tA, (int)tB, M, N, K,
                         alpha, A, lda, B, ldb, beta, C, ldc);
    else
      hipLaunchKernelGGL((k_dgemm<128, 128>), dim3((unsigned)big_tiles), dim3(256), 0, ctx->stream, (int)tA, (int)tB, M, N, K,
                         alpha, A, lda, B, ldb, beta, C, ldc);
  } else {
    long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (mfma4())
      hipLaunchKernelGGL((k_dgemm<64, 64, 0>), dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (int)tA, (int)tB, M, N, K, alpha, A,
                         lda, B, ldb, beta, C, ldc);
    else
      hipLaunchKernelGGL((k_dgemm<64, 64>), dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (int)tA, (int)tB, M, N, K, alpha, A,
                         lda, B, ldb, beta, C, ldc);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

}  // namespace hfg
