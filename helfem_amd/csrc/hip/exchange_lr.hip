// Exact exchange for low-rank densities (gfx950): the fast path of TwoDBasis::exchange
// (/root/reference/src/diatomic/basis.cpp:1532-1733, src/atomic/TwoDBasis.cpp:957-1140).
//
// Every density matrix an SCF run hands to exchange() is P = sum_o s_o l_o l_o^T with a handful of terms
// (occupied orbitals).  The reference does not use this: it forms, for every output shell pair (j,k) and
// channel, the R x R matrix  Rmat = sum_{i,l} cpl P_il  and pushes it through the radial integrals.  With the
// factors the same sums reorganise into dense products:
//
//   V^t[(j n),(c o)]   = sum_{i: m_i = m_j - M_c} c_t(j,i,L_c) l_o[(i n)]           channel c = (L,M), t = 0,2
//   cross-element part:  aP[(j e a),(c o)] = (P0_e V^0 - P2_e V^2)(a),  aQ likewise with the Q integrals,
//        G = (aQ diag(w)) aP^T   (one GEMM, w = LMfac_c s_o),   K(e>f) -= G[(j e a),(k f b)],  K(e<f) -= G^T
//   in-element part: for every primitive-table slot tau and element e one GEMM
//        C_tau,e[(a b),(j k)] = sum_{tt,(i' l')} ktei_tau,e,tt[(a b),(i' l')] RB_tau,e[(tt i' l'),(j k)]
//        with RB = +-sum_{c in tau, o} w V^t_j[i'] V^t'_k[l'] and ktei the exchange-ordered table
//        (utils::exchange_tei, libhelfem/src/utils.cpp:130), restricted to the shells that have a channel in tau.
//
// The factors come from a diagonally pivoted LDL^T of P (exact for the positive semi-definite, rank-nocc
// matrices of an SCF run; signs s_o make rank-deficient indefinite inputs work too).  The factorisation is
// verified (max |P - L S L^T|); if it does not reproduce P to 1e-13 within HFG_EXL_RMAX columns the caller falls
// back to the general kernels of exchange.hip.  Same sums as the reference, different association order.
#include "tables.h"
#include <algorithm>
#include <cstdlib>

namespace hfg {

void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);
void gemm_tasklist_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_split2_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_wl_dev(hfg_ctx *ctx, const GemmTask *dtasks, const int2 *dwl, int nwg, int tiles);
void gemm_tasklist_wl_split2_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, const int2 *dwl, int nwg);
void gemm_tasklist_split2_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);

constexpr int EXL_RMAX = 64;
constexpr int EXL_GMAX = 16;  // factor groups (residual factorisations) at most
constexpr int EXL_QMAX = 16;  // rows per thread in the factorisation kernel: N <= 1024 * EXL_QMAX

// -------------------------------------------------------------------------------------------------
// pivoted LDL^T, one workgroup
// -------------------------------------------------------------------------------------------------
// dscale: |first pivot| of the first group; the later groups (factorisations of the residual matrices, see
// exchange_lowrank_dev) stop relative to THAT, not to their own first pivot
__global__ __launch_bounds__(1024) void k_exl_factor(const double *__restrict__ P, int N, int rmax, double tol,
                                                     double *__restrict__ L, double *__restrict__ sgn,
                                                     int *__restrict__ info, double *__restrict__ dscale, int first) {
  __shared__ double red_v[16];
  __shared__ int red_i[16];
  __shared__ double lrow[EXL_RMAX];
  __shared__ double ssgn[EXL_RMAX];
  __shared__ double piv_d;
  __shared__ int piv_i;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double d[EXL_QMAX];
#pragma unroll
  for (int q = 0; q < EXL_QMAX; q++) {
    int i = tid + 1024 * q;
    d[q] = (i < N) ? P[(size_t)i * N + i] : 0.0;
  }
  double dmax0 = 0.0;
  int r = 0;
  for (int k = 0; k < rmax; k++) {
    // ---- pivot: largest |d|, ties to the smallest index (deterministic) ----
    double bv = -1.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < EXL_QMAX; q++) {
      int i = tid + 1024 * q;
      double a = fabs(d[q]);
      if (i < N && (a > bv || (a == bv && i < bi))) {
        bv = a;
        bi = i;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      double ov = __shfl_down(bv, o, 64);
      int oi = __shfl_down(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      red_v[wave] = bv;
      red_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      double v = red_v[0];
      int ii = red_i[0];
      for (int w = 1; w < 16; w++)
        if (red_v[w] > v || (red_v[w] == v && red_i[w] < ii)) {
          v = red_v[w];
          ii = red_i[w];
        }
      piv_i = ii;
    }
    __syncthreads();
    const int p = piv_i;
    // the owner of row p publishes d_p; the row of L built so far (stores of earlier steps, each followed by a workgroup
    // barrier) is fetched by k threads at once instead of one thread's k loads in a row
    if (tid == (p & 1023)) {
      double dv = 0.0;
#pragma unroll
      for (int q = 0; q < EXL_QMAX; q++)
        if (q == (p >> 10)) dv = d[q];
      piv_d = dv;
    }
    if (tid < k) lrow[tid] = L[(size_t)tid * N + p];
    __syncthreads();
    const double dp = piv_d;
    if (k == 0) dmax0 = first ? fabs(dp) : dscale[0];
    if (!(fabs(dp) > tol * dmax0) || dmax0 == 0.0) break;
    const double sk = (dp > 0.0) ? 1.0 : -1.0;
    const double inv = 1.0 / sqrt(fabs(dp));
    if (tid == 0) ssgn[k] = sk;
#pragma unroll
    for (int q = 0; q < EXL_QMAX; q++) {
      int i = tid + 1024 * q;
      if (i < N) {
        double v = P[(size_t)p * N + i];
        // eight loads of earlier columns in flight, subtracted in the same order as one by one
        int j = 0;
        for (; j + 8 <= k; j += 8) {
          double lv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) lv[u] = L[(size_t)(j + u) * N + i];
#pragma unroll
          for (int u = 0; u < 8; u++) v -= ssgn[j + u] * lv[u] * lrow[j + u];
        }
        for (; j < k; j++) v -= ssgn[j] * L[(size_t)j * N + i] * lrow[j];
        v *= inv;
        L[(size_t)k * N + i] = v;
        d[q] -= sk * v * v;
        if (i == p) d[q] = 0.0;
      }
    }
    r = k + 1;
    __syncthreads();
  }
  if (tid == 0) {
    info[0] = r;
    if (first) dscale[0] = dmax0;
  }
  __syncthreads();
  for (int j = tid; j < r; j += 1024) sgn[j] = ssgn[j];
}

__device__ inline void atomic_max_nonneg(double *addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

// LS[k][i] = sgn[k] L[k][i]  (operand of the residual update P <- P - L S L^T)
__global__ void k_exl_scale_cols(const double *__restrict__ L, const double *__restrict__ sgn, int N, int r, double *__restrict__ LS) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i < N && k < r) LS[(size_t)k * N + i] = sgn[k] * L[(size_t)k * N + i];
}

// dinfo[0] = max |P - L S L^T|, dinfo[1] = max |P|.  Workgroup = 256 rows x 32 columns: a thread keeps its row of L
// in registers, the 32 scaled rows of the column block sit in LDS.
__global__ __launch_bounds__(256) void k_exl_resid(const double *__restrict__ P, int N, const double *__restrict__ L,
                                                   const double *__restrict__ sgn, const int *__restrict__ info,
                                                   double *__restrict__ dinfo) {
  __shared__ double lj[32][EXL_RMAX + 1];
  const int r = info[0];
  const int j0 = blockIdx.y * 32;
  for (int t = threadIdx.x; t < 32 * r; t += 256) {
    int jj = t % 32, k = t / 32;
    lj[jj][k] = (j0 + jj < N) ? sgn[k] * L[(size_t)k * N + j0 + jj] : 0.0;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  double li[EXL_RMAX];
#pragma unroll
  for (int k = 0; k < EXL_RMAX; k++) li[k] = (k < r && i < N) ? L[(size_t)k * N + i] : 0.0;
  __syncthreads();
  double res = 0.0, pa = 0.0;
  if (i < N)
    for (int jj = 0; jj < 32 && j0 + jj < N; jj++) {
      double pv = P[(size_t)(j0 + jj) * N + i];
      double acc = pv;
#pragma unroll
      for (int k = 0; k < EXL_RMAX; k++)
        if (k < r) acc -= li[k] * lj[jj][k];
      res = fmax(res, fabs(acc));
      pa = fmax(pa, fabs(pv));
    }
  for (int o = 32; o > 0; o >>= 1) {
    res = fmax(res, __shfl_down(res, o, 64));
    pa = fmax(pa, __shfl_down(pa, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomic_max_nonneg(dinfo, res);
    atomic_max_nonneg(dinfo + 1, pa);
  }
}

// Ld[(x,n), o] (Nd x r): the factor in the (shell, dummy radial index) numbering
__global__ void k_exl_expand(const double *__restrict__ L, int N, int Nd, int R, int r, const int *__restrict__ shell_off,
                             const int *__restrict__ shell_skip, double *__restrict__ Ld) {
  int row = blockIdx.x * blockDim.x + threadIdx.x;
  int o = blockIdx.y;
  if (row >= Nd || o >= r) return;
  int x = row / R, n = row % R;
  double v = 0.0;
  if (!(shell_skip[x] && n == 0)) v = L[(size_t)o * N + shell_off[x] + n];
  Ld[(size_t)o * Nd + row] = v;
}

// V^t[(c,o)][(j,n)] = sum_{i: m_i = m_j - M_c} c_t(j,i,L_c) Ld[(i,n),o]
__global__ __launch_bounds__(256) void k_exl_V(const double *__restrict__ Ld, int Nd, int R, int A, int r, const int *__restrict__ LM_L,
                        const int *__restrict__ LM_M, const int *__restrict__ shell_m,
                        const double *__restrict__ c0tab, const double *__restrict__ c2tab, int Lp1, int two,
                        double *__restrict__ V0, double *__restrict__ V2) {
  // the shells i that couple to (channel c, shell j) and their two coefficients, once per workgroup (every output used to
  // walk all A shells through three dependent scalar loads each: 12 us per output)
  __shared__ int si[256];
  __shared__ double sa0[256], sa2[256];
  __shared__ int scount;
  const int c = blockIdx.x, j = blockIdx.y;
  const int L = LM_L[c], need = shell_m[j] - LM_M[c];
  if (threadIdx.x == 0) scount = 0;
  __syncthreads();
  for (int i0 = 0; i0 < A; i0 += 256) {  // (A <= 256 in one pass; the list keeps the order of i: fixed summation order)
    const int i = i0 + threadIdx.x;
    double a0 = 0.0, a2 = 0.0;
    bool on = false;
    if (i < A && shell_m[i] == need) {
      a0 = c0tab[((size_t)j * A + i) * Lp1 + L];
      a2 = two ? c2tab[((size_t)j * A + i) * Lp1 + L] : 0.0;
      on = (a0 != 0.0 || a2 != 0.0);
    }
    // ordered compaction: position = number of active threads before this one (wave ballots, four waves in order)
    __shared__ int wcnt[4];
    const unsigned long long bal = __ballot(on);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int base = scount;
    for (int w = 0; w < wave; w++) base += wcnt[w];
    if (on) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (pos < 256) {
        si[pos] = i;
        sa0[pos] = a0;
        sa2[pos] = a2;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) scount = min(256, scount + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
    __syncthreads();
  }
  const int cnt = scount;
  for (int t = threadIdx.x; t < R * r; t += blockDim.x) {
    int n = t % R, o = t / R;
    double u0 = 0.0, u2 = 0.0;
    for (int q = 0; q < cnt; q++) {
      const double lv = Ld[(size_t)o * Nd + si[q] * R + n];
      u0 += sa0[q] * lv;
      u2 += sa2[q] * lv;
    }
    size_t off = ((size_t)c * r + o) * Nd + (size_t)j * R + n;
    V0[off] = u0;
    if (two) V2[off] = u2;
  }
}

// general form (any p, any number of elements): one workgroup per (column, shell)
__global__ void k_exl_alpha_gen(const double *__restrict__ V0, const double *__restrict__ V2, const double *__restrict__ disj,
                            const int *__restrict__ LM_tab, const int *__restrict__ LM_ilm,
                            const double *__restrict__ LM_fac, const double *__restrict__ sgn, int Nd, int R, int A, int E,
                            int p, int r, int Ntab, int two, int rank, int nranks, double *__restrict__ aP,
                            double *__restrict__ aQw) {
  const int col = blockIdx.x, j = blockIdx.y;
  const int c = col / r, o = col % r;
  const int tab = LM_tab[c];
  const int pp = p * p;
  // multi-GPU: the (L,M) channels of the cross-element part are dealt out over the ranks
  const double w = (LM_ilm[c] % nranks == rank) ? LM_fac[c] * sgn[o] : 0.0;
  const int tQ0 = two ? 2 : 1;
  const size_t Na = (size_t)A * E * p;
  const double *v0 = V0 + (size_t)col * Nd + (size_t)j * R;
  const double *v2 = two ? V2 + (size_t)col * Nd + (size_t)j * R : nullptr;
  for (int t = threadIdx.x; t < E * p; t += blockDim.x) {
    int e = t / p, a = t % p;
    const double *P0 = disj + (((size_t)0 * Ntab + tab) * E + e) * pp;
    const double *Q0 = disj + (((size_t)tQ0 * Ntab + tab) * E + e) * pp;
    const double *P2 = two ? disj + (((size_t)1 * Ntab + tab) * E + e) * pp : nullptr;
    const double *Q2 = two ? disj + (((size_t)3 * Ntab + tab) * E + e) * pp : nullptr;
    double sp = 0.0, sq = 0.0;
    for (int cc = 0; cc < p; cc++) {
      int n = e * (p - 1) + cc;
      if (n >= R) continue;
      double x0 = v0[n];
      sp += P0[cc * p + a] * x0;
      sq += Q0[cc * p + a] * x0;
      if (two) {
        double x2 = v2[n];
        sp -= P2[cc * p + a] * x2;
        sq -= Q2[cc * p + a] * x2;
      }
    }
    size_t off = (size_t)col * Na + ((size_t)e * A + j) * p + a;  // rows ordered (element, shell, primitive)
    aP[off] = sp;
    aQw[off] = w * sq;
  }
}

// aP[(c,o)][(j,e,a)] = sum_c' P0_e(a,c') V0[j, e(p-1)+c'] - P2_e(a,c') V2[...];  aQw = w_(c,o) * (same with Q0, Q2)
// One workgroup per (channel, factor) column: thread (group g, element e, row a) keeps row a of the four p x p tables of
// its element in registers and walks over the shells j = g, g + EXL_AG, ...; the column of V is staged in LDS (the p
// values a thread needs per shell are the same for the p lanes of an (e, j): broadcast reads).  (One workgroup per (column, shell) with the tables read
// from L2 per output took 1.5 ms at Nbf = 4230: 268 000 workgroups of 75 active threads.)
constexpr int EXL_AG = 4;
constexpr int EXL_AP = 16;  // p at most
__global__ __launch_bounds__(512) void k_exl_alpha(const double *__restrict__ V0, const double *__restrict__ V2,
                                                   const double *__restrict__ disj, const int *__restrict__ LM_tab,
                                                   const int *__restrict__ LM_ilm, const double *__restrict__ LM_fac,
                                                   const double *__restrict__ sgn, int Nd, int R, int A, int E, int p, int r,
                                                   int Ntab, int two, int rank, int nranks, const int *__restrict__ ch_perm,
                                                   const int *__restrict__ sh_perm, const int *__restrict__ ch_q,
                                                   const int *__restrict__ sh_lo, const int *__restrict__ sh_hi,
                                                   double *__restrict__ aP, double *__restrict__ aQw) {
  extern __shared__ double xs[];  // this column of V0 (and of V2)
  const int col = blockIdx.x;
  const int c = col / r, o = col % r;
  const size_t colp = (size_t)(ch_perm ? ch_perm[c] : c) * r + o;  // output column: channels M-major
  const int tab = LM_tab[c];
  const int pp = p * p;
  // multi-GPU: the (L,M) channels of the cross-element part are dealt out over the ranks
  const double w = (LM_ilm[c] % nranks == rank) ? LM_fac[c] * sgn[o] : 0.0;
  const int tQ0 = two ? 2 : 1;
  const size_t Na = (size_t)A * E * p;
  const int ng = min(EXL_AG, (int)blockDim.x / (E * p));
  // thread (e, g, a): the lanes of one pass write ng * p consecutive doubles per element
  const int e = threadIdx.x / (ng * p), g = (threadIdx.x / p) % ng, a = threadIdx.x % p;
  const bool act = e < E;
  double tp0[EXL_AP], tq0[EXL_AP], tp2[EXL_AP], tq2[EXL_AP];
  {
    const int ee = act ? e : 0;
    const double *P0 = disj + (((size_t)0 * Ntab + tab) * E + ee) * pp;
    const double *Q0 = disj + (((size_t)tQ0 * Ntab + tab) * E + ee) * pp;
    const double *P2 = disj + (((size_t)1 * Ntab + tab) * E + ee) * pp;
    const double *Q2 = disj + (((size_t)3 * Ntab + tab) * E + ee) * pp;
    // unconditional loads with clamped indices, masked afterwards: loads under divergent branches are waited for one
    // by one (60 round trips per thread: most of this kernel's 0.6 ms)
    const double *P2c = two ? P2 : P0, *Q2c = two ? Q2 : Q0;
#pragma unroll
    for (int cc = 0; cc < EXL_AP; cc++) {
      const int ix = min(cc, p - 1) * p + a;
      tp0[cc] = P0[ix];
      tq0[cc] = Q0[ix];
      tp2[cc] = P2c[ix];
      tq2[cc] = Q2c[ix];
    }
#pragma unroll
    for (int cc = 0; cc < EXL_AP; cc++) {
      const bool in = act && cc < p && ee * (p - 1) + cc < R;
      if (!in) tp0[cc] = tq0[cc] = 0.0;
      if (!in || !two) tp2[cc] = tq2[cc] = 0.0;
    }
  }
  double *x0s = xs, *x2s = xs + Nd;
  {
    // eight rounds of loads in flight (a rolled "load, store to LDS" loop pays one memory round trip per round, and this
    // column of V was written a moment ago: it comes from beyond the L2)
    constexpr int UN = 8;
    const double *g0 = V0 + (size_t)col * Nd, *g2 = V2 + (size_t)col * Nd;
    for (int t0 = threadIdx.x; t0 < Nd; t0 += UN * blockDim.x) {
      double v0[UN], v2[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int t = min(t0 + u * (int)blockDim.x, Nd - 1);
        v0[u] = g0[t];
        v2[u] = g2[t];  // (V2 = V0 when there is one table type)
      }
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int t = t0 + u * (int)blockDim.x;
        if (t < Nd) {
          x0s[t] = v0[u];
          if (two) x2s[t] = v2[u];
        }
      }
    }
  }
  __syncthreads();
  if (!act) return;
  const int n0 = e * (p - 1);
  const int cq = sh_perm ? ch_q[c] : 0;
  for (int j = g; j < A; j += ng) {
    // blocks of equal m: the products never read a (shell, channel) whose V vanishes identically
    if (sh_perm && (cq < sh_lo[j] || cq > sh_hi[j])) continue;
    const double *x0 = x0s + j * R + n0, *x2 = x2s + j * R + n0;
    double sp = 0.0, sq = 0.0;
#pragma unroll
    for (int cc = 0; cc < EXL_AP; cc++) {
      const int n = max(0, min(cc, R - 1 - n0));  // clamped: the table entries beyond the basis are zero
      const double y0 = x0[n];
      sp += tp0[cc] * y0;
      sq += tq0[cc] * y0;
      if (two) {
        const double y2 = x2[n];
        sp -= tp2[cc] * y2;
        sq -= tq2[cc] * y2;
      }
    }
    const size_t off = colp * Na + ((size_t)e * A + (sh_perm ? sh_perm[j] : j)) * p + a;  // rows ordered (element, shell, primitive)
    aP[off] = sp;
    aQw[off] = w * sq;
  }
}

// ktei[tab][e][tt][(i' + p l')][(a + p b)] = tei_tt[tab][e][(a p + i'), (l' p + b)]   (utils::exchange_tei)
// One column-major block of Mld rows x Kcols columns per (tab, e), column tt pp + i' + p l', row a + p b; Mld >= pp and
// Kcols >= ntt pp pad the block for the GEMM's 16-byte loads (the padding is zeroed by the caller).
__global__ void k_exl_permute_tei(const double *__restrict__ tei, int Ntab, int E, int p, int ntt, int Mld, int Kcols,
                                  double *__restrict__ ktei) {
  const int blk = blockIdx.x;  // (tt*Ntab + tab)*E + e
  const int e = blk % E, tab = (blk / E) % Ntab, tt = blk / (E * Ntab);
  const int pp = p * p;
  const double *T = tei + (size_t)blk * pp * pp;
  double *K = ktei + (((size_t)tab * E + e) * Kcols + (size_t)tt * pp) * Mld;
  const int lp = blockIdx.y;
  for (int t = threadIdx.x; t < p * pp; t += blockDim.x) {
    int m = t % pp, ip = t / pp;
    int a = m % p, b = m / p;
    K[(size_t)(ip + p * lp) * Mld + m] = T[(size_t)(lp * p + b) * pp + a * p + ip];
  }
}

// RB_tau,e[(tt, i', l'), (pj, pk)] = sign_tt sum_{c in tau} sum_o w V^t[(c,o)][(j,e,i')] V^t'[(c,o)][(k,e,l')]
__global__ void k_exl_RB(const double *__restrict__ V0, const double *__restrict__ V2, const int *__restrict__ tab_ch_off,
                         const int *__restrict__ tab_ch, const double *__restrict__ LM_fac,
                         const double *__restrict__ sgn, const int *__restrict__ S_off, const int *__restrict__ S_list,
                         const long long *__restrict__ rb_off, int tau0, int Nd, int R, int E, int p, int r, int ntt,
                         double *__restrict__ RB) {
  extern __shared__ double sh[];  // vj[2][nco][p], vk[2][nco][p], w[nco]
  const int e = blockIdx.y;
  const int tau = tau0 + blockIdx.z;  // one launch covers all table slots (grid.z); slots of other ranks have no offset
  if (rb_off[tau] < 0) return;
  const int ns = S_off[tau + 1] - S_off[tau];
  const int n = blockIdx.x;  // pair pj <= pk, n = pk (pk + 1) / 2 + pj  (K is symmetric: K_kj = K_jk^T)
  const int npair = ns * (ns + 1) / 2;
  if (n >= npair) return;
  int pk = (int)((sqrt(8.0 * n + 1.0) - 1.0) * 0.5);
  while ((pk + 1) * (pk + 2) / 2 <= n) pk++;
  while (pk * (pk + 1) / 2 > n) pk--;
  const int pj = n - pk * (pk + 1) / 2;
  const int j = S_list[S_off[tau] + pj], k = S_list[S_off[tau] + pk];
  const int c0 = tab_ch_off[tau], nch = tab_ch_off[tau + 1] - c0;
  const int nco = nch * r;
  double *vj = sh, *vk = sh + 2 * nco * p, *w = sh + 4 * nco * p;
  const bool two = (ntt == 4);
  for (int t = threadIdx.x; t < nco * p; t += blockDim.x) {
    int ii = t % p, co = t / p;
    int c = tab_ch[c0 + co / r], o = co % r;
    int nn = e * (p - 1) + ii;
    size_t col = (size_t)c * r + o;
    bool ok = nn < R;
    vj[t] = ok ? V0[col * Nd + (size_t)j * R + nn] : 0.0;
    vk[t] = ok ? V0[col * Nd + (size_t)k * R + nn] : 0.0;
    vj[nco * p + t] = (ok && two) ? V2[col * Nd + (size_t)j * R + nn] : 0.0;
    vk[nco * p + t] = (ok && two) ? V2[col * Nd + (size_t)k * R + nn] : 0.0;
    if (ii == 0) w[co] = LM_fac[c] * sgn[o];
  }
  __syncthreads();
  const int pp = p * p;
  const int Kt = ntt * pp;
  double *out = RB + rb_off[tau] + ((size_t)e * npair + n) * Kt;
  // thread (i', l') forms all type combinations at once: 4 LDS reads per 4 FMAs
  for (int il = threadIdx.x; il < pp; il += blockDim.x) {
    int ip = il % p, lp = il / p;
    const double *a0 = vj, *a2 = vj + (size_t)nco * p, *b0 = vk, *b2 = vk + (size_t)nco * p;
    double s00 = 0.0, s02 = 0.0, s20 = 0.0, s22 = 0.0;
    if (two) {
      for (int co = 0; co < nco; co++) {
        double wa0 = w[co] * a0[co * p + ip], wa2 = w[co] * a2[co * p + ip];
        double x0 = b0[co * p + lp], x2 = b2[co * p + lp];
        s00 += wa0 * x0;
        s02 += wa0 * x2;
        s20 += wa2 * x0;
        s22 += wa2 * x2;
      }
      out[il] = s00;            // tt = 00
      out[pp + il] = -s02;      // 02
      out[2 * pp + il] = -s20;  // 20
      out[3 * pp + il] = s22;   // 22
    } else {
      for (int co = 0; co < nco; co++) s00 += w[co] * a0[co * p + ip] * b0[co * p + lp];
      out[il] = s00;
    }
  }
}

// The same RB blocks, four shells j against four shells k per workgroup: the factors of the eight shells are staged once
// for sixteen shell pairs (512 threads: staging with four rounds of loads in flight), and a thread (i', l') of each half
// keeps 4 x 2 pairs x 4 type combinations = 32 sums in registers -- 13 LDS reads per 32 FMAs instead of 5 per 4.  (The
// one-pair kernel above moves 20 KB from L2 into LDS per pair and is bound by the LDS port: 14 TFLOP/s.)
constexpr int EXL_SB = 4;  // shells per side of a block
constexpr int EXL_CK = 24;  // (channel, factor) columns staged at a time: 2 * 4 shells * 2 types * 24 * p doubles of LDS
__global__ __launch_bounds__(512) void k_exl_RB4(const double *__restrict__ V0, const double *__restrict__ V2,
                                                 const int *__restrict__ tab_ch_off, const int *__restrict__ tab_ch,
                                                 const double *__restrict__ LM_fac, const double *__restrict__ sgn,
                                                 const int *__restrict__ S_off, const int *__restrict__ S_list,
                                                 const long long *__restrict__ rb_off, int tau0, int Nd, int R, int E, int p, int r,
                                                 int ntt, double *__restrict__ RB) {
  extern __shared__ double sh[];  // vj[SB][2][CK][p], vk[SB][2][CK][p], w[CK]
  const int e = blockIdx.y;
  const int tau = tau0 + blockIdx.z;
  if (rb_off[tau] < 0) return;
  const int ns = S_off[tau + 1] - S_off[tau];
  const int nb = (ns + EXL_SB - 1) / EXL_SB;
  const int nblk = nb * (nb + 1) / 2;
  if ((int)blockIdx.x >= nblk) return;
  int bk = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((bk + 1) * (bk + 2) / 2 <= (int)blockIdx.x) bk++;
  while (bk * (bk + 1) / 2 > (int)blockIdx.x) bk--;
  const int bj = blockIdx.x - bk * (bk + 1) / 2;  // bj <= bk
  const int c0 = tab_ch_off[tau], nch = tab_ch_off[tau + 1] - c0;
  const int nco = nch * r;
  const bool two = (ntt == 4);
  const int half_slab = EXL_CK * p, slab = 2 * half_slab;  // doubles per type / per shell
  double *vj = sh, *vk = sh + EXL_SB * slab, *w = sh + 2 * EXL_SB * slab;
  const int pp = p * p, Kt = ntt * pp;
  const int npair = ns * (ns + 1) / 2;
  // the two halves of the 512 threads take two of the four k shells each: thread (i', l') of a half keeps 4 x 2 pairs x
  // 4 type combinations = 32 sums across the chunks (one (i', l') per thread: p <= 16)
  constexpr int KB = EXL_SB / 2;
  const int half = threadIdx.x >> 8, il = threadIdx.x & 255;
  const bool act = il < pp;
  const int ip = act ? il % p : 0, lp = act ? il / p : 0;
  double acc[EXL_SB][KB][4];
#pragma unroll
  for (int a = 0; a < EXL_SB; a++)
#pragma unroll
    for (int b = 0; b < KB; b++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[a][b][q] = 0.0;
  const double *vkh = vk + (size_t)half * KB * slab;
  for (int cb = 0; cb < nco; cb += EXL_CK) {
    const int nck = min(EXL_CK, nco - cb);
    if (cb) __syncthreads();  // the previous chunk has been consumed
    // staging: four rounds of loads in flight per thread (a rolled "load, store to LDS" loop pays one L2 round trip per round)
    {
      const int total = EXL_SB * nck * p;
      constexpr int UN = 4;
      for (int t0 = threadIdx.x; t0 < total; t0 += UN * blockDim.x) {
        double a0[UN], a2[UN], b0[UN], b2[UN];
        int at[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int t = t0 + u * blockDim.x;
          const bool in = t < total;
          const int tt = in ? t : 0;
          const int ii = tt % p, cl = (tt / p) % nck, sidx = tt / (p * nck);
          const int co = cb + cl;
          const int c = tab_ch[c0 + co / r], o = co % r;
          const int nn = e * (p - 1) + ii;
          const size_t col = (size_t)c * r + o;
          const int pj = EXL_SB * bj + sidx, pk = EXL_SB * bk + sidx;
          const bool okj = in && nn < R && pj < ns, okk = in && nn < R && pk < ns;
          const int j = S_list[S_off[tau] + min(pj, ns - 1)], k = S_list[S_off[tau] + min(pk, ns - 1)];
          const size_t oj = col * Nd + (size_t)j * R + min(nn, R - 1), ok_ = col * Nd + (size_t)k * R + min(nn, R - 1);
          at[u] = in ? sidx * slab + cl * p + ii : -1;
          a0[u] = V0[oj];
          b0[u] = V0[ok_];
          a2[u] = two ? V2[oj] : 0.0;
          b2[u] = two ? V2[ok_] : 0.0;
          if (!okj) a0[u] = a2[u] = 0.0;
          if (!okk) b0[u] = b2[u] = 0.0;
          if (in && ii == 0 && sidx == 0) w[cl] = LM_fac[c] * sgn[o];
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
          if (at[u] >= 0) {
            vj[at[u]] = a0[u];
            vk[at[u]] = b0[u];
            vj[at[u] + half_slab] = a2[u];
            vk[at[u] + half_slab] = b2[u];
          }
      }
    }
    __syncthreads();
    if (act)
      for (int cl = 0; cl < nck; cl++) {
        const double wc = w[cl];
        double wa0[EXL_SB], wa2[EXL_SB], x0[KB], x2[KB];
#pragma unroll
        for (int a = 0; a < EXL_SB; a++) {
          wa0[a] = wc * vj[a * slab + cl * p + ip];
          wa2[a] = two ? wc * vj[a * slab + half_slab + cl * p + ip] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < KB; b++) {
          x0[b] = vkh[b * slab + cl * p + lp];
          x2[b] = two ? vkh[b * slab + half_slab + cl * p + lp] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < EXL_SB; a++)
#pragma unroll
          for (int b = 0; b < KB; b++) {
            acc[a][b][0] += wa0[a] * x0[b];
            acc[a][b][1] += wa0[a] * x2[b];
            acc[a][b][2] += wa2[a] * x0[b];
            acc[a][b][3] += wa2[a] * x2[b];
          }
      }
  }
  if (!act) return;
#pragma unroll
  for (int a = 0; a < EXL_SB; a++)
#pragma unroll
    for (int b = 0; b < KB; b++) {
      const int pj = EXL_SB * bj + a, pk = EXL_SB * bk + half * KB + b;
      if (pj > pk || pk >= ns) continue;  // only pj <= pk is stored (K is symmetric)
      const int n = pk * (pk + 1) / 2 + pj;
      double *out = RB + rb_off[tau] + ((size_t)e * npair + n) * Kt;
      out[il] = acc[a][b][0];
      if (two) {
        out[pp + il] = -acc[a][b][1];
        out[2 * pp + il] = -acc[a][b][2];
        out[3 * pp + il] = acc[a][b][3];
      }
    }
}

// The same blocks on the matrix cores.  For one block of 4 x 4 shells the sums are one small product
//   D[(t2, b, l'), (t1, a, i')] = sum_cl  V_t2[k_b][l'; cl] * (w_cl V_t1[j_a][i'; cl]),     cl = (channel, factor),
// i.e. 8 x 8 tiles of 16 x 16 (p <= 16 rows of a shell per tile) with K = nco: v_mfma_f64_16x16x4_f64 with the a-operand
// taken from the k shells and the b-operand from the weighted j shells (operand maps: hip/gemm.hip), so that a result
// register of lane l holds (l' = (l >> 4) + 4 reg, i' = l & 15) and one store instruction writes four consecutive rows
// l' of a p x p block: 4 p contiguous doubles.  No LDS: the operands of a 16 x 4 fragment are p contiguous doubles per
// (channel, factor) column of V0 / V2 and go from L2 straight into the operand registers, two k-steps ahead of the
// instructions that use them.  512 threads = 8 waves of 4 x 2 tiles (types: wave & 1 for k, (wave >> 1) & 1 for j; j
// shells 2 (wave >> 2) + {0, 1}); without the second table type (atomic basis) 8 waves of 1 x 2 tiles.
typedef double exl_d4 __attribute__((ext_vector_type(4)));
template <bool TWO>
__global__ __launch_bounds__(512) void k_exl_RBm(const double *__restrict__ V0, const double *__restrict__ V2,
                                                 const int *__restrict__ tab_ch_off, const int *__restrict__ tab_ch,
                                                 const double *__restrict__ LM_fac, const double *__restrict__ sgn,
                                                 const int *__restrict__ S_off, const int *__restrict__ S_list,
                                                 const long long *__restrict__ rb_off, const int4 *__restrict__ wlist, int nwg,
                                                 int Nd, int R, int E, int p, int r, int Kld, double *__restrict__ RB) {
  // workgroup list (slot, element, block pair), slot-major: consecutive workgroup ids go round the 8 XCDs, so every XCD
  // takes one contiguous eighth of the list -- the blocks of a (slot, element) share their shells' columns of V in ONE L2
  int id = blockIdx.x;
  {
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = id % 8;
    id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + id / 8;
  }
  const int4 wd = wlist[id];
  const int tau = wd.x, e = wd.y, bj = wd.z, bk = wd.w;  // bj <= bk
  const int ns = S_off[tau + 1] - S_off[tau];
  const int c0 = tab_ch_off[tau], nco = (tab_ch_off[tau + 1] - c0) * r;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  constexpr int NR = TWO ? 4 : 1, NC = 2;
  const int t2 = TWO ? (wave & 1) : 0, t1 = TWO ? ((wave >> 1) & 1) : 0;
  const int b0 = TWO ? 0 : (wave >> 1), a0 = TWO ? 2 * (wave >> 2) : 2 * (wave & 1);
  const double *Vk = t2 ? V2 : V0, *Vj = t1 ? V2 : V0;
  const int nn = e * (p - 1) + l15;
  const bool okrow = l15 < p && nn < R;
  // element offsets of this lane's row in the k shells (a-operand) and the j shells (b-operand); -1: beyond the list
  long long offk[NR], offj[NC];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const int pk = EXL_SB * bk + b0 + i;
    offk[i] = (okrow && pk < ns) ? (long long)S_list[S_off[tau] + pk] * R + nn : -1;
  }
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const int pj = EXL_SB * bj + a0 + i;
    offj[i] = (okrow && pj < ns) ? (long long)S_list[S_off[tau] + pj] * R + nn : -1;
  }
  exl_d4 acc[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) acc[i][j] = exl_d4{0.0, 0.0, 0.0, 0.0};
  const int nsteps = (nco + 3) / 4;
  // column offset and weight of every (channel, factor) column once per workgroup (the divisions and the three dependent
  // index loads per fetch kept the vector ALU busier than the matrix pipe); columns nco .. 4 nsteps: weight 0, offset 0
  extern __shared__ double2 colw[];  // {bit pattern of the offset, weight}
  for (int cl = threadIdx.x; cl < 4 * nsteps; cl += blockDim.x) {
    double2 v = {0.0, 0.0};
    if (cl < nco) {
      const int c = tab_ch[c0 + cl / r], o = cl % r;
      v.x = __longlong_as_double((long long)(((size_t)c * r + o) * Nd));
      v.y = LM_fac[c] * sgn[o];
    }
    colw[cl] = v;
  }
  __syncthreads();
  auto fetch = [&](int s, double (&fa)[NR], double (&fb)[NC]) {
    const double2 cw = colw[4 * s + l4];
    const long long col = __double_as_longlong(cw.x);
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const double v = Vk[col + (offk[i] >= 0 ? offk[i] : 0)];
      fa[i] = offk[i] >= 0 ? v : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const double v = Vj[col + (offj[i] >= 0 ? offj[i] : 0)];
      fb[i] = offj[i] >= 0 ? cw.y * v : 0.0;
    }
  };
  auto mma = [&](const double (&fa)[NR], const double (&fb)[NC]) {
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < NC; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  double fa0[NR], fb0[NC], fa1[NR], fb1[NC];
  if (nsteps > 0) fetch(0, fa0, fb0);
  if (nsteps > 1) fetch(1, fa1, fb1);
  for (int s = 0; s < nsteps; s += 2) {
    mma(fa0, fb0);
    if (s + 2 < nsteps) fetch(s + 2, fa0, fb0);
    if (s + 1 < nsteps) {
      mma(fa1, fb1);
      if (s + 3 < nsteps) fetch(s + 3, fa1, fb1);
    }
  }
  const int pp = p * p, Kt = (TWO ? 4 : 1) * pp;
  const int npair = ns * (ns + 1) / 2;
  const double sg = (t1 != t2) ? -1.0 : 1.0;
  const int q = 2 * t1 + t2;
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) {
      const int pk = EXL_SB * bk + b0 + i, pj = EXL_SB * bj + a0 + j;
      if (pj > pk || pk >= ns) continue;  // only pj <= pk is stored (K is symmetric)
      double *col = RB + rb_off[tau] + ((size_t)e * npair + (size_t)pk * (pk + 1) / 2 + pj) * Kld;
      if (q == 0 && lane < Kld - Kt) col[Kt + lane] = 0.0;  // rows Kt .. Kld of the column: padding of the GEMM's k steps
      if (l15 >= p) continue;
      double *out = col + q * pp + l15;
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int lp = l4 + 4 * reg;
        if (lp < p) out[lp * p] = sg * acc[i][j][reg];
      }
    }
}

// Element-pair variant for kernels that do not factorise over elements (erfc, TwoDBasis.cpp:1262): one block per
// (e >= f) and ORDERED shell pair,
//   RB_tau,ef[(tt, i', l'), (pj, pk)] = sign_tt sum_{c in tau} sum_o w V^t[(c,o)][(j,e,i')] V^t'[(c,o)][(k,f,l')]
__device__ inline void exl_unpack_tri(int n, int &hi, int &lo) {  // n = hi (hi + 1) / 2 + lo, lo <= hi
  hi = (int)((sqrt(8.0 * n + 1.0) - 1.0) * 0.5);
  while ((hi + 1) * (hi + 2) / 2 <= n) hi++;
  while (hi * (hi + 1) / 2 > n) hi--;
  lo = n - hi * (hi + 1) / 2;
}

__global__ void k_exl_RB_pair(const double *__restrict__ V0, const double *__restrict__ V2,
                              const int *__restrict__ tab_ch_off, const int *__restrict__ tab_ch,
                              const double *__restrict__ LM_fac, const double *__restrict__ sgn,
                              const int *__restrict__ S_off, const int *__restrict__ S_list,
                              const long long *__restrict__ rb_off, int tau0, int Nd, int R, int E, int p, int r, int ntt,
                              double *__restrict__ RB) {
  extern __shared__ double sh[];  // vj[2][nco][p], vk[2][nco][p], w[nco]
  const int tau = tau0 + blockIdx.z;
  if (rb_off[tau] < 0) return;
  const int ef = blockIdx.y;
  int e, f;
  exl_unpack_tri(ef, e, f);
  const int ns = S_off[tau + 1] - S_off[tau];
  const int n = blockIdx.x;  // n = pk ns + pj
  if (n >= ns * ns) return;
  const int pj = n % ns, pk = n / ns;
  const int j = S_list[S_off[tau] + pj], k = S_list[S_off[tau] + pk];
  const int c0 = tab_ch_off[tau], nch = tab_ch_off[tau + 1] - c0;
  const int nco = nch * r;
  double *vj = sh, *vk = sh + 2 * nco * p, *w = sh + 4 * nco * p;
  const bool two = (ntt == 4);
  for (int t = threadIdx.x; t < nco * p; t += blockDim.x) {
    int ii = t % p, co = t / p;
    int c = tab_ch[c0 + co / r], o = co % r;
    int nj = e * (p - 1) + ii, nk = f * (p - 1) + ii;
    size_t col = (size_t)c * r + o;
    bool okj = nj < R, okk = nk < R;
    vj[t] = okj ? V0[col * Nd + (size_t)j * R + nj] : 0.0;
    vk[t] = okk ? V0[col * Nd + (size_t)k * R + nk] : 0.0;
    vj[nco * p + t] = (okj && two) ? V2[col * Nd + (size_t)j * R + nj] : 0.0;
    vk[nco * p + t] = (okk && two) ? V2[col * Nd + (size_t)k * R + nk] : 0.0;
    if (ii == 0) w[co] = LM_fac[c] * sgn[o];
  }
  __syncthreads();
  const int pp = p * p;
  const int Kt = ntt * pp;
  double *out = RB + rb_off[tau] + ((size_t)ef * ns * ns + n) * Kt;
  for (int il = threadIdx.x; il < pp; il += blockDim.x) {
    int ip = il % p, lp = il / p;
    const double *a0 = vj, *a2 = vj + (size_t)nco * p, *b0 = vk, *b2 = vk + (size_t)nco * p;
    double s00 = 0.0, s02 = 0.0, s20 = 0.0, s22 = 0.0;
    if (two) {
      for (int co = 0; co < nco; co++) {
        double wa0 = w[co] * a0[co * p + ip], wa2 = w[co] * a2[co * p + ip];
        double x0 = b0[co * p + lp], x2 = b2[co * p + lp];
        s00 += wa0 * x0;
        s02 += wa0 * x2;
        s20 += wa2 * x0;
        s22 += wa2 * x2;
      }
      out[il] = s00;
      out[pp + il] = -s02;
      out[2 * pp + il] = -s20;
      out[3 * pp + il] = s22;
    } else {
      for (int co = 0; co < nco; co++) s00 += w[co] * a0[co * p + ip] * b0[co * p + lp];
      out[il] = s00;
    }
  }
}

// sums the slots of the element-pair products: e == f goes to Kin (as k_exl_reduce), e > f to the cross-element block
// G_ef[(j,a),(k,b)] that k_exl_assemble reads (the e < f half of K is its transpose)
__global__ void k_exl_reduce_pair(const double *__restrict__ C, const long long *__restrict__ c_off,
                                  const int *__restrict__ S_off, const int *__restrict__ pos, int A, int E, int p, int Ntab,
                                  double *__restrict__ Kin, double *__restrict__ G) {
  const int jk = blockIdx.x, ef = blockIdx.y;
  int e, f;
  exl_unpack_tri(ef, e, f);
  const int j = jk / A, k = jk % A;
  const int pp = p * p;
  const size_t Ap = (size_t)A * p;
  for (int t = threadIdx.x; t < pp; t += blockDim.x) {
    const int a = t % p, b = t / p;
    double s = 0.0;
    for (int tau = 0; tau < Ntab; tau++) {
      if (c_off[tau] < 0) continue;
      int pj = pos[tau * A + j], pk = pos[tau * A + k];
      if (pj < 0 || pk < 0) continue;
      int ns = S_off[tau + 1] - S_off[tau];
      s += C[c_off[tau] + (((size_t)ef * ns + pk) * ns + pj) * pp + t];
    }
    if (e == f) Kin[((size_t)jk * E + e) * pp + t] = s;
    else G[((size_t)e * (e - 1) / 2 + f) * Ap * Ap + ((size_t)k * p + b) * Ap + ((size_t)j * p + a)] = s;
  }
}

// Kin[(j,k)][e][(a + p b)] = sum over the table slots that contain both shells of C_tau,e[(a b),(pj,pk)]; the GEMMs
// only cover pj <= pk, the other half is the transpose
__global__ __launch_bounds__(256) void k_exl_reduce(const double *__restrict__ C, const long long *__restrict__ c_off,
                                                    const int *__restrict__ S_off, const int *__restrict__ pos, int A, int E, int p,
                                                    int Ntab, double *__restrict__ Kin) {
  // the slots' block offsets for this (j, k, e) first (one thread per slot), then the sums with four loads in flight: the
  // rolled loop over the slots paid the index loads and the load of C one after the other, 126 times
  __shared__ long long soff[1024];  // offset of the block, -1: no contribution; bit 62: transposed
  const int jk = blockIdx.x, e = blockIdx.y;
  const int j = jk / A, k = jk % A;
  const int pp = p * p;
  constexpr long long TR = 1ll << 62;
  for (int t0 = 0; t0 < Ntab; t0 += 1024) {
    const int nt = min(1024, Ntab - t0);
    if (t0) __syncthreads();
    for (int u = threadIdx.x; u < nt; u += blockDim.x) {
      const int tau = t0 + u;
      long long o = -1;
      if (c_off[tau] >= 0) {
        const int pj = pos[tau * A + j], pk = pos[tau * A + k];
        if (pj >= 0 && pk >= 0) {
          const int ns = S_off[tau + 1] - S_off[tau];
          const size_t npair = (size_t)ns * (ns + 1) / 2;
          if (pj <= pk) o = c_off[tau] + (long long)(((size_t)e * npair + (size_t)pk * (pk + 1) / 2 + pj) * pp);
          else o = (c_off[tau] + (long long)(((size_t)e * npair + (size_t)pj * (pj + 1) / 2 + pk) * pp)) | TR;
        }
      }
      soff[u] = o;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < pp; t += blockDim.x) {
      const int a = t % p, b = t / p;
      const int in_n = a + p * b, in_t = b + p * a;
      double s = t0 ? Kin[((size_t)jk * E + e) * pp + t] : 0.0;
      int u = 0;
      for (; u + 4 <= nt; u += 4) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const long long o = soff[u + q];
          v[q] = (o < 0) ? 0.0 : C[(o & ~TR) + ((o & TR) ? in_t : in_n)];
        }
        s += v[0];
        s += v[1];
        s += v[2];
        s += v[3];
      }
      for (; u < nt; u++) {
        const long long o = soff[u];
        if (o >= 0) s += C[(o & ~TR) + ((o & TR) ? in_t : in_n)];
      }
      Kin[((size_t)jk * E + e) * pp + t] = s;
    }
  }
}

// K(pure row, pure col) = -(in-element + cross-element contributions)
// G holds, for every element pair e > f, the block G_ef[(j,a),(k,b)] (A p x A p, column-major) at ((e (e-1))/2 + f)
__global__ void k_exl_assemble(const double *__restrict__ Kin, const double *__restrict__ G, int N, int A, int E, int p,
                               const int *__restrict__ pure_shell, const int *__restrict__ pure_n,
                               const int *__restrict__ sh_perm, double *__restrict__ K, int accumulate) {
  int row = blockIdx.x * 64 + (threadIdx.x & 63);
  int col = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= N || col >= N) return;
  const int j = pure_shell[row], n = pure_n[row], k = pure_shell[col], m = pure_n[col];
  const int pm = p - 1, pp = p * p;
  const size_t Ap = (size_t)A * p;
  double v = 0.0;
  for (int ce = 0; ce < 2; ce++) {
    int e = n / pm - ce;
    if (e < 0 || e >= E) continue;
    int a = n - e * pm;
    if (a < 0 || a > pm) continue;
    for (int cf = 0; cf < 2; cf++) {
      int f = m / pm - cf;
      if (f < 0 || f >= E) continue;
      int b = m - f * pm;
      if (b < 0 || b > pm) continue;
      size_t ra = (size_t)(sh_perm ? sh_perm[j] : j) * p + a, rb = (size_t)(sh_perm ? sh_perm[k] : k) * p + b;  // G: shells m-major
      if (e == f) v += Kin[(((size_t)j * A + k) * E + e) * pp + a + p * b];
      else if (e > f) v += G[((size_t)e * (e - 1) / 2 + f) * Ap * Ap + rb * Ap + ra];
      else v += G[((size_t)f * (f - 1) / 2 + e) * Ap * Ap + ra * Ap + rb];
    }
  }
  if (accumulate) K[(size_t)col * N + row] -= v;
  else K[(size_t)col * N + row] = -v;
}

struct ExLRAux {
  DevBuf<double> c0tab, c2tab, ktei, L, sgn, dinfo, Ld, V0, V2, aP, aQw, G, RB, C, Kin, Pwork, LS;
  int kM = 0, kK = 0;  // rows and columns of one exchange-ordered element table (padded, see exlr_for)
  DevBuf<int4> rbm_list;  // workgroups of k_exl_RBm: (slot, element, block pair)
  DevBuf<double> ones;     // EXL_RMAX ones: the signs of factors handed in by the caller
  DevBuf<int2> gwl;        // workgroups (task, tile) of the element GEMM, XCD-contiguous
  std::vector<int2> h_gwl;
  DevBuf<int2> cwl;        // the same for the cross-element products (two entries per tile: split K)
  std::vector<int2> h_cwl;
  int rbm_n = 0, rbm_shard = -1;
  DevBuf<int> info, LM_L, LM_M, tab_ch_off, tab_ch, S_off, S_list, pos, pure_shell, pure_n;
  DevBuf<long long> rb_off, c_off;
  DevBuf<GemmTask> tasks, ctasks;
  std::vector<int> hS_off;
  std::vector<int> h_nch;  // channels per slot
  // cross-element products by blocks of equal m: in aP / aQw / G the shells are renumbered m-major (sh_perm: the shells
  // of one m are a run [j0, j0 + nj)) and the channels M-major (ch_perm), chM0[q] .. chM0[q + 1] are the channels of the q-th value of M, and run g
  // has a non-zero V on the M values act_lo[g] .. act_hi[g] only (a contiguous interval: M = m_j - m_i)
  bool cross_ok = false;
  DevBuf<int> ch_perm, sh_perm;  // channel -> M-major position, shell -> m-major position
  DevBuf<int> ch_q, sh_lo, sh_hi;  // channel -> index of its M; shell -> the interval of M indices its run reaches
  struct MRun { int j0, nj, lo, hi; };
  std::vector<MRun> runs;
  std::vector<int> chM0;
  int max_nch = 0;
};
static std::map<hfg_dev_tables *, ExLRAux *> g_exlr;

void exchange_lr_release(hfg_dev_tables *t) {
  auto it = g_exlr.find(t);
  if (it != g_exlr.end()) {
    delete it->second;
    g_exlr.erase(it);
  }
}

static ExLRAux &exlr_for(hfg_ctx *ctx, hfg_dev_tables *t) {
  auto it = g_exlr.find(t);
  if (it != g_exlr.end()) return *it->second;
  ExLRAux *a = new ExLRAux();
  hipStream_t s = ctx->stream;
  const int A = t->A, Ntab = t->Ntab, NLM = t->NLM, Lp1 = t->Lp1;
  a->c0tab.upload(t->h_c0tab, s);
  a->c2tab.upload(t->h_c2tab, s);
  a->LM_L.upload(t->h_LM_L, s);
  a->LM_M.upload(t->h_LM_M, s);
  // channels per table slot, shells per table slot
  std::vector<std::vector<int> > ch_of(Ntab), S_of(Ntab);
  std::vector<int> LM_tab(NLM);
  for (int c = 0; c < NLM; c++) {
    LM_tab[c] = t->h_lm_tab[t->h_LM_ilm[c]];
    ch_of[LM_tab[c]].push_back(c);
  }
  std::vector<int> pos((size_t)Ntab * A, -1);
  for (int tau = 0; tau < Ntab; tau++)
    for (int j = 0; j < A; j++) {
      bool has = false;
      for (int c : ch_of[tau]) {
        int L = t->h_LM_L[c], need = t->h_shell_m[j] - t->h_LM_M[c];
        for (int i = 0; i < A && !has; i++)
          if (t->h_shell_m[i] == need && (t->h_c0tab[((size_t)j * A + i) * Lp1 + L] != 0.0 ||
                                          t->h_c2tab[((size_t)j * A + i) * Lp1 + L] != 0.0))
            has = true;
        if (has) break;
      }
      if (has) {
        pos[(size_t)tau * A + j] = (int)S_of[tau].size();
        S_of[tau].push_back(j);
      }
    }
  std::vector<int> ch_off(Ntab + 1, 0), ch_list, S_off(Ntab + 1, 0), S_list;
  a->h_nch.resize(Ntab);
  for (int tau = 0; tau < Ntab; tau++) {
    ch_off[tau] = (int)ch_list.size();
    ch_list.insert(ch_list.end(), ch_of[tau].begin(), ch_of[tau].end());
    S_off[tau] = (int)S_list.size();
    S_list.insert(S_list.end(), S_of[tau].begin(), S_of[tau].end());
    a->h_nch[tau] = (int)ch_of[tau].size();
    a->max_nch = std::max(a->max_nch, a->h_nch[tau]);
  }
  ch_off[Ntab] = (int)ch_list.size();
  S_off[Ntab] = (int)S_list.size();
  a->hS_off = S_off;
  a->tab_ch_off.upload(ch_off, s);
  a->tab_ch.upload(ch_list, s);
  a->S_off.upload(S_off, s);
  a->S_list.upload(S_list, s);
  a->pos.upload(pos, s);
  {
    // channels M-major (stable), runs of shells with equal m, and for every run the interval of M values on which
    // V^t_j is not identically zero (the same test as for the slots above)
    std::vector<int> Mv(t->h_LM_M.begin(), t->h_LM_M.begin() + NLM);
    std::vector<int> Ms = Mv;
    std::sort(Ms.begin(), Ms.end());
    Ms.erase(std::unique(Ms.begin(), Ms.end()), Ms.end());
    std::vector<int> perm(NLM), cnt(Ms.size() + 1, 0);
    for (int c = 0; c < NLM; c++) cnt[(std::lower_bound(Ms.begin(), Ms.end(), Mv[c]) - Ms.begin()) + 1]++;
    for (size_t q = 0; q < Ms.size(); q++) cnt[q + 1] += cnt[q];
    a->chM0 = cnt;
    {
      std::vector<int> fill(cnt.begin(), cnt.end() - 1);
      for (int c = 0; c < NLM; c++) perm[c] = fill[std::lower_bound(Ms.begin(), Ms.end(), Mv[c]) - Ms.begin()]++;
    }
    a->ch_perm.upload(perm, s);
    // shells m-major (stable)
    std::vector<int> ms(t->h_shell_m.begin(), t->h_shell_m.begin() + A);
    std::vector<int> mu = ms;
    std::sort(mu.begin(), mu.end());
    mu.erase(std::unique(mu.begin(), mu.end()), mu.end());
    std::vector<int> sperm(A);
    int posn = 0;
    for (int mval : mu) {
      ExLRAux::MRun run{posn, 0, (int)Ms.size(), -1};
      std::vector<char> act(Ms.size(), 0);
      for (int jj = 0; jj < A; jj++) {
        if (ms[jj] != mval) continue;
        sperm[jj] = posn++;
        run.nj++;
        for (int c = 0; c < NLM; c++) {
          const int q = (int)(std::lower_bound(Ms.begin(), Ms.end(), Mv[c]) - Ms.begin());
          if (act[q]) continue;
          const int L = t->h_LM_L[c], need = ms[jj] - Mv[c];
          for (int i = 0; i < A; i++)
            if (ms[i] == need && (t->h_c0tab[((size_t)jj * A + i) * Lp1 + L] != 0.0 ||
                                  t->h_c2tab[((size_t)jj * A + i) * Lp1 + L] != 0.0)) {
              act[q] = 1;
              break;
            }
        }
      }
      for (size_t q = 0; q < Ms.size(); q++)
        if (act[q]) {
          run.lo = std::min(run.lo, (int)q);
          run.hi = std::max(run.hi, (int)q);
        }
      a->runs.push_back(run);
    }
    a->sh_perm.upload(sperm, s);
    {
      std::vector<int> chq(NLM), slo(A), shi(A);
      for (int c = 0; c < NLM; c++) chq[c] = (int)(std::lower_bound(Ms.begin(), Ms.end(), Mv[c]) - Ms.begin());
      for (int jj = 0; jj < A; jj++) {
        const ExLRAux::MRun &run = a->runs[std::lower_bound(mu.begin(), mu.end(), ms[jj]) - mu.begin()];
        slo[jj] = run.lo;
        shi[jj] = run.hi;
      }
      a->ch_q.upload(chq, s);
      a->sh_lo.upload(slo, s);
      a->sh_hi.upload(shi, s);
    }
    a->cross_ok = a->runs.size() > 1;
  }
  std::vector<int> ps, pn;
  for (int x = 0; x < A; x++)
    for (int n = (t->h_shell_skip[x] ? 1 : 0); n < t->R; n++) {
      ps.push_back(x);
      pn.push_back(n);
    }
  a->pure_shell.upload(ps, s);
  a->pure_n.upload(pn, s);
  // exchange-ordered primitive tables
  // (pair tables, erfc: one block per ordered element pair, the permutation is the same with E^2 "elements")
  const size_t pp = (size_t)t->p * t->p;
  const int nper = t->pair_tei ? t->E * t->E : t->E;
  const std::vector<double> ones_h((size_t)EXL_RMAX, 1.0);  // alive until the synchronisation below
  a->ones.upload(ones_h, s);
  // element tables: rows padded to whole 128-row tiles, columns to a multiple of the GEMM's k step (zeros)
  a->kM = t->pair_tei ? (int)pp : (int)((pp + 127) / 128 * 128);
  a->kK = t->pair_tei ? (int)(t->ntt * pp) : (int)((t->ntt * pp + 15) / 16 * 16);
  a->ktei.resize((size_t)Ntab * nper * a->kM * a->kK);
  HFG_HIP_CHECK(hipMemsetAsync(a->ktei.p, 0, sizeof(double) * (size_t)Ntab * nper * a->kM * a->kK, s));
  hipLaunchKernelGGL(k_exl_permute_tei, dim3(t->ntt * Ntab * nper, t->p), dim3(256), 0, s, t->tei.p, Ntab, nper, t->p,
                     t->ntt, a->kM, a->kK, a->ktei.p);
  HFG_HIP_CHECK(hipStreamSynchronize(s));
  g_exlr[t] = a;
  return *a;
}

/// K from P through the low-rank factors.  Returns false (nothing written) when P is not reproduced by at most
/// EXL_RMAX factors; the caller then runs the general kernels.
/// Lknown / rknown: the caller KNOWS P = sum_o l_o l_o^T (the device-resident SCF loop: P = C_occ C_occ^T, formed from the
/// same columns one line earlier): the factorisation, its verification and their host synchronisation are skipped.
bool exchange_lowrank_dev(hfg_ctx *ctx, hfg_dev_tables *t, const double *dP, double *dK, const double *Lknown, int rknown) {
  const int A = t->A, R = t->R, E = t->E, p = t->p, Nd = t->Nd, N = t->N, NLM = t->NLM, Ntab = t->Ntab, ntt = t->ntt;
  if (N > 1024 * EXL_QMAX) return false;
  // erfc kernel (pair_tei): no factorisation over elements; the element-pair products below need an exchange-ordered
  // copy of the pair tables
  if (t->pair_tei && (double)ntt * Ntab * E * E * p * p * p * p * sizeof(double) > 32e9) return false;
  static const bool pair_off = getenv("HELFEM_EXL_PAIR") && atoi(getenv("HELFEM_EXL_PAIR")) == 0;
  if (t->pair_tei && pair_off) return false;
  ExLRAux &a = exlr_for(ctx, t);
  for (const ExLRAux::MRun &run : a.runs)
    if (run.nj > 256) return false;  // k_exl_V lists at most 256 coupled shells
  hipStream_t s = ctx->stream;
  const int two = (ntt == 4) ? 1 : 0;
  const int pp = p * p;

  // ---- factorise and verify ----
  // Up to EXL_GMAX groups of EXL_RMAX factors: when the first group does not reproduce P, the residual matrix
  // P - L S L^T is factorised in turn, and so on; K is linear in P, so the groups' exchange matrices add up (a density of
  // rank 65 costs two passes, not the general kernels' seventy-fold).  All factorisations come first: an input that is not
  // of low rank is handed to the general kernels before any exchange work is done.
  static const int gmax = getenv("HELFEM_EXL_GROUPS") ? std::max(1, std::min(EXL_GMAX, atoi(getenv("HELFEM_EXL_GROUPS")))) : 4;
  a.L.resize((size_t)N * EXL_RMAX * gmax);
  a.sgn.resize((size_t)EXL_RMAX * gmax);
  a.info.resize(4);
  a.dinfo.resize(4);
  std::vector<int> rg;
  double pmax = 0.0;
  bool reproduced = false;
  const double *Pcur = dP;
  static const bool no_hint = getenv("HELFEM_EXL_HINT") && atoi(getenv("HELFEM_EXL_HINT")) == 0;  // checker: always factorise
  if (Lknown && rknown >= 0 && rknown <= EXL_RMAX && !no_hint) {
    if (rknown > 0) {
      HFG_HIP_CHECK(hipMemcpyAsync(a.L.p, Lknown, sizeof(double) * (size_t)N * rknown, hipMemcpyDeviceToDevice, s));
      HFG_HIP_CHECK(hipMemcpyAsync(a.sgn.p, a.ones.p, sizeof(double) * (size_t)rknown, hipMemcpyDeviceToDevice, s));
    }
    rg.push_back(rknown);
    reproduced = true;
  }
  for (int g = 0; g < gmax && !reproduced; g++) {
    double *Lg = a.L.p + (size_t)g * N * EXL_RMAX, *sg = a.sgn.p + (size_t)g * EXL_RMAX;
    HFG_HIP_CHECK(hipMemsetAsync(a.dinfo.p, 0, 2 * sizeof(double), s));
    hipLaunchKernelGGL(k_exl_factor, dim3(1), dim3(1024), 0, s, Pcur, N, EXL_RMAX, 1e-14, Lg, sg, a.info.p, a.dinfo.p + 2, g == 0 ? 1 : 0);
    hipLaunchKernelGGL(k_exl_resid, dim3((N + 255) / 256, (N + 31) / 32), dim3(256), 0, s, Pcur, N, Lg, sg, a.info.p, a.dinfo.p);
    int r = 0;
    double dinfo[2] = {0, 0};
    HFG_HIP_CHECK(hipMemcpyAsync(&r, a.info.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipMemcpyAsync(dinfo, a.dinfo.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    if (g == 0) pmax = dinfo[1];
    rg.push_back(r);
    reproduced = (dinfo[0] <= 1e-13 * pmax);
    if (reproduced || r == 0 || g + 1 == gmax) break;
    // residual matrix for the next group
    a.Pwork.resize((size_t)N * N);
    a.LS.resize((size_t)N * EXL_RMAX);
    if (g == 0) HFG_HIP_CHECK(hipMemcpyAsync(a.Pwork.p, dP, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_exl_scale_cols, dim3((N + 255) / 256, r), dim3(256), 0, s, Lg, sg, N, r, a.LS.p);
    gemm_dev(ctx, false, true, N, N, r, -1.0, Lg, N, a.LS.p, N, 1.0, a.Pwork.p, N);
    Pcur = a.Pwork.p;
  }
  if (!reproduced) return false;
  int rmax_g = 0;
  for (int r : rg) rmax_g = std::max(rmax_g, r);
  {
    // the one-pair kernels (erfc pair tables, p > 16, HELFEM_EXL_RB=1) stage all (channel, factor) columns at once
    const bool one_pair = t->pair_tei || p > 16 || (getenv("HELFEM_EXL_RB") && atoi(getenv("HELFEM_EXL_RB")) == 1);
    if (one_pair && (size_t)(4 * a.max_nch * rmax_g * p + a.max_nch * rmax_g) * sizeof(double) > 150 * 1024) return false;
  }
  if (rmax_g == 0) {  // P == 0
    HFG_HIP_CHECK(hipMemsetAsync(dK, 0, sizeof(double) * (size_t)N * N, s));
    return true;
  }
  bool wrote = false;
  for (size_t g = 0; g < rg.size(); g++) {
  const int r = rg[g];
  if (r == 0) continue;
  const double *Lgrp = a.L.p + g * (size_t)N * EXL_RMAX, *sgrp = a.sgn.p + g * (size_t)EXL_RMAX;
  const int accumulate = wrote ? 1 : 0;
  wrote = true;

  // ---- angular stage ----
  const size_t ncol = (size_t)NLM * r;
  const size_t Na = (size_t)A * E * p;
  a.Ld.resize((size_t)Nd * r);
  a.V0.resize(ncol * Nd);
  if (two) a.V2.resize(ncol * Nd);
  a.aP.resize(ncol * Na);
  a.aQw.resize(ncol * Na);
  const size_t Ap = (size_t)A * p;
  const bool pair = t->pair_tei != 0;
  a.G.resize(std::max<size_t>((size_t)E * (E - 1) / 2 * Ap * Ap, 1));
  hipLaunchKernelGGL(k_exl_expand, dim3((Nd + 255) / 256, r), dim3(256), 0, s, Lgrp, N, Nd, R, r, t->shell_off.p,
                     t->shell_skip.p, a.Ld.p);
  hipLaunchKernelGGL(k_exl_V, dim3(NLM, A), dim3(256), 0, s, a.Ld.p, Nd, R, A, r, a.LM_L.p, a.LM_M.p, t->shell_m.p,
                     a.c0tab.p, a.c2tab.p, t->Lp1, two, a.V0.p, a.V2.p);
  // cross products by blocks of equal m (below) need the M-major column order, which only the fast alpha kernel writes
  static const bool group_off = getenv("HELFEM_EXL_MGROUPS") && atoi(getenv("HELFEM_EXL_MGROUPS")) == 0;
  const size_t alds = (size_t)(two ? 2 : 1) * Nd * sizeof(double);
  const bool alpha_fast = !(p > EXL_AP || E * p > 512 || alds > 150 * 1024);
  const bool grouped = !pair && a.cross_ok && alpha_fast && !group_off;
  if (!pair) {
  if (!alpha_fast)
    hipLaunchKernelGGL(k_exl_alpha_gen, dim3((unsigned)ncol, A), dim3(128), 0, s, a.V0.p, a.V2.p, t->disj.p, t->LM_tab.p,
                       t->LM_ilm.p, t->LM_fac.p, sgrp, Nd, R, A, E, p, r, Ntab, two, ctx->shard_rank, ctx->shard_n, a.aP.p,
                       a.aQw.p);
  else {
    // threads: (element, group, row) -- as many groups of shells as fit into 512 threads, EXL_AG at most
    const int per = E * p, ng = std::max(1, std::min(EXL_AG, 512 / per));
    if (alds > 64 * 1024)
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_exl_alpha, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alds));
    hipLaunchKernelGGL(k_exl_alpha, dim3((unsigned)ncol), dim3(ng * per), alds, s, a.V0.p, two ? a.V2.p : a.V0.p, t->disj.p, t->LM_tab.p,
                       t->LM_ilm.p, t->LM_fac.p, sgrp, Nd, R, A, E, p, r, Ntab, two, ctx->shard_rank, ctx->shard_n,
                       grouped ? a.ch_perm.p : nullptr, grouped ? a.sh_perm.p : nullptr, a.ch_q.p, a.sh_lo.p, a.sh_hi.p, a.aP.p,
                       a.aQw.p);
  }
  }
  // ---- cross-element part: G_ef = aQw_e aP_f^T for e > f (the other half of K is its transpose) ----
  if (!pair) {
    std::vector<GemmTask> ct;
    int maxMN = 0;
    long tiles = 0;
    for (int e = 1; e < E; e++)
      for (int f = 0; f < e; f++) {
        GemmTask g;
        g.lda = g.ldb = (int)Na;
        g.ldc = (int)Ap;
        g.tB = 1;
        double *Gef = a.G.p + ((size_t)e * (e - 1) / 2 + f) * Ap * Ap;
        if (!grouped) {
          g.A = a.aQw.p + (size_t)e * Ap;
          g.B = a.aP.p + (size_t)f * Ap;
          g.C = Gef;
          g.M = g.N = (int)Ap;
          g.K = (int)ncol;
          ct.push_back(g);
          maxMN = (int)Ap;
          tiles += (long)((Ap + 127) / 128) * ((Ap + 127) / 128);
          continue;
        }
        // V^t_j vanishes unless some shell i has m_i = m_j - M: a block (run of m_j, run of m_k) of G only sums over the
        // channels whose M both runs reach (43 % of the flops of the full products for sigma + pi shells)
        for (const ExLRAux::MRun &rj : a.runs)
          for (const ExLRAux::MRun &rk : a.runs) {
            const int lo = std::max(rj.lo, rk.lo), hi = std::min(rj.hi, rk.hi);
            if (lo > hi) continue;  // G was zeroed
            const int k0 = a.chM0[lo], k1 = a.chM0[hi + 1];
            g.A = a.aQw.p + (size_t)k0 * r * Na + (size_t)e * Ap + (size_t)rj.j0 * p;
            g.B = a.aP.p + (size_t)k0 * r * Na + (size_t)f * Ap + (size_t)rk.j0 * p;
            g.C = Gef + (size_t)rk.j0 * p * Ap + (size_t)rj.j0 * p;
            g.M = rj.nj * p;
            g.N = rk.nj * p;
            g.K = (k1 - k0) * r;
            ct.push_back(g);
            maxMN = std::max(maxMN, std::max(g.M, g.N));
            tiles += (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
          }
      }
    if (grouped) HFG_HIP_CHECK(hipMemsetAsync(a.G.p, 0, sizeof(double) * (size_t)E * (E - 1) / 2 * Ap * Ap, s));
    if (!ct.empty()) {
      a.ctasks.upload(ct, s);
      HFG_HIP_CHECK(hipStreamSynchronize(s));  // ct lives on this stack frame
      // few large tiles (ten products of 8 x 8 tiles at Nbf = 4230: 640 tiles on 512 slots run as two rounds): two
      // half-K workgroups per tile into a zeroed G fill the slots evenly (two addends per element: deterministic)
      static const bool nosplit = getenv("HELFEM_EXL_SPLITK") && atoi(getenv("HELFEM_EXL_SPLITK")) == 0;
      if (!nosplit && tiles < 2048 && ncol >= 512) {
        if (!grouped) HFG_HIP_CHECK(hipMemsetAsync(a.G.p, 0, sizeof(double) * ct.size() * Ap * Ap, s));
        // 128 x 64 tiles: the blocks of ~300 columns pad to 320 instead of 384 (11.6 -> 11.5 ms per build; HELFEM_EXL_CRECT=0: square)
        static const bool crect = !(getenv("HELFEM_EXL_CRECT") && atoi(getenv("HELFEM_EXL_CRECT")) == 0);
        static const bool cwl_off = getenv("HELFEM_EXL_WL") && atoi(getenv("HELFEM_EXL_WL")) == 0;
        if (crect && grouped && !cwl_off) {
          // all tiles of a product on one XCD: its operand blocks are fetched once (2.4 GB were fetched for 0.3 GB of aP, aQw)
          std::vector<int2> &wl = a.h_cwl;
          wl.clear();
          for (size_t k = 0; k < ct.size(); k++) {
            const int nt = ((ct[k].M + 127) / 128) * ((ct[k].N + 63) / 64);
            for (int q = 0; q < 2 * nt; q++) wl.push_back(make_int2((int)k, q));
          }
          a.cwl.upload(wl, s);
          gemm_tasklist_wl_split2_rect_dev(ctx, a.ctasks.p, a.cwl.p, (int)wl.size());
        } else if (crect && grouped) gemm_tasklist_split2_rect_dev(ctx, a.ctasks.p, (int)ct.size(), maxMN, maxMN);
        else gemm_tasklist_split2_dev(ctx, a.ctasks.p, (int)ct.size(), maxMN, maxMN);
      } else
        gemm_tasklist_dev(ctx, a.ctasks.p, (int)ct.size(), maxMN, maxMN);
    }
  }

  // ---- in-element part: one GEMM per (table slot, element); slots are dealt out over the ranks ----
  std::vector<long long> rb_off(Ntab, -1), c_off(Ntab, -1);
  std::vector<GemmTask> tasks;
  const int Kt = ntt * pp;
  // the matrix-core RB kernel writes columns of a.kK >= Kt rows (zero padded like the element tables): every k step of
  // the GEMM is then a full one; the vector kernels (checkers, p > 16) and the pair tables keep the unpadded columns
  static const int rb_env = getenv("HELFEM_EXL_RB") ? atoi(getenv("HELFEM_EXL_RB")) : 0;
  const bool rb_mfma = !pair && p <= 16 && rb_env != 1 && rb_env != 4;
  const int Kld = rb_mfma ? a.kK : Kt;
  size_t rb_tot = 0, c_tot = 0;
  int maxN = 0;
  for (int tau = 0; tau < Ntab; tau++) {
    int ns = a.hS_off[tau + 1] - a.hS_off[tau];
    if (ns == 0 || (tau % ctx->shard_n) != ctx->shard_rank) continue;
    rb_off[tau] = (long long)rb_tot;
    c_off[tau] = (long long)c_tot;
    // blocks per slot: elements x unordered shell pairs, or (pair tables) element pairs e >= f x ordered shell pairs
    const size_t nblk = pair ? (size_t)E * (E + 1) / 2 : (size_t)E, ncols = pair ? (size_t)ns * ns : (size_t)ns * (ns + 1) / 2;
    rb_tot += nblk * ncols * Kld;
    c_tot += nblk * ncols * pp;
    maxN = std::max(maxN, (int)ncols);
  }
  a.RB.resize(rb_tot + 128 * (size_t)Kld);  // slack: the GEMM may read up to the edge of its last column tile
  a.C.resize(std::max<size_t>(c_tot, 1));
  for (int tau = 0; tau < Ntab; tau++) {
    if (rb_off[tau] < 0) continue;
    int ns = a.hS_off[tau + 1] - a.hS_off[tau];
    if (pair) {
      const size_t ncols = (size_t)ns * ns;
      for (int e = 0; e < E; e++)
        for (int f = 0; f <= e; f++) {
          const size_t ef = (size_t)e * (e + 1) / 2 + f;
          GemmTask g;
          g.A = a.ktei.p + ((((size_t)tau * E + e) * E + f) * ntt) * (size_t)pp * pp;
          g.B = a.RB.p + rb_off[tau] + ef * ncols * Kt;
          g.C = a.C.p + c_off[tau] + ef * ncols * pp;
          g.M = pp;
          g.N = (int)ncols;
          g.K = Kt;
          g.lda = pp;
          g.ldb = Kt;
          g.ldc = pp;
          tasks.push_back(g);
        }
      continue;
    }
    const size_t npair = (size_t)ns * (ns + 1) / 2;
    for (int e = 0; e < E; e++) {
      GemmTask g;
      g.A = a.ktei.p + ((size_t)tau * E + e) * (size_t)a.kM * a.kK;
      g.B = a.RB.p + rb_off[tau] + (size_t)e * npair * Kld;
      g.C = a.C.p + c_off[tau] + (size_t)e * npair * pp;
      g.M = pp;
      g.N = (int)npair;
      g.K = Kld;
      g.lda = a.kM;
      g.ldb = Kld;
      g.ldc = pp;
      g.over = 3;  // the table's rows are padded to whole tiles; RB has slack behind its last column
      tasks.push_back(g);
    }
  }
  a.rb_off.upload(rb_off, s);
  a.c_off.upload(c_off, s);
  a.Kin.resize((size_t)A * A * E * pp);
  if (!tasks.empty()) {
    a.tasks.upload(tasks, s);
    size_t shb = (size_t)(4 * a.max_nch * r * p + a.max_nch * r) * sizeof(double);
    if (shb > 64 * 1024) {
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_exl_RB, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_exl_RB_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
    }
    // one launch for all table slots (grid.z, at most 65535 per launch): 126 launches of ~40 us each before
    int max_ns = 0;
    for (int tau = 0; tau < Ntab; tau++)
      if (rb_off[tau] >= 0) max_ns = std::max(max_ns, a.hS_off[tau + 1] - a.hS_off[tau]);
    for (int tau0 = 0; tau0 < Ntab; tau0 += 65535) {
      const int nz = std::min(65535, Ntab - tau0);
      if (pair)
        hipLaunchKernelGGL(k_exl_RB_pair, dim3(max_ns * max_ns, E * (E + 1) / 2, nz), dim3(256), shb, s, a.V0.p, a.V2.p,
                           a.tab_ch_off.p, a.tab_ch.p, t->LM_fac.p, sgrp, a.S_off.p, a.S_list.p, a.rb_off.p, tau0, Nd, R, E,
                           p, r, ntt, a.RB.p);
      else {
        static const bool rb_one = getenv("HELFEM_EXL_RB") && atoi(getenv("HELFEM_EXL_RB")) == 1;  // the one-pair kernel (checker)
        static const bool rb_four = getenv("HELFEM_EXL_RB") && atoi(getenv("HELFEM_EXL_RB")) == 4;  // the 4 x 4 vector kernel (checker)
        const size_t shb4 = (size_t)(4 * EXL_SB * EXL_CK * p + EXL_CK) * sizeof(double);
        if (!rb_one && !rb_four && p <= 16) {
          // the list of workgroups depends on the tables and on the shard only
          if (a.rbm_shard != ctx->shard_rank * 65536 + ctx->shard_n) {
            std::vector<int4> wl;
            for (int tau = 0; tau < Ntab; tau++) {
              if (rb_off[tau] < 0) continue;
              const int nbk = (a.hS_off[tau + 1] - a.hS_off[tau] + EXL_SB - 1) / EXL_SB;
              for (int e = 0; e < E; e++)
                for (int bk = 0; bk < nbk; bk++)
                  for (int bj = 0; bj <= bk; bj++) wl.push_back(make_int4(tau, e, bj, bk));
            }
            a.rbm_n = (int)wl.size();
            a.rbm_list.upload(wl, s);
            HFG_HIP_CHECK(hipStreamSynchronize(s));  // wl lives on this stack frame
            a.rbm_shard = ctx->shard_rank * 65536 + ctx->shard_n;
          }
          const size_t shm = (size_t)((a.max_nch * r + 3) / 4 * 4) * sizeof(double2);
          if (a.rbm_n > 0) {
            if (two)
              hipLaunchKernelGGL(k_exl_RBm<true>, dim3(a.rbm_n), dim3(512), shm, s, a.V0.p, a.V2.p, a.tab_ch_off.p, a.tab_ch.p,
                                 t->LM_fac.p, sgrp, a.S_off.p, a.S_list.p, a.rb_off.p, a.rbm_list.p, a.rbm_n, Nd, R, E, p, r, Kld,
                                 a.RB.p);
            else
              hipLaunchKernelGGL(k_exl_RBm<false>, dim3(a.rbm_n), dim3(512), shm, s, a.V0.p, a.V2.p, a.tab_ch_off.p, a.tab_ch.p,
                                 t->LM_fac.p, sgrp, a.S_off.p, a.S_list.p, a.rb_off.p, a.rbm_list.p, a.rbm_n, Nd, R, E, p, r, Kld,
                                 a.RB.p);
          }
        } else if (rb_one || p > 16)
          hipLaunchKernelGGL(k_exl_RB, dim3(max_ns * (max_ns + 1) / 2, E, nz), dim3(256), shb, s, a.V0.p, a.V2.p, a.tab_ch_off.p,
                             a.tab_ch.p, t->LM_fac.p, sgrp, a.S_off.p, a.S_list.p, a.rb_off.p, tau0, Nd, R, E, p, r, ntt,
                             a.RB.p);
        else {
          if (shb4 > 64 * 1024)
            HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_exl_RB4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb4));
          const int nb = (max_ns + EXL_SB - 1) / EXL_SB;
          hipLaunchKernelGGL(k_exl_RB4, dim3(nb * (nb + 1) / 2, E, nz), dim3(512), shb4, s, a.V0.p, a.V2.p, a.tab_ch_off.p, a.tab_ch.p,
                             t->LM_fac.p, sgrp, a.S_off.p, a.S_list.p, a.rb_off.p, tau0, Nd, R, E, p, r, ntt, a.RB.p);
        }
      }
    }
    // 128 x 64 tiles: the pair counts of the tasks are padded to 64 instead of 128 columns (3.5 % instead of 7 % of idle
    // columns) and twice as many workgroups share the slots -- 6.6 -> 6.2 ms at Nbf = 4230 (HELFEM_EXL_RECT=0: 128 x 128)
    // ... for the shorter pair lists; with the longer lists of larger angular bases the padding is small either way and
    // the square tiles' lower LDS traffic per flop wins (measured: Nbf = 4230, 1000 pairs per task on average, 11.6 against
    // 12.0 ms per build; Nbf = 6102, 28.6 against 29.4 the other way round)
    static const int rect_env = getenv("HELFEM_EXL_RECT") ? atoi(getenv("HELFEM_EXL_RECT")) : -1;
    double ncols_tot = 0.0;
    for (const GemmTask &q : tasks) ncols_tot += (double)q.N;
    const bool rect_tiles = rect_env >= 0 ? rect_env != 0 : (ncols_tot < 1500.0 * (double)tasks.size());
    static const bool wl_off = getenv("HELFEM_EXL_WL") && atoi(getenv("HELFEM_EXL_WL")) == 0;  // checker: plain task-list grid
    // bench.py: HIP events around this launch alone ("exl_element_gemm") and its USEFUL work in GFLOP, 2 p^2 pairs (ntt p^2)
    // per task without any padding, accumulated in the `ms` field of "exl_element_gemm_gflop"
    ProfScope pgemm(ctx, "exl_element_gemm");
    if (ctx->profiling) {
      double gf = 0.0;
      for (const GemmTask &q : tasks) gf += 2.0 * (double)pp * (double)q.N * (double)Kt * 1e-9;
      ctx->prof["exl_element_gemm_gflop"].ms += gf;
      ctx->prof["exl_element_gemm_gflop"].launches += 1;
    }
    if (!wl_off) {
      // all tiles of a task on one XCD (its element table is then fetched from HBM once, not by all eight L2s)
      // short pair lists: 64 x 64 tiles (three workgroups per CU; 5.71 against 6.08 ms with 128 x 64 at Nbf = 4230); long lists:
      // 128 x 128 (15.5 against 15.8 ms with 64 x 64 at Nbf = 6102).  HELFEM_EXL_RECT = 0 / 1 / 2 forces 128 x 128 / 128 x 64 / 64 x 64
      const int tile_mode = rect_env >= 0 ? rect_env : (rect_tiles ? 2 : 0);
      const int BMt = tile_mode == 2 ? 64 : 128, BNt = tile_mode == 0 ? 128 : 64;
      std::vector<int2> &wl = a.h_gwl;
      wl.clear();
      for (size_t k = 0; k < tasks.size(); k++) {
        const int nt = ((tasks[k].M + BMt - 1) / BMt) * ((tasks[k].N + BNt - 1) / BNt);
        for (int q = 0; q < nt; q++) wl.push_back(make_int2((int)k, q));
      }
      a.gwl.upload(wl, s);  // (h_gwl lives in the aux until the synchronisation at the end of the build)
      gemm_tasklist_wl_dev(ctx, a.tasks.p, a.gwl.p, (int)wl.size(), tile_mode);
    } else if (rect_tiles) gemm_tasklist_rect_dev(ctx, a.tasks.p, (int)tasks.size(), pp, maxN);
    else gemm_tasklist_dev(ctx, a.tasks.p, (int)tasks.size(), pp, maxN);
  }
  if (pair)
    hipLaunchKernelGGL(k_exl_reduce_pair, dim3(A * A, E * (E + 1) / 2), dim3(256), 0, s, a.C.p, a.c_off.p, a.S_off.p, a.pos.p,
                       A, E, p, Ntab, a.Kin.p, a.G.p);
  else
    hipLaunchKernelGGL(k_exl_reduce, dim3(A * A, E), dim3(256), 0, s, a.C.p, a.c_off.p, a.S_off.p, a.pos.p, A, E, p, Ntab,
                       a.Kin.p);
  dim3 grid((N + 63) / 64, (N + 3) / 4);
  hipLaunchKernelGGL(k_exl_assemble, grid, dim3(256), 0, s, a.Kin.p, a.G.p, N, A, E, p, a.pure_shell.p, a.pure_n.p,
                     grouped ? a.sh_perm.p : nullptr, dK, accumulate);
  HFG_HIP_CHECK(hipGetLastError());
  HFG_HIP_CHECK(hipStreamSynchronize(s));  // host task list and offsets live on this stack frame
  }  // factor groups
  return true;
}

}  // namespace hfg
