// Coulomb and exchange-correlation Fock-matrix kernels for the diatomic basis (gfx950).
//
// What they replace (reference, /root/reference):
//   TwoDBasis::coulomb                src/diatomic/basis.cpp:1359-1530
//   DFTGrid::eval_Fxc (restricted)    src/diatomic/dftgrid.cpp:769-810 and the worker it drives
//                                     (:51-117 update_density, :343-458 compute_xc, :499-545 eval_Fxc,
//                                      :669-755 compute_bf, dftgrid.h:190-253 increment_lda/gga)
//
// Design notes (DESIGN.md has the long version):
//  * Both J and the XC matrix are block-banded in the radial index (two FEM functions only overlap
//    inside one element), and both only read the element-diagonal p x p blocks of P.  All work is
//    done on the "compact" layout X_c[x][y][e][j][i] (tables.h); dense N x N matrices appear only at
//    the API boundary (gather_compact / scatter_dense).
//  * XC: the reference forms the complex (A*p) x (ntheta*nphi) basis-function matrix for every radial
//    point and runs zgemm on it.  Here the product structure  bf = B_n(mu) Theta_lm(theta) e^{i m phi}
//    is used: contract the radial index (X1), the theta index per (m,m') group pair (X2), evaluate
//    the functional on the grid with the phi sums folded in (X3), and go back up (X4, X5).  Same
//    sums, different association order; ~2000x fewer flops, no O(Ng * ne) intermediates.
//  * No atomics anywhere: every output element has exactly one writer and a fixed summation order,
//    so results are bitwise reproducible run to run.
#include "tables.h"
#include "xc_device.h"

namespace hfg {

// -------------------------------------------------------------------------------------------------
// dense <-> compact
// -------------------------------------------------------------------------------------------------
// Xc[x][y][e][j][i] = P(pure(x,e*(p-1)+i), pure(y,e*(p-1)+j))
__global__ void k_gather_compact(const double *__restrict__ P, int N, int A, int R, int E, int p,
                                 const int *__restrict__ shell_off, const int *__restrict__ shell_skip,
                                 double *__restrict__ Xc) {
  int blk = blockIdx.x;  // (x*A+y)*E+e
  int e = blk % E;
  int xy = blk / E;
  int y = xy % A, x = xy / A;
  int pp = p * p;
  for (int t = threadIdx.x; t < pp; t += blockDim.x) {
    int i = t % p, j = t / p;
    int ni = e * (p - 1) + i, nj = e * (p - 1) + j;
    double v = 0.0;
    bool ok = (ni < R) && (nj < R) && !(shell_skip[x] && ni == 0) && !(shell_skip[y] && nj == 0);
    if (ok) v = P[(size_t)(shell_off[y] + nj) * N + (shell_off[x] + ni)];
    Xc[(size_t)blk * pp + t] = v;
  }
}

// out(pure row, pure col) = sum over the (one or two) elements that contain both radial functions
__global__ void k_scatter_dense(const double *__restrict__ Xc, int N, int A, int R, int E, int p,
                                const int *__restrict__ pure_shell, const int *__restrict__ pure_n,
                                double *__restrict__ out) {
  int row = blockIdx.x * 64 + (threadIdx.x & 63);
  int col = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= N || col >= N) return;
  int x = pure_shell[row], n = pure_n[row];
  int y = pure_shell[col], m = pure_n[col];
  int pm = p - 1, pp = p * p;
  // candidate elements of n: e1=n/pm (local n%pm) and, on a shared node, e1-1 (local pm)
  int e1 = n / pm, f1 = m / pm;
  double v = 0.0;
  const double *base = Xc + (size_t)(x * A + y) * E * pp;
  for (int ce = 0; ce < 2; ce++) {
    int e = e1 - ce;
    if (e < 0 || e >= E) continue;
    int i = n - e * pm;
    if (i < 0 || i > pm) continue;
    for (int cf = 0; cf < 2; cf++) {
      int f = f1 - cf;
      if (f != e) continue;
      int j = m - f * pm;
      if (j < 0 || j > pm) continue;
      v += base[(size_t)e * pp + j * p + i];
    }
  }
  out[(size_t)col * N + row] = v;
}

// F(row,col) = [blockid(row)==blockid(col)] * (H0(row,col) + scatter(Xc)(row,col))
// (Fock assembly + scf::enforce_fock_symmetry, src/diatomic/main.cpp:871-900, scf_helpers.cpp:249)
__global__ void k_fock_finish(const double *__restrict__ Xc, const double *__restrict__ H0,
                              const int *__restrict__ blockid, int N, int A, int R, int E, int p,
                              const int *__restrict__ pure_shell, const int *__restrict__ pure_n,
                              double *__restrict__ out) {
  int row = blockIdx.x * 64 + (threadIdx.x & 63);
  int col = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= N || col >= N) return;
  size_t o = (size_t)col * N + row;
  if (blockid && blockid[row] != blockid[col]) {
    out[o] = 0.0;
    return;
  }
  int x = pure_shell[row], n = pure_n[row];
  int y = pure_shell[col], m = pure_n[col];
  int pm = p - 1, pp = p * p;
  int e1 = n / pm, f1 = m / pm;
  double v = H0 ? H0[o] : 0.0;
  const double *base = Xc + (size_t)(x * A + y) * E * pp;
  for (int ce = 0; ce < 2; ce++) {
    int e = e1 - ce;
    if (e < 0 || e >= E) continue;
    int i = n - e * pm;
    if (i < 0 || i > pm) continue;
    for (int cf = 0; cf < 2; cf++) {
      int f = f1 - cf;
      if (f != e) continue;
      int j = m - f * pm;
      if (j < 0 || j > pm) continue;
      v += base[(size_t)e * pp + j * p + i];
    }
  }
  out[o] = v;
}

__global__ void k_add_inplace(double *__restrict__ a, const double *__restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) a[i] += b[i];
}

// -------------------------------------------------------------------------------------------------
// Coulomb
// -------------------------------------------------------------------------------------------------
// K1  ket contraction (basis.cpp:1380-1405), restricted to the element-diagonal blocks:
//     Paux{0,2}[iLM][e][t] = sum_{(x,y) in iLM} c{0,2} * Pc[x][y][e][t]
__global__ void k_coulomb_ket(const double *__restrict__ Pc, int A, int E, int pp, const int *__restrict__ lm_off,
                              const int *__restrict__ lm_x, const int *__restrict__ lm_y,
                              const double *__restrict__ lm_c0, const double *__restrict__ lm_c2, int NLM,
                              const int *__restrict__ LM_ilm, int rank, int nranks, double *__restrict__ Paux) {
  int iLM = blockIdx.x, e = blockIdx.y;
  if (LM_ilm[iLM] % nranks != rank) return;  // (L,|M|) channels are the multi-GPU shards of J
  int beg = lm_off[iLM], end = lm_off[iLM + 1];
  for (int t = threadIdx.x; t < pp; t += blockDim.x) {
    double a0 = 0.0, a2 = 0.0;
    // eight entries in flight: the rolled "load P, two FMAs" loop paid one memory round trip per entry (~330 entries per
    // channel: 146 us for the kernel); the order of the additions is unchanged
    int k = beg;
    for (; k + 8 <= end; k += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = Pc[((size_t)(lm_x[k + u] * A + lm_y[k + u]) * E + e) * pp + t];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 += lm_c0[k + u] * v[u];
        a2 += lm_c2[k + u] * v[u];
      }
    }
    for (; k < end; k++) {
      double v = Pc[((size_t)(lm_x[k] * A + lm_y[k]) * E + e) * pp + t];
      a0 += lm_c0[k] * v;
      a2 += lm_c2[k] * v;
    }
    Paux[((size_t)(0 * NLM + iLM) * E + e) * pp + t] = a0;
    Paux[((size_t)(1 * NLM + iLM) * E + e) * pp + t] = a2;
  }
}

// K2b in-element integrals (basis.cpp:1472-1485): Y[tt][iLM][e][:] = tei_tt[ilm][e] * vec(Paux_{tt&1 ? 2:0}[iLM][e])
//     for iLM=(L,+|M|) and its partner (L,-|M|) in one pass over the p^2 x p^2 table (HBM-bound:
//     this is where the 4*Nlm*E*p^4*8 bytes of primitive integrals are streamed once per build).
__global__ void k_coulomb_tei(const double *__restrict__ tei, const double *__restrict__ Paux, int Ntab, int NLM,
                              int E, int pp, const int *__restrict__ lmpos /* [Nlm][2] iLM of +M and -M */,
                              const int *__restrict__ lm_tab, int rank, int nranks, double *__restrict__ Y) {
  extern __shared__ double sh[];  // x_plus[pp], x_minus[pp]
  int ilm = blockIdx.x / E, e = blockIdx.x % E;
  if (ilm % nranks != rank) return;
  int tt = blockIdx.y;  // 0:00 1:02 2:20 3:22
  int iLMp = lmpos[2 * ilm], iLMm = lmpos[2 * ilm + 1];
  int which = (tt & 1);  // 00,20 act on Paux0 ; 02,22 act on Paux2
  double *xp = sh, *xm = sh + pp;
  for (int t = threadIdx.x; t < pp; t += blockDim.x) {
    xp[t] = (iLMp >= 0) ? Paux[((size_t)(which * NLM + iLMp) * E + e) * pp + t] : 0.0;
    xm[t] = (iLMm >= 0) ? Paux[((size_t)(which * NLM + iLMm) * E + e) * pp + t] : 0.0;
  }
  __syncthreads();
  const double *T = tei + (((size_t)tt * Ntab + lm_tab[ilm]) * E + e) * (size_t)pp * pp;
  for (int r = threadIdx.x; r < pp; r += blockDim.x) {
    double yp = 0.0, ym = 0.0;
#pragma unroll 5
    for (int c = 0; c < pp; c++) {
      double v = T[(size_t)c * pp + r];
      yp += v * xp[c];
      ym += v * xm[c];
    }
    if (iLMp >= 0) Y[((size_t)(tt * NLM + iLMp) * E + e) * pp + r] = yp;
    if (iLMm >= 0) Y[((size_t)(tt * NLM + iLMm) * E + e) * pp + r] = ym;
  }
}

// K2c per (L,M): disjoint (cross-element) part via the trace scalars + in-element part (basis.cpp:1424-1494).
//     full = 1: prolate kernel with the four P0/P2/Q0/Q2 and 00/02/20/22 tables; full = 0: spherical kernel
//     r_<^L/r_>^{L+1} with P0/Q0 and 00 only (atomic TwoDBasis.cpp:884-936), disj then holds [P0][Q0].
__global__ void k_coulomb_radial(const double *__restrict__ Paux, const double *__restrict__ Y,
                                 const double *__restrict__ disj, const int *__restrict__ LM_ilm,
                                 const int *__restrict__ LM_tab, const double *__restrict__ LM_fac, int Ntab, int NLM,
                                 int E, int p, int full, int rank, int nranks, double *__restrict__ Jaux) {
  extern __shared__ double sh[];  // red[4*E*nwave], sc[4*E], big[E], small[E]
  int iLM = blockIdx.x;
  if (LM_ilm[iLM] % nranks != rank) return;
  int tab = LM_tab[iLM];
  double fac = LM_fac[iLM];
  int pp = p * p;
  int nwave = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  double *red = sh;
  double *sc = red + 4 * E * nwave;
  double *big = sc + 4 * E, *small = big + E;
  const int tQ0 = full ? 2 : 1;
  const double *dP0 = disj + ((size_t)0 * Ntab + tab) * E * pp;
  const double *dQ0 = disj + ((size_t)tQ0 * Ntab + tab) * E * pp;
  const double *dP2 = full ? disj + ((size_t)1 * Ntab + tab) * E * pp : nullptr;
  const double *dQ2 = full ? disj + ((size_t)3 * Ntab + tab) * E * pp : nullptr;
  // traces: js[k][e], k: 0 small0=tr(P0*Psub0) 1 big0=tr(Q0*Psub0) 2 small2=tr(P2*Psub2) 3 big2=tr(Q2*Psub2)
  for (int e = 0; e < E; e++) {
    double a[4] = {0, 0, 0, 0};
    for (int t = threadIdx.x; t < pp; t += blockDim.x) {
      int i = t % p, j = t / p;
      double x0 = Paux[((size_t)(0 * NLM + iLM) * E + e) * pp + (i * p + j)];  // Psub(j,i)
      a[0] += dP0[(size_t)e * pp + t] * x0;
      a[1] += dQ0[(size_t)e * pp + t] * x0;
      if (full) {
        double x2 = Paux[((size_t)(1 * NLM + iLM) * E + e) * pp + (i * p + j)];
        a[2] += dP2[(size_t)e * pp + t] * x2;
        a[3] += dQ2[(size_t)e * pp + t] * x2;
      }
    }
    for (int k = 0; k < 4; k++) {
      double v = a[k];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) red[(k * E + e) * nwave + wave] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < 4 * E) {
    double v = 0.0;
    for (int w = 0; w < nwave; w++) v += red[threadIdx.x * nwave + w];
    sc[threadIdx.x] = fac * v;
  }
  __syncthreads();
  if (threadIdx.x < E) {
    int e = threadIdx.x;
    // contributions to element e from jel>e use "big" of jel, from jel<e use "small" of jel
    double sb = 0.0, ss = 0.0;
    for (int jel = e + 1; jel < E; jel++) sb += sc[1 * E + jel] - sc[3 * E + jel];  // jbig0 - jbig2
    for (int jel = 0; jel < e; jel++) ss += sc[0 * E + jel] - sc[2 * E + jel];      // jsmall0 - jsmall2
    big[e] = sb;
    small[e] = ss;
  }
  __syncthreads();
  for (int e = 0; e < E; e++)
    for (int t = threadIdx.x; t < pp; t += blockDim.x) {
      double P0 = dP0[(size_t)e * pp + t], Q0 = dQ0[(size_t)e * pp + t];
      double y00 = Y[((size_t)(0 * NLM + iLM) * E + e) * pp + t];
      double j0 = P0 * big[e] + Q0 * small[e] + fac * y00;
      double j2 = 0.0;
      if (full) {
        double P2 = dP2[(size_t)e * pp + t], Q2 = dQ2[(size_t)e * pp + t];
        double y02 = Y[((size_t)(1 * NLM + iLM) * E + e) * pp + t];
        double y20 = Y[((size_t)(2 * NLM + iLM) * E + e) * pp + t], y22 = Y[((size_t)(3 * NLM + iLM) * E + e) * pp + t];
        j0 -= fac * y02;
        j2 = -P2 * big[e] - Q2 * small[e] - fac * y20 + fac * y22;
      }
      Jaux[((size_t)(0 * NLM + iLM) * E + e) * pp + t] = j0;
      Jaux[((size_t)(1 * NLM + iLM) * E + e) * pp + t] = j2;
    }
}

// K3 bra expansion (basis.cpp:1498-1527): Jc[i][j][e][:] = sum_L c0(j,i,L) Jaux0[iLM] + c2(j,i,L) Jaux2[iLM]
__global__ void k_coulomb_bra(const double *__restrict__ Jaux, int A, int E, int pp, int NLM,
                              const int *__restrict__ pair_off, const int *__restrict__ ent_iLM,
                              const double *__restrict__ ent_c0, const double *__restrict__ ent_c2,
                              const int *__restrict__ LM_ilm, int rank, int nranks, double *__restrict__ Jc) {
  int ij = blockIdx.x, e = blockIdx.y;
  int iang = ij / A, jang = ij % A;
  int pr = jang * A + iang;  // pair (x=jang, y=iang)
  int beg = pair_off[pr], end = pair_off[pr + 1];
  for (int t = threadIdx.x; t < pp; t += blockDim.x) {
    double acc = 0.0;
    // four entries (eight loads) in flight; entries of other ranks' channels contribute an exact zero in the same place
    int k = beg;
    for (; k + 4 <= end; k += 4) {
      double j0[4], j2[4], c0[4], c2[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int iLM = ent_iLM[k + u];
        const bool mine = (LM_ilm[iLM] % nranks == rank);
        j0[u] = Jaux[((size_t)(0 * NLM + iLM) * E + e) * pp + t];
        j2[u] = Jaux[((size_t)(1 * NLM + iLM) * E + e) * pp + t];
        c0[u] = mine ? ent_c0[k + u] : 0.0;
        c2[u] = mine ? ent_c2[k + u] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (c0[u] != 0.0 || c2[u] != 0.0) acc += c0[u] * j0[u] + c2[u] * j2[u];
    }
    for (; k < end; k++) {
      int iLM = ent_iLM[k];
      if (LM_ilm[iLM] % nranks != rank) continue;
      acc += ent_c0[k] * Jaux[((size_t)(0 * NLM + iLM) * E + e) * pp + t] +
             ent_c2[k] * Jaux[((size_t)(1 * NLM + iLM) * E + e) * pp + t];
    }
    Jc[((size_t)ij * E + e) * pp + t] = acc;
  }
}

// -------------------------------------------------------------------------------------------------
// XC
// -------------------------------------------------------------------------------------------------
// X1 radial contraction of the density: D0[Q][x][y] = sum_ij B_i(q) Pc[x][y][e][j][i] B_j(q),
//    D1[Q][x][y] = sum_ij B'_i(q) Pc[..][j][i] B_j(q)            (replaces Pv = P conj(bf), dftgrid.cpp:62)
//    (meta-GGA: D2[Q][x][y] = sum_ij B'_i Pc B'_j for the kinetic energy density)
__global__ __launch_bounds__(256) void k_xc_density_radial(const double *__restrict__ Pc, const double *__restrict__ B,
                                                           const double *__restrict__ dB, int A, int E, int p, int nq,
                                                           int do_grad, int do_tau, int rank, int nranks,
                                                           double *__restrict__ D0, double *__restrict__ D1,
                                                           double *__restrict__ D2) {
  // One workgroup per (shell pair, element).  The p x p block of P and the element's B, B' tables sit in LDS (the
  // tables transposed to [i][q]: consecutive q <-> consecutive banks); thread (q, jc) forms the rows j = jc, jc+NJ, ...
  // of T = P b(q), contracts them with b_j, b'_j, and the NJ partial results of a point are summed through LDS in a
  // fixed order.
  extern __shared__ double sh[];  // P[pp], Bt[p][nq], dBt[p][nq], part[3][NJ][nq]
  const int xy = blockIdx.x, e = blockIdx.y;
  const int pp = p * p;
  double *sP = sh, *sB = sh + pp, *sdB = sB + p * nq;
  const int NJ = blockDim.x / nq > 0 ? min((int)(blockDim.x / nq), p) : 1;
  double *part = sdB + p * nq;
  for (int t = threadIdx.x; t < pp; t += blockDim.x) sP[t] = Pc[((size_t)xy * E + e) * pp + t];
  for (int t = threadIdx.x; t < p * nq; t += blockDim.x) {
    int q = t / p, i = t % p;
    sB[i * nq + q] = B[((size_t)e * nq + q) * p + i];
    sdB[i * nq + q] = dB[((size_t)e * nq + q) * p + i];
  }
  __syncthreads();
  const size_t AA = (size_t)A * A;
  // work items (q, jc), q fastest; with nq <= blockDim.x every point is finished in one pass
  for (int q0 = 0; q0 < nq; q0 += blockDim.x / NJ) {
    const int ql = threadIdx.x % (blockDim.x / NJ), jc = threadIdx.x / (blockDim.x / NJ);
    const int q = q0 + ql;
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    if (q < nq && jc < NJ)
      for (int j = jc; j < p; j += NJ) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < p; i++) {
          double pv = sP[j * p + i];
          s0 += sB[i * nq + q] * pv;
          s1 += sdB[i * nq + q] * pv;
        }
        d0 += s0 * sB[j * nq + q];
        d1 += s1 * sB[j * nq + q];
        d2 += s1 * sdB[j * nq + q];
      }
    if (q < nq && jc < NJ) {
      part[(0 * NJ + jc) * nq + q] = d0;
      part[(1 * NJ + jc) * nq + q] = d1;
      part[(2 * NJ + jc) * nq + q] = d2;
    }
    __syncthreads();
    if (jc == 0 && q < nq && (e * nq + q) % nranks == rank) {  // radial quadrature points are the multi-GPU shards of XC
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int k = 0; k < NJ; k++) {
        a0 += part[(0 * NJ + k) * nq + q];
        a1 += part[(1 * NJ + k) * nq + q];
        a2 += part[(2 * NJ + k) * nq + q];
      }
      const size_t Q = (size_t)e * nq + q;
      D0[Q * AA + xy] = a0;
      if (do_grad) D1[Q * AA + xy] = a1;
      if (do_tau) D2[Q * AA + xy] = a2;
    }
    __syncthreads();
  }
}

// X2 theta contraction per (m-group pair): V[k][Q][ga][gb][i],
//    k=0: sum Theta_a D0_ab Theta_b ; k=1: sum dTheta_a D0_ab Theta_b ; k=2: sum Theta_a D1_ab Theta_b
//    meta-GGA: k=3: sum Theta_a D2_ab Theta_b ; k=4: sum dTheta_a D0_ab dTheta_b
__global__ void k_xc_density_theta(const double *__restrict__ D0, const double *__restrict__ D1,
                                   const double *__restrict__ D2, const double *__restrict__ Th,
                                   const double *__restrict__ dTh, int A, int nth, int G,
                                   const int *__restrict__ grp_off, const int *__restrict__ grp_shell, int do_grad,
                                   int do_tau, size_t NQ, int rank, int nranks, double *__restrict__ V) {
  extern __shared__ double sh[];  // d0[na*nb], d1[na*nb], d2[na*nb]
  size_t Q = blockIdx.x;
  if ((int)(Q % nranks) != rank) return;
  int ga = blockIdx.y / G, gb = blockIdx.y % G;
  int a0 = grp_off[ga], na = grp_off[ga + 1] - a0;
  int b0 = grp_off[gb], nb = grp_off[gb + 1] - b0;
  double *d0 = sh, *d1 = sh + na * nb, *d2 = sh + 2 * na * nb;
  size_t AA = (size_t)A * A;
  for (int t = threadIdx.x; t < na * nb; t += blockDim.x) {
    int ia = t / nb, ib = t % nb;
    int a = grp_shell[a0 + ia], b = grp_shell[b0 + ib];
    d0[t] = D0[Q * AA + (size_t)a * A + b];
    d1[t] = do_grad ? D1[Q * AA + (size_t)a * A + b] : 0.0;
    d2[t] = do_tau ? D2[Q * AA + (size_t)a * A + b] : 0.0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nth; i += blockDim.x) {
    double r = 0.0, s = 0.0, u = 0.0, k3 = 0.0, k4 = 0.0;
    for (int ia = 0; ia < na; ia++) {
      int a = grp_shell[a0 + ia];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int ib = 0; ib < nb; ib++) {
        double tb = Th[(size_t)grp_shell[b0 + ib] * nth + i];
        s0 += d0[ia * nb + ib] * tb;
        s1 += d1[ia * nb + ib] * tb;
        if (do_tau) {
          s2 += d2[ia * nb + ib] * tb;
          s3 += d0[ia * nb + ib] * dTh[(size_t)grp_shell[b0 + ib] * nth + i];
        }
      }
      double ta = Th[(size_t)a * nth + i];
      r += ta * s0;
      if (do_grad) {
        s += dTh[(size_t)a * nth + i] * s0;
        u += ta * s1;
      }
      if (do_tau) {
        k3 += ta * s2;
        k4 += dTh[(size_t)a * nth + i] * s3;
      }
    }
    size_t o = ((Q * G + ga) * G + gb) * nth + i;
    size_t stride = NQ * G * G * nth;
    V[o] = r;
    if (do_grad) {
      V[stride + o] = s;
      V[2 * stride + o] = u;
    }
    if (do_tau) {
      V[3 * stride + o] = k3;
      V[4 * stride + o] = k4;
    }
  }
}

// X3 grid evaluation at one radial point: density & gradient on the (theta,phi) grid, functional,
//    energy/electron partial sums, and the phi-transformed potentials
//       Fo[0][Q][ga][gb][i] = sum_j (1/2 w vrho cos(D phi_j) - m_ga gr_phi sin(D phi_j))
//       Fo[1][..]           = sum_j gr_nu cos(D phi_j)
//       Fo[2][..]           = sum_j gr_mu cos(D phi_j)            D = m_ga - m_gb
//    (dftgrid.cpp:69-86, 412-416, 471-477, 510-531, 693-707)
__global__ void k_xc_grid(const double *__restrict__ V, const double *__restrict__ rad_w,
                          const double *__restrict__ rad_sh, const double *__restrict__ th_s,
                          const double *__restrict__ th_w, const int *__restrict__ grp_m,
                          const double *__restrict__ cosd, const double *__restrict__ sind, int Dmax, int G, int nth,
                          int nphi, double Rh, int geom, int x_func, int c_func, int do_grad, int do_tau, double thr,
                          size_t NQ, int rank, int nranks, double *__restrict__ Fo,
                          double *__restrict__ partial /* [3][NQ] */) {
  extern __shared__ double sh[];  // pot[5][nth*nphi], red[3*nwave]
  size_t Q = blockIdx.x;
  if ((int)(Q % nranks) != rank) {
    if (threadIdx.x == 0) {
      partial[Q] = 0.0;
      partial[NQ + Q] = 0.0;
      partial[2 * NQ + Q] = 0.0;
    }
    return;
  }
  int ng = nth * nphi;
  double *p0 = sh, *p1 = sh + ng, *p2 = sh + 2 * ng, *p3 = sh + 3 * ng, *p4 = sh + 4 * ng;
  double *red = sh + 5 * ng;
  double shm = rad_sh[Q], wr = rad_w[Q];
  double dphi = 2.0 * HFG_PI / nphi;
  size_t stride = NQ * G * G * nth;
  double nel = 0.0, exc_sum = 0.0, kin_sum = 0.0;
  for (int pt = threadIdx.x; pt < ng; pt += blockDim.x) {
    int i = pt / nphi, j = pt % nphi;
    double sth = th_s[i];
    // scale factors of the radial-like, polar-like and azimuthal coordinates and the volume weight:
    // prolate spheroidal (diatomic dftgrid.cpp:693-707), spherical (atomic dftgrid.cpp:724-743; shm holds r)
    double hmu, hnu, hphi, w;
    if (geom == 0) {
      double h2 = shm * shm + sth * sth;
      hmu = hnu = Rh * sqrt(h2);
      hphi = Rh * shm * sth;
      w = th_w[i] * dphi * wr * Rh * Rh * Rh * shm * h2;
    } else {
      hmu = 1.0;
      hnu = shm;
      hphi = shm * sth;
      w = th_w[i] * dphi * wr * shm * shm;
    }
    double rho = 0.0, gmu = 0.0, gnu = 0.0, gphi = 0.0, tau = 0.0;
    for (int ga = 0; ga < G; ga++)
      for (int gb = 0; gb < G; gb++) {
        int D = grp_m[ga] - grp_m[gb];
        double cd = cosd[(size_t)(D + Dmax) * nphi + j];
        size_t o = ((Q * G + ga) * G + gb) * nth + i;
        double vr = V[o];
        rho += cd * vr;
        if (do_tau)  // tau = 1/2 sum_c Re[(P conj d_c bf) . d_c bf] / h_c^2   (dftgrid.cpp:90-112)
          tau += 0.5 * cd * (V[3 * stride + o] / (hmu * hmu) + V[4 * stride + o] / (hnu * hnu) +
                             (double)(grp_m[ga] * grp_m[gb]) * vr / (hphi * hphi));
        if (do_grad) {
          double sd = sind[(size_t)(D + Dmax) * nphi + j];
          gnu += cd * V[stride + o];
          gmu += cd * V[2 * stride + o];
          gphi -= grp_m[ga] * sd * vr;
        }
      }
    double sigma = 0.0;
    if (do_grad) {
      gmu *= 2.0 / hmu;
      gnu *= 2.0 / hnu;
      gphi *= 2.0 / hphi;
      sigma = gmu * gmu + gnu * gnu + gphi * gphi;
    }
    double exc = 0.0, vrho = 0.0, vsig = 0.0, vtau = 0.0;
    if (rho >= thr && rho > 0.0) {
      const bool live = 0.5 * rho >= thr;  // the spin channels of the exchange sum carry rho/2 each
      if (x_func > 0) {
        if (xc::is_mgga(x_func)) xc::eval_add_mgga(x_func, rho, sigma, tau, live, exc, vrho, vsig, vtau);
        else xc::eval_add(x_func, rho, sigma, live, exc, vrho, vsig);
      }
      if (c_func > 0) {
        if (xc::is_mgga(c_func)) xc::eval_add_mgga(c_func, rho, sigma, tau, live, exc, vrho, vsig, vtau);
        else xc::eval_add(c_func, rho, sigma, live, exc, vrho, vsig);
      }
    }
    nel += w * rho;
    exc_sum += w * exc * rho;
    kin_sum += w * tau;
    if (do_tau) p4[pt] = 0.5 * w * vtau;  // vt of dftgrid.cpp:534-535
    p0[pt] = w * vrho;
    if (do_grad) {
      double f = 2.0 * w * vsig;
      p1[pt] = f * gmu / hmu;
      p2[pt] = f * gnu / hnu;
      p3[pt] = f * gphi / hphi;
    }
  }
  // block reduction of the two scalars (fixed order -> deterministic)
  int nwave = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  for (int o = 32; o > 0; o >>= 1) {
    nel += __shfl_down(nel, o, 64);
    exc_sum += __shfl_down(exc_sum, o, 64);
    kin_sum += __shfl_down(kin_sum, o, 64);
  }
  if (lane == 0) {
    red[wave] = nel;
    red[nwave + wave] = exc_sum;
    red[2 * nwave + wave] = kin_sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c2 = 0.0;
    for (int w = 0; w < nwave; w++) {
      a += red[w];
      b += red[nwave + w];
      c2 += red[2 * nwave + w];
    }
    partial[Q] = a;
    partial[NQ + Q] = b;
    partial[2 * NQ + Q] = c2;
  }
  // phi transforms
  int nout = G * G * nth;
  for (int t = threadIdx.x; t < nout; t += blockDim.x) {
    int i = t % nth;
    int gab = t / nth;
    int ga = gab / G, gb = gab % G;
    int D = grp_m[ga] - grp_m[gb];
    const double *cd = cosd + (size_t)(D + Dmax) * nphi;
    const double *sd = sind + (size_t)(D + Dmax) * nphi;
    double fa = 0.0, fs = 0.0, fb = 0.0, ft = 0.0;
    double mga = grp_m[ga];
    for (int j = 0; j < nphi; j++) {
      int pt = i * nphi + j;
      fa += 0.5 * p0[pt] * cd[j];
      if (do_grad) {
        fa -= mga * p3[pt] * sd[j];
        fs += p2[pt] * cd[j];
        fb += p1[pt] * cd[j];
      }
      if (do_tau) ft += p4[pt] * cd[j];
    }
    size_t o = ((Q * G + ga) * G + gb) * nth + i;
    if (do_tau) {
      // the three tau terms of eval_Fxc (dftgrid.cpp:533-540) share the phi sum; the scale factors depend on
      // theta only.  Halves as for the LDA term: the radial stage adds the (x,y) and (y,x) blocks.
      double sth = th_s[i], hmu, hnu, hphi;
      if (geom == 0) {
        hmu = hnu = Rh * sqrt(shm * shm + sth * sth);
        hphi = Rh * shm * sth;
      } else {
        hmu = 1.0;
        hnu = shm;
        hphi = shm * sth;
      }
      fa += 0.5 * mga * (double)grp_m[gb] * ft / (hphi * hphi);
      Fo[3 * stride + o] = 0.5 * ft / (hmu * hmu);
      Fo[4 * stride + o] = 0.5 * ft / (hnu * hnu);
    }
    Fo[o] = fa;
    if (do_grad) {
      Fo[stride + o] = fs;
      Fo[2 * stride + o] = fb;
    }
  }
}

// X3 for a spin-polarised density (DFTGridWorker::update_density(Pa,Pb), compute_xc, eval_Fxc(Ha,Hb,beta);
//    dftgrid.cpp:119-170, 343-458, 547-613): V and Fo hold the alpha planes (0..2) followed by the beta planes (3..5).
__global__ void k_xc_grid_pol(const double *__restrict__ V, const double *__restrict__ rad_w,
                              const double *__restrict__ rad_sh, const double *__restrict__ th_s,
                              const double *__restrict__ th_w, const int *__restrict__ grp_m,
                              const double *__restrict__ cosd, const double *__restrict__ sind, int Dmax, int G, int nth,
                              int nphi, double Rh, int geom, int x_func, int c_func, int do_grad, int do_tau, double thr,
                              size_t NQ, int rank, int nranks, double *__restrict__ Fo,
                              double *__restrict__ partial /* [3][NQ] */, int rowc) {
  // meta-GGA (do_tau): V and Fo carry five planes per spin (the tau planes 3, 4 as in k_xc_grid), LDS one more potential
  // plane per spin.  The theta rows go through LDS rowc at a time (the phi transform is row by row): rowc = nth unless the
  // planes of the whole grid exceed a CU's LDS (lmax beyond ~26 with mmax = 2).
  extern __shared__ double sh[];  // pot[npot][rowc*nphi] (npot = 8, or 10 with tau), red[3*nwave]
  size_t Q = blockIdx.x;
  if ((int)(Q % nranks) != rank) {
    if (threadIdx.x == 0) {
      partial[Q] = 0.0;
      partial[NQ + Q] = 0.0;
      partial[2 * NQ + Q] = 0.0;
    }
    return;
  }
  const int ng = rowc * nphi;        // plane stride in LDS
  const int npl = do_tau ? 5 : 3;    // planes per spin in V and Fo
  const int npot = do_tau ? 5 : 4;   // potential planes per spin in LDS
  double *red = sh + 2 * npot * ng;
  double shm = rad_sh[Q], wr = rad_w[Q];
  double dphi = 2.0 * HFG_PI / nphi;
  size_t stride = NQ * G * G * nth;
  double nel = 0.0, exc_sum = 0.0, kin_sum = 0.0;
  for (int i0 = 0; i0 < nth; i0 += rowc) {
  const int rows = min(rowc, nth - i0);
  if (i0) __syncthreads();  // the previous chunk's planes have been transformed
  for (int pt = threadIdx.x; pt < rows * nphi; pt += blockDim.x) {
    int i = i0 + pt / nphi, j = pt % nphi;
    double sth = th_s[i];
    double hmu, hnu, hphi, w;
    if (geom == 0) {
      double h2 = shm * shm + sth * sth;
      hmu = hnu = Rh * sqrt(h2);
      hphi = Rh * shm * sth;
      w = th_w[i] * dphi * wr * Rh * Rh * Rh * shm * h2;
    } else {
      hmu = 1.0;
      hnu = shm;
      hphi = shm * sth;
      w = th_w[i] * dphi * wr * shm * shm;
    }
    double rho[2] = {0.0, 0.0}, gmu[2] = {0.0, 0.0}, gnu[2] = {0.0, 0.0}, gphi[2] = {0.0, 0.0}, tau[2] = {0.0, 0.0};
    for (int ga = 0; ga < G; ga++)
      for (int gb = 0; gb < G; gb++) {
        int D = grp_m[ga] - grp_m[gb];
        double cd = cosd[(size_t)(D + Dmax) * nphi + j];
        double sd = sind[(size_t)(D + Dmax) * nphi + j];
        size_t o = ((Q * G + ga) * G + gb) * nth + i;
        for (int sp = 0; sp < 2; sp++) {
          const double *Vs = V + (size_t)sp * npl * stride;
          double vr = Vs[o];
          rho[sp] += cd * vr;
          if (do_tau)  // dftgrid.cpp:159-200: tau of each spin density
            tau[sp] += 0.5 * cd * (Vs[3 * stride + o] / (hmu * hmu) + Vs[4 * stride + o] / (hnu * hnu) +
                                   (double)(grp_m[ga] * grp_m[gb]) * vr / (hphi * hphi));
          if (do_grad) {
            gnu[sp] += cd * Vs[stride + o];
            gmu[sp] += cd * Vs[2 * stride + o];
            gphi[sp] -= grp_m[ga] * sd * vr;
          }
        }
      }
    double saa = 0.0, sab = 0.0, sbb = 0.0;
    if (do_grad) {
      for (int sp = 0; sp < 2; sp++) {
        gmu[sp] *= 2.0 / hmu;
        gnu[sp] *= 2.0 / hnu;
        gphi[sp] *= 2.0 / hphi;
      }
      saa = gmu[0] * gmu[0] + gnu[0] * gnu[0] + gphi[0] * gphi[0];
      sab = gmu[0] * gmu[1] + gnu[0] * gnu[1] + gphi[0] * gphi[1];
      sbb = gmu[1] * gmu[1] + gnu[1] * gnu[1] + gphi[1] * gphi[1];
    }
    double exc = 0.0, va = 0.0, vb = 0.0, vsaa = 0.0, vsab = 0.0, vsbb = 0.0, vta = 0.0, vtb = 0.0;
    const double rt = rho[0] + rho[1];
    if (rt >= thr && rt > 0.0) {
      double ra = fmax(rho[0], thr), rb = fmax(rho[1], thr);
      for (int f = 0; f < 2; f++) {
        const int id = f ? c_func : x_func;
        if (id <= 0) continue;
        if (xc::is_mgga(id)) xc::eval_add_mgga_pol(id, ra, rb, saa, sab, sbb, tau[0], tau[1], rho[0] >= thr, rho[1] >= thr, exc, va, vb, vsaa, vsab, vsbb, vta, vtb);
        else xc::eval_add_pol(id, ra, rb, saa, sab, sbb, rho[0] >= thr, rho[1] >= thr, exc, va, vb, vsaa, vsab, vsbb);
      }
    }
    nel += w * rt;
    exc_sum += w * exc * rt;
    kin_sum += w * (tau[0] + tau[1]);
    sh[0 * ng + pt] = w * va;
    sh[npot * ng + pt] = w * vb;
    if (do_tau) {
      sh[4 * ng + pt] = 0.5 * w * vta;
      sh[(npot + 4) * ng + pt] = 0.5 * w * vtb;
    }
    if (do_grad) {
      // gr_a = w (2 vs_aa grad rho_a + vs_ab grad rho_b) / h ; gr_b likewise   (dftgrid.cpp:583-601)
      sh[1 * ng + pt] = w * (2.0 * vsaa * gmu[0] + vsab * gmu[1]) / hmu;
      sh[2 * ng + pt] = w * (2.0 * vsaa * gnu[0] + vsab * gnu[1]) / hnu;
      sh[3 * ng + pt] = w * (2.0 * vsaa * gphi[0] + vsab * gphi[1]) / hphi;
      sh[(npot + 1) * ng + pt] = w * (2.0 * vsbb * gmu[1] + vsab * gmu[0]) / hmu;
      sh[(npot + 2) * ng + pt] = w * (2.0 * vsbb * gnu[1] + vsab * gnu[0]) / hnu;
      sh[(npot + 3) * ng + pt] = w * (2.0 * vsbb * gphi[1] + vsab * gphi[0]) / hphi;
    }
  }
  __syncthreads();
  int nout = G * G * rows;
  for (int t = threadIdx.x; t < 2 * nout; t += blockDim.x) {
    int sp = t / nout, tt = t % nout;
    const int il = tt % rows;
    int i = i0 + il;
    int gab = tt / rows;
    int ga = gab / G, gb = gab % G;
    int D = grp_m[ga] - grp_m[gb];
    const double *cd = cosd + (size_t)(D + Dmax) * nphi;
    const double *sd = sind + (size_t)(D + Dmax) * nphi;
    const double *p0 = sh + (size_t)(npot * sp) * ng, *p1 = p0 + ng, *p2 = p0 + 2 * ng, *p3 = p0 + 3 * ng, *p4 = p0 + 4 * ng;
    double fa = 0.0, fs = 0.0, fb = 0.0, ft = 0.0;
    double mga = grp_m[ga];
    for (int j = 0; j < nphi; j++) {
      int pt = il * nphi + j;
      fa += 0.5 * p0[pt] * cd[j];
      if (do_grad) {
        fa -= mga * p3[pt] * sd[j];
        fs += p2[pt] * cd[j];
        fb += p1[pt] * cd[j];
      }
      if (do_tau) ft += p4[pt] * cd[j];
    }
    size_t o = ((Q * G + ga) * G + gb) * nth + i;
    double *Fs = Fo + (size_t)sp * npl * stride;
    if (do_tau) {  // the three tau terms of eval_Fxc, as in k_xc_grid
      double sth = th_s[i], hmu, hnu, hphi;
      if (geom == 0) {
        hmu = hnu = Rh * sqrt(shm * shm + sth * sth);
        hphi = Rh * shm * sth;
      } else {
        hmu = 1.0;
        hnu = shm;
        hphi = shm * sth;
      }
      fa += 0.5 * mga * (double)grp_m[gb] * ft / (hphi * hphi);
      Fs[3 * stride + o] = 0.5 * ft / (hmu * hmu);
      Fs[4 * stride + o] = 0.5 * ft / (hnu * hnu);
    }
    Fs[o] = fa;
    if (do_grad) {
      Fs[stride + o] = fs;
      Fs[2 * stride + o] = fb;
    }
  }
  }  // theta chunks
  int nwave = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  for (int o = 32; o > 0; o >>= 1) {
    nel += __shfl_down(nel, o, 64);
    exc_sum += __shfl_down(exc_sum, o, 64);
    kin_sum += __shfl_down(kin_sum, o, 64);
  }
  if (lane == 0) {
    red[wave] = nel;
    red[nwave + wave] = exc_sum;
    red[2 * nwave + wave] = kin_sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c2 = 0.0;
    for (int w = 0; w < nwave; w++) {
      a += red[w];
      b += red[nwave + w];
      c2 += red[2 * nwave + w];
    }
    partial[Q] = a;
    partial[NQ + Q] = b;
    partial[2 * NQ + Q] = c2;
  }
}

// X4 theta expansion: GA[Q][a][b] = sum_i Theta_a Theta_b Fo0 + dTheta_a Theta_b Fo1 ; GB[Q][a][b] = sum_i Theta_a Theta_b Fo2
//    meta-GGA: GA += sum_i dTheta_a dTheta_b Fo4 ; GC[Q][a][b] = sum_i Theta_a Theta_b Fo3
__global__ void k_xc_fock_theta(const double *__restrict__ Fo, const double *__restrict__ Th,
                                const double *__restrict__ dTh, int A, int nth, int G,
                                const int *__restrict__ grp_off, const int *__restrict__ grp_shell, int do_grad,
                                int do_tau, size_t NQ, int rank, int nranks, double *__restrict__ GA,
                                double *__restrict__ GB, double *__restrict__ GC) {
  // LDS: f0..f4[nth], then the Theta / dTheta rows of the two shell groups, [i][shell] (the pair loop reads them with
  // consecutive b across threads; from global memory the loop was latency-bound: 0.26 ms per build at Nbf = 4230)
  extern __shared__ double sh[];
  size_t Q = blockIdx.x;
  if ((int)(Q % nranks) != rank) return;
  int ga = blockIdx.y / G, gb = blockIdx.y % G;
  int a0 = grp_off[ga], na = grp_off[ga + 1] - a0;
  int b0 = grp_off[gb], nb = grp_off[gb + 1] - b0;
  size_t stride = NQ * G * G * nth;
  size_t o = ((Q * G + ga) * G + gb) * nth;
  double *f0 = sh, *f1 = sh + nth, *f2 = sh + 2 * nth, *f3 = sh + 3 * nth, *f4 = sh + 4 * nth;
  double *sTa = sh + 5 * nth, *sDa = sTa + (size_t)nth * na, *sTb = sDa + (size_t)nth * na, *sDb = sTb + (size_t)nth * nb;
  for (int i = threadIdx.x; i < nth; i += blockDim.x) {
    f0[i] = Fo[o + i];
    f1[i] = do_grad ? Fo[stride + o + i] : 0.0;
    f2[i] = do_grad ? Fo[2 * stride + o + i] : 0.0;
    f3[i] = do_tau ? Fo[3 * stride + o + i] : 0.0;
    f4[i] = do_tau ? Fo[4 * stride + o + i] : 0.0;
  }
  for (int t = threadIdx.x; t < nth * na; t += blockDim.x) {
    int ia = t / nth, i = t % nth;  // consecutive threads read consecutive theta points of one shell
    int a = grp_shell[a0 + ia];
    sTa[i * na + ia] = Th[(size_t)a * nth + i];
    sDa[i * na + ia] = dTh[(size_t)a * nth + i];
  }
  for (int t = threadIdx.x; t < nth * nb; t += blockDim.x) {
    int ib = t / nth, i = t % nth;
    int b = grp_shell[b0 + ib];
    sTb[i * nb + ib] = Th[(size_t)b * nth + i];
    sDb[i * nb + ib] = dTh[(size_t)b * nth + i];
  }
  __syncthreads();
  size_t AA = (size_t)A * A;
  for (int t = threadIdx.x; t < na * nb; t += blockDim.x) {
    const int ia = t / nb, ib = t % nb;
    int a = grp_shell[a0 + ia], b = grp_shell[b0 + ib];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < nth; i++) {
      const double tai = sTa[i * na + ia], dai = sDa[i * na + ia], tbi = sTb[i * nb + ib];
      s0 += (tai * f0[i] + dai * f1[i]) * tbi;
      s1 += tai * f2[i] * tbi;
      if (do_tau) {
        s0 += dai * sDb[i * nb + ib] * f4[i];
        s2 += tai * tbi * f3[i];
      }
    }
    GA[Q * AA + (size_t)a * A + b] = s0;
    if (do_grad) GB[Q * AA + (size_t)a * A + b] = s1;
    if (do_tau) GC[Q * AA + (size_t)a * A + b] = s2;
  }
}

// X4 for shell groups whose Theta tables do not fit a CU's LDS (lmax beyond ~33 with mmax = 2): the theta points go
// through LDS nthc at a time and every thread keeps the sums of its (a, b) pairs in registers across the chunks
// (at most XC_FT_MAXPP pairs per thread: groups of up to 64 shells)
constexpr int XC_FT_MAXPP = 16;
__global__ __launch_bounds__(256) void k_xc_fock_theta_chunked(const double *__restrict__ Fo, const double *__restrict__ Th,
                                                               const double *__restrict__ dTh, int A, int nth, int G,
                                                               const int *__restrict__ grp_off, const int *__restrict__ grp_shell,
                                                               int do_grad, int do_tau, size_t NQ, int rank, int nranks,
                                                               double *__restrict__ GA, double *__restrict__ GB,
                                                               double *__restrict__ GC, int nthc) {
  extern __shared__ double sh[];
  size_t Q = blockIdx.x;
  if ((int)(Q % nranks) != rank) return;
  int ga = blockIdx.y / G, gb = blockIdx.y % G;
  int a0 = grp_off[ga], na = grp_off[ga + 1] - a0;
  int b0 = grp_off[gb], nb = grp_off[gb + 1] - b0;
  size_t stride = NQ * G * G * nth;
  size_t o = ((Q * G + ga) * G + gb) * nth;
  double *f0 = sh, *f1 = sh + nthc, *f2 = sh + 2 * nthc, *f3 = sh + 3 * nthc, *f4 = sh + 4 * nthc;
  double *sTa = sh + 5 * nthc, *sDa = sTa + (size_t)nthc * na, *sTb = sDa + (size_t)nthc * na, *sDb = sTb + (size_t)nthc * nb;
  double s0[XC_FT_MAXPP], s1[XC_FT_MAXPP], s2[XC_FT_MAXPP];
#pragma unroll
  for (int q = 0; q < XC_FT_MAXPP; q++) s0[q] = s1[q] = s2[q] = 0.0;
  for (int c0 = 0; c0 < nth; c0 += nthc) {
    const int len = min(nthc, nth - c0);
    if (c0) __syncthreads();
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      f0[i] = Fo[o + c0 + i];
      f1[i] = do_grad ? Fo[stride + o + c0 + i] : 0.0;
      f2[i] = do_grad ? Fo[2 * stride + o + c0 + i] : 0.0;
      f3[i] = do_tau ? Fo[3 * stride + o + c0 + i] : 0.0;
      f4[i] = do_tau ? Fo[4 * stride + o + c0 + i] : 0.0;
    }
    for (int t = threadIdx.x; t < len * na; t += blockDim.x) {
      int ia = t / len, i = t % len;
      int a = grp_shell[a0 + ia];
      sTa[i * na + ia] = Th[(size_t)a * nth + c0 + i];
      sDa[i * na + ia] = dTh[(size_t)a * nth + c0 + i];
    }
    for (int t = threadIdx.x; t < len * nb; t += blockDim.x) {
      int ib = t / len, i = t % len;
      int b = grp_shell[b0 + ib];
      sTb[i * nb + ib] = Th[(size_t)b * nth + c0 + i];
      sDb[i * nb + ib] = dTh[(size_t)b * nth + c0 + i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < XC_FT_MAXPP; q++) {
      const int t = threadIdx.x + q * 256;
      if (t < na * nb) {
        const int ia = t / nb, ib = t % nb;
        double t0 = s0[q], t1 = s1[q], t2 = s2[q];
        for (int i = 0; i < len; i++) {
          const double tai = sTa[i * na + ia], dai = sDa[i * na + ia], tbi = sTb[i * nb + ib];
          t0 += (tai * f0[i] + dai * f1[i]) * tbi;
          t1 += tai * f2[i] * tbi;
          if (do_tau) {
            t0 += dai * sDb[i * nb + ib] * f4[i];
            t2 += tai * tbi * f3[i];
          }
        }
        s0[q] = t0;
        s1[q] = t1;
        s2[q] = t2;
      }
    }
  }
  size_t AA = (size_t)A * A;
#pragma unroll
  for (int q = 0; q < XC_FT_MAXPP; q++) {
    const int t = threadIdx.x + q * 256;
    if (t < na * nb) {
      const int ia = t / nb, ib = t % nb;
      int a = grp_shell[a0 + ia], b = grp_shell[b0 + ib];
      GA[Q * AA + (size_t)a * A + b] = s0[q];
      if (do_grad) GB[Q * AA + (size_t)a * A + b] = s1[q];
      if (do_tau) GC[Q * AA + (size_t)a * A + b] = s2[q];
    }
  }
}

// X5 radial expansion into the compact Fock blocks:
//    Hc[x][y][e][n'][n] = sum_q B_n B_n' (GA_xy + GA_yx) + B'_n B_n' GB_xy + B_n B'_n' GB_yx
//    meta-GGA: + B'_n B'_n' (GC_xy + GC_yx)
__global__ void k_xc_fock_radial(const double *__restrict__ GA, const double *__restrict__ GB,
                                 const double *__restrict__ GC, const double *__restrict__ B,
                                 const double *__restrict__ dB, int A, int E, int p, int nq, int do_grad, int do_tau,
                                 int rank, int nranks, double *__restrict__ Hc) {
  extern __shared__ double sh[];  // gs[nq], g1[nq], g2[nq], g3[nq], B[nq][p], dB[nq][p] of this element
  int xy = blockIdx.x, e = blockIdx.y;
  int x = xy / A, y = xy % A;
  int yx = y * A + x;
  size_t AA = (size_t)A * A;
  double *gs = sh, *g1 = sh + nq, *g2 = sh + 2 * nq, *g3 = sh + 3 * nq;
  double *sB = sh + 4 * nq, *sdB = sB + nq * p;
  for (int q = threadIdx.x; q < nq; q += blockDim.x) {
    size_t Q = (size_t)e * nq + q;
    bool own = ((int)(Q % nranks) == rank);
    gs[q] = own ? GA[Q * AA + xy] + GA[Q * AA + yx] : 0.0;
    g1[q] = (own && do_grad) ? GB[Q * AA + xy] : 0.0;
    g2[q] = (own && do_grad) ? GB[Q * AA + yx] : 0.0;
    g3[q] = (own && do_tau) ? GC[Q * AA + xy] + GC[Q * AA + yx] : 0.0;
  }
  for (int t = threadIdx.x; t < nq * p; t += blockDim.x) {
    sB[t] = B[(size_t)e * nq * p + t];
    sdB[t] = dB[(size_t)e * nq * p + t];
  }
  __syncthreads();
  // U[q][m] = gs B_m + g2 B'_m,  W[q][m] = g1 B_m + g3 B'_m:  H[n][m] = sum_q B_n U[q][m] + B'_n W[q][m]  (four LDS reads
  // per point and output instead of eight: the loop is LDS-bound)
  double *sU = sdB + nq * p, *sW = sU + nq * p;
  for (int t = threadIdx.x; t < nq * p; t += blockDim.x) {
    const int q = t / p;
    sU[t] = gs[q] * sB[t] + g2[q] * sdB[t];
    sW[t] = g1[q] * sB[t] + g3[q] * sdB[t];
  }
  __syncthreads();
  int pp = p * p;
  for (int t = threadIdx.x; t < pp; t += blockDim.x) {
    int n = t % p, m = t / p;
    double acc = 0.0;
    for (int q = 0; q < nq; q++) acc += sB[q * p + n] * sU[q * p + m] + sdB[q * p + n] * sW[q * p + m];
    Hc[((size_t)xy * E + e) * pp + t] = acc;
  }
}

__global__ void k_xc_sum_partials(const double *__restrict__ partial, size_t NQ, double *__restrict__ scal) {
  // one wave: lane l adds the points l, l + 64, ... in order, then a fixed shuffle tree (deterministic)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double nel = 0.0, exc = 0.0, kin = 0.0;
  for (size_t q = threadIdx.x; q < NQ; q += 64) {
    nel += partial[q];
    exc += partial[NQ + q];
    kin += partial[2 * NQ + q];
  }
  for (int o = 32; o > 0; o >>= 1) {
    nel += __shfl_down(nel, o, 64);
    exc += __shfl_down(exc, o, 64);
    kin += __shfl_down(kin, o, 64);
  }
  if (threadIdx.x == 0) {
    scal[0] = exc;  // Exc
    scal[1] = nel;  // Nel
    scal[2] = kin;  // Ekin = integral of tau (zero unless a meta-GGA asked for tau; dftgrid.cpp:227-240)
  }
}

// -------------------------------------------------------------------------------------------------
// Model-potential matrix of the initial guess (TwoDGrid::model_potential, src/diatomic/twodquadrature.cpp:213-232,
// 351-375): H_ij = int phi_i [V_1(r_1) + V_2(r_2)] phi_j on the same (mu, nu) product grid as the XC quadrature.  The
// integrand does not depend on phi, so the phi transform of k_xc_grid collapses to the m-diagonal planes
// Fo[0][Q][g][g][i] = 1/2 nphi w v; the theta and radial expansions X4, X5 are reused unchanged.
struct DevModelPot {
  int kind, Z;
  double d, H;
};
__device__ inline double mp_potential(const DevModelPot &p, double r) {
  double zeff;
  if (p.kind == 0) zeff = (double)p.Z;
  else if (p.kind == 1) {
    const double Hz = (p.H > 0.0) ? p.H : p.d * pow((double)(p.Z - 1), 0.4);
    zeff = 1.0 + (p.Z - 1) / (1.0 + (exp(r / p.d) - 1.0) * Hz);
  } else {
    const double alpha = 0.7280642371, beta = -0.5430794693, gamma = 0.3612163121;
    const double x = r * cbrt(128.0 * p.Z / (9.0 * HFG_PI * HFG_PI)), sx = sqrt(x);
    const double f = 1.0 + alpha * sx + beta * x * exp(-gamma * sx);
    zeff = p.Z * f * f * exp(-2.0 * alpha * sx);
  }
  const double v = -zeff / r;
  return (isfinite(v) && v != 0.0 && fabs(v) >= 2.2250738585072014e-308) ? v : 0.0;  // std::isnormal, as the reference
}
__global__ void k_mp_fill(const double *__restrict__ rad_w, const double *__restrict__ rad_sh,
                          const double *__restrict__ th_c, const double *__restrict__ th_s,
                          const double *__restrict__ th_w, int G, int nth, int nphi, double Rh, int geom, DevModelPot p1,
                          DevModelPot p2, double *__restrict__ Fo) {
  const size_t Q = blockIdx.x;
  const double shm = rad_sh[Q], wr = rad_w[Q];
  const double dphi = 2.0 * HFG_PI / nphi;
  for (int t = threadIdx.x; t < G * G * nth; t += blockDim.x) {
    const int i = t % nth, gab = t / nth, ga = gab / G, gb = gab % G;
    double f = 0.0;
    if (ga == gb) {
      const double sth = th_s[i], cth = th_c[i];
      double w, v;
      if (geom == 0) {
        const double chm = sqrt(1.0 + shm * shm);
        w = th_w[i] * dphi * wr * Rh * Rh * Rh * shm * (shm * shm + sth * sth);
        v = mp_potential(p1, Rh * (chm + cth)) + mp_potential(p2, Rh * (chm - cth));
      } else {
        w = th_w[i] * dphi * wr * shm * shm;
        v = mp_potential(p1, shm);
      }
      f = 0.5 * nphi * w * v;
    }
    Fo[((Q * G + ga) * G + gb) * nth + i] = f;
  }
}

// launchers
// -------------------------------------------------------------------------------------------------
static int round_up64(int n) { return ((n + 63) / 64) * 64; }
static size_t xc_fock_radial_lds(int p, int nq) {
  const size_t shb = (size_t)(4 * nq + 4 * nq * p) * sizeof(double);
  if (shb > 150 * 1024) throw std::runtime_error("radial quadrature too large for the XC Fock kernel's LDS tables");
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_xc_fock_radial, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  return shb;
}
// LDS of k_xc_fock_theta: five potential rows and the Theta, dTheta rows of two shell groups
static size_t xc_fock_theta_lds(int nth, int maxgrp) {
  size_t shb = (size_t)(5 * nth + 4 * nth * maxgrp) * sizeof(double);
  if (shb > 160 * 1024) throw std::runtime_error("angular grid too large for the XC Fock kernel's LDS tables");  // callers use launch_xc_fock_theta
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_xc_fock_theta, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  return shb;
}
// LDS the angular kernels may plan with; HELFEM_XC_LDS_LIMIT (bytes) lowers it so that the chunked paths can be exercised on
// small bases (tests)
static size_t xc_lds_limit() {
  static const size_t v = getenv("HELFEM_XC_LDS_LIMIT") ? (size_t)atol(getenv("HELFEM_XC_LDS_LIMIT")) : (size_t)150 * 1024;
  return v;
}
// X4 launch: the one-pass kernel when its tables fit a CU's LDS, the chunked one otherwise
static void launch_xc_fock_theta(hfg_ctx *ctx, size_t NQ, int G, int nth, int maxgrp, const double *Fo, const double *Th, const double *dTh,
                                 int A, const int *grp_off, const int *grp_shell, int do_grad, int do_tau, int rank, int nranks,
                                 double *GA, double *GB, double *GC) {
  const size_t need = (size_t)(5 * nth + 4 * nth * maxgrp) * sizeof(double);
  if (need <= xc_lds_limit()) {
    hipLaunchKernelGGL(k_xc_fock_theta, dim3((unsigned)NQ, G * G), dim3(256), xc_fock_theta_lds(nth, maxgrp), ctx->stream, Fo, Th, dTh, A, nth,
                       G, grp_off, grp_shell, do_grad, do_tau, NQ, rank, nranks, GA, GB, GC);
    return;
  }
  if (maxgrp * maxgrp > XC_FT_MAXPP * 256) throw std::runtime_error("angular basis too large for the XC Fock kernels (more than 64 shells of one m)");
  int nthc = (int)(xc_lds_limit() / sizeof(double) / (5 + 4 * maxgrp));
  if (nthc < 4) throw std::runtime_error("angular basis too large for the XC Fock kernels' LDS tables");
  nthc = std::min(nthc, nth);
  const size_t shb = (size_t)(5 * nthc + 4 * nthc * maxgrp) * sizeof(double);
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_xc_fock_theta_chunked, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  hipLaunchKernelGGL(k_xc_fock_theta_chunked, dim3((unsigned)NQ, G * G), dim3(256), shb, ctx->stream, Fo, Th, dTh, A, nth, G, grp_off, grp_shell,
                     do_grad, do_tau, NQ, rank, nranks, GA, GB, GC, nthc);
}

// LDS of k_xc_density_radial (256 threads): P block, the two transposed tables, the partial sums of the j classes
static size_t xc_fock_radial_lds(int p, int nq);
static size_t xc_density_radial_lds(int p, int nq) {
  const int NJ = (256 / nq > 0) ? std::min(256 / nq, p) : 1;
  size_t shb = (size_t)(p * p + 2 * p * nq + 3 * NJ * nq) * sizeof(double);
  if (shb > 150 * 1024) throw std::runtime_error("radial quadrature too large for the XC density kernel's LDS tables");
  if (shb > 64 * 1024)  // elements of more than ~24 nodes with the default 5 quadrature points per node
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_xc_density_radial, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  return shb;
}

struct FockAux {
  DevBuf<int> pure_shell, pure_n, lmpos;
  DevBuf<double> Pc, Jc, Pc2, Jc2, Paux, Y, Jaux, D0, D1, D2, V, Fo, GA, GB, GC, partial, scal;
};

static std::map<hfg_dev_tables *, FockAux *> g_aux;

static FockAux &aux_for(hfg_ctx *ctx, hfg_basis *basis) {
  hfg_dev_tables *t = basis->dev;
  auto it = g_aux.find(t);
  if (it != g_aux.end()) return *it->second;
  FockAux *a = new FockAux();
  std::vector<int> ps(t->N), pn(t->N);
  {
    size_t k = 0;
    for (int s = 0; s < t->A; s++)
      for (int n = (t->h_shell_skip[s] ? 1 : 0); n < t->R; n++, k++) {
        ps[k] = s;
        pn[k] = n;
      }
  }
  a->pure_shell.upload(ps, ctx->stream);
  a->pure_n.upload(pn, ctx->stream);
  std::vector<int> lmpos(2 * t->Nlm, -1);
  for (int i = 0; i < t->NLM; i++) {
    int M = t->h_LM_M[i];
    int ilm = t->h_LM_ilm[i];
    if (M >= 0) lmpos[2 * ilm] = i;
    if (M < 0) lmpos[2 * ilm + 1] = i;
  }
  a->lmpos.upload(lmpos, ctx->stream);
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  g_aux[t] = a;
  return *a;
}

void fock_release(hfg_dev_tables *t) {
  auto it = g_aux.find(t);
  if (it != g_aux.end()) {
    delete it->second;
    g_aux.erase(it);
  }
}

static hfg_dev_tables *tables_of(hfg_ctx *ctx, hfg_basis *basis) {
  if (!basis->dev || !basis->dev->have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
  if (basis->dev_device != ctx->device) throw std::logic_error("basis tables live on a different device\n");
  return basis->dev;
}

void gather_compact(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dPc) {
  hfg_dev_tables *t = basis->dev;
  hipLaunchKernelGGL(k_gather_compact, dim3(t->A * t->A * t->E), dim3(256), 0, ctx->stream, dP, t->N, t->A, t->R, t->E,
                     t->p, t->shell_off.p, t->shell_skip.p, dPc);
}

void scatter_dense(hfg_ctx *ctx, hfg_basis *basis, const double *dXc, double *dOut) {
  hfg_dev_tables *t = basis->dev;
  FockAux &a = aux_for(ctx, basis);
  dim3 grid((t->N + 63) / 64, (t->N + 3) / 4);
  hipLaunchKernelGGL(k_scatter_dense, grid, dim3(256), 0, ctx->stream, dXc, t->N, t->A, t->R, t->E, t->p,
                     a.pure_shell.p, a.pure_n.p, dOut);
}

// J (compact) from P (compact)
void coulomb_compact(hfg_ctx *ctx, hfg_basis *basis, const double *dPc, double *dJc) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  FockAux &a = aux_for(ctx, basis);
  const int pp = t->p * t->p;
  const size_t nb = (size_t)t->NLM * t->E * pp;
  a.Paux.resize(2 * nb);
  a.Y.resize(4 * nb);
  a.Jaux.resize(2 * nb);
  int bs = std::min(256, round_up64(pp));
  hipLaunchKernelGGL(k_coulomb_ket, dim3(t->NLM, t->E), dim3(bs), 0, ctx->stream, dPc, t->A, t->E, pp, t->lm_off.p,
                     t->lm_x.p, t->lm_y.p, t->lm_c0.p, t->lm_c2.p, t->NLM, t->LM_ilm.p, ctx->shard_rank, ctx->shard_n, a.Paux.p);
  hipLaunchKernelGGL(k_coulomb_tei, dim3(t->Nlm * t->E, t->ntt), dim3(bs), 2 * pp * sizeof(double), ctx->stream,
                     t->tei.p, a.Paux.p, t->Ntab, t->NLM, t->E, pp, a.lmpos.p, t->lm_tab.p, ctx->shard_rank, ctx->shard_n,
                     a.Y.p);
  int nwave = bs / 64;
  size_t shb = (size_t)(4 * t->E * nwave + 4 * t->E + 2 * t->E) * sizeof(double);
  hipLaunchKernelGGL(k_coulomb_radial, dim3(t->NLM), dim3(bs), shb, ctx->stream, a.Paux.p, a.Y.p, t->disj.p,
                     t->LM_ilm.p, t->LM_tab.p, t->LM_fac.p, t->Ntab, t->NLM, t->E, t->p, t->ntt == 4 ? 1 : 0,
                     ctx->shard_rank, ctx->shard_n, a.Jaux.p);
  hipLaunchKernelGGL(k_coulomb_bra, dim3(t->A * t->A, t->E), dim3(bs), 0, ctx->stream, a.Jaux.p, t->A, t->E, pp,
                     t->NLM, t->pair_off.p, t->ent_iLM.p, t->ent_c0.p, t->ent_c2.p, t->LM_ilm.p, ctx->shard_rank, ctx->shard_n, dJc);
  HFG_HIP_CHECK(hipGetLastError());
}

void coulomb_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dJ) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  FockAux &a = aux_for(ctx, basis);
  const size_t nc = (size_t)t->A * t->A * t->E * t->p * t->p;
  a.Pc.resize(nc);
  a.Jc.resize(nc);
  ProfScope ps(ctx, "coulomb");
  gather_compact(ctx, basis, dP, a.Pc.p);
  coulomb_compact(ctx, basis, a.Pc.p, a.Jc.p);
  scatter_dense(ctx, basis, a.Jc.p, dJ);
  HFG_HIP_CHECK(hipGetLastError());
}

void xc_compact(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dPc, double *dHc, double *dScal,
                double thr) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  if (!t->have_xc) throw std::runtime_error("XC grid tables were not uploaded (hfg_basis_upload with ldft,mdft > 0)\n");
  if ((x_func > 0 && !xc::is_supported(x_func)) || (c_func > 0 && !xc::is_supported(c_func)))
    throw std::runtime_error("Functional not found!");
  FockAux &a = aux_for(ctx, basis);
  const int A = t->A, E = t->E, p = t->p, nq = t->nq, G = t->G, nth = t->ntheta, nphi = t->nphi;
  const size_t NQ = (size_t)E * nq, AA = (size_t)A * A;
  int do_grad = ((x_func > 0 && xc::is_gga(x_func)) || (c_func > 0 && xc::is_gga(c_func))) ? 1 : 0;
  int do_tau = ((x_func > 0 && xc::is_mgga(x_func)) || (c_func > 0 && xc::is_mgga(c_func))) ? 1 : 0;
  // meta-GGAs: a floor under the density threshold.  Below 1e-50 the tau-dependent expressions overflow (tau_unif ~ n^(5/3),
  // p ~ sigma / n^(8/3)) and return NaN where the point carries nothing; libxc keeps such points out with its own tau and
  // sigma thresholds, --dftthr 0 would switch the density threshold off.  Far-field densities of an SCF density are
  // rounding noise of the eigensolver at that level (tests/test_gpu_fullsize.py::test_fullsize_xc_without_density_threshold).
  if (do_tau) thr = std::max(thr, 1e-40);
  a.D0.resize(NQ * AA);
  a.D1.resize(NQ * AA);
  a.GA.resize(NQ * AA);
  a.GB.resize(NQ * AA);
  if (do_tau) {
    a.D2.resize(NQ * AA);
    a.GC.resize(NQ * AA);
  }
  const size_t nv = NQ * G * G * nth;
  a.V.resize(5 * nv);
  a.Fo.resize(5 * nv);
  a.partial.resize(3 * NQ);
  int maxgrp = 0;
  for (int g = 0; g < G; g++) maxgrp = std::max(maxgrp, t->h_grp_off[g + 1] - t->h_grp_off[g]);

  hipLaunchKernelGGL(k_xc_density_radial, dim3(A * A, E), dim3(256), xc_density_radial_lds(p, nq),
                     ctx->stream, dPc, t->rad_B.p, t->rad_dB.p, A, E, p, nq, do_grad, do_tau, ctx->shard_rank, ctx->shard_n, a.D0.p,
                     a.D1.p, a.D2.p);
  hipLaunchKernelGGL(k_xc_density_theta, dim3((unsigned)NQ, G * G), dim3(std::min(256, round_up64(nth))),
                     3 * maxgrp * maxgrp * sizeof(double), ctx->stream, a.D0.p, a.D1.p, a.D2.p, t->Th.p, t->dTh.p, A, nth, G,
                     t->grp_off.p, t->grp_shell.p, do_grad, do_tau, NQ, ctx->shard_rank, ctx->shard_n, a.V.p);
  size_t shb = (size_t)(5 * nth * nphi + 3 * 4) * sizeof(double);
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_xc_grid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  hipLaunchKernelGGL(k_xc_grid, dim3((unsigned)NQ), dim3(256), shb, ctx->stream, a.V.p, t->rad_w.p, t->rad_sh.p,
                     t->th_s.p, t->th_w.p, t->grp_m.p, t->cosd.p, t->sind.p, t->Dmax, G, nth, nphi, t->Rhalf, t->geom,
                     x_func, c_func, do_grad, do_tau, thr, NQ, ctx->shard_rank, ctx->shard_n, a.Fo.p, a.partial.p);
  launch_xc_fock_theta(ctx, NQ, G, nth, maxgrp, a.Fo.p, t->Th.p, t->dTh.p, A, t->grp_off.p, t->grp_shell.p, do_grad, do_tau, ctx->shard_rank,
                       ctx->shard_n, a.GA.p, a.GB.p, a.GC.p);
  hipLaunchKernelGGL(k_xc_fock_radial, dim3(A * A, E), dim3(std::min(256, round_up64(p * p))), xc_fock_radial_lds(p, nq),
                     ctx->stream, a.GA.p, a.GB.p, a.GC.p, t->rad_B.p, t->rad_dB.p, A, E, p, nq, do_grad, do_tau,
                     ctx->shard_rank, ctx->shard_n, dHc);
  hipLaunchKernelGGL(k_xc_sum_partials, dim3(1), dim3(64), 0, ctx->stream, a.partial.p, NQ, dScal);
  HFG_HIP_CHECK(hipGetLastError());
}

/// External functional parameters for the following XC builds on this stream (NULL / 0: the functional's defaults).
/// Supported: lda_x {alpha}, gga_x_pbe {kappa, mu}, gga_c_pbe {beta, gamma, BB} -- libxc's parameter lists; anything
/// else throws (std::runtime_error, as libxc's "number of parameters" check does through the reference).
void set_xc_params(hfg_ctx *ctx, int x_func, const double *x_pars, int nx, int c_func, const double *c_pars, int nc) {
  xc::XCPar par = HFG_XCPAR_DEFAULTS;
  if (nx > 0) {
    if (!x_pars) throw std::runtime_error("Exchange functional parameters missing.\n");
    if (x_func == 1 && nx == 1) par.x_alpha = x_pars[0];
    else if (x_func == 101 && nx == 2) {
      par.x_kappa = x_pars[0];
      par.x_mu = x_pars[1];
    } else
      throw std::runtime_error("External parameters are not supported for exchange functional " + std::to_string(x_func) + " with " +
                               std::to_string(nx) + " values (supported: lda_x {alpha}, gga_x_pbe {kappa, mu}).\n");
  }
  if (nc > 0) {
    if (!c_pars) throw std::runtime_error("Correlation functional parameters missing.\n");
    if (c_func == 130 && nc == 3) {
      par.c_beta = c_pars[0];
      par.c_gamma = c_pars[1];
      par.c_BB = c_pars[2];
    } else
      throw std::runtime_error("External parameters are not supported for correlation functional " + std::to_string(c_func) + " with " +
                               std::to_string(nc) + " values (supported: gga_c_pbe {beta, gamma, BB}).\n");
  }
  // the host copy must stay valid until the asynchronous copy has run: a small per-thread ring
  static thread_local xc::XCPar staged[8];
  static thread_local int slot = 0;
  staged[slot] = par;
  HFG_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(xc::c_xcpar), &staged[slot], sizeof(par), 0, hipMemcpyHostToDevice, ctx->stream));
  slot = (slot + 1) % 8;
}

void xc_fock_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dH, double *dScal,
                 double thr) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  FockAux &a = aux_for(ctx, basis);
  const size_t nc = (size_t)t->A * t->A * t->E * t->p * t->p;
  a.Pc.resize(nc);
  a.Jc.resize(nc);
  ProfScope ps(ctx, "xc");
  gather_compact(ctx, basis, dP, a.Pc.p);
  xc_compact(ctx, basis, x_func, c_func, a.Pc.p, a.Jc.p, dScal, thr);
  scatter_dense(ctx, basis, a.Jc.p, dH);
  HFG_HIP_CHECK(hipGetLastError());
}

// spin-polarised XC: Hc_a, Hc_b (compact) from Pc_a, Pc_b (compact); dScal = (Exc, Nel, 0)
void xc_compact_pol(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dPca, const double *dPcb,
                    double *dHca, double *dHcb, double *dScal, double thr) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  if (!t->have_xc) throw std::runtime_error("XC grid tables were not uploaded (hfg_basis_upload with ldft,mdft > 0)\n");
  if ((x_func > 0 && !xc::is_supported(x_func)) || (c_func > 0 && !xc::is_supported(c_func)))
    throw std::runtime_error("Functional not found!");
  FockAux &a = aux_for(ctx, basis);
  const int A = t->A, E = t->E, p = t->p, nq = t->nq, G = t->G, nth = t->ntheta, nphi = t->nphi;
  const size_t NQ = (size_t)E * nq, AA = (size_t)A * A;
  int do_grad = ((x_func > 0 && xc::is_gga(x_func)) || (c_func > 0 && xc::is_gga(c_func))) ? 1 : 0;
  int do_tau = ((x_func > 0 && xc::is_mgga(x_func)) || (c_func > 0 && xc::is_mgga(c_func))) ? 1 : 0;
  // meta-GGAs: a floor under the density threshold.  Below 1e-50 the tau-dependent expressions overflow (tau_unif ~ n^(5/3),
  // p ~ sigma / n^(8/3)) and return NaN where the point carries nothing; libxc keeps such points out with its own tau and
  // sigma thresholds, --dftthr 0 would switch the density threshold off.  Far-field densities of an SCF density are
  // rounding noise of the eigensolver at that level (tests/test_gpu_fullsize.py::test_fullsize_xc_without_density_threshold).
  if (do_tau) thr = std::max(thr, 1e-40);
  const int npl = do_tau ? 5 : 3;
  a.D0.resize(NQ * AA);
  a.D1.resize(NQ * AA);
  a.GA.resize(NQ * AA);
  a.GB.resize(NQ * AA);
  if (do_tau) {
    a.D2.resize(NQ * AA);
    a.GC.resize(NQ * AA);
  }
  const size_t nv = NQ * G * G * nth;
  a.V.resize(2 * npl * nv);
  a.Fo.resize(2 * npl * nv);
  a.partial.resize(3 * NQ);
  int maxgrp = 0;
  for (int g = 0; g < G; g++) maxgrp = std::max(maxgrp, t->h_grp_off[g + 1] - t->h_grp_off[g]);
  for (int sp = 0; sp < 2; sp++) {
    hipLaunchKernelGGL(k_xc_density_radial, dim3(A * A, E), dim3(256), xc_density_radial_lds(p, nq),
                       ctx->stream, sp ? dPcb : dPca, t->rad_B.p, t->rad_dB.p, A, E, p, nq, do_grad, do_tau, ctx->shard_rank,
                       ctx->shard_n, a.D0.p, a.D1.p, a.D2.p);
    hipLaunchKernelGGL(k_xc_density_theta, dim3((unsigned)NQ, G * G), dim3(std::min(256, round_up64(nth))),
                       3 * maxgrp * maxgrp * sizeof(double), ctx->stream, a.D0.p, a.D1.p, (const double *)a.D2.p, t->Th.p, t->dTh.p,
                       A, nth, G, t->grp_off.p, t->grp_shell.p, do_grad, do_tau, NQ, ctx->shard_rank, ctx->shard_n,
                       a.V.p + (size_t)sp * npl * nv);
  }
  int rowc = nth;  // theta rows per pass through LDS
  while ((size_t)((do_tau ? 10 : 8) * rowc * nphi + 3 * 4) * sizeof(double) > xc_lds_limit() && rowc > 1) rowc = (rowc + 1) / 2;
  size_t shb = (size_t)((do_tau ? 10 : 8) * rowc * nphi + 3 * 4) * sizeof(double);
  if (shb > 150 * 1024) throw std::runtime_error("XC angular grid too large for the polarised grid kernel's LDS tile");
  if (shb > 64 * 1024)
    HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_xc_grid_pol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  hipLaunchKernelGGL(k_xc_grid_pol, dim3((unsigned)NQ), dim3(256), shb, ctx->stream, a.V.p, t->rad_w.p, t->rad_sh.p,
                     t->th_s.p, t->th_w.p, t->grp_m.p, t->cosd.p, t->sind.p, t->Dmax, G, nth, nphi, t->Rhalf, t->geom,
                     x_func, c_func, do_grad, do_tau, thr, NQ, ctx->shard_rank, ctx->shard_n, a.Fo.p, a.partial.p, rowc);
  for (int sp = 0; sp < 2; sp++) {
    launch_xc_fock_theta(ctx, NQ, G, nth, maxgrp, a.Fo.p + (size_t)sp * npl * nv, t->Th.p, t->dTh.p, A, t->grp_off.p, t->grp_shell.p, do_grad,
                         do_tau, ctx->shard_rank, ctx->shard_n, a.GA.p, a.GB.p, a.GC.p);
    hipLaunchKernelGGL(k_xc_fock_radial, dim3(A * A, E), dim3(std::min(256, round_up64(p * p))), xc_fock_radial_lds(p, nq),
                       ctx->stream, a.GA.p, a.GB.p, (const double *)a.GC.p, t->rad_B.p, t->rad_dB.p, A, E, p, nq, do_grad, do_tau,
                       ctx->shard_rank, ctx->shard_n, sp ? dHcb : dHca);
  }
  hipLaunchKernelGGL(k_xc_sum_partials, dim3(1), dim3(64), 0, ctx->stream, a.partial.p, NQ, dScal);
  HFG_HIP_CHECK(hipGetLastError());
}

void xc_fock_pol_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dPa, const double *dPb,
                     double *dHa, double *dHb, double *dScal, double thr) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  FockAux &a = aux_for(ctx, basis);
  const size_t nc = (size_t)t->A * t->A * t->E * t->p * t->p;
  a.Pc.resize(nc);
  a.Jc.resize(nc);
  a.Pc2.resize(nc);
  a.Jc2.resize(nc);
  ProfScope ps(ctx, "xc");
  gather_compact(ctx, basis, dPa, a.Pc.p);
  gather_compact(ctx, basis, dPb, a.Pc2.p);
  xc_compact_pol(ctx, basis, x_func, c_func, a.Pc.p, a.Pc2.p, a.Jc.p, a.Jc2.p, dScal, thr);
  scatter_dense(ctx, basis, a.Jc.p, dHa);
  scatter_dense(ctx, basis, a.Jc2.p, dHb);
  HFG_HIP_CHECK(hipGetLastError());
}

void model_potential_dev(hfg_ctx *ctx, hfg_basis *basis, int kind1, int Z1, double d1, double H1, int kind2, int Z2,
                         double d2, double H2, double *dH) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  if (!t->have_xc) throw std::runtime_error("XC grid tables were not uploaded (hfg_basis_upload with ldft,mdft > 0)\n");
  for (int k : {kind1, kind2})
    if (k != 0 && k != 1 && k != 3) throw std::logic_error("Unsupported guess\n");
  if ((kind1 == 1 && !(d1 > 0.0)) || (kind2 == 1 && !(d2 > 0.0)))
    throw std::logic_error("GSZ guess: the screening length d_Z must be given\n");
  FockAux &a = aux_for(ctx, basis);
  const int A = t->A, E = t->E, p = t->p, nq = t->nq, G = t->G, nth = t->ntheta, nphi = t->nphi;
  const size_t NQ = (size_t)E * nq, AA = (size_t)A * A;
  const size_t nc = (size_t)A * A * E * p * p;
  a.Jc.resize(nc);
  a.GA.resize(NQ * AA);
  a.GB.resize(NQ * AA);
  a.Fo.resize(5 * NQ * G * G * nth);
  int maxgrp = 0;
  for (int g = 0; g < G; g++) maxgrp = std::max(maxgrp, t->h_grp_off[g + 1] - t->h_grp_off[g]);
  DevModelPot p1{kind1, Z1, d1, H1}, p2{kind2, Z2, d2, H2};
  hipLaunchKernelGGL(k_mp_fill, dim3((unsigned)NQ), dim3(256), 0, ctx->stream, t->rad_w.p, t->rad_sh.p, t->th_c.p, t->th_s.p,
                     t->th_w.p, G, nth, nphi, t->Rhalf, t->geom, p1, p2, a.Fo.p);
  launch_xc_fock_theta(ctx, NQ, G, nth, maxgrp, a.Fo.p, t->Th.p, t->dTh.p, A, t->grp_off.p, t->grp_shell.p, 0, 0, 0, 1, a.GA.p, a.GB.p,
                       (double *)nullptr);
  hipLaunchKernelGGL(k_xc_fock_radial, dim3(A * A, E), dim3(std::min(256, round_up64(p * p))),
                     xc_fock_radial_lds(p, nq), ctx->stream, a.GA.p, a.GB.p, (const double *)nullptr, t->rad_B.p,
                     t->rad_dB.p, A, E, p, nq, 0, 0, 0, 1, a.Jc.p);
  scatter_dense(ctx, basis, a.Jc.p, dH);
  HFG_HIP_CHECK(hipGetLastError());
}

size_t fock_compact_size(hfg_basis *basis) {
  return (size_t)basis->Nang() * basis->Nang() * basis->Nel() * basis->max_Nprim() * basis->max_Nprim();
}

// This shard's part of J + XC in the compact layout (to be summed over ranks), dScal = partial (Exc, Nel, 0)
void fock_compact_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dFc,
                      double *dScal, double thr) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  FockAux &a = aux_for(ctx, basis);
  const size_t nc = fock_compact_size(basis);
  a.Pc.resize(nc);
  a.Jc.resize(nc);
  // J and the XC matrix are independent given the compact density: the Coulomb kernels (one of them streams the 1 GB of
  // primitive integrals: HBM-bound) run on the context's side stream beside the XC kernels (LDS- and latency-bound)
  // -- HELFEM_FOCK_OVERLAP=0: one after the other on the main stream
  static const bool overlap = !(getenv("HELFEM_FOCK_OVERLAP") && atoi(getenv("HELFEM_FOCK_OVERLAP")) == 0);
  if ((x_func > 0 || c_func > 0) && overlap && !ctx->avoid_side) {
    hipStream_t main = ctx->stream, q = ctx->side();
    gather_compact(ctx, basis, dP, a.Pc.p);
    HFG_HIP_CHECK(hipEventRecord(ctx->side_ev[0], main));
    HFG_HIP_CHECK(hipStreamWaitEvent(q, ctx->side_ev[0], 0));
    ctx->stream = q;
    try {
      ProfScope ps(ctx, "coulomb");  // (its events go to the side stream with the kernels)
      coulomb_compact(ctx, basis, a.Pc.p, dFc);
    } catch (...) {
      ctx->stream = main;
      throw;
    }
    ctx->stream = main;
    HFG_HIP_CHECK(hipEventRecord(ctx->side_ev[1], q));
    ProfScope ps(ctx, "xc");
    xc_compact(ctx, basis, x_func, c_func, a.Pc.p, a.Jc.p, dScal, thr);
    HFG_HIP_CHECK(hipStreamWaitEvent(main, ctx->side_ev[1], 0));
    hipLaunchKernelGGL(k_add_inplace, dim3(2048), dim3(256), 0, ctx->stream, dFc, a.Jc.p, nc);
    HFG_HIP_CHECK(hipGetLastError());
    return;
  }
  {
    ProfScope ps(ctx, "coulomb");
    gather_compact(ctx, basis, dP, a.Pc.p);
    coulomb_compact(ctx, basis, a.Pc.p, dFc);
  }
  if (x_func > 0 || c_func > 0) {
    ProfScope ps(ctx, "xc");
    xc_compact(ctx, basis, x_func, c_func, a.Pc.p, a.Jc.p, dScal, thr);
    hipLaunchKernelGGL(k_add_inplace, dim3(2048), dim3(256), 0, ctx->stream, dFc, a.Jc.p, nc);
  } else {
    HFG_HIP_CHECK(hipMemsetAsync(dScal, 0, 3 * sizeof(double), ctx->stream));
  }
  (void)t;
  HFG_HIP_CHECK(hipGetLastError());
}

// F = enforce_fock_symmetry(H0 + J + XC) from the (rank-summed) compact matrix
void fock_finish_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dFc, const double *dH0, const int *dBlockId,
                     double *dF) {
  hfg_dev_tables *t = tables_of(ctx, basis);
  FockAux &a = aux_for(ctx, basis);
  ProfScope ps(ctx, "scatter");
  dim3 grid((t->N + 63) / 64, (t->N + 3) / 4);
  hipLaunchKernelGGL(k_fock_finish, grid, dim3(256), 0, ctx->stream, dFc, dH0, dBlockId, t->N, t->A, t->R, t->E, t->p,
                     a.pure_shell.p, a.pure_n.p, dF);
  HFG_HIP_CHECK(hipGetLastError());
}

}  // namespace hfg
