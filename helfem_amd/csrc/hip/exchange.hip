// Exact-exchange matrix for the diatomic basis (gfx950).
// Replaces TwoDBasis::exchange, /root/reference/src/diatomic/basis.cpp:1532-1733.
//
// K(j n, k n') = - sum_{L,M} LMfac sum_{i,l} cpl(j,i;k,l;L,M) * radial[ P_il ]        (basis.cpp:1601-1727)
//
// The reference loops over output shell pairs (j,k) and, for each, over all density blocks (i,l)
// with m_j-m_i = m_k-m_l (A^4 work).  Here the angular sum is split in two A^3 steps,
//     U_t[c][l]  = sum_i  c_t(j,i,L) P_il            (c = coupling channel (L,M) of shell j)
//     R_tt'[k][ilm] = +-LMfac sum_l c_t'(k,l,L) U_t[c][l],
// shell j being a host loop so that the intermediates stay O(NLM * A * R^2).  The radial stage
// follows the reference: in-element blocks contract the primitive integrals (index-permuted reads
// of prim_tei replace the reference's prim_ktei copies), cross-element blocks use the factorised
// disjoint P/Q integrals.
#include "tables.h"

namespace hfg {

// Pd (Ndummy x Ndummy) <- expand_boundaries(P)   (basis.cpp:1754)
__global__ void k_expand(const double *__restrict__ P, int N, int Nd, int R, const int *__restrict__ shell_off,
                         const int *__restrict__ shell_skip, double *__restrict__ Pd) {
  int row = blockIdx.x * blockDim.x + threadIdx.x;
  int col = blockIdx.y;
  if (row >= Nd) return;
  int x = row / R, n = row % R, y = col / R, m = col % R;
  double v = 0.0;
  if (!(shell_skip[x] && n == 0) && !(shell_skip[y] && m == 0))
    v = P[(size_t)(shell_off[y] + m) * N + shell_off[x] + n];
  Pd[(size_t)col * Nd + row] = v;
}

// K (N x N) <- remove_boundaries(Kd)   (basis.cpp:1735)
__global__ void k_remove(const double *__restrict__ Kd, int N, int Nd, const int *__restrict__ pure_idx,
                         double *__restrict__ K) {
  int row = blockIdx.x * blockDim.x + threadIdx.x;
  int col = blockIdx.y;
  if (row >= N) return;
  K[(size_t)col * N + row] = Kd[(size_t)pure_idx[col] * Nd + pure_idx[row]];
}

// EXa: U_t[c][l][n'][n] = sum_{i: m_i = m_j - M_c} c_t(j,i,L_c) Pd[(i,n),(l,n')]
__global__ void k_ex_U(const double *__restrict__ Pd, int Nd, int R, int A, int j, const int *__restrict__ chanL,
                       const int *__restrict__ chanM, const int *__restrict__ shell_m,
                       const double *__restrict__ c0tab, const double *__restrict__ c2tab, int Lp1,
                       double *__restrict__ U /* [2][nchan][A][R*R] */, int nchan) {
  int c = blockIdx.x, l = blockIdx.y;
  int L = chanL[c], M = chanM[c];
  int mi_need = shell_m[j] - M;
  size_t RR = (size_t)R * R;
  for (int t = threadIdx.x; t < (int)RR; t += blockDim.x) {
    int n = t % R, np = t / R;
    double u0 = 0.0, u2 = 0.0;
    for (int i = 0; i < A; i++) {
      if (shell_m[i] != mi_need) continue;
      double a0 = c0tab[((size_t)j * A + i) * Lp1 + L], a2 = c2tab[((size_t)j * A + i) * Lp1 + L];
      if (a0 == 0.0 && a2 == 0.0) continue;
      double pv = Pd[(size_t)(l * R + np) * Nd + i * R + n];
      u0 += a0 * pv;
      u2 += a2 * pv;
    }
    U[((size_t)(0 * nchan + c) * A + l) * RR + t] = u0;
    U[((size_t)(1 * nchan + c) * A + l) * RR + t] = u2;
  }
}

// EXb: Rm[k][ilm][tt][n'][n], tt = 00,02,20,22:
//   R00 = LMfac sum_l c0(k,l) U0 ; R02 = -LMfac sum_l c2(k,l) U0 ; R20 = -LMfac sum_l c0(k,l) U2 ; R22 = LMfac sum_l c2(k,l) U2
//   summed over the channels c of shell j that map onto ilm (M and -M)
__global__ void k_ex_R(const double *__restrict__ U, int R, int A, int nchan, const int *__restrict__ chanL,
                       const int *__restrict__ chanM, const int *__restrict__ chan_ilm,
                       const double *__restrict__ chan_fac, const int *__restrict__ shell_m,
                       const double *__restrict__ c0tab, const double *__restrict__ c2tab, int Lp1, int Nlm, int ntt,
                       double *__restrict__ Rm, int *__restrict__ couple /* [A][Nlm] */) {
  // "ilm" here and below is the primitive-table slot ((L,|M|) for the diatomic tables, L for the atomic ones)
  int k = blockIdx.x, ilm = blockIdx.y;
  size_t RR = (size_t)R * R;
  int any = 0;
  for (int t = threadIdx.x; t < (int)RR; t += blockDim.x) {
    double r00 = 0.0, r02 = 0.0, r20 = 0.0, r22 = 0.0;
    for (int c = 0; c < nchan; c++) {
      if (chan_ilm[c] != ilm) continue;
      int L = chanL[c], M = chanM[c];
      int ml_need = shell_m[k] - M;
      double fac = chan_fac[c];
      for (int l = 0; l < A; l++) {
        if (shell_m[l] != ml_need) continue;
        double b0 = c0tab[((size_t)k * A + l) * Lp1 + L], b2 = c2tab[((size_t)k * A + l) * Lp1 + L];
        if (b0 == 0.0 && b2 == 0.0) continue;
        any = 1;
        double u0 = U[((size_t)(0 * nchan + c) * A + l) * RR + t];
        double u2 = U[((size_t)(1 * nchan + c) * A + l) * RR + t];
        r00 += fac * b0 * u0;
        r02 -= fac * b2 * u0;
        r20 -= fac * b0 * u2;
        r22 += fac * b2 * u2;
      }
    }
    double *o = Rm + (((size_t)k * Nlm + ilm) * ntt) * RR + t;
    o[0] = r00;
    if (ntt == 4) {
      o[RR] = r02;
      o[2 * RR] = r20;
      o[3 * RR] = r22;
    }
  }
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) couple[k * Nlm + ilm] = any;
}

// EXc/EXd: radial stage for output block (j,k), element pair (iel,jel)
__global__ void k_ex_radial(const double *__restrict__ Rm, const int *__restrict__ couple,
                            const double *__restrict__ tei, const double *__restrict__ disj, int R, int E, int p,
                            int Nlm, int ntt, int pair_tei, double *__restrict__ Kc /* [A][E][E][p*p] */) {
  extern __shared__ double sh[];  // T[p*p]
  int k = blockIdx.x;
  int iel = blockIdx.y / E, jel = blockIdx.y % E;
  int pp = p * p;
  size_t RR = (size_t)R * R;
  int ifirst = iel * (p - 1), jfirst = jel * (p - 1);
  int t = threadIdx.x;
  int a = t % p, b = t / p;  // output (row a in iel, col b in jel)
  bool active = t < pp;
  int ga = ifirst + a, gb = jfirst + b;
  bool inrange = active && ga < R && gb < R;
  double acc = 0.0;
  for (int ilm = 0; ilm < Nlm; ilm++) {
    if (!couple[k * Nlm + ilm]) continue;
    const double *R00 = Rm + (((size_t)k * Nlm + ilm) * ntt) * RR;
    const double *R02 = R00 + RR, *R20 = R00 + 2 * RR, *R22 = R00 + 3 * RR;  // only read when ntt == 4
    if (iel == jel || pair_tei) {
      // Ksub(a,b) = sum_{i',l'} tei[(i' a),(b l')] R(i',l') :  ktei(b*p+a, l'*p+i') = tei(a*p+i', l'*p+b)
      // pair_tei (erfc kernel, TwoDBasis.cpp:1262): every element pair has its own block, rows in iel, columns in jel
      if (inrange) {
        double s = 0.0;
        for (int tt = 0; tt < ntt; tt++) {
          const double *T = pair_tei ? tei + ((((size_t)tt * Nlm + ilm) * E + iel) * E + jel) * (size_t)pp * pp
                                     : tei + (((size_t)tt * Nlm + ilm) * E + iel) * (size_t)pp * pp;
          const double *Rt = R00 + (size_t)tt * RR;
          for (int lp = 0; lp < p; lp++) {
            int gl = jfirst + lp;
            if (gl >= R) continue;
            const double *Tc = T + (size_t)(lp * p + b) * pp + a * p;
            for (int ip = 0; ip < p; ip++) {
              int gi = ifirst + ip;
              if (gi >= R) continue;
              s += Tc[ip] * Rt[(size_t)gl * R + gi];
            }
          }
        }
        acc -= s;
      }
    } else {
      // disjoint integrals: the outer element gets Q, the inner one P  (basis.cpp:1709-1713)
      const bool full = (ntt == 4);
      const int tQ0 = full ? 2 : 1;  // disj types: 0=P0 1=P2 2=Q0 3=Q2 (diatomic); 0=P0 1=Q0 (atomic)
      int it0 = (iel > jel) ? tQ0 : 0, it2 = (iel > jel) ? 3 : 1;
      int jt0 = (iel > jel) ? 0 : tQ0, jt2 = (iel > jel) ? 1 : 3;
      const double *ii0 = disj + (((size_t)it0 * Nlm + ilm) * E + iel) * pp;
      const double *ii2 = full ? disj + (((size_t)it2 * Nlm + ilm) * E + iel) * pp : ii0;
      const double *jj0 = disj + (((size_t)jt0 * Nlm + ilm) * E + jel) * pp;
      const double *jj2 = full ? disj + (((size_t)jt2 * Nlm + ilm) * E + jel) * pp : jj0;
      for (int pass = 0; pass < (full ? 2 : 1); pass++) {
        const double *Ra = pass ? R20 : R00, *Rb = pass ? R22 : R02;
        const double *ii = pass ? ii2 : ii0;
        // T(a,b) = sum_c Ra(a,c) jj0(b,c) + Rb(a,c) jj2(b,c)
        double tv = 0.0;
        if (active && ga < R) {
          for (int c = 0; c < p; c++) {
            int gc = jfirst + c;
            if (gc >= R) continue;
            tv += Ra[(size_t)gc * R + ga] * jj0[c * p + b];
            if (full) tv += Rb[(size_t)gc * R + ga] * jj2[c * p + b];
          }
        }
        __syncthreads();
        if (active) sh[b * p + a] = tv;
        __syncthreads();
        if (inrange) {
          double s = 0.0;
          for (int c = 0; c < p; c++) s += ii[c * p + a] * sh[b * p + c];
          acc -= s;
        }
      }
    }
  }
  // one writer per entry of the compact block; shared boundary functions are summed in k_ex_assemble
  if (active) Kc[((size_t)(k * E + iel) * E + jel) * pp + b * p + a] = inrange ? acc : 0.0;
}

// Kd[(j,n),(k,n')] = sum of the (one to four) element-pair blocks that contain (n,n')
__global__ void k_ex_assemble(const double *__restrict__ Kc, int R, int E, int p, int Nd, int j,
                              double *__restrict__ Kd) {
  int k = blockIdx.x, np = blockIdx.y;
  int pm = p - 1, pp = p * p;
  for (int n = threadIdx.x; n < R; n += blockDim.x) {
    double v = 0.0;
    for (int ce = 0; ce < 2; ce++) {
      int e = n / pm - ce;
      if (e < 0 || e >= E) continue;
      int a = n - e * pm;
      if (a < 0 || a > pm) continue;
      for (int cf = 0; cf < 2; cf++) {
        int f = np / pm - cf;
        if (f < 0 || f >= E) continue;
        int b = np - f * pm;
        if (b < 0 || b > pm) continue;
        v += Kc[((size_t)(k * E + e) * E + f) * pp + b * p + a];
      }
    }
    Kd[(size_t)(k * R + np) * Nd + (size_t)j * R + n] = v;
  }
}

struct ExAux {
  DevBuf<double> c0tab, c2tab, Pd, Kd, Kc, U, Rm, chan_fac;
  DevBuf<int> chanL, chanM, chan_ilm, couple, pure_idx;
  std::vector<std::vector<int> > hL, hM, hilm;
  std::vector<std::vector<double> > hfac;
  int Lp1 = 0;
};
static std::map<hfg_dev_tables *, ExAux *> g_ex;

void exchange_release(hfg_dev_tables *t) {
  auto it = g_ex.find(t);
  if (it != g_ex.end()) {
    delete it->second;
    g_ex.erase(it);
  }
}

static ExAux &exaux_for(hfg_ctx *ctx, hfg_dev_tables *t) {
  auto it = g_ex.find(t);
  if (it != g_ex.end()) return *it->second;
  ExAux *a = new ExAux();
  const int A = t->A;
  a->Lp1 = t->Lp1;
  a->hL.resize(A);
  a->hM.resize(A);
  a->hilm.resize(A);
  a->hfac.resize(A);
  // coupling channels (L,M) of every shell x: those with a non-zero coefficient against some shell y
  std::map<std::pair<int, int>, int> LMpos;
  for (int i = 0; i < t->NLM; i++) LMpos[std::make_pair(t->h_LM_L[i], t->h_LM_M[i])] = i;
  for (int x = 0; x < A; x++) {
    std::vector<std::pair<int, int> > chans;
    for (int y = 0; y < A; y++) {
      int M = t->h_shell_m[x] - t->h_shell_m[y];
      for (int L = 0; L < a->Lp1; L++)
        if (t->h_c0tab[((size_t)x * A + y) * a->Lp1 + L] != 0.0 || t->h_c2tab[((size_t)x * A + y) * a->Lp1 + L] != 0.0)
          chans.push_back(std::make_pair(L, M));
    }
    std::sort(chans.begin(), chans.end());
    chans.erase(std::unique(chans.begin(), chans.end()), chans.end());
    for (auto &c : chans) {
      int iLM = LMpos.at(c);
      a->hL[x].push_back(c.first);
      a->hM[x].push_back(c.second);
      a->hilm[x].push_back(t->h_lm_tab[t->h_LM_ilm[iLM]]);
      a->hfac[x].push_back(t->h_LM_fac[iLM]);
    }
  }
  a->c0tab.upload(t->h_c0tab, ctx->stream);
  a->c2tab.upload(t->h_c2tab, ctx->stream);
  // positions of the real functions inside the (shell, dummy radial index) numbering
  std::vector<int> pidx;
  for (int s = 0; s < A; s++)
    for (int n = (t->h_shell_skip[s] ? 1 : 0); n < t->R; n++) pidx.push_back(s * t->R + n);
  if ((int)pidx.size() != t->N) throw std::logic_error("pure index list does not match the basis size");
  a->pure_idx.upload(pidx, ctx->stream);
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  g_ex[t] = a;
  return *a;
}

bool exchange_lowrank_dev(hfg_ctx *ctx, hfg_dev_tables *t, const double *dP, double *dK, const double *Lknown, int rknown);

// rs: the range-separated kernel of TwoDBasis::rs_exchange (src/atomic/TwoDBasis.cpp:1142) through basis->dev_rs
void exchange_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dK, bool rs, const double *Lknown, int rknown) {
  hfg_dev_tables *t = rs ? basis->dev_rs : basis->dev;
  if (!t || !t->have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
  if (basis->dev_device != ctx->device) throw std::logic_error("basis tables live on a different device\n");
  ProfScope ps(ctx, "exchange");
  // fast path for the low-rank densities of SCF runs (exchange_lr.hip); HELFEM_EXCHANGE=general forces the
  // general kernels below, which take any symmetric P
  {
    const char *mode = getenv("HELFEM_EXCHANGE");
    if (!(mode && std::string(mode) == "general") && exchange_lowrank_dev(ctx, t, dP, dK, Lknown, rknown)) return;
  }
  ExAux &a = exaux_for(ctx, t);
  hipStream_t s = ctx->stream;
  const int A = t->A, R = t->R, E = t->E, p = t->p, Nd = t->Nd, N = t->N, Nlm = t->Ntab, ntt = t->ntt;
  const size_t RR = (size_t)R * R;
  a.Pd.resize((size_t)Nd * Nd);
  a.Kd.resize((size_t)Nd * Nd);
  a.Rm.resize((size_t)A * Nlm * ntt * RR);
  a.couple.resize((size_t)A * Nlm);
  a.Kc.resize((size_t)A * E * E * p * p);
  hipLaunchKernelGGL(k_expand, dim3((Nd + 255) / 256, Nd), dim3(256), 0, s, dP, N, Nd, R, t->shell_off.p,
                     t->shell_skip.p, a.Pd.p);
  HFG_HIP_CHECK(hipMemsetAsync(a.Kd.p, 0, sizeof(double) * (size_t)Nd * Nd, s));
  for (int j = 0; j < A; j++) {
    if ((j % ctx->shard_n) != ctx->shard_rank) continue;  // output row-blocks are sharded over ranks
    int nchan = (int)a.hL[j].size();
    if (!nchan) continue;
    a.chanL.upload(a.hL[j], s);
    a.chanM.upload(a.hM[j], s);
    a.chan_ilm.upload(a.hilm[j], s);
    a.chan_fac.upload(a.hfac[j], s);
    a.U.resize((size_t)2 * nchan * A * RR);
    hipLaunchKernelGGL(k_ex_U, dim3(nchan, A), dim3(256), 0, s, a.Pd.p, Nd, R, A, j, a.chanL.p, a.chanM.p,
                       t->shell_m.p, a.c0tab.p, a.c2tab.p, a.Lp1, a.U.p, nchan);
    hipLaunchKernelGGL(k_ex_R, dim3(A, Nlm), dim3(256), 0, s, a.U.p, R, A, nchan, a.chanL.p, a.chanM.p, a.chan_ilm.p,
                       a.chan_fac.p, t->shell_m.p, a.c0tab.p, a.c2tab.p, a.Lp1, Nlm, ntt, a.Rm.p, a.couple.p);
    int bs = std::max(64, ((p * p + 63) / 64) * 64);
    hipLaunchKernelGGL(k_ex_radial, dim3(A, E * E), dim3(bs), p * p * sizeof(double), s, a.Rm.p, a.couple.p, t->tei.p,
                       t->disj.p, R, E, p, Nlm, ntt, t->pair_tei, a.Kc.p);
    hipLaunchKernelGGL(k_ex_assemble, dim3(A, R), dim3(128), 0, s, a.Kc.p, R, E, p, Nd, j, a.Kd.p);
  }
  hipLaunchKernelGGL(k_remove, dim3((N + 255) / 256, N), dim3(256), 0, s, a.Kd.p, N, Nd, a.pure_idx.p, dK);
  HFG_HIP_CHECK(hipGetLastError());
}

}  // namespace hfg
