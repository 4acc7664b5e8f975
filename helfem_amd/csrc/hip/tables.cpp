#include "tables.h"
#include <algorithm>
#include <cmath>
#include <functional>
#include <set>

namespace hfg {


namespace {
// What the table builder needs from either basis
struct BasisView {
  int geom = 0;
  int A = 0, R = 0, E = 0, p = 0, nq = 0, N = 0, Nd = 0;
  double Rhalf = 0.0;
  helfem::IVec lval, mval;
  std::vector<int> skip;                          // first radial function of the shell absent
  std::vector<std::pair<int, int> > LM_map, lm_map;  // sorted (L,M) and (L,|M|) lists
  std::vector<int> lm_tab;
  std::vector<double> LM_fac;
  int Ntab = 0, ntt = 4, ndt = 4, Lp1 = 0;
  int rs_kind = 0, pair_tei = 0;
  std::function<const helfem::Mat &(int, int, int)> tei_pair;  // (tab, e, f), pair_tei only
  std::function<void(int, int, int &, int &)> Lrange;  // (x,y) -> Lmin,Lmax
  std::function<double(int, int, int)> c0, c2;         // (x,y,L)
  std::function<const helfem::Mat &(int, int, int)> disj, tei;  // (type, tab, e)
  std::function<int(int)> lo;                          // first primitive held by element e
  std::function<helfem::Mat(int)> bf, df;
  std::function<helfem::Vec(int)> wrad, rcoord;  // rcoord: sinh(mu) (prolate) or r (spherical)

  int LMind(int L, int M) const {
    auto it = std::lower_bound(LM_map.begin(), LM_map.end(), std::make_pair(L, M));
    if (it == LM_map.end() || *it != std::make_pair(L, M)) return -1;
    return (int)(it - LM_map.begin());
  }
  int lmind(int L, int M) const {
    auto it = std::lower_bound(lm_map.begin(), lm_map.end(), std::make_pair(L, std::abs(M)));
    if (it == lm_map.end() || *it != std::make_pair(L, std::abs(M))) return -1;
    return (int)(it - lm_map.begin());
  }
};

BasisView view_of(const hfg_basis *basis, bool rs = false) {
  BasisView v;
  if (rs && (basis->kind == 0 || basis->ab.rs_kind == 0))
    throw std::logic_error("Primitive teis have not been computed!\n");
  if (basis->kind == 0) {
    const helfem::diatomic::TwoDBasis *b = &basis->b;
    v.geom = 0;
    v.A = (int)b->Nang();
    v.R = (int)b->Nrad();
    v.E = (int)b->Nel();
    v.p = (int)b->max_Nprim();
    v.nq = b->nquad();
    v.N = (int)b->Nbf();
    v.Nd = (int)b->Ndummy();
    v.Rhalf = b->Rhalf;
    v.lval = b->lval;
    v.mval = b->mval;
    for (int m : b->mval) v.skip.push_back(m != 0);
    for (auto &x : b->LM_map) v.LM_map.push_back(std::make_pair(x.first, x.second));
    for (auto &x : b->lm_map) v.lm_map.push_back(std::make_pair(x.first, x.second));
    v.Ntab = (int)b->lm_map.size();
    for (int i = 0; i < v.Ntab; i++) v.lm_tab.push_back(i);
    for (auto &x : v.LM_map) v.LM_fac.push_back(b->LMfac(x.first, x.second));
    v.ntt = 4;
    v.ndt = 4;
    v.Lp1 = b->Lmax + 1;
    v.Lrange = [b](int x, int y, int &Lmin, int &Lmax) {
      Lmin = std::max(std::abs(b->lval[x] - b->lval[y]) - 2, std::abs(b->mval[x] - b->mval[y]));
      Lmax = b->lval[x] + b->lval[y] + 2;
    };
    v.c0 = [b](int x, int y, int L) {
      return b->gaunt.mod_coeff(b->lval[x], b->mval[x], L, b->mval[x] - b->mval[y], b->lval[y], b->mval[y]);
    };
    v.c2 = [b](int x, int y, int L) {
      return b->gaunt.coeff(b->lval[x], b->mval[x], L, b->mval[x] - b->mval[y], b->lval[y], b->mval[y]);
    };
    const int E = v.E;
    v.disj = [b, E](int t, int tab, int e) -> const helfem::Mat & {
      const std::vector<helfem::Mat> *src[4] = {&b->disjoint_P0, &b->disjoint_P2, &b->disjoint_Q0, &b->disjoint_Q2};
      return (*src[t])[tab * E + e];
    };
    v.tei = [b, E](int t, int tab, int e) -> const helfem::Mat & {
      const std::vector<helfem::Mat> *src[4] = {&b->prim_tei00, &b->prim_tei02, &b->prim_tei20, &b->prim_tei22};
      return (*src[t])[tab * E + e];
    };
    v.lo = [](int) { return 0; };
    v.bf = [b](int e) { return b->get_bf(e); };
    v.df = [b](int e) { return b->get_df(e); };
    v.wrad = [b](int e) { return b->get_wrad(e); };
    v.rcoord = [b](int e) {
      helfem::Vec r = b->get_r(e);
      for (auto &x : r) x = std::sinh(x);
      return r;
    };
  } else {
    const helfem::atomic::TwoDBasis *b = &basis->ab;
    v.geom = 1;
    v.A = (int)b->Nang();
    v.E = (int)b->Nel();
    v.p = (int)b->max_Nprim();
    v.R = v.E * (v.p - 1);  // "dummy" radial count: function 0 (dropped at the nucleus) + the Nrad real ones
    if ((int)b->Nrad() != v.R - 1) throw std::logic_error("unexpected atomic radial basis size");
    v.nq = b->nquad();
    v.N = (int)b->Nbf();
    v.Nd = v.A * v.R;
    v.Rhalf = 1.0;
    v.lval = b->lval;
    v.mval = b->mval;
    v.skip.assign(v.A, 1);
    const int NL = b->N_L(), Mmax = b->Mmax();
    for (int L = 0; L < NL; L++) {
      for (int M = -std::min(L, Mmax); M <= std::min(L, Mmax); M++) {
        v.LM_map.push_back(std::make_pair(L, M));
        v.LM_fac.push_back(4.0 * M_PI / (2 * L + 1));
      }
      for (int M = 0; M <= std::min(L, Mmax); M++) {
        v.lm_map.push_back(std::make_pair(L, M));
        v.lm_tab.push_back(L);
      }
    }
    v.Ntab = NL;
    v.ntt = 1;
    v.ndt = 2;
    v.Lp1 = NL;
    v.Lrange = [b](int x, int y, int &Lmin, int &Lmax) {
      Lmin = std::max(std::abs(b->lval[x] - b->lval[y]), std::abs(b->mval[x] - b->mval[y]));
      Lmax = b->lval[x] + b->lval[y];
    };
    v.c0 = [b](int x, int y, int L) {
      return b->gaunt.coeff(b->lval[x], b->mval[x], L, b->mval[x] - b->mval[y], b->lval[y], b->mval[y]);
    };
    v.c2 = [](int, int, int) { return 0.0; };
    const int E = v.E;
    v.disj = [b, E](int t, int tab, int e) -> const helfem::Mat & {
      return (t == 0 ? b->disjoint_L : b->disjoint_m1L)[tab * E + e];
    };
    v.tei = [b, E](int, int tab, int e) -> const helfem::Mat & { return b->prim_tei[tab * E + e]; };
    v.lo = [](int e) { return e == 0 ? 1 : 0; };
    v.bf = [b](int e) { return b->get_bf(e); };
    v.df = [b](int e) { return b->get_df(e); };
    v.wrad = [b](int e) { return b->get_wrad(e); };
    v.rcoord = [b](int e) { return b->get_r(e); };
    if (rs) {
      // TwoDBasis.cpp:1240: Lfac = 4 pi lambda (Yukawa) or 4 pi mu/(2L+1) (erfc)
      v.rs_kind = b->rs_kind;
      const double lam = b->rs_lambda;
      for (size_t i = 0; i < v.LM_map.size(); i++)
        v.LM_fac[i] = (b->rs_kind == 1) ? 4.0 * M_PI * lam : 4.0 * M_PI * lam / (2 * v.LM_map[i].first + 1);
      if (b->rs_kind == 1) {
        v.disj = [b, E](int t, int tab, int e) -> const helfem::Mat & {
          return (t == 0 ? b->disjoint_iL : b->disjoint_kL)[tab * E + e];
        };
        v.tei = [b, E](int, int tab, int e) -> const helfem::Mat & { return b->rs_tei[tab * E + e]; };
      } else {
        v.pair_tei = 1;
        v.ndt = 0;
        v.tei_pair = [b, E](int tab, int e, int f) -> const helfem::Mat & { return b->rs_tei[((size_t)tab * E + e) * E + f]; };
      }
    }
  }
  return v;
}
}  // namespace

static void fill_tables(hfg_ctx *ctx, hfg_basis *basis, const BasisView &b, hfg_dev_tables *t, int ldft, int mdft);

void upload_tables(hfg_ctx *ctx, hfg_basis *basis, int ldft, int mdft) {
  if (!basis->have_tei()) throw std::logic_error("Primitive teis have not been computed!\n");
  const BasisView b = view_of(basis);
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  if (basis->dev) delete basis->dev;
  hfg_dev_tables *t = new hfg_dev_tables();
  basis->dev = t;
  basis->dev_device = ctx->device;
  fill_tables(ctx, basis, b, t, ldft, mdft);
}

void upload_rs_tables(hfg_ctx *ctx, hfg_basis *basis) {
  const BasisView b = view_of(basis, true);
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  if (basis->dev_rs) delete basis->dev_rs;
  hfg_dev_tables *t = new hfg_dev_tables();
  basis->dev_rs = t;
  if (basis->dev && basis->dev_device != ctx->device) throw std::logic_error("basis tables live on a different device\n");
  basis->dev_device = ctx->device;
  fill_tables(ctx, basis, b, t, 0, 0);
}

static void fill_tables(hfg_ctx *ctx, hfg_basis *basis, const BasisView &b, hfg_dev_tables *t, int ldft, int mdft) {
  hipStream_t s = ctx->stream;
  const bool use_dev_tei = basis->tei_on_device && !b.rs_kind;
  t->rs_kind = b.rs_kind;
  t->pair_tei = b.pair_tei;

  const int A = t->A = b.A;
  const int R = t->R = b.R;
  const int E = t->E = b.E;
  const int p = t->p = b.p;
  t->nq = b.nq;
  t->N = b.N;
  t->Nd = b.Nd;
  const int NLM = t->NLM = (int)b.LM_map.size();
  const int Nlm = t->Nlm = (int)b.lm_map.size();
  t->Rhalf = b.Rhalf;
  t->geom = b.geom;
  t->Ntab = b.Ntab;
  t->ntt = b.ntt;
  t->ndt = b.ndt;
  t->dQ0 = (b.ndt == 4) ? 2 : 1;
  t->Lp1 = b.Lp1;
  if (R != E * (p - 1)) throw std::logic_error("unexpected radial basis size");

  // ---- shells ----
  std::vector<int> sl(A), sm(A), soff(A), sskip(A);
  {
    int off = 0;
    for (int a = 0; a < A; a++) {
      sl[a] = b.lval[a];
      sm[a] = b.mval[a];
      sskip[a] = b.skip[a];
      soff[a] = off - sskip[a];  // pure(a,n) = soff[a] + n
      off += R - sskip[a];
    }
    if (off != t->N) throw std::logic_error("basis size mismatch");
  }
  t->h_shell_l = sl;
  t->h_shell_m = sm;
  t->h_shell_skip = sskip;
  t->shell_l.upload(sl, s);
  t->shell_m.upload(sm, s);
  t->shell_off.upload(soff, s);
  t->shell_skip.upload(sskip, s);

  // ---- Gaunt couplings ----
  std::vector<int> pair_off(A * A + 1, 0), ent_iLM;
  std::vector<double> ent_c0, ent_c2;
  std::vector<std::vector<int> > byLMx(NLM), byLMy(NLM);
  std::vector<std::vector<double> > byLMc0(NLM), byLMc2(NLM);
  t->h_c0tab.assign((size_t)A * A * b.Lp1, 0.0);
  t->h_c2tab.assign((size_t)A * A * b.Lp1, 0.0);
  for (int x = 0; x < A; x++)
    for (int y = 0; y < A; y++) {
      int M = b.mval[x] - b.mval[y];
      int Lmin, Lmax;
      b.Lrange(x, y, Lmin, Lmax);
      pair_off[x * A + y] = (int)ent_iLM.size();
      for (int L = Lmin; L <= Lmax; L++) {
        double c0 = b.c0(x, y, L);
        double c2 = b.c2(x, y, L);
        if (c0 == 0.0 && c2 == 0.0) continue;
        int iLM = b.LMind(L, M);
        if (iLM < 0 || L >= b.Lp1) throw std::logic_error("coupling channel outside the (L,M) table");
        t->h_c0tab[((size_t)x * A + y) * b.Lp1 + L] = c0;
        t->h_c2tab[((size_t)x * A + y) * b.Lp1 + L] = c2;
        ent_iLM.push_back(iLM);
        ent_c0.push_back(c0);
        ent_c2.push_back(c2);
        byLMx[iLM].push_back(x);
        byLMy[iLM].push_back(y);
        byLMc0[iLM].push_back(c0);
        byLMc2[iLM].push_back(c2);
      }
    }
  pair_off[A * A] = (int)ent_iLM.size();
  t->T = (int)ent_iLM.size();
  t->pair_off.upload(pair_off, s);
  t->ent_iLM.upload(ent_iLM, s);
  t->ent_c0.upload(ent_c0, s);
  t->ent_c2.upload(ent_c2, s);
  std::vector<int> lm_off(NLM + 1, 0), lm_x, lm_y;
  std::vector<double> lm_c0, lm_c2;
  for (int i = 0; i < NLM; i++) {
    lm_off[i] = (int)lm_x.size();
    lm_x.insert(lm_x.end(), byLMx[i].begin(), byLMx[i].end());
    lm_y.insert(lm_y.end(), byLMy[i].begin(), byLMy[i].end());
    lm_c0.insert(lm_c0.end(), byLMc0[i].begin(), byLMc0[i].end());
    lm_c2.insert(lm_c2.end(), byLMc2[i].begin(), byLMc2[i].end());
  }
  lm_off[NLM] = (int)lm_x.size();
  t->lm_off.upload(lm_off, s);
  t->lm_x.upload(lm_x, s);
  t->lm_y.upload(lm_y, s);
  t->lm_c0.upload(lm_c0, s);
  t->lm_c2.upload(lm_c2, s);

  std::vector<int> LM_ilm(NLM), LM_partner(NLM, -1), LM_tab(NLM);
  t->h_LM_L.resize(NLM);
  t->h_LM_M.resize(NLM);
  for (int i = 0; i < NLM; i++) {
    int L = b.LM_map[i].first, M = b.LM_map[i].second;
    LM_ilm[i] = b.lmind(L, M);
    if (LM_ilm[i] < 0) throw std::logic_error("(L,|M|) table incomplete");
    LM_tab[i] = b.lm_tab[LM_ilm[i]];
    t->h_LM_L[i] = L;
    t->h_LM_M[i] = M;
    LM_partner[i] = b.LMind(L, -M);
  }
  t->h_LM_ilm = LM_ilm;
  t->h_LM_fac = b.LM_fac;
  t->h_lm_tab = b.lm_tab;
  t->LM_ilm.upload(LM_ilm, s);
  t->LM_partner.upload(LM_partner, s);
  t->LM_fac.upload(b.LM_fac, s);
  t->lm_tab.upload(b.lm_tab, s);
  t->LM_tab.upload(LM_tab, s);

  // ---- primitive integrals, zero padded to p (element e holds primitives lo(e)..lo(e)+n-1) ----
  const int Ntab = b.Ntab;
  {
    const size_t pp = (size_t)p * p;
    std::vector<double> disj(std::max<size_t>((size_t)b.ndt * Ntab * E * pp, 1), 0.0);
    for (int tt = 0; tt < b.ndt; tt++)
      for (int tab = 0; tab < Ntab; tab++)
        for (int e = 0; e < E; e++) {
          const helfem::Mat &m = b.disj(tt, tab, e);
          const int lo = b.lo(e);
          if ((int)m.n_rows + lo > p) throw std::logic_error("primitive block larger than the padded size");
          double *dst = &disj[(((size_t)tt * Ntab + tab) * E + e) * pp];
          for (size_t j = 0; j < m.n_cols; j++)
            for (size_t i = 0; i < m.n_rows; i++) {
              double val = m(i, j);
              dst[(j + lo) * p + (i + lo)] = std::isfinite(val) ? val : 0.0;
            }
        }
    t->disj.upload(disj, s);
    HFG_HIP_CHECK(hipStreamSynchronize(s));
  }
  {
    const size_t pp = (size_t)p * p;
    const size_t blk = pp * pp;
    const int nper = b.pair_tei ? E * E : E;  // blocks per (type, slot)
    t->tei.resize((size_t)b.ntt * Ntab * nper * blk);
    if (use_dev_tei) {
      if (basis->dev_tei.n < (size_t)b.ntt * Ntab * E * blk) throw std::logic_error("device tei buffer has the wrong size");
      HFG_HIP_CHECK(hipMemcpyAsync(t->tei.p, basis->dev_tei.p, sizeof(double) * (size_t)b.ntt * Ntab * E * blk,
                                   hipMemcpyDeviceToDevice, s));
      HFG_HIP_CHECK(hipStreamSynchronize(s));
    }
    std::vector<double> stage(blk);
    if (!use_dev_tei)
    for (int tt = 0; tt < b.ntt; tt++)
      for (int tab = 0; tab < Ntab; tab++)
        for (int eb = 0; eb < nper; eb++) {
          const int e = b.pair_tei ? eb / E : eb, f = b.pair_tei ? eb % E : eb;
          const helfem::Mat &m = b.pair_tei ? b.tei_pair(tab, e, f) : b.tei(tt, tab, e);
          const int lo = b.lo(e), lof = b.lo(f);  // rows: primitives of e, columns: primitives of f
          const size_t Ni = (size_t)std::lround(std::sqrt((double)m.n_rows)), Nf = (size_t)std::lround(std::sqrt((double)m.n_cols));
          if (Ni * Ni != m.n_rows || Nf * Nf != m.n_cols || (int)Ni + lo > p || (int)Nf + lof > p)
            throw std::logic_error("unexpected primitive tei block size");
          double *dst = t->tei.p + (((size_t)tt * Ntab + tab) * nper + eb) * blk;
          if ((int)Ni == p && (int)Nf == p) {
            HFG_HIP_CHECK(hipMemcpyAsync(dst, m.memptr(), blk * sizeof(double), hipMemcpyHostToDevice, s));
          } else {
            std::fill(stage.begin(), stage.end(), 0.0);
            for (size_t cj = 0; cj < Nf; cj++)
              for (size_t ci = 0; ci < Nf; ci++)
                for (size_t rj = 0; rj < Ni; rj++)
                  for (size_t ri = 0; ri < Ni; ri++)
                    stage[((cj + lof) * p + (ci + lof)) * pp + (rj + lo) * p + (ri + lo)] = m(rj * Ni + ri, cj * Nf + ci);
            HFG_HIP_CHECK(hipMemcpyAsync(dst, stage.data(), blk * sizeof(double), hipMemcpyHostToDevice, s));
            HFG_HIP_CHECK(hipStreamSynchronize(s));
          }
        }
    HFG_HIP_CHECK(hipStreamSynchronize(s));
  }
  t->have_tei = true;

  // ---- XC grid tables ----
  if (ldft > 0 && mdft > 0) {
    const int nq = t->nq;
    t->ntheta = ldft;
    t->nphi = mdft;
    std::vector<double> B((size_t)E * nq * p, 0.0), dB((size_t)E * nq * p, 0.0), rw((size_t)E * nq), rsh((size_t)E * nq);
    for (int e = 0; e < E; e++) {
      helfem::Mat bf = b.bf(e), df = b.df(e);
      helfem::Vec w = b.wrad(e), r = b.rcoord(e);
      const int lo = b.lo(e);
      for (int q = 0; q < nq; q++) {
        for (size_t i = 0; i < bf.n_cols; i++) {
          B[((size_t)e * nq + q) * p + i + lo] = bf(q, i);
          dB[((size_t)e * nq + q) * p + i + lo] = df(q, i);
        }
        rw[(size_t)e * nq + q] = w[q];
        rsh[(size_t)e * nq + q] = r[q];
      }
    }
    t->rad_B.upload(B, s);
    t->rad_dB.upload(dB, s);
    t->rad_w.upload(rw, s);
    t->rad_sh.upload(rsh, s);

    helfem::Vec xc, wc;
    helfem::chebyshev_rule(ldft, xc, wc);
    std::vector<double> ths(ldft), Th((size_t)A * ldft), dTh((size_t)A * ldft);
    for (int i = 0; i < ldft; i++) ths[i] = sqrt(1.0 - xc[i] * xc[i]);
    for (int a = 0; a < A; a++)
      for (int i = 0; i < ldft; i++) {
        Th[(size_t)a * ldft + i] = helfem::theta_lm(b.lval[a], b.mval[a], xc[i]);
        dTh[(size_t)a * ldft + i] = helfem::dtheta_lm(b.lval[a], b.mval[a], xc[i]);
      }
    t->th_c.upload(xc, s);
    t->th_s.upload(ths, s);
    t->th_w.upload(wc, s);
    t->Th.upload(Th, s);
    t->dTh.upload(dTh, s);

    // m groups
    std::set<int> mset(b.mval.begin(), b.mval.end());
    std::vector<int> gm(mset.begin(), mset.end());
    t->G = (int)gm.size();
    std::vector<int> goff(t->G + 1, 0), gshell, sgrp(A);
    for (int g = 0; g < t->G; g++) {
      goff[g] = (int)gshell.size();
      for (int a = 0; a < A; a++)
        if (b.mval[a] == gm[g]) {
          gshell.push_back(a);
          sgrp[a] = g;
        }
    }
    goff[t->G] = (int)gshell.size();
    t->h_grp_off = goff;
    t->grp_m.upload(gm, s);
    t->grp_off.upload(goff, s);
    t->grp_shell.upload(gshell, s);
    t->shell_grp.upload(sgrp, s);
    int Dmax = gm.back() - gm.front();
    t->Dmax = Dmax;
    std::vector<double> cosd((size_t)(2 * Dmax + 1) * mdft), sind((size_t)(2 * Dmax + 1) * mdft);
    double dphi = 2.0 * M_PI / mdft;
    for (int D = -Dmax; D <= Dmax; D++)
      for (int j = 0; j < mdft; j++) {
        cosd[(size_t)(D + Dmax) * mdft + j] = cos(D * (j * dphi));
        sind[(size_t)(D + Dmax) * mdft + j] = sin(D * (j * dphi));
      }
    t->cosd.upload(cosd, s);
    t->sind.upload(sind, s);
    t->have_xc = true;
  }
  HFG_HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace hfg
