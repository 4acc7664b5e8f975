#include "tables.h"
#include <algorithm>
#include <cmath>
#include <set>

namespace hfg {

using helfem::diatomic::TwoDBasis;

void upload_tables(hfg_ctx *ctx, hfg_basis *basis, int ldft, int mdft) {
  const TwoDBasis &b = basis->b;
  if (!b.have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  if (basis->dev) delete basis->dev;
  hfg_dev_tables *t = new hfg_dev_tables();
  basis->dev = t;
  basis->dev_device = ctx->device;
  hipStream_t s = ctx->stream;

  const int A = t->A = (int)b.Nang();
  const int R = t->R = (int)b.Nrad();
  const int E = t->E = (int)b.Nel();
  const int p = t->p = (int)b.max_Nprim();
  t->nq = b.nquad();
  t->N = (int)b.Nbf();
  t->Nd = (int)b.Ndummy();
  const int NLM = t->NLM = (int)b.LM_map.size();
  const int Nlm = t->Nlm = (int)b.lm_map.size();
  t->Rhalf = b.Rhalf;
  if (R != E * (p - 1)) throw std::logic_error("unexpected radial basis size");

  // ---- shells ----
  std::vector<int> sl(A), sm(A), soff(A), sskip(A);
  {
    int off = 0;
    for (int a = 0; a < A; a++) {
      sl[a] = b.lval[a];
      sm[a] = b.mval[a];
      sskip[a] = (b.mval[a] != 0);
      soff[a] = off - sskip[a];  // pure(a,n) = soff[a] + n
      off += R - sskip[a];
    }
  }
  t->shell_l.upload(sl, s);
  t->shell_m.upload(sm, s);
  t->shell_off.upload(soff, s);
  t->shell_skip.upload(sskip, s);

  // ---- Gaunt couplings ----
  std::vector<int> pair_off(A * A + 1, 0), ent_iLM;
  std::vector<double> ent_c0, ent_c2;
  std::vector<std::vector<int> > byLMx(NLM), byLMy(NLM);
  std::vector<std::vector<double> > byLMc0(NLM), byLMc2(NLM);
  for (int x = 0; x < A; x++)
    for (int y = 0; y < A; y++) {
      int lx = b.lval[x], mx = b.mval[x], ly = b.lval[y], my = b.mval[y];
      int M = mx - my;
      int Lmin = std::max(std::abs(lx - ly) - 2, std::abs(M));
      int Lmax = lx + ly + 2;
      pair_off[x * A + y] = (int)ent_iLM.size();
      for (int L = Lmin; L <= Lmax; L++) {
        double c0 = b.gaunt.mod_coeff(lx, mx, L, M, ly, my);
        double c2 = b.gaunt.coeff(lx, mx, L, M, ly, my);
        if (c0 == 0.0 && c2 == 0.0) continue;
        int iLM = (int)b.LMind(L, M);
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

  std::vector<int> LM_ilm(NLM), LM_partner(NLM, -1);
  std::vector<double> LM_fac(NLM);
  t->h_LM_M.resize(NLM);
  for (int i = 0; i < NLM; i++) {
    int L = b.LM_map[i].first, M = b.LM_map[i].second;
    LM_ilm[i] = (int)b.lmind(L, M);
    LM_fac[i] = b.LMfac(L, M);
    t->h_LM_M[i] = M;
    auto it = std::lower_bound(b.LM_map.begin(), b.LM_map.end(), helfem::diatomic::lmidx_t(L, -M));
    if (it != b.LM_map.end() && it->first == L && it->second == -M) LM_partner[i] = (int)(it - b.LM_map.begin());
  }
  t->LM_ilm.upload(LM_ilm, s);
  t->LM_partner.upload(LM_partner, s);
  t->LM_fac.upload(LM_fac, s);

  // ---- primitive integrals, zero padded to p ----
  {
    const size_t pp = (size_t)p * p;
    std::vector<double> disj(4 * (size_t)Nlm * E * pp, 0.0);
    const std::vector<helfem::Mat> *dsrc[4] = {&b.disjoint_P0, &b.disjoint_P2, &b.disjoint_Q0, &b.disjoint_Q2};
    for (int tt = 0; tt < 4; tt++)
      for (int ilm = 0; ilm < Nlm; ilm++)
        for (int e = 0; e < E; e++) {
          const helfem::Mat &m = (*dsrc[tt])[ilm * E + e];
          double *dst = &disj[(((size_t)tt * Nlm + ilm) * E + e) * pp];
          for (size_t j = 0; j < m.n_cols; j++)
            for (size_t i = 0; i < m.n_rows; i++) dst[j * p + i] = m(i, j);
        }
    t->disj.upload(disj, s);
    HFG_HIP_CHECK(hipStreamSynchronize(s));
  }
  {
    const size_t pp = (size_t)p * p;
    const size_t blk = pp * pp;
    t->tei.resize(4 * (size_t)Nlm * E * blk);
    const std::vector<helfem::Mat> *tsrc[4] = {&b.prim_tei00, &b.prim_tei02, &b.prim_tei20, &b.prim_tei22};
    std::vector<double> stage(blk);
    for (int tt = 0; tt < 4; tt++)
      for (int ilm = 0; ilm < Nlm; ilm++)
        for (int e = 0; e < E; e++) {
          const helfem::Mat &m = (*tsrc[tt])[ilm * E + e];
          const size_t Ni = b.fem.nprim(e);
          double *dst = t->tei.p + (((size_t)tt * Nlm + ilm) * E + e) * blk;
          if ((int)Ni == p) {
            HFG_HIP_CHECK(hipMemcpyAsync(dst, m.memptr(), blk * sizeof(double), hipMemcpyHostToDevice, s));
          } else {
            std::fill(stage.begin(), stage.end(), 0.0);
            for (size_t cj = 0; cj < Ni; cj++)
              for (size_t ci = 0; ci < Ni; ci++)
                for (size_t rj = 0; rj < Ni; rj++)
                  for (size_t ri = 0; ri < Ni; ri++)
                    stage[(cj * p + ci) * pp + rj * p + ri] = m(rj * Ni + ri, cj * Ni + ci);
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
      helfem::Mat bf = b.get_bf(e), df = b.get_df(e);
      helfem::Vec w = b.get_wrad(e), r = b.get_r(e);
      for (int q = 0; q < nq; q++) {
        for (size_t i = 0; i < bf.n_cols; i++) {
          B[((size_t)e * nq + q) * p + i] = bf(q, i);
          dB[((size_t)e * nq + q) * p + i] = df(q, i);
        }
        rw[(size_t)e * nq + q] = w[q];
        rsh[(size_t)e * nq + q] = std::sinh(r[q]);
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
