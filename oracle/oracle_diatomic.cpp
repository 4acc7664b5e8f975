// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Loop-for-loop CPU restatement of the diatomic Fock build of the reference:
//   TwoDBasis::coulomb   src/diatomic/basis.cpp:1359-1530
//   TwoDBasis::exchange  src/diatomic/basis.cpp:1532-1733
//   DFTGridWorker / DFTGrid::eval_Fxc  src/diatomic/dftgrid.cpp:51-117, 343-545, 669-810,
//                                      increment_lda / increment_gga src/diatomic/dftgrid.h:190-253
// The dense (complex basis-function matrix) formulation of the reference is kept on purpose:
// this is the checker for the sum-factorised GPU kernels.
#include "oracle.h"
#include "oracle_grid.h"
#include <cfloat>
#include <cmath>
#include <complex>

namespace oracle {
using helfem::diatomic::TwoDBasis;

static Mat submat(const Mat &M, size_t r0, size_t c0, size_t nr, size_t nc) {
  Mat S(nr, nc);
  for (size_t j = 0; j < nc; j++)
    for (size_t i = 0; i < nr; i++) S(i, j) = M(r0 + i, c0 + j);
  return S;
}
static void add_submat(Mat &M, size_t r0, size_t c0, const Mat &S, double f) {
  for (size_t j = 0; j < S.n_cols; j++)
    for (size_t i = 0; i < S.n_rows; i++) M(r0 + i, c0 + j) += f * S(i, j);
}
static Vec matvec(const Mat &A, const Vec &x) {
  Vec y(A.n_rows, 0.0);
  for (size_t j = 0; j < A.n_cols; j++)
    for (size_t i = 0; i < A.n_rows; i++) y[i] += A(i, j) * x[j];
  return y;
}

Mat coulomb(const TwoDBasis &b, const Mat &P0, int shard_rank, int shard_n) {
  if (!b.have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
  Mat P(b.expand_boundaries(P0));
  const size_t Nel = b.Nel(), Nrad = b.Nrad(), NLM = b.LM_map.size();
  const helfem::IVec &lval = b.lval, &mval = b.mval;

  // ket contraction (:1380-1405)
  std::vector<Mat> Paux0(NLM, Mat(Nrad, Nrad)), Paux2(NLM, Mat(Nrad, Nrad));
  for (size_t kang = 0; kang < lval.size(); kang++)
    for (size_t lang = 0; lang < lval.size(); lang++) {
      int lk = lval[kang], mk = mval[kang], ll = lval[lang], ml = mval[lang];
      int M = mk - ml;
      int Lmin = std::max(std::abs(lk - ll) - 2, std::abs(M));
      int Lmax = lk + ll + 2;
      for (int L = Lmin; L <= Lmax; L++) {
        const size_t iLM = b.LMind(L, M);
        double cpl0 = b.gaunt.mod_coeff(lk, mk, L, M, ll, ml);
        double cpl2 = b.gaunt.coeff(lk, mk, L, M, ll, ml);
        Mat Prad = submat(P, kang * Nrad, lang * Nrad, Nrad, Nrad);
        if (cpl0 != 0.0) add_submat(Paux0[iLM], 0, 0, Prad, cpl0);
        if (cpl2 != 0.0) add_submat(Paux2[iLM], 0, 0, Prad, cpl2);
      }
    }

  // radial contraction per (L,M) (:1414-1495)
  std::vector<Mat> Jaux0(NLM, Mat(Nrad, Nrad)), Jaux2(NLM, Mat(Nrad, Nrad));
  for (size_t iLM = 0; iLM < NLM; iLM++) {
    int L = b.LM_map[iLM].first, M = b.LM_map[iLM].second;
    const size_t ilm = b.lmind(L, M);
    if ((int)(ilm % shard_n) != shard_rank) continue;  // channel owned by another rank: contributes nothing here
    const double LMfac = b.LMfac(L, M);
    for (size_t jel = 0; jel < Nel; jel++) {
      size_t jfirst, jlast;
      b.fem.get_idx(jel, jfirst, jlast);
      size_t Nj = jlast - jfirst + 1;
      Mat Psub0 = submat(Paux0[iLM], jfirst, jfirst, Nj, Nj);
      Mat Psub2 = submat(Paux2[iLM], jfirst, jfirst, Nj, Nj);

      double jsmall0 = LMfac * helfem::trace_prod(b.disjoint_P0[ilm * Nel + jel], Psub0);
      double jbig0 = LMfac * helfem::trace_prod(b.disjoint_Q0[ilm * Nel + jel], Psub0);
      double jsmall2 = LMfac * helfem::trace_prod(b.disjoint_P2[ilm * Nel + jel], Psub2);
      double jbig2 = LMfac * helfem::trace_prod(b.disjoint_Q2[ilm * Nel + jel], Psub2);

      double ifac0 = jbig0 - jbig2, ifac2 = -jbig0 + jbig2;
      for (size_t iel = 0; iel < jel; iel++) {
        size_t ifirst, ilast;
        b.fem.get_idx(iel, ifirst, ilast);
        add_submat(Jaux0[iLM], ifirst, ifirst, b.disjoint_P0[ilm * Nel + iel], ifac0);
        add_submat(Jaux2[iLM], ifirst, ifirst, b.disjoint_P2[ilm * Nel + iel], ifac2);
      }
      ifac0 = jsmall0 - jsmall2;
      ifac2 = -jsmall0 + jsmall2;
      for (size_t iel = jel + 1; iel < Nel; iel++) {
        size_t ifirst, ilast;
        b.fem.get_idx(iel, ifirst, ilast);
        add_submat(Jaux0[iLM], ifirst, ifirst, b.disjoint_Q0[ilm * Nel + iel], ifac0);
        add_submat(Jaux2[iLM], ifirst, ifirst, b.disjoint_Q2[ilm * Nel + iel], ifac2);
      }
      {  // in-element contribution
        const size_t idx = ilm * Nel + jel;
        Vec t00 = matvec(b.prim_tei00[idx], Psub0.d), t02 = matvec(b.prim_tei02[idx], Psub2.d);
        Vec t20 = matvec(b.prim_tei20[idx], Psub0.d), t22 = matvec(b.prim_tei22[idx], Psub2.d);
        Mat Jsub0(Nj, Nj), Jsub2(Nj, Nj);
        for (size_t k = 0; k < Nj * Nj; k++) {
          Jsub0.d[k] = LMfac * t00[k] - LMfac * t02[k];
          Jsub2.d[k] = -LMfac * t20[k] + LMfac * t22[k];
        }
        add_submat(Jaux0[iLM], jfirst, jfirst, Jsub0, 1.0);
        add_submat(Jaux2[iLM], jfirst, jfirst, Jsub2, 1.0);
      }
    }
  }

  // bra expansion (:1498-1527)
  Mat J(b.Ndummy(), b.Ndummy());
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      int M = mj - mi;
      int Lmin = std::max(std::abs(lj - li) - 2, std::abs(M));
      int Lmax = lj + li + 2;
      for (int L = Lmin; L <= Lmax; L++) {
        const size_t iLM = b.LMind(L, M);
        double cpl0 = b.gaunt.mod_coeff(lj, mj, L, M, li, mi);
        if (cpl0 != 0.0) add_submat(J, iang * Nrad, jang * Nrad, Jaux0[iLM], cpl0);
        double cpl2 = b.gaunt.coeff(lj, mj, L, M, li, mi);
        if (cpl2 != 0.0) add_submat(J, iang * Nrad, jang * Nrad, Jaux2[iLM], cpl2);
      }
    }
  return b.remove_boundaries(J);
}

Mat exchange(const TwoDBasis &b, const Mat &P0, const std::vector<std::pair<int, int> > *only) {
  if (!b.have_ktei) throw std::logic_error("Primitive teis have not been computed!\n");
  Mat P(b.expand_boundaries(P0));
  const size_t Nel = b.Nel(), Nrad = b.Nrad(), Nlm = b.lm_map.size();
  const helfem::IVec &lval = b.lval, &mval = b.mval;
  Mat K(b.Ndummy(), b.Ndummy());

  for (size_t jang = 0; jang < lval.size(); jang++)
    for (size_t kang = 0; kang < lval.size(); kang++) {
      // output-block filter of the full-size parity tests: the reference's loop is per (jang, kang) (basis.cpp:1575-1579),
      // each block's arithmetic is independent of the others, so a subset of blocks is still the reference algorithm
      if (only) {
        bool sel = false;
        for (const auto &jk : *only) sel = sel || ((size_t)jk.first == jang && (size_t)jk.second == kang);
        if (!sel) continue;
      }
      int lj = lval[jang], mj = mval[jang], lk = lval[kang], mk = mval[kang];
      std::vector<Mat> R00(Nlm, Mat(Nrad, Nrad)), R02(Nlm, Mat(Nrad, Nrad)), R20(Nlm, Mat(Nrad, Nrad)),
          R22(Nlm, Mat(Nrad, Nrad));
      std::vector<bool> couple(Nlm, false);

      // angular sums (:1601-1651)
      for (size_t iang = 0; iang < lval.size(); iang++) {
        int li = lval[iang], mi = mval[iang];
        for (size_t lang = 0; lang < lval.size(); lang++) {
          int ll = lval[lang], ml = mval[lang];
          int M = mj - mi, Mp = mk - ml;
          if (M != Mp) continue;
          Mat Psub = submat(P, iang * Nrad, lang * Nrad, Nrad, Nrad);
          double bdens = 0.0;
          for (double v : Psub.d) bdens += v * v;
          bdens = sqrt(bdens);
          if (bdens < 10 * DBL_EPSILON) continue;
          int Lmin = std::max(std::max(std::abs(li - lj), std::abs(lk - ll)) - 2, std::abs(M));
          int Lmax = std::min(li + lj, lk + ll) + 2;
          for (int L = Lmin; L <= Lmax; L++) {
            double mj_i = b.gaunt.mod_coeff(lj, mj, L, M, li, mi), c_j_i = b.gaunt.coeff(lj, mj, L, M, li, mi);
            double mk_l = b.gaunt.mod_coeff(lk, mk, L, M, ll, ml), c_k_l = b.gaunt.coeff(lk, mk, L, M, ll, ml);
            double cpl00 = mj_i * mk_l, cpl02 = -mj_i * c_k_l, cpl20 = -c_j_i * mk_l, cpl22 = c_j_i * c_k_l;
            if (cpl00 == 0.0 && cpl02 == 0.0 && cpl20 == 0.0 && cpl22 == 0.0) continue;
            const size_t ilm = b.lmind(L, M);
            const double LMfac = b.LMfac(L, M);
            add_submat(R00[ilm], 0, 0, Psub, LMfac * cpl00);
            add_submat(R02[ilm], 0, 0, Psub, LMfac * cpl02);
            add_submat(R20[ilm], 0, 0, Psub, LMfac * cpl20);
            add_submat(R22[ilm], 0, 0, Psub, LMfac * cpl22);
            couple[ilm] = true;
          }
        }
      }

      // element loops (:1654-1727)
      for (size_t iel = 0; iel < Nel; iel++) {
        size_t ifirst, ilast;
        b.fem.get_idx(iel, ifirst, ilast);
        for (size_t jel = 0; jel < Nel; jel++) {
          size_t jfirst, jlast;
          b.fem.get_idx(jel, jfirst, jlast);
          size_t Ni = ilast - ifirst + 1, Nj = jlast - jfirst + 1;
          if (iel == jel) {
            Vec Ksub(Ni * Nj, 0.0);
            for (size_t ilm = 0; ilm < Nlm; ilm++) {
              if (!couple[ilm]) continue;
              size_t idx = ilm * Nel + iel;
              const Mat *kt[4] = {&b.prim_ktei00[idx], &b.prim_ktei02[idx], &b.prim_ktei20[idx], &b.prim_ktei22[idx]};
              const Mat *Rm[4] = {&R00[ilm], &R02[ilm], &R20[ilm], &R22[ilm]};
              for (int t = 0; t < 4; t++) {
                Mat Rs = submat(*Rm[t], ifirst, jfirst, Ni, Nj);
                Vec y = matvec(*kt[t], Rs.d);
                for (size_t k = 0; k < Ksub.size(); k++) Ksub[k] += y[k];
              }
            }
            for (size_t jj = 0; jj < Nj; jj++)
              for (size_t ii = 0; ii < Ni; ii++)
                K(jang * Nrad + ifirst + ii, kang * Nrad + jfirst + jj) -= Ksub[jj * Ni + ii];
          } else {
            Mat Ksub(Ni, Nj);
            for (size_t ilm = 0; ilm < Nlm; ilm++) {
              if (!couple[ilm]) continue;
              // when r(iel)>r(jel), iel gets Q, jel gets P
              const Mat &iint0 = (iel > jel) ? b.disjoint_Q0[ilm * Nel + iel] : b.disjoint_P0[ilm * Nel + iel];
              const Mat &iint2 = (iel > jel) ? b.disjoint_Q2[ilm * Nel + iel] : b.disjoint_P2[ilm * Nel + iel];
              const Mat &jint0 = (iel > jel) ? b.disjoint_P0[ilm * Nel + jel] : b.disjoint_Q0[ilm * Nel + jel];
              const Mat &jint2 = (iel > jel) ? b.disjoint_P2[ilm * Nel + jel] : b.disjoint_Q2[ilm * Nel + jel];
              Mat T = helfem::matmul(submat(R00[ilm], ifirst, jfirst, Ni, Nj), false, jint0, true) +
                      helfem::matmul(submat(R02[ilm], ifirst, jfirst, Ni, Nj), false, jint2, true);
              Ksub -= helfem::matmul(iint0, false, T, false);
              T = helfem::matmul(submat(R20[ilm], ifirst, jfirst, Ni, Nj), false, jint0, true) +
                  helfem::matmul(submat(R22[ilm], ifirst, jfirst, Ni, Nj), false, jint2, true);
              Ksub -= helfem::matmul(iint2, false, T, false);
            }
            add_submat(K, jang * Nrad + ifirst, kang * Nrad + jfirst, Ksub, 1.0);
          }
        }
      }
    }
  return b.remove_boundaries(K);
}

// -------------------------------------------------------------------------------------------------
// XC quadrature, dense formulation of the reference
// -------------------------------------------------------------------------------------------------
namespace {
struct GridWorker : public DenseGrid {
  const TwoDBasis &b;
  Vec cth, phi, wang;

  GridWorker(const TwoDBasis &b_, int lang, int mang) : b(b_) { helfem::angular_chebyshev(lang, mang, cth, phi, wang); }

  void compute_bf(size_t iel, size_t irad) {
    // dftgrid.cpp:669-755
    size_t ifirst, ilast;
    b.fem.get_idx(iel, ifirst, ilast);
    size_t Nr = ilast - ifirst + 1;
    bf_ind.resize(Nr * b.Nang());
    for (size_t iam = 0; iam < b.Nang(); iam++)
      for (size_t j = 0; j < Nr; j++) bf_ind[iam * Nr + j] = b.Nrad() * iam + ifirst + j;
    ne = bf_ind.size();
    Ng = wang.size();

    double wrad = b.get_wrad(iel)[irad];
    double mu = b.get_r(iel)[irad];
    double shmu = std::sinh(mu), Rh = b.Rhalf;
    scale_r.resize(Ng);
    scale_phi.resize(Ng);
    wtot.resize(Ng);
    for (size_t ia = 0; ia < Ng; ia++) {
      double sth = sqrt(1.0 - cth[ia] * cth[ia]);
      scale_r[ia] = Rh * sqrt(shmu * shmu + sth * sth);
      scale_phi[ia] = Rh * shmu * sth;
      wtot[ia] = wang[ia] * wrad * std::pow(Rh, 3) * shmu * (shmu * shmu + sth * sth);
    }
    scale_theta = scale_r;

    Mat frad = b.get_bf(iel), drad = b.get_df(iel);  // nq x Nr
    bf.assign(ne * Ng, cplx(0));
    if (do_grad) {
      bf_rho.assign(ne * Ng, cplx(0));
      bf_theta.assign(ne * Ng, cplx(0));
      bf_phi.assign(ne * Ng, cplx(0));
    }
    for (size_t ia = 0; ia < Ng; ia++) {
      double cotth = cth[ia] / sqrt(1.0 - cth[ia] * cth[ia]);
      for (size_t i = 0; i < b.Nang(); i++) {
        int l = b.lval[i], m = b.mval[i];
        cplx sph = helfem::spherical_harmonics(l, m, cth[ia], phi[ia]);
        cplx angfac(0);
        if (do_grad) {
          // basis.cpp:1914-1926
          angfac = m * cotth * sph;
          if (m < l)
            angfac += sqrt((double)((l - m) * (l + m + 1))) * std::exp(cplx(0, -phi[ia])) *
                      helfem::spherical_harmonics(l, m + 1, cth[ia], phi[ia]);
        }
        for (size_t j = 0; j < Nr; j++) {
          size_t u = i * Nr + j;
          bf[ia * ne + u] = sph * frad(irad, j);
          if (do_grad) {
            bf_rho[ia * ne + u] = sph * drad(irad, j);
            bf_phi[ia * ne + u] = cplx(0.0, m) * sph * frad(irad, j);
            bf_theta[ia * ne + u] = angfac * frad(irad, j);
          }
        }
      }
    }
  }
};
}  // namespace

void eval_Fxc(const TwoDBasis &b, int lang, int mang, int x_func, int c_func, const Mat &P0, Mat &Hout, double &Exc,
              double &Nel, double &Ekin, double thr, long q_begin, long q_end, int shard_rank, int shard_n) {
  Mat H(b.Ndummy(), b.Ndummy());
  Mat P(b.expand_boundaries(P0));
  double exc = 0.0, nel = 0.0, ekin = 0.0;
  GridWorker grid(b, lang, mang);
  grid.do_grad = (x_func > 0 && xc_is_gga(x_func)) || (c_func > 0 && xc_is_gga(c_func));
  grid.do_tau = (x_func > 0 && xc_is_mgga(x_func)) || (c_func > 0 && xc_is_mgga(c_func));
  long q = 0;
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++, q++) {
      if (q < q_begin || (q_end >= 0 && q >= q_end)) continue;
      if ((int)(q % shard_n) != shard_rank) continue;
      grid.compute_bf(iel, irad);
      grid.update_density(P);
      for (size_t ip = 0; ip < grid.Ng; ip++) nel += grid.wtot[ip] * grid.rho[ip];
      ekin += grid.compute_Ekin();
      grid.compute_xc(x_func, c_func, thr);
      for (size_t ip = 0; ip < grid.Ng; ip++) exc += grid.wtot[ip] * grid.exc[ip] * grid.rho[ip];
      grid.eval_Fxc(H);
    }
  Exc = exc;
  Nel = nel;
  Ekin = ekin;  // only meta-GGAs integrate tau (dftgrid.cpp:227-240)
  Hout = b.remove_boundaries(H);
}

void eval_Fxc_pol(const TwoDBasis &b, int lang, int mang, int x_func, int c_func, const Mat &Pa0, const Mat &Pb0, Mat &Haout,
                  Mat &Hbout, double &Exc, double &Nel, double &Ekin, double thr, long q_begin, long q_end, int shard_rank,
                  int shard_n) {
  Mat Ha(b.Ndummy(), b.Ndummy()), Hb(b.Ndummy(), b.Ndummy());
  Mat Pa(b.expand_boundaries(Pa0)), Pb(b.expand_boundaries(Pb0));
  double exc = 0.0, nel = 0.0, ekin = 0.0;
  GridWorker grid(b, lang, mang);
  grid.do_grad = (x_func > 0 && xc_is_gga(x_func)) || (c_func > 0 && xc_is_gga(c_func));
  grid.do_tau = (x_func > 0 && xc_is_mgga(x_func)) || (c_func > 0 && xc_is_mgga(c_func));
  long q = 0;
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++, q++) {
      if (q < q_begin || (q_end >= 0 && q >= q_end)) continue;
      if ((int)(q % shard_n) != shard_rank) continue;
      grid.compute_bf(iel, irad);
      grid.update_density(Pa, Pb);
      nel += grid.compute_Nel();
      ekin += grid.compute_Ekin();
      grid.compute_xc(x_func, c_func, thr);
      exc += grid.eval_Exc();
      grid.eval_Fxc(Ha, Hb);
    }
  Exc = exc;
  Nel = nel;
  Ekin = ekin;
  Haout = b.remove_boundaries(Ha);
  Hbout = b.remove_boundaries(Hb);
}

Mat grid_overlap(const TwoDBasis &b, int lang, int mang) {
  Mat S(b.Ndummy(), b.Ndummy());
  GridWorker grid(b, lang, mang);
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++) {
      grid.compute_bf(iel, irad);
      Mat Ssub(grid.ne, grid.ne);
      grid.increment_lda(Ssub, grid.wtot, grid.bf);
      for (size_t j = 0; j < grid.ne; j++)
        for (size_t i = 0; i < grid.ne; i++) S(grid.bf_ind[i], grid.bf_ind[j]) += Ssub(i, j);
    }
  return b.remove_boundaries(S);
}

Mat model_potential(const TwoDBasis &b, int lang, int mang, const helfem::ModelPotential &p1, const helfem::ModelPotential &p2) {
  // twodquadrature.cpp:213-232 (integrand V_1(r_1) + V_2(r_2), r_{1,2} = Rh (cosh mu +- cos theta), non-normal values
  // skipped) and :351-375 (sum over the radial points); dense form H += Re[(bf o w v) bf^H] of the XC worker
  Mat H(b.Ndummy(), b.Ndummy());
  GridWorker grid(b, lang, mang);
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++) {
      grid.compute_bf(iel, irad);
      const double chmu = std::cosh(b.get_r(iel)[irad]);
      Vec wv(grid.Ng);
      for (size_t ia = 0; ia < grid.Ng; ia++) {
        const double r1 = b.Rhalf * (chmu + grid.cth[ia]), r2 = b.Rhalf * (chmu - grid.cth[ia]);
        const double V1 = p1.V(r1), V2 = p2.V(r2);
        double v = 0.0;
        if (std::isnormal(V1)) v += V1;
        if (std::isnormal(V2)) v += V2;
        wv[ia] = grid.wtot[ia] * v;
      }
      Mat Hsub(grid.ne, grid.ne);
      grid.increment_lda(Hsub, wv, grid.bf);
      for (size_t j = 0; j < grid.ne; j++)
        for (size_t i = 0; i < grid.ne; i++) H(grid.bf_ind[i], grid.bf_ind[j]) += Hsub(i, j);
    }
  return b.remove_boundaries(H);
}

Mat grid_kinetic(const TwoDBasis &b, int lang, int mang) {
  Mat T(b.Ndummy(), b.Ndummy());
  GridWorker grid(b, lang, mang);
  grid.do_grad = true;
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++) {
      grid.compute_bf(iel, irad);
      Mat Tsub(grid.ne, grid.ne);
      Vec w0(grid.Ng), w1(grid.Ng), w2(grid.Ng);
      for (size_t i = 0; i < grid.Ng; i++) {
        w0[i] = grid.wtot[i] / (grid.scale_r[i] * grid.scale_r[i]);
        w1[i] = grid.wtot[i] / (grid.scale_theta[i] * grid.scale_theta[i]);
        w2[i] = grid.wtot[i] / (grid.scale_phi[i] * grid.scale_phi[i]);
      }
      grid.increment_lda(Tsub, w0, grid.bf_rho);
      grid.increment_lda(Tsub, w1, grid.bf_theta);
      grid.increment_lda(Tsub, w2, grid.bf_phi);
      for (size_t j = 0; j < grid.ne; j++)
        for (size_t i = 0; i < grid.ne; i++) T(grid.bf_ind[i], grid.bf_ind[j]) += 0.5 * Tsub(i, j);
    }
  return b.remove_boundaries(T);
}

}  // namespace oracle
