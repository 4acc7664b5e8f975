// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Loop-for-loop CPU restatement of the atomic Fock build of the reference:
//   atomic::basis::TwoDBasis::coulomb   src/atomic/TwoDBasis.cpp:817-955
//   atomic::basis::TwoDBasis::exchange  src/atomic/TwoDBasis.cpp:957-1140
//   atomic::basis::TwoDBasis::rs_exchange  src/atomic/TwoDBasis.cpp:1142-1322
//   atomic::dftgrid::DFTGridWorker / DFTGrid::eval_Fxc  src/atomic/dftgrid.cpp:710-790 (compute_bf), :810-870,
//       with the same update_density / compute_xc / eval_Fxc algebra as the diatomic worker
//       (w = w_ang w_rad r^2, h_r = 1, h_theta = r, h_phi = r sin(theta);  atomic/dftgrid.cpp:724-743).
// The reference processes a whole radial element per compute_bf call; the sums are the same, here they are
// accumulated one radial point at a time (the dense complex formulation is kept).
#include "oracle.h"
#include "oracle_grid.h"
#include "../helfem_amd/csrc/host/atomic_basis.h"
#include <cfloat>
#include <cmath>
#include <complex>

namespace oracle {
using helfem::atomic::TwoDBasis;

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

Mat atomic_coulomb(const TwoDBasis &b, const Mat &P) {
  if (!b.have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
  const size_t Nel = b.Nel(), Nrad = b.Nrad();
  const helfem::IVec &lval = b.lval, &mval = b.mval;
  const int NL = b.N_L(), Mmax = b.Mmax();
  auto idx = [&](int L, int M) { return (size_t)L * (2 * Mmax + 1) + (M + Mmax); };
  std::vector<Mat> Paux((size_t)NL * (2 * Mmax + 1), Mat(Nrad, Nrad)), Jaux((size_t)NL * (2 * Mmax + 1), Mat(Nrad, Nrad));
  // contract ket (:852-873)
  for (size_t kang = 0; kang < lval.size(); kang++)
    for (size_t lang = 0; lang < lval.size(); lang++) {
      int lk = lval[kang], mk = mval[kang], ll = lval[lang], ml = mval[lang];
      int M = mk - ml;
      int Lmin = std::max(std::abs(lk - ll), std::abs(M)), Lmax = lk + ll;
      for (int L = Lmin; L <= Lmax; L++) {
        double cpl = b.gaunt.coeff(lk, mk, L, M, ll, ml);
        add_submat(Paux[idx(L, M)], 0, 0, submat(P, kang * Nrad, lang * Nrad, Nrad, Nrad), cpl);
      }
    }
  // contract integrals (:884-936)
  for (int L = 0; L < NL; L++) {
    const double Lfac = 4.0 * M_PI / (2 * L + 1);
    for (int M = -std::min(L, Mmax); M <= std::min(L, Mmax); M++)
      for (size_t jel = 0; jel < Nel; jel++) {
        size_t jfirst, jlast;
        b.fem.get_idx(jel, jfirst, jlast);
        size_t Nj = jlast - jfirst + 1;
        Mat Psub = submat(Paux[idx(L, M)], jfirst, jfirst, Nj, Nj);
        double jsmall = Lfac * helfem::trace_prod(b.disjoint_L[L * Nel + jel], Psub);
        double jbig = Lfac * helfem::trace_prod(b.disjoint_m1L[L * Nel + jel], Psub);
        for (size_t iel = 0; iel < jel; iel++)
          add_submat(Jaux[idx(L, M)], b.fem.first[iel], b.fem.first[iel], b.disjoint_L[L * Nel + iel], jbig);
        for (size_t iel = jel + 1; iel < Nel; iel++)
          add_submat(Jaux[idx(L, M)], b.fem.first[iel], b.fem.first[iel], b.disjoint_m1L[L * Nel + iel], jsmall);
        Vec y = matvec(b.prim_tei[L * Nel + jel], Psub.d);
        Mat Jsub(Nj, Nj);
        for (size_t k = 0; k < y.size(); k++) Jsub.d[k] = Lfac * y[k];
        add_submat(Jaux[idx(L, M)], jfirst, jfirst, Jsub, 1.0);
      }
  }
  // full Coulomb matrix (:938-954)
  Mat J(b.Nbf(), b.Nbf());
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      int M = mj - mi;
      int Lmin = std::max(std::abs(lj - li), std::abs(M)), Lmax = lj + li;
      for (int L = Lmin; L <= Lmax; L++) {
        double cpl = b.gaunt.coeff(lj, mj, L, M, li, mi);
        if (cpl != 0.0) add_submat(J, iang * Nrad, jang * Nrad, Jaux[idx(L, M)], cpl);
      }
    }
  return J;
}

Mat atomic_exchange(const TwoDBasis &b, const Mat &P) {
  if (!b.have_ktei) throw std::logic_error("Primitive teis have not been computed!\n");
  const size_t Nel = b.Nel(), Nrad = b.Nrad();
  const helfem::IVec &lval = b.lval, &mval = b.mval;
  const size_t NL = b.N_L();
  Mat K(b.Nbf(), b.Nbf());
  for (size_t jang = 0; jang < lval.size(); jang++)
    for (size_t kang = 0; kang < lval.size(); kang++) {
      int lj = lval[jang], mj = mval[jang], lk = lval[kang], mk = mval[kang];
      std::vector<Mat> Rmat(NL, Mat(Nrad, Nrad));
      std::vector<bool> couple(NL, false);
      for (size_t iang = 0; iang < lval.size(); iang++) {
        int li = lval[iang], mi = mval[iang];
        for (size_t lang = 0; lang < lval.size(); lang++) {
          int ll = lval[lang], ml = mval[lang];
          int M = mj - mi, Mp = mk - ml;
          if (M != Mp) continue;
          Mat Psub = submat(P, iang * Nrad, lang * Nrad, Nrad, Nrad);
          double bdens = 0.0;
          for (double v : Psub.d) bdens += v * v;
          if (sqrt(bdens) < 10 * DBL_EPSILON) continue;
          int Lmin = std::max(std::max(std::abs(li - lj), std::abs(lk - ll)), std::abs(M));
          int Lmax = std::min(li + lj, lk + ll);
          for (int L = Lmin; L <= Lmax; L++) {
            double cpl = b.gaunt.coeff(lj, mj, L, M, li, mi) * b.gaunt.coeff(lk, mk, L, M, ll, ml);
            if (cpl == 0.0) continue;
            double Lfac = 4.0 * M_PI / (2 * L + 1);
            add_submat(Rmat[L], 0, 0, Psub, Lfac * cpl);
            couple[L] = true;
          }
        }
      }
      for (size_t iel = 0; iel < Nel; iel++) {
        size_t ifirst, ilast;
        b.fem.get_idx(iel, ifirst, ilast);
        for (size_t jel = 0; jel < Nel; jel++) {
          size_t jfirst, jlast;
          b.fem.get_idx(jel, jfirst, jlast);
          size_t Ni = ilast - ifirst + 1, Nj = jlast - jfirst + 1;
          if (iel == jel) {
            Vec Ksub(Ni * Nj, 0.0);
            for (size_t L = 0; L < NL; L++) {
              if (!couple[L]) continue;
              Vec y = matvec(b.prim_ktei[L * Nel + iel], submat(Rmat[L], ifirst, jfirst, Ni, Nj).d);
              for (size_t k = 0; k < Ksub.size(); k++) Ksub[k] += y[k];
            }
            for (size_t jj = 0; jj < Nj; jj++)
              for (size_t ii = 0; ii < Ni; ii++)
                K(jang * Nrad + ifirst + ii, kang * Nrad + jfirst + jj) -= Ksub[jj * Ni + ii];
          } else {
            Mat Ksub(Ni, Nj);
            for (size_t L = 0; L < NL; L++) {
              if (!couple[L]) continue;
              const Mat &iint = (iel > jel) ? b.disjoint_m1L[L * Nel + iel] : b.disjoint_L[L * Nel + iel];
              const Mat &jint = (iel > jel) ? b.disjoint_L[L * Nel + jel] : b.disjoint_m1L[L * Nel + jel];
              Mat T = helfem::matmul(submat(Rmat[L], ifirst, jfirst, Ni, Nj), false, jint, true);
              Ksub += helfem::matmul(iint, false, T, false);
            }
            add_submat(K, jang * Nrad + ifirst, kang * Nrad + jfirst, Ksub, -1.0);
          }
        }
      }
    }
  return K;
}

Mat atomic_rs_exchange(const TwoDBasis &b, const Mat &P) {
  if (b.rs_ktei.empty()) throw std::logic_error("Primitive teis have not been computed!\n");
  const bool yukawa = (b.rs_kind == 1);
  const double lambda = b.rs_lambda;
  const size_t Nel = b.Nel(), Nrad = b.Nrad();
  const helfem::IVec &lval = b.lval, &mval = b.mval;
  const size_t NL = b.N_L();
  Mat K(b.Nbf(), b.Nbf());
  for (size_t jang = 0; jang < lval.size(); jang++)
    for (size_t kang = 0; kang < lval.size(); kang++) {
      int lj = lval[jang], mj = mval[jang], lk = lval[kang], mk = mval[kang];
      // radial helpers: angular sums (:1207-1245)
      std::vector<Mat> Rmat(NL, Mat(Nrad, Nrad));
      std::vector<bool> couple(NL, false);
      for (size_t iang = 0; iang < lval.size(); iang++) {
        int li = lval[iang], mi = mval[iang];
        for (size_t lang = 0; lang < lval.size(); lang++) {
          int ll = lval[lang], ml = mval[lang];
          int M = mj - mi, Mp = mk - ml;
          if (M != Mp) continue;
          Mat Psub = submat(P, iang * Nrad, lang * Nrad, Nrad, Nrad);
          double bdens = 0.0;
          for (double v : Psub.d) bdens += v * v;
          if (sqrt(bdens) < 10 * DBL_EPSILON) continue;
          int Lmin = std::max(std::max(std::abs(li - lj), std::abs(lk - ll)), std::abs(M));
          int Lmax = std::min(li + lj, lk + ll);
          for (int L = Lmin; L <= Lmax; L++) {
            double cpl = b.gaunt.coeff(lj, mj, L, M, li, mi) * b.gaunt.coeff(lk, mk, L, M, ll, ml);
            if (cpl == 0.0) continue;
            double Lfac = yukawa ? 4.0 * M_PI * lambda : 4.0 * M_PI * lambda / (2 * L + 1);  // :1240
            add_submat(Rmat[L], 0, 0, Psub, Lfac * cpl);
            couple[L] = true;
          }
        }
      }
      for (size_t iel = 0; iel < Nel; iel++) {
        size_t ifirst, ilast;
        b.fem.get_idx(iel, ifirst, ilast);
        for (size_t jel = 0; jel < Nel; jel++) {
          size_t jfirst, jlast;
          b.fem.get_idx(jel, jfirst, jlast);
          size_t Ni = ilast - ifirst + 1, Nj = jlast - jfirst + 1;
          if (!yukawa || iel == jel) {  // the error-function kernel does not factorise (:1262)
            Vec Ksub(Ni * Nj, 0.0);
            for (size_t L = 0; L < NL; L++) {
              if (!couple[L]) continue;
              const Mat &ktei = yukawa ? b.rs_ktei[L * Nel + iel] : b.rs_ktei[(L * Nel + iel) * Nel + jel];
              Vec y = matvec(ktei, submat(Rmat[L], ifirst, jfirst, Ni, Nj).d);
              for (size_t k = 0; k < Ksub.size(); k++) Ksub[k] += y[k];
            }
            for (size_t jj = 0; jj < Nj; jj++)
              for (size_t ii = 0; ii < Ni; ii++)
                K(jang * Nrad + ifirst + ii, kang * Nrad + jfirst + jj) -= Ksub[jj * Ni + ii];
          } else {
            Mat Ksub(Ni, Nj);
            for (size_t L = 0; L < NL; L++) {
              if (!couple[L]) continue;
              const Mat &iint = (iel > jel) ? b.disjoint_kL[L * Nel + iel] : b.disjoint_iL[L * Nel + iel];
              const Mat &jint = (iel > jel) ? b.disjoint_iL[L * Nel + jel] : b.disjoint_kL[L * Nel + jel];
              Mat T = helfem::matmul(submat(Rmat[L], ifirst, jfirst, Ni, Nj), false, jint, true);
              Ksub += helfem::matmul(iint, false, T, false);
            }
            add_submat(K, jang * Nrad + ifirst, kang * Nrad + jfirst, Ksub, -1.0);
          }
        }
      }
    }
  return K;
}

// ---- XC quadrature --------------------------------------------------------------------------------------
namespace {
struct AtomicGridWorker : public DenseGrid {
  const TwoDBasis &b;
  Vec cth, phi, wang;
  AtomicGridWorker(const TwoDBasis &b_, int lang, int mang) : b(b_) { helfem::angular_chebyshev(lang, mang, cth, phi, wang); }

  // atomic/dftgrid.cpp:710-790, one radial point of element iel
  void compute_bf(size_t iel, size_t irad) {
    size_t ifirst, ilast;
    b.fem.get_idx(iel, ifirst, ilast);
    const size_t Nr = ilast - ifirst + 1, A = b.Nang();
    ne = Nr * A;
    Ng = wang.size();
    bf_ind.resize(ne);
    for (size_t iam = 0; iam < A; iam++)
      for (size_t j = 0; j < Nr; j++) bf_ind[iam * Nr + j] = b.Nrad() * iam + ifirst + j;
    Mat frad = b.get_bf(iel), drad = b.get_df(iel);
    const double r = b.get_r(iel)[irad], wrad = b.get_wrad(iel)[irad];
    bf.assign(ne * Ng, cplx(0));
    if (do_grad) {
      bf_rho.assign(ne * Ng, cplx(0));
      bf_theta.assign(ne * Ng, cplx(0));
      bf_phi.assign(ne * Ng, cplx(0));
    }
    wtot.resize(Ng);
    scale_r.assign(Ng, 1.0);
    scale_theta.assign(Ng, r);
    scale_phi.resize(Ng);
    for (size_t ia = 0; ia < Ng; ia++) {
      double sth = sqrt(1.0 - cth[ia] * cth[ia]);
      wtot[ia] = wang[ia] * wrad * r * r;
      scale_phi[ia] = r * sth;
      double cotth = cth[ia] / sth;
      for (size_t i = 0; i < A; i++) {
        int l = b.lval[i], m = b.mval[i];
        cplx sph = helfem::spherical_harmonics(l, m, cth[ia], phi[ia]);
        cplx angfac(0);
        if (do_grad) {
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

void atomic_eval_Fxc(const TwoDBasis &b, int lang, int mang, int x_func, int c_func, const Mat &P, Mat &H, double &Exc,
                     double &Nel, double &Ekin, double thr) {
  AtomicGridWorker grid(b, lang, mang);
  grid.do_grad = (x_func > 0 && xc_is_gga(x_func)) || (c_func > 0 && xc_is_gga(c_func));
  grid.do_tau = (x_func > 0 && xc_is_mgga(x_func)) || (c_func > 0 && xc_is_mgga(c_func));
  H.zeros(b.Nbf(), b.Nbf());
  double exc = 0.0, nel = 0.0, ekin = 0.0;
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++) {
      grid.compute_bf(iel, irad);
      grid.update_density(P);
      nel += grid.compute_Nel();
      ekin += grid.compute_Ekin();
      grid.compute_xc(x_func, c_func, thr);
      exc += grid.eval_Exc();
      grid.eval_Fxc(H);
    }
  Exc = exc;
  Nel = nel;
  Ekin = ekin;
}

void atomic_eval_Fxc_pol(const TwoDBasis &b, int lang, int mang, int x_func, int c_func, const Mat &Pa, const Mat &Pb,
                         Mat &Ha, Mat &Hb, double &Exc, double &Nel, double &Ekin, double thr) {
  AtomicGridWorker grid(b, lang, mang);
  grid.do_grad = (x_func > 0 && xc_is_gga(x_func)) || (c_func > 0 && xc_is_gga(c_func));
  grid.do_tau = (x_func > 0 && xc_is_mgga(x_func)) || (c_func > 0 && xc_is_mgga(c_func));
  Ha.zeros(b.Nbf(), b.Nbf());
  Hb.zeros(b.Nbf(), b.Nbf());
  double exc = 0.0, nel = 0.0, ekin = 0.0;
  for (size_t iel = 0; iel < b.Nel(); iel++)
    for (size_t irad = 0; irad < (size_t)b.nquad(); irad++) {
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
}

}  // namespace oracle
