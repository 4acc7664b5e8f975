#include "diatomic_basis.h"
#include "parallel.h"
#include <algorithm>
#include <cmath>
#include <set>
#include <sstream>

namespace helfem {
namespace diatomic {

void lm_to_l_m(const IVec &lmmax, IVec &lval, IVec &mval) {
  lval.clear();
  mval.clear();
  for (size_t mabs = 0; mabs < lmmax.size(); mabs++)
    for (int l = (int)mabs; l <= lmmax[mabs]; l++) {
      lval.push_back(l);
      mval.push_back((int)mabs);
      if (mabs > 0) {
        lval.push_back(l);
        mval.push_back(-(int)mabs);
      }
    }
}

TwoDBasis::TwoDBasis(int Z1_, int Z2_, double Rhalf_, int nnodes_, int n_quad, const Vec &bval,
                     const IVec &lval_, const IVec &mval_, int lpad_)
    : Z1(Z1_), Z2(Z2_), Rhalf(Rhalf_), lpad(lpad_), nnodes(nnodes_), lval(lval_), mval(mval_) {
  if (nnodes < 2) throw std::logic_error("Can't have finite element basis with less than two nodes per element.\n");
  if (lval.size() != mval.size() || lval.empty()) throw std::logic_error("Invalid angular basis\n");
  // sigma orbitals may reach the nuclear axis: no function dropped on the left; function (and
  // derivative, irrelevant for LIPs) dropped at the practical infinity (basis.cpp:314-318)
  LIPBasis poly(lobatto_nodes(nnodes));
  fem = FEMBasis(poly, bval, false, true);
  chebyshev_rule(n_quad, xq, wq);

  // L|M| and LM maps (basis.cpp:333-378)
  std::set<lmidx_t> lmset, LMset;
  Lmax = 0;
  Mmax = 0;
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      int M = mj - mi;
      int Lstart = std::max(std::abs(lj - li) - 2, std::abs(M));
      int Lend = lj + li + 2;
      for (int L = Lstart; L <= Lend; L++) {
        Lmax = std::max(Lmax, L);
        Mmax = std::max(Mmax, std::abs(M));
        lmset.insert(lmidx_t(L, std::abs(M)));
        LMset.insert(lmidx_t(L, M));
      }
    }
  lm_map.assign(lmset.begin(), lmset.end());
  LM_map.assign(LMset.begin(), LMset.end());
}

size_t TwoDBasis::Nbf() const {
  size_t nbf = 0;
  for (size_t i = 0; i < mval.size(); i++) nbf += (mval[i] != 0) ? Nrad() - 1 : Nrad();
  return nbf;
}

std::vector<size_t> TwoDBasis::pure_indices() const {
  // basis.cpp:482-499: non-sigma shells lose their first radial function
  std::vector<size_t> idx;
  idx.reserve(Nbf());
  for (size_t i = 0; i < mval.size(); i++)
    for (size_t j = (mval[i] == 0 ? 0 : 1); j < Nrad(); j++) idx.push_back(i * Nrad() + j);
  return idx;
}

std::vector<size_t> TwoDBasis::m_indices(int m) const {
  std::vector<size_t> idx;
  size_t ibf = 0;
  for (size_t i = 0; i < mval.size(); i++) {
    size_t nsh = (mval[i] == 0) ? Nrad() : Nrad() - 1;
    if (mval[i] == m)
      for (size_t j = 0; j < nsh; j++) idx.push_back(ibf + j);
    ibf += nsh;
  }
  return idx;
}

std::vector<size_t> TwoDBasis::m_indices(int m, bool odd) const {
  std::vector<size_t> idx;
  size_t ibf = 0;
  for (size_t i = 0; i < mval.size(); i++) {
    size_t nsh = (mval[i] == 0) ? Nrad() : Nrad() - 1;
    if (mval[i] == m && (lval[i] % 2 == (int)odd))
      for (size_t j = 0; j < nsh; j++) idx.push_back(ibf + j);
    ibf += nsh;
  }
  return idx;
}

std::vector<std::vector<size_t> > TwoDBasis::get_sym_idx(int symm) const {
  // basis.cpp:561-588
  std::vector<std::vector<size_t> > idx;
  std::set<int> mset(mval.begin(), mval.end());
  if (symm == 0) {
    idx.resize(1);
    for (size_t i = 0; i < Nbf(); i++) idx[0].push_back(i);
  } else if (symm == 1) {
    for (int m : mset) idx.push_back(m_indices(m));
  } else if (symm == 2) {
    for (int m : mset) {
      idx.push_back(m_indices(m, false));
      idx.push_back(m_indices(m, true));
    }
  } else
    throw std::logic_error("Unknown symmetry\n");
  return idx;
}

size_t TwoDBasis::lmind(int L, int M) const {
  lmidx_t p(L, std::abs(M));
  auto low = std::lower_bound(lm_map.begin(), lm_map.end(), p);
  if (low == lm_map.end() || !(*low == p)) {
    std::ostringstream oss;
    oss << "Could not find L=" << p.first << ", |M|= " << p.second << " on the list!\n";
    throw std::logic_error(oss.str());
  }
  return low - lm_map.begin();
}

size_t TwoDBasis::LMind(int L, int M) const {
  lmidx_t p(L, M);
  auto low = std::lower_bound(LM_map.begin(), LM_map.end(), p);
  if (low == LM_map.end() || !(*low == p)) {
    std::ostringstream oss;
    oss << "Could not find L=" << p.first << ", M= " << p.second << " on the list!\n";
    throw std::logic_error(oss.str());
  }
  return low - LM_map.begin();
}

double TwoDBasis::LMfac(int L, int M) const {
  // 4 pi Rh^5 (-1)^M / [(L+|M|)!/(L-|M|)!]
  int aM = std::abs(M);
  double ratio = 1.0;
  for (int p = L + aM; p > L - aM; p--) ratio *= p;
  return 4.0 * M_PI * std::pow(Rhalf, 5) * ((aM % 2) ? -1.0 : 1.0) / ratio;
}

Mat TwoDBasis::radial_integral(int m, int n) const {
  // basis.cpp:92-102
  std::function<double(double)> chsh;
  if (m != 0 && n != 0)
    chsh = [m, n](double mu) { return std::pow(std::sinh(mu), m) * std::pow(std::cosh(mu), n); };
  else if (m != 0 && n == 0)
    chsh = [m](double mu) { return std::pow(std::sinh(mu), m); };
  else if (m == 0 && n != 0)
    chsh = [n](double mu) { return std::pow(std::cosh(mu), n); };
  return fem.matrix_element(0, 0, xq, wq, chsh);
}

Mat TwoDBasis::radial_overlap(const TwoDBasis &rh, int n) const {
  // the products of two finite element bases are polynomial only on the intersections of their elements: one rule per
  // intersection, with the larger of the two numbers of quadrature points (basis.cpp:104-200)
  Vec xp, wp;
  chebyshev_rule((int)std::max(xq.size(), rh.xq.size()), xp, wp);
  Mat S(fem.nbf(), rh.fem.nbf());
  for (size_t iel = 0; iel < fem.nelem(); iel++)
    for (size_t jel = 0; jel < rh.fem.nelem(); jel++) {
      const double imin = fem.element_begin(iel), imax = fem.element_end(iel);
      const double jmin = rh.fem.element_begin(jel), jmax = rh.fem.element_end(jel);
      if (!((jmin >= imin && jmin < imax) || (imin >= jmin && imin < jmax))) continue;
      const double a = std::max(imin, jmin), b = std::min(imax, jmax);
      const double mid = 0.5 * (b + a), len = 0.5 * (b - a);
      Vec xi(xp.size()), xj(xp.size()), w(xp.size());
      for (size_t q = 0; q < xp.size(); q++) {
        const double mu = mid + len * xp[q];
        xi[q] = (mu - fem.element_midpoint(iel)) / fem.scaling_factor(iel);
        xj[q] = (mu - rh.fem.element_midpoint(jel)) / rh.fem.scaling_factor(jel);
        w[q] = wp[q] * len * std::sinh(mu) * (n != 0 ? std::pow(std::cosh(mu), n) : 1.0);
      }
      const Mat ibf = fem.eval_dnf(xi, 0, iel), jbf = rh.fem.eval_dnf(xj, 0, jel);
      size_t i0, i1, j0, j1;
      fem.get_idx(iel, i0, i1);
      rh.fem.get_idx(jel, j0, j1);
      for (size_t fj = 0; fj < jbf.n_cols; fj++)
        for (size_t fi = 0; fi < ibf.n_cols; fi++) {
          double acc = 0.0;
          for (size_t q = 0; q < xp.size(); q++) acc += w[q] * ibf(q, fi) * jbf(q, fj);
          S(i0 + fi, j0 + fj) += acc;
        }
    }
  return S;
}

Mat TwoDBasis::overlap(const TwoDBasis &rh) const {
  // basis.cpp:713-750
  const Mat I10(radial_overlap(rh, 0)), I12(radial_overlap(rh, 2));
  const size_t Nr = Nrad(), Nr2 = rh.Nrad();
  Mat S(Ndummy(), rh.Ndummy());
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < rh.lval.size(); jang++) {
      const int li = lval[iang], mi = mval[iang], lj = rh.lval[jang], mj = rh.mval[jang];
      if (mi != mj) continue;
      const double cpl = gaunt.cosine2_coupling(lj, mj, li, mi);
      if (li != lj && cpl == 0.0) continue;
      for (size_t j = 0; j < Nr2; j++)
        for (size_t i = 0; i < Nr; i++) {
          double v = (li == lj) ? I12(i, j) : 0.0;
          if (cpl != 0.0) v -= I10(i, j) * cpl;
          S(iang * Nr + i, jang * Nr2 + j) = v;
        }
    }
  S *= std::pow(Rhalf, 3);
  const std::vector<size_t> pi(pure_indices()), pj(rh.pure_indices());
  Mat out(pi.size(), pj.size());
  for (size_t j = 0; j < pj.size(); j++)
    for (size_t i = 0; i < pi.size(); i++) out(i, j) = S(pi[i], pj[j]);
  return out;
}

static void set_sub(Mat &M, size_t Nrad, size_t iang, size_t jang, const Mat &Mrad, double fac, bool add) {
  for (size_t j = 0; j < Nrad; j++)
    for (size_t i = 0; i < Nrad; i++) {
      double v = fac * Mrad(i, j);
      if (add)
        M(iang * Nrad + i, jang * Nrad + j) += v;
      else
        M(iang * Nrad + i, jang * Nrad + j) = v;
    }
}

// One-electron matrices are sums of (angular coupling) x (radial matrix) blocks.  The terms are collected first (the
// Gaunt cache is not thread safe) and the boundary-cleaned matrix is then filled shell column by shell column on all
// host threads: value = scale * (fac_1 R_1(i,j) + fac_2 R_2(i,j) + ...) in the order the terms were added, which is
// what "set_sub ...; M *= scale; remove_boundaries(M)" computes -- without the Ndummy^2 intermediate and its passes.
namespace {
struct BlockTerm {
  size_t iang, jang;
  const Mat *rad;
  double fac;
};
}  // namespace
static Mat assemble_pure(const TwoDBasis &b, const std::vector<BlockTerm> &terms, double scale) {
  const size_t nang = b.lval.size(), Nr = b.Nrad();
  std::vector<size_t> off(nang + 1, 0), skip(nang, 0);
  for (size_t a = 0; a < nang; a++) {
    skip[a] = (b.mval[a] != 0) ? 1 : 0;
    off[a + 1] = off[a] + Nr - skip[a];
  }
  std::vector<std::vector<BlockTerm> > bycol(nang);
  for (const BlockTerm &t : terms) bycol[t.jang].push_back(t);
  Mat F(off[nang], off[nang]);
  parallel_for(nang, [&](size_t jang) {
    const std::vector<BlockTerm> &ts = bycol[jang];
    // the terms of one (iang, jang) block are consecutive in ts (they were added pair by pair)
    for (size_t k = 0; k < ts.size();) {
      size_t k1 = k;
      while (k1 < ts.size() && ts[k1].iang == ts[k].iang) k1++;
      const size_t iang = ts[k].iang;
      for (size_t j = skip[jang]; j < Nr; j++) {
        double *col = &F(off[iang], off[jang] + j - skip[jang]);
        for (size_t i = skip[iang]; i < Nr; i++) {
          double v = ts[k].fac * (*ts[k].rad)(i, j);
          for (size_t q = k + 1; q < k1; q++) v += ts[q].fac * (*ts[q].rad)(i, j);
          col[i - skip[iang]] = v * scale;
        }
      }
      k = k1;
    }
  });
  return F;
}

Mat TwoDBasis::overlap() const {
  // basis.cpp:677-711:  S = Rh^3 [ delta_ll' I_{1,2} - c_2 I_{1,0} ]
  Mat I10(radial_integral(1, 0)), I12(radial_integral(1, 2));
  std::vector<BlockTerm> terms;
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      if (mi == mj) {
        if (li == lj) terms.push_back({iang, jang, &I12, 1.0});
        double cpl = gaunt.cosine2_coupling(lj, mj, li, mi);
        if (cpl != 0.0) terms.push_back({iang, jang, &I10, -cpl});
      }
    }
  return assemble_pure(*this, terms, std::pow(Rhalf, 3));
}

Mat TwoDBasis::kinetic() const {
  // basis.cpp:752-778
  std::function<double(double)> sinhmu = [](double mu) { return std::sinh(mu); };
  Mat Trad(fem.matrix_element(1, 1, xq, wq, sinhmu));
  Mat Ip1(radial_integral(1, 0)), Im1(radial_integral(-1, 0));
  std::vector<BlockTerm> terms;
  for (size_t iang = 0; iang < lval.size(); iang++) {
    terms.push_back({iang, iang, &Trad, 1.0});
    if (lval[iang] != 0) terms.push_back({iang, iang, &Ip1, (double)(lval[iang] * (lval[iang] + 1))});
    if (mval[iang] != 0) terms.push_back({iang, iang, &Im1, (double)(mval[iang] * mval[iang])});
  }
  return assemble_pure(*this, terms, Rhalf / 2.0);
}

Mat TwoDBasis::nuclear() const {
  // basis.cpp:780-817
  Mat I10(radial_integral(1, 0)), I11(radial_integral(1, 1));
  std::vector<BlockTerm> terms;
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      if (mi == mj) {
        if (li == lj) terms.push_back({iang, jang, &I11, (double)(Z1 + Z2)});
        if (Z1 != Z2) {
          double cpl = gaunt.cosine_coupling(lj, mj, li, mi);
          if (cpl != 0.0) terms.push_back({iang, jang, &I10, (Z2 - Z1) * cpl});
        }
      }
    }
  return assemble_pure(*this, terms, -std::pow(Rhalf, 2));
}

Mat TwoDBasis::dipole_z() const {
  // basis.cpp:819-856
  Mat I11(radial_integral(1, 1)), I13(radial_integral(1, 3));
  Mat V(Ndummy(), Ndummy());
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      if (mi == mj) {
        double cpl1 = gaunt.cosine_coupling(lj, mj, li, mi);
        if (cpl1 != 0.0) set_sub(V, Nrad(), iang, jang, I13, cpl1, true);
        double cpl3 = gaunt.cosine3_coupling(lj, mj, li, mi);
        if (cpl3 != 0.0) set_sub(V, Nrad(), iang, jang, I11, -cpl3, true);
      }
    }
  V *= std::pow(Rhalf, 4);
  return remove_boundaries(V);
}

Mat TwoDBasis::quadrupole_zz() const {
  // basis.cpp:858-900
  Mat I10(radial_integral(1, 0)), I12(radial_integral(1, 2)), I14(radial_integral(1, 4));
  Mat A(I10 - 3.0 * I12), B(3.0 * I14 - I10), C(I12 - I14);
  Mat V(Ndummy(), Ndummy());
  for (size_t iang = 0; iang < lval.size(); iang++)
    for (size_t jang = 0; jang < lval.size(); jang++) {
      int li = lval[iang], mi = mval[iang], lj = lval[jang], mj = mval[jang];
      if (mi == mj) {
        double cpl4 = gaunt.cosine4_coupling(lj, mj, li, mi);
        if (cpl4 != 0.0) set_sub(V, Nrad(), iang, jang, A, cpl4, true);
        double cpl2 = gaunt.cosine2_coupling(lj, mj, li, mi);
        if (cpl2 != 0.0) set_sub(V, Nrad(), iang, jang, B, cpl2, true);
        if (li == lj) set_sub(V, Nrad(), iang, jang, C, 1.0, true);
      }
    }
  V *= std::pow(Rhalf, 5) / 2;
  return remove_boundaries(V);
}

Mat TwoDBasis::remove_boundaries(const Mat &Fnob) const {
  if (Fnob.n_rows != Ndummy() || Fnob.n_cols != Ndummy()) {
    std::ostringstream oss;
    oss << "Matrix does not have expected size! Got " << Fnob.n_rows << " x " << Fnob.n_cols << ", expected "
        << Ndummy() << " x " << Ndummy() << "!\n";
    throw std::logic_error(oss.str());
  }
  std::vector<size_t> idx(pure_indices());
  Mat F(idx.size(), idx.size());
  for (size_t j = 0; j < idx.size(); j++)
    for (size_t i = 0; i < idx.size(); i++) F(i, j) = Fnob(idx[i], idx[j]);
  return F;
}

Mat TwoDBasis::expand_boundaries(const Mat &Ppure) const {
  if (Ppure.n_rows != Nbf() || Ppure.n_cols != Nbf()) {
    std::ostringstream oss;
    oss << "Matrix does not have expected size! Got " << Ppure.n_rows << " x " << Ppure.n_cols << ", expected "
        << Nbf() << " x " << Nbf() << "!\n";
    throw std::logic_error(oss.str());
  }
  std::vector<size_t> idx(pure_indices());
  Mat P(Ndummy(), Ndummy());
  for (size_t j = 0; j < idx.size(); j++)
    for (size_t i = 0; i < idx.size(); i++) P(idx[i], idx[j]) = Ppure(i, j);
  return P;
}

Vec TwoDBasis::get_wrad(size_t iel) const {
  Vec w(wq);
  for (auto &x : w) x *= fem.scaling_factor(iel);
  return w;
}

Mat exchange_tei(const Mat &tei, size_t Ni, size_t Nj, size_t Nk, size_t Nl) {
  Mat ktei(Nj * Nk, Ni * Nl);
  for (size_t ii = 0; ii < Ni; ii++)
    for (size_t jj = 0; jj < Nj; jj++)
      for (size_t kk = 0; kk < Nk; kk++)
        for (size_t ll = 0; ll < Nl; ll++) ktei(kk * Nj + jj, ll * Ni + ii) = tei(jj * Ni + ii, ll * Nk + kk);
  return ktei;
}

// -------------------------------------------------------------------------------------------------
// Primitive two-electron integrals
// -------------------------------------------------------------------------------------------------
namespace {
inline double clean(double v) { return std::isnormal(v) ? v : 0.0; }  // legendretable.cpp:83-89
}

void TwoDBasis::tei_element_tables(size_t iel, TeiElementTables &t) const {
  const size_t nq = xq.size(), Nlm = lm_map.size();
  const int ld = Lmax + 1;
  const legendre_provider_t provider = get_legendre_provider();
  const int Lm = Lmax, Mm = Mmax, lp = lpad;
  auto legPQ = [provider, Lm, Mm, lp](double xi, double *P, double *Q) {
    if (provider)
      provider(Lm, Mm, lp, xi, P, Q);
    else
      legendre_PQ(Lm, Mm, xi, P, Q);
  };
  const double mumin0 = fem.element_begin(iel), mumax0 = fem.element_end(iel);
  const double mumid0 = 0.5 * (mumax0 + mumin0), mulen0 = 0.5 * (mumax0 - mumin0);
  LIPBasis poly = fem.get_basis(iel);
  const size_t Ni = poly.nbf(), Np = Ni * Ni;
  t.Ni = Ni;
  t.Np = Np;
  t.nq = nq;
  t.Nlm = Nlm;
  t.bb0.zeros(Np, nq);
  t.bbs.zeros(Np, nq * nq);
  t.wQ.assign(2 * Nlm * nq, 0.0);
  t.wP.assign(2 * Nlm * nq * nq, 0.0);
  Vec mu0(nq);
  for (size_t q = 0; q < nq; q++) mu0[q] = mumid0 + mulen0 * xq[q];
  Mat bf0 = poly.eval_dnf(xq, 0, mulen0);
  const size_t tab = (size_t)ld * (Mmax + 1);
  parallel_for(nq, [&](size_t isub) {
    double mumin = (isub == 0) ? mumin0 : mu0[isub - 1];
    double mumax = mu0[isub];
    double mumid = 0.5 * (mumax + mumin), mulen = 0.5 * (mumax - mumin);
    Vec xpoly(nq);
    std::vector<double> Pt(tab), Qt(tab);
    for (size_t q = 0; q < nq; q++) {
      double mu = mumid + mulen * xq[q];
      double ch = std::cosh(mu);
      size_t s = isub * nq + q;
      double w = wq[q] * mulen * std::sinh(mu);
      xpoly[q] = (mu - mumid0) / mulen0;
      legPQ(ch, Pt.data(), Qt.data());
      for (size_t ilm = 0; ilm < Nlm; ilm++) {
        double pl = clean(Pt[(size_t)lm_map[ilm].second * ld + lm_map[ilm].first]);
        t.wP[(0 * Nlm + ilm) * nq * nq + s] = w * pl;
        t.wP[(1 * Nlm + ilm) * nq * nq + s] = w * pl * ch * ch;
      }
    }
    Mat bf = poly.eval_dnf(xpoly, 0, mulen0);
    for (size_t q = 0; q < nq; q++)
      for (size_t j = 0; j < Ni; j++)
        for (size_t i = 0; i < Ni; i++) t.bbs(j * Ni + i, isub * nq + q) = bf(q, i) * bf(q, j);
  }, provider ? 1 : 0);
  std::vector<double> Pt(tab), Qt(tab);
  for (size_t q = 0; q < nq; q++) {
    double ch = std::cosh(mu0[q]);
    double w = wq[q] * mulen0 * std::sinh(mu0[q]);
    legPQ(ch, Pt.data(), Qt.data());
    for (size_t ilm = 0; ilm < Nlm; ilm++) {
      double ql = clean(Qt[(size_t)lm_map[ilm].second * ld + lm_map[ilm].first]);
      t.wQ[(0 * Nlm + ilm) * nq + q] = w * ql;
      t.wQ[(1 * Nlm + ilm) * nq + q] = w * ql * ch * ch;
    }
    for (size_t j = 0; j < Ni; j++)
      for (size_t i = 0; i < Ni; i++) t.bb0(j * Ni + i, q) = bf0(q, i) * bf0(q, j);
  }
}

void TwoDBasis::compute_disjoint() {
  const size_t Ne = Nel(), Nlm = lm_map.size(), nq = xq.size();
  const int ld = Lmax + 1;
  const legendre_provider_t provider = get_legendre_provider();
  const int Lm = Lmax, Mm = Mmax, lp = lpad;
  disjoint_P0.assign(Ne * Nlm, Mat());
  disjoint_P2.assign(Ne * Nlm, Mat());
  disjoint_Q0.assign(Ne * Nlm, Mat());
  disjoint_Q2.assign(Ne * Nlm, Mat());
  const size_t tab = (size_t)ld * (Mmax + 1);
  for (size_t iel = 0; iel < Ne; iel++) {
    const double mumin0 = fem.element_begin(iel), mumax0 = fem.element_end(iel);
    const double mumid0 = 0.5 * (mumax0 + mumin0), mulen0 = 0.5 * (mumax0 - mumin0);
    LIPBasis poly = fem.get_basis(iel);
    const size_t Ni = poly.nbf(), Np = Ni * Ni;
    Mat bf0 = poly.eval_dnf(xq, 0, mulen0);
    std::vector<double> Pm(nq * tab), Qm(nq * tab);
    Vec wmain(nq), chmain(nq);
    Mat bb0(Np, nq);
    for (size_t q = 0; q < nq; q++) {
      double mu = mumid0 + mulen0 * xq[q];
      chmain[q] = std::cosh(mu);
      wmain[q] = wq[q] * mulen0 * std::sinh(mu);
      if (provider) provider(Lm, Mm, lp, chmain[q], &Pm[q * tab], &Qm[q * tab]);
      else legendre_PQ(Lm, Mm, chmain[q], &Pm[q * tab], &Qm[q * tab]);
      for (size_t j = 0; j < Ni; j++)
        for (size_t i = 0; i < Ni; i++) bb0(j * Ni + i, q) = bf0(q, i) * bf0(q, j);
    }
    for (size_t ilm = 0; ilm < Nlm; ilm++) {
      const size_t off = (size_t)lm_map[ilm].second * ld + lm_map[ilm].first;
      Mat P0(Ni, Ni), P2(Ni, Ni), Q0(Ni, Ni), Q2(Ni, Ni);
      for (size_t q = 0; q < nq; q++) {
        double pl = clean(Pm[q * tab + off]), ql = clean(Qm[q * tab + off]);
        double c2 = chmain[q] * chmain[q], w = wmain[q];
        const double *b = &bb0.d[q * Np];
        for (size_t k = 0; k < Np; k++) {
          P0.d[k] += w * pl * b[k];
          P2.d[k] += w * pl * c2 * b[k];
          Q0.d[k] += w * ql * b[k];
          Q2.d[k] += w * ql * c2 * b[k];
        }
      }
      disjoint_P0[ilm * Ne + iel] = P0;
      disjoint_P2[ilm * Ne + iel] = P2;
      disjoint_Q0[ilm * Ne + iel] = Q0;
      disjoint_Q2[ilm * Ne + iel] = Q2;
    }
  }
  have_disjoint = true;
}

void TwoDBasis::compute_tei(bool exchange) {
  const size_t Ne = Nel();
  const size_t Nlm = lm_map.size();
  const size_t nq = xq.size();
  const int ld = Lmax + 1;
  // P_L^M / Q_L^M evaluation (own implementation unless a test installed the reference's library)
  const legendre_provider_t provider = get_legendre_provider();
  const int Lm = Lmax, Mm = Mmax, lp = lpad;
  auto legPQ = [provider, Lm, Mm, lp](double xi, double *P, double *Q) {
    if (provider)
      provider(Lm, Mm, lp, xi, P, Q);
    else
      legendre_PQ(Lm, Mm, xi, P, Q);
  };

  disjoint_P0.assign(Ne * Nlm, Mat());
  disjoint_P2.assign(Ne * Nlm, Mat());
  disjoint_Q0.assign(Ne * Nlm, Mat());
  disjoint_Q2.assign(Ne * Nlm, Mat());
  prim_tei00.assign(Ne * Nlm, Mat());
  prim_tei02.assign(Ne * Nlm, Mat());
  prim_tei20.assign(Ne * Nlm, Mat());
  prim_tei22.assign(Ne * Nlm, Mat());

  for (size_t iel = 0; iel < Ne; iel++) {
    const double mumin0 = fem.element_begin(iel), mumax0 = fem.element_end(iel);
    const double mumid0 = 0.5 * (mumax0 + mumin0), mulen0 = 0.5 * (mumax0 - mumin0);
    LIPBasis poly = fem.get_basis(iel);
    const size_t Ni = poly.nbf();
    const size_t Np = Ni * Ni;

    // ---- points: main rule and the nq sub-interval rules (quadrature.cpp:61-77, basis.cpp:229-264)
    Vec mu0(nq);
    for (size_t q = 0; q < nq; q++) mu0[q] = mumid0 + mulen0 * xq[q];
    Mat bf0 = poly.eval_dnf(xq, 0, mulen0);  // nq x Ni

    // Legendre values at the main points, all (L,M)
    std::vector<double> Pm(nq * ld * (Mmax + 1)), Qm(nq * ld * (Mmax + 1));
    // and at the sub-interval points
    std::vector<double> Ps(nq * nq * ld * (Mmax + 1));
    Vec wsub(nq * nq), chsub(nq * nq);  // w*mulen*sinh(mu), cosh(mu)
    Mat bbs(Np, nq * nq);               // products B_i B_j at sub-interval points, column = point
    parallel_for(nq, [&](size_t isub) {
      double mumin = (isub == 0) ? mumin0 : mu0[isub - 1];
      double mumax = mu0[isub];
      double mumid = 0.5 * (mumax + mumin), mulen = 0.5 * (mumax - mumin);
      Vec xpoly(nq);
      std::vector<double> Qdummy(ld * (Mmax + 1));
      for (size_t q = 0; q < nq; q++) {
        double mu = mumid + mulen * xq[q];
        double ch = std::cosh(mu);
        size_t s = isub * nq + q;
        wsub[s] = wq[q] * mulen * std::sinh(mu);
        chsub[s] = ch;
        xpoly[q] = (mu - mumid0) / mulen0;
        legPQ(ch, &Ps[s * ld * (Mmax + 1)], Qdummy.data());
      }
      Mat bf = poly.eval_dnf(xpoly, 0, mulen0);
      for (size_t q = 0; q < nq; q++)
        for (size_t j = 0; j < Ni; j++)
          for (size_t i = 0; i < Ni; i++) bbs(j * Ni + i, isub * nq + q) = bf(q, i) * bf(q, j);
    }, provider ? 1 : 0);  // an installed test provider (Fortran library) is not re-entrant
    Vec wmain(nq), chmain(nq);
    for (size_t q = 0; q < nq; q++) {
      chmain[q] = std::cosh(mu0[q]);
      wmain[q] = wq[q] * mulen0 * std::sinh(mu0[q]);
      legPQ(chmain[q], &Pm[q * ld * (Mmax + 1)], &Qm[q * ld * (Mmax + 1)]);
    }
    Mat bb0(Np, nq);  // products at main points
    for (size_t q = 0; q < nq; q++)
      for (size_t j = 0; j < Ni; j++)
        for (size_t i = 0; i < Ni; i++) bb0(j * Ni + i, q) = bf0(q, i) * bf0(q, j);

    parallel_for(Nlm, [&](size_t ilm) {
      const int L = lm_map[ilm].first, M = lm_map[ilm].second;
      const size_t off = (size_t)M * ld + L;

      // ---- disjoint integrals (basis.cpp:193-211): \int B_i B_j sinh cosh^k {P|Q}_L^M
      Mat P0(Ni, Ni), P2(Ni, Ni), Q0(Ni, Ni), Q2(Ni, Ni);
      for (size_t q = 0; q < nq; q++) {
        double pl = clean(Pm[q * ld * (Mmax + 1) + off]), ql = clean(Qm[q * ld * (Mmax + 1) + off]);
        double c2 = chmain[q] * chmain[q];
        double w = wmain[q];
        const double *b = &bb0.d[q * Np];
        for (size_t k = 0; k < Np; k++) {
          P0.d[k] += w * pl * b[k];
          P2.d[k] += w * pl * c2 * b[k];
          Q0.d[k] += w * ql * b[k];
          Q2.d[k] += w * ql * c2 * b[k];
        }
      }
      disjoint_P0[ilm * Ne + iel] = P0;
      disjoint_P2[ilm * Ne + iel] = P2;
      disjoint_Q0[ilm * Ne + iel] = Q0;
      disjoint_Q2[ilm * Ne + iel] = Q2;

      // ---- in-element integrals (quadrature.cpp:22-123)
      // inner_l(q,(ij)) = \int_{mumin0}^{mu_q} sinh cosh^l P_L^M B_i B_j, accumulated sub-interval by sub-interval
      Mat inner[2];
      for (int il = 0; il < 2; il++) {
        inner[il].zeros(Np, nq);  // column q
        Vec acc(Np, 0.0);
        for (size_t isub = 0; isub < nq; isub++) {
          for (size_t q = 0; q < nq; q++) {
            size_t s = isub * nq + q;
            double w = wsub[s] * clean(Ps[s * ld * (Mmax + 1) + off]);
            if (il == 1) w *= chsub[s] * chsub[s];
            const double *b = &bbs.d[s * Np];
            for (size_t k = 0; k < Np; k++) acc[k] += w * b[k];
          }
          for (size_t k = 0; k < Np; k++) inner[il].d[isub * Np + k] = acc[k];
        }
      }
      // wrk(k,l)((ij),(i'j')) = sum_q w_q sinh cosh^k Q_L^M B_iB_j(q) inner_l(q,(i'j'))
      Mat wrk[2][2];
      for (int ik = 0; ik < 2; ik++)
        for (int il = 0; il < 2; il++) {
          Mat &W = wrk[ik][il];
          W.zeros(Np, Np);
          for (size_t q = 0; q < nq; q++) {
            double w = wmain[q] * clean(Qm[q * ld * (Mmax + 1) + off]);
            if (ik == 1) w *= chmain[q] * chmain[q];
            const double *b = &bb0.d[q * Np];
            const double *in = &inner[il].d[q * Np];
            for (size_t c = 0; c < Np; c++) {
              double wi = w * in[c];
              double *col = &W.d[c * Np];
              for (size_t r = 0; r < Np; r++) col[r] += b[r] * wi;
            }
          }
        }
      // twoe_integral(k,l) = wrk(k,l) + wrk(l,k)^T
      Mat t00(wrk[0][0] + wrk[0][0].t());
      Mat t02(wrk[0][1] + wrk[1][0].t());
      Mat t20(wrk[1][0] + wrk[0][1].t());
      Mat t22(wrk[1][1] + wrk[1][1].t());
      prim_tei00[ilm * Ne + iel] = t00;
      prim_tei02[ilm * Ne + iel] = t02;
      prim_tei20[ilm * Ne + iel] = t20;
      prim_tei22[ilm * Ne + iel] = t22;
    });
  }
  have_tei = true;
  have_disjoint = true;

  if (exchange) {
    prim_ktei00.assign(Ne * Nlm, Mat());
    prim_ktei02.assign(Ne * Nlm, Mat());
    prim_ktei20.assign(Ne * Nlm, Mat());
    prim_ktei22.assign(Ne * Nlm, Mat());
    parallel_for(Ne * Nlm, [&](size_t idx) {
      size_t iel = idx % Ne;
      size_t Ni = fem.nprim(iel);
      prim_ktei00[idx] = exchange_tei(prim_tei00[idx], Ni, Ni, Ni, Ni);
      prim_ktei02[idx] = exchange_tei(prim_tei02[idx], Ni, Ni, Ni, Ni);
      prim_ktei20[idx] = exchange_tei(prim_tei20[idx], Ni, Ni, Ni, Ni);
      prim_ktei22[idx] = exchange_tei(prim_tei22[idx], Ni, Ni, Ni, Ni);
    });
    have_ktei = true;
  }
}

}  // namespace diatomic
}  // namespace helfem
