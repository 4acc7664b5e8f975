#include "atomic_basis.h"
#include "diatomic_basis.h"  // exchange_tei
#include "parallel.h"
#include <algorithm>
#include <cmath>

namespace helfem {
namespace atomic {

void angular_basis(int lmax, int mmax, IVec &lval, IVec &mval) {
  lval.clear();
  mval.clear();
  for (int mabs = 0; mabs <= mmax; mabs++)
    for (int l = mabs; l <= lmax; l++) {
      lval.push_back(l);
      mval.push_back(mabs);
      if (mabs > 0) {
        lval.push_back(l);
        mval.push_back(-mabs);
      }
    }
}

TwoDBasis::TwoDBasis(int Z_, int nnodes_, int n_quad, const Vec &bval, const IVec &lval_, const IVec &mval_)
    : Z(Z_), nnodes(nnodes_), lval(lval_), mval(mval_) {
  if (nnodes < 2) throw std::logic_error("Can't have finite element basis with less than two nodes per element.\n");
  if (bval.empty() || bval[0] != 0.0) throw std::logic_error("radial grid must start from zero\n");
  LIPBasis poly(lobatto_nodes(nnodes));
  // functions vanish at the nucleus (u = r R(r)) and at the practical infinity (TwoDBasis.cpp:48-52)
  fem = FEMBasis(poly, bval, true, true);
  chebyshev_rule(n_quad, xq, wq);
}

int TwoDBasis::N_L() const { return 2 * *std::max_element(lval.begin(), lval.end()) + 1; }
int TwoDBasis::Mmax() const {
  return *std::max_element(mval.begin(), mval.end()) - *std::min_element(mval.begin(), mval.end());
}

std::vector<size_t> TwoDBasis::m_indices(int m) const {
  std::vector<size_t> idx;
  for (size_t i = 0; i < mval.size(); i++)
    if (mval[i] == m)
      for (size_t j = 0; j < Nrad(); j++) idx.push_back(i * Nrad() + j);
  return idx;
}

std::vector<size_t> TwoDBasis::lm_indices(int l, int m) const {
  std::vector<size_t> idx;
  for (size_t i = 0; i < mval.size(); i++)
    if (mval[i] == m && lval[i] == l)
      for (size_t j = 0; j < Nrad(); j++) idx.push_back(i * Nrad() + j);
  return idx;
}

std::vector<std::vector<size_t> > TwoDBasis::get_sym_idx(int symm) const {
  // TwoDBasis.cpp:202-224; symm 1: one block per m in order of first appearance (find_unique), symm 2: per (l,m)
  std::vector<std::vector<size_t> > idx;
  if (symm == 0) {
    idx.resize(1);
    for (size_t i = 0; i < Nbf(); i++) idx[0].push_back(i);
  } else if (symm == 1) {
    std::vector<int> seen;
    for (int m : mval)
      if (std::find(seen.begin(), seen.end(), m) == seen.end()) seen.push_back(m);
    for (int m : seen) idx.push_back(m_indices(m));
  } else if (symm == 2) {
    for (size_t i = 0; i < mval.size(); i++) idx.push_back(lm_indices(lval[i], mval[i]));
  } else
    throw std::logic_error("Unknown symmetry\n");
  return idx;
}

// g_i(x) = B_i(x)/(x-x_0) and its derivative for the enabled functions of the first element
static void reduced_lip(const LIPBasis &p, const Vec &x, Mat &g, Mat &dg) {
  const Vec &x0 = p.x0;
  const size_t np = x0.size();
  g.zeros(x.size(), p.enabled.size());
  dg.zeros(x.size(), p.enabled.size());
  for (size_t ix = 0; ix < x.size(); ix++)
    for (size_t c = 0; c < p.enabled.size(); c++) {
      size_t fi = p.enabled[c];
      if (fi == 0) throw std::logic_error("reduced LIP needs the first function dropped");
      double val = 1.0 / (x0[fi] - x0[0]);
      for (size_t ip = 1; ip < np; ip++)
        if (ip != fi) val *= (x[ix] - x0[ip]) / (x0[fi] - x0[ip]);
      double der = 0.0;
      for (size_t d1 = 1; d1 < np; d1++) {
        if (d1 == fi) continue;
        double t = 1.0 / (x0[fi] - x0[0]);
        for (size_t ip = 1; ip < np; ip++) {
          if (ip == fi || ip == d1) continue;
          t *= (x[ix] - x0[ip]) / (x0[fi] - x0[ip]);
        }
        der += t / (x0[fi] - x0[d1]);
      }
      g(ix, c) = val;
      dg(ix, c) = der;
    }
}

Mat TwoDBasis::get_bf(size_t iel) const {
  double sc = fem.scaling_factor(iel);
  if (iel == 0) {
    Mat g, dg;
    reduced_lip(fem.get_basis(0), xq, g, dg);
    for (auto &v : g.d) v /= sc;  // r = sc (x - x_0)
    return g;
  }
  Mat f = fem.eval_dnf(xq, 0, iel);
  Vec r = get_r(iel);
  for (size_t j = 0; j < f.n_cols; j++)
    for (size_t i = 0; i < f.n_rows; i++) f(i, j) /= r[i];
  return f;
}

Mat TwoDBasis::get_df(size_t iel) const {
  double sc = fem.scaling_factor(iel);
  if (iel == 0) {
    Mat g, dg;
    reduced_lip(fem.get_basis(0), xq, g, dg);
    for (auto &v : dg.d) v /= sc * sc;
    return dg;
  }
  // RadialBasis.cpp:676-700: (-f/r + f')/r
  Mat f = fem.eval_dnf(xq, 0, iel), d = fem.eval_dnf(xq, 1, iel);
  Vec r = get_r(iel);
  for (size_t j = 0; j < f.n_cols; j++)
    for (size_t i = 0; i < f.n_rows; i++) {
      double invr = 1.0 / r[i];
      d(i, j) = (-f(i, j) * invr + d(i, j)) * invr;
    }
  return d;
}

Vec TwoDBasis::get_wrad(size_t iel) const {
  Vec w(wq);
  for (auto &x : w) x *= fem.scaling_factor(iel);
  return w;
}

Mat TwoDBasis::radial_integral(int Rexp, size_t iel) const {
  Mat bf = get_bf(iel);
  Vec r = get_r(iel), w = get_wrad(iel);
  Mat wbf(bf);
  for (size_t q = 0; q < bf.n_rows; q++) {
    double wp = w[q] * std::pow(r[q], Rexp + 2);
    for (size_t j = 0; j < bf.n_cols; j++) wbf(q, j) *= wp;
  }
  return matmul(wbf, true, bf, false);
}

static Mat assemble_radial(const TwoDBasis &b, const std::function<Mat(size_t)> &el) {
  Mat M(b.Nrad(), b.Nrad());
  for (size_t iel = 0; iel < b.Nel(); iel++) {
    Mat m = el(iel);
    size_t i0 = b.fem.first[iel];
    for (size_t j = 0; j < m.n_cols; j++)
      for (size_t i = 0; i < m.n_rows; i++) M(i0 + i, i0 + j) += m(i, j);
  }
  return M;
}

static Mat place_diag(const TwoDBasis &b, const std::vector<Mat> &rad) {
  size_t R = b.Nrad();
  Mat O(b.Nbf(), b.Nbf());
  for (size_t a = 0; a < b.Nang(); a++)
    for (size_t j = 0; j < R; j++)
      for (size_t i = 0; i < R; i++) O(a * R + i, a * R + j) = rad[a](i, j);
  return O;
}

Mat TwoDBasis::overlap() const {
  Mat Orad = assemble_radial(*this, [this](size_t iel) { return radial_integral(0, iel); });
  return place_diag(*this, std::vector<Mat>(Nang(), Orad));
}

Mat TwoDBasis::overlap(const TwoDBasis &rh) const {
  // radial part: int (B_i / r)(B'_j / r) r^2 dr = int B_i B'_j dr over the intersections of the two element grids
  Vec xp, wp;
  chebyshev_rule((int)std::max(xq.size(), rh.xq.size()), xp, wp);
  Mat Srad(fem.nbf(), rh.fem.nbf());
  for (size_t iel = 0; iel < fem.nelem(); iel++)
    for (size_t jel = 0; jel < rh.fem.nelem(); jel++) {
      const double imin = fem.element_begin(iel), imax = fem.element_end(iel);
      const double jmin = rh.fem.element_begin(jel), jmax = rh.fem.element_end(jel);
      if (!((jmin >= imin && jmin < imax) || (imin >= jmin && imin < jmax))) continue;
      const double a = std::max(imin, jmin), b = std::min(imax, jmax);
      const double mid = 0.5 * (b + a), len = 0.5 * (b - a);
      Vec xi(xp.size()), xj(xp.size());
      for (size_t q = 0; q < xp.size(); q++) {
        const double r = mid + len * xp[q];
        xi[q] = (r - fem.element_midpoint(iel)) / fem.scaling_factor(iel);
        xj[q] = (r - rh.fem.element_midpoint(jel)) / rh.fem.scaling_factor(jel);
      }
      const Mat ibf = fem.eval_dnf(xi, 0, iel), jbf = rh.fem.eval_dnf(xj, 0, jel);
      const size_t i0 = fem.first[iel], j0 = rh.fem.first[jel];
      for (size_t fj = 0; fj < jbf.n_cols; fj++)
        for (size_t fi = 0; fi < ibf.n_cols; fi++) {
          double acc = 0.0;
          for (size_t q = 0; q < xp.size(); q++) acc += wp[q] * len * ibf(q, fi) * jbf(q, fj);
          Srad(i0 + fi, j0 + fj) += acc;
        }
    }
  // angular part: the same (l, m) only (atomic/TwoDBasis.cpp:338-341)
  const size_t R = Nrad(), R2 = rh.Nrad();
  Mat S(Nbf(), rh.Nbf());
  for (size_t ia = 0; ia < Nang(); ia++)
    for (size_t ja = 0; ja < rh.Nang(); ja++)
      if (lval[ia] == rh.lval[ja] && mval[ia] == rh.mval[ja])
        for (size_t j = 0; j < R2; j++)
          for (size_t i = 0; i < R; i++) S(ia * R + i, ja * R2 + j) = Srad(i, j);
  return S;
}

Mat TwoDBasis::kinetic() const {
  // TwoDBasis.cpp:349-380: 1/2 int B'B' + l(l+1) 1/2 int (B/r)(B/r)
  std::function<double(double)> none;
  Mat Trad = assemble_radial(*this, [&](size_t iel) { return 0.5 * fem.matrix_element(iel, 1, 1, xq, wq, none); });
  Mat Tl = assemble_radial(*this, [&](size_t iel) {
    Mat bf = get_bf(iel);
    Vec w = get_wrad(iel);
    Mat wbf(bf);
    for (size_t q = 0; q < bf.n_rows; q++)
      for (size_t j = 0; j < bf.n_cols; j++) wbf(q, j) *= w[q];
    return 0.5 * matmul(wbf, true, bf, false);
  });
  std::vector<Mat> rad;
  for (size_t a = 0; a < Nang(); a++) rad.push_back(Trad + (double)(lval[a] * (lval[a] + 1)) * Tl);
  return place_diag(*this, rad);
}

Mat TwoDBasis::nuclear() const {
  Mat Vrad = assemble_radial(*this, [this](size_t iel) { return radial_integral(-1, iel); });
  return place_diag(*this, std::vector<Mat>(Nang(), (-(double)Z) * Vrad));
}

// quadrature::twoe_inner_integral + twoe_integral / yukawa_integral (libhelfem/src/quadrature.cpp:22-169) for a kernel
// g(r<, r>) = fsmallbig(r<, r>) whose r> dependence is fbig(r>): the inner integral up to each outer quadrature point
// is built segment by segment (each segment with a fresh rule, scaled by the kernel at its upper end) and carried
// to the next point with the ratio fbig(r_ip)/fbig(r_ip-1).
static Mat twoe_integral_kernel(double rmin, double rmax, const Vec &xq, const Vec &wq, const LIPBasis &poly,
                                const std::function<double(double, double)> &fsmallbig,
                                const std::function<double(double)> &fbig) {
  const size_t nq = xq.size();
  const double rmid0 = 0.5 * (rmax + rmin), rlen0 = 0.5 * (rmax - rmin);
  const size_t Ni = poly.nbf(), Np = Ni * Ni;
  Vec r0(nq);
  for (size_t q = 0; q < nq; q++) r0[q] = rmid0 + rlen0 * xq[q];
  Mat inner(Np, nq);
  for (size_t ip = 0; ip < nq; ip++) {
    double a = (ip == 0) ? rmin : r0[ip - 1], bnd = r0[ip];
    double rmid = 0.5 * (bnd + a), rlen = 0.5 * (bnd - a);
    Vec xpoly(nq), wp(nq);
    for (size_t q = 0; q < nq; q++) {
      double r = rmid + rlen * xq[q];
      wp[q] = wq[q] * fsmallbig(r, bnd) * rlen;
      xpoly[q] = (r - rmid0) / rlen0;
    }
    Mat bf = poly.eval_dnf(xpoly, 0, rlen0);
    double *col = &inner.d[ip * Np];
    for (size_t q = 0; q < nq; q++)
      for (size_t l = 0; l < Ni; l++)
        for (size_t k = 0; k < Ni; k++) col[l * Ni + k] += wp[q] * bf(q, k) * bf(q, l);
    if (ip > 0) {
      double ratio = fbig(r0[ip]) / fbig(r0[ip - 1]);
      const double *prev = &inner.d[(ip - 1) * Np];
      for (size_t k = 0; k < Np; k++) col[k] += prev[k] * ratio;
    }
  }
  Mat bf0 = poly.eval_dnf(xq, 0, rlen0);
  Mat ints(Np, Np);
  for (size_t q = 0; q < nq; q++) {
    double w = wq[q] * rlen0;
    const double *in = &inner.d[q * Np];
    for (size_t c = 0; c < Np; c++) {
      double wi = w * in[c];
      double *colp = &ints.d[c * Np];
      for (size_t fj = 0; fj < Ni; fj++)
        for (size_t fi = 0; fi < Ni; fi++) colp[fi * Ni + fj] += bf0(q, fi) * bf0(q, fj) * wi;
    }
  }
  return ints + ints.t();
}

Mat twoe_integral(double rmin, double rmax, const Vec &xq, const Vec &wq, const LIPBasis &poly, int L) {
  // r_<^L / r_>^{L+1}: inner(q,(kl)) = r_q^{-L-1} int_{rmin}^{r_q} r^L B_k B_l dr
  return twoe_integral_kernel(
      rmin, rmax, xq, wq, poly, [L](double r, double R) { return std::pow(r / R, (double)L) / R; },
      [L](double r) { return std::pow(r, -(double)L - 1.0); });
}

Mat yukawa_integral(double rmin, double rmax, const Vec &xq, const Vec &wq, const LIPBasis &poly, int L, double lambda) {
  return twoe_integral_kernel(
      rmin, rmax, xq, wq, poly,
      [L, lambda](double r, double R) { return bessel_il(r * lambda, L) * bessel_kl(R * lambda, L); },
      [L, lambda](double r) { return bessel_kl(r * lambda, L); });
}

void TwoDBasis::compute_disjoint() {
  const size_t Ne = Nel(), NL = (size_t)N_L();
  disjoint_L.assign(Ne * NL, Mat());
  disjoint_m1L.assign(Ne * NL, Mat());
  for (size_t L = 0; L < NL; L++)
    for (size_t iel = 0; iel < Ne; iel++) {
      disjoint_L[L * Ne + iel] = radial_integral((int)L, iel);
      disjoint_m1L[L * Ne + iel] = radial_integral(-(int)L - 1, iel);
    }
  have_disjoint = true;
}

void TwoDBasis::tei_element_tables(size_t iel, diatomic::TwoDBasis::TeiElementTables &t) const {
  const size_t nq = xq.size(), NL = (size_t)N_L();
  const double rmin = fem.element_begin(iel), rmax = fem.element_end(iel);
  const double rmid0 = 0.5 * (rmax + rmin), rlen0 = 0.5 * (rmax - rmin);
  LIPBasis poly = fem.get_basis(iel);
  const size_t Ni = poly.nbf(), Np = Ni * Ni;
  t.Ni = Ni;
  t.Np = Np;
  t.nq = nq;
  t.Nlm = NL;
  t.bb0.zeros(Np, nq);
  t.bbs.zeros(Np, nq * nq);
  t.wQ.assign(NL * nq, 0.0);
  t.wP.assign(NL * nq * nq, 0.0);
  Vec r0(nq);
  for (size_t q = 0; q < nq; q++) r0[q] = rmid0 + rlen0 * xq[q];
  Mat bf0 = poly.eval_dnf(xq, 0, rlen0);
  for (size_t q = 0; q < nq; q++) {
    const double w = wq[q] * rlen0;
    for (size_t L = 0; L < NL; L++) t.wQ[L * nq + q] = w * std::pow(r0[q], -(double)L - 1.0);
    for (size_t j = 0; j < Ni; j++)
      for (size_t i = 0; i < Ni; i++) t.bb0(j * Ni + i, q) = bf0(q, i) * bf0(q, j);
  }
  for (size_t isub = 0; isub < nq; isub++) {
    const double a = (isub == 0) ? rmin : r0[isub - 1], bnd = r0[isub];
    const double rmid = 0.5 * (bnd + a), rlen = 0.5 * (bnd - a);
    Vec xpoly(nq);
    for (size_t q = 0; q < nq; q++) {
      const double r = rmid + rlen * xq[q];
      xpoly[q] = (r - rmid0) / rlen0;
      const double w = wq[q] * rlen;
      for (size_t L = 0; L < NL; L++) t.wP[L * nq * nq + isub * nq + q] = w * std::pow(r, (double)L);
    }
    Mat bf = poly.eval_dnf(xpoly, 0, rlen0);
    for (size_t q = 0; q < nq; q++)
      for (size_t j = 0; j < Ni; j++)
        for (size_t i = 0; i < Ni; i++) t.bbs(j * Ni + i, isub * nq + q) = bf(q, i) * bf(q, j);
  }
}

void TwoDBasis::compute_tei(bool exchange) {
  const size_t Ne = Nel(), NL = (size_t)N_L();
  compute_disjoint();
  prim_tei.assign(Ne * NL, Mat());
  parallel_for(Ne * NL, [&](size_t idx) {
    const size_t L = idx / Ne, iel = idx % Ne;
    prim_tei[L * Ne + iel] =
        twoe_integral(fem.element_begin(iel), fem.element_end(iel), xq, wq, fem.get_basis(iel), (int)L);
  });
  have_tei = true;
  if (exchange) {
    prim_ktei.assign(Ne * NL, Mat());
    for (size_t idx = 0; idx < Ne * NL; idx++) {
      size_t Ni = fem.nprim(idx % Ne);
      prim_ktei[idx] = diatomic::exchange_tei(prim_tei[idx], Ni, Ni, Ni, Ni);
    }
    have_ktei = true;
  }
}

Mat TwoDBasis::model_potential(const ModelPotential &pot) const {
  // RadialBasis::model_potential: (B_i/r)(B_j/r) V(r) r^2 = B_i B_j V(r)
  Mat Vrad = assemble_radial(*this, [&](size_t iel) {
    return fem.matrix_element(iel, 0, 0, xq, wq, [&pot](double r) { return pot.V(r); });
  });
  return place_diag(*this, std::vector<Mat>(Nang(), Vrad));
}

Mat TwoDBasis::bessel_il_integral(int L, double lambda, size_t iel) const {
  return fem.matrix_element(iel, 0, 0, xq, wq, [L, lambda](double r) { return bessel_il(r * lambda, L); });
}

Mat TwoDBasis::bessel_kl_integral(int L, double lambda, size_t iel) const {
  return fem.matrix_element(iel, 0, 0, xq, wq, [L, lambda](double r) { return bessel_kl(r * lambda, L); });
}

Mat TwoDBasis::erfc_integral(int L, double mu, size_t iel, size_t kel) const {
  // RadialBasis.cpp:502-558 + quadrature.cpp:171-222.  The kernel has a cusp at r = r', so the in-element integral
  // uses nq sub-intervals (each with its own nq-point rule) for the second coordinate; distinct elements use one.
  const size_t nq = xq.size();
  const size_t Nint = (iel == kel) ? nq : 1;
  Vec xk(nq * Nint), wk(nq * Nint);
  for (size_t ii = 0; ii < Nint; ii++) {
    double istart = ii * 2.0 / Nint - 1.0, iend = (ii + 1) * 2.0 / Nint - 1.0;
    double imid = 0.5 * (iend + istart), ilen = 0.5 * (iend - istart);
    for (size_t q = 0; q < nq; q++) {
      xk[ii * nq + q] = imid + xq[q] * ilen;
      wk[ii * nq + q] = wq[q] * ilen;
    }
  }
  Mat ibf = fem.eval_dnf(xq, 0, iel), kbf = fem.eval_dnf(xk, 0, kel);
  const double rleni = fem.scaling_factor(iel), rlenk = fem.scaling_factor(kel);
  Vec ri = fem.eval_coord(xq, iel), rk = fem.eval_coord(xk, kel);
  const size_t Ni = ibf.n_cols, Nk = kbf.n_cols, nqi = nq, nqk = nq * Nint;
  // Green's function and weighted product functions
  Mat Fn(nqi, nqk);
  for (size_t k = 0; k < nqk; k++)
    for (size_t i = 0; i < nqi; i++) Fn(i, k) = erfc_Phi(L, mu * ri[i], mu * rk[k]);
  Mat pij(nqi, Ni * Ni), pkl(nqk, Nk * Nk);
  for (size_t fi = 0; fi < Ni; fi++)
    for (size_t fj = 0; fj < Ni; fj++)
      for (size_t q = 0; q < nqi; q++) pij(q, fi * Ni + fj) = ibf(q, fi) * ibf(q, fj) * wq[q] * rleni;
  for (size_t fi = 0; fi < Nk; fi++)
    for (size_t fj = 0; fj < Nk; fj++)
      for (size_t q = 0; q < nqk; q++) pkl(q, fi * Nk + fj) = kbf(q, fi) * kbf(q, fj) * wk[q] * rlenk;
  Mat tei = matmul(pij, true, matmul(Fn, false, pkl, false), false);
  if (iel == kel) tei = 0.5 * (tei + tei.t());
  return tei;
}

void TwoDBasis::compute_yukawa(double lambda) {
  // TwoDBasis.cpp:741-778
  rs_kind = 1;
  rs_lambda = lambda;
  const size_t Ne = Nel(), NL = (size_t)N_L();
  disjoint_iL.assign(Ne * NL, Mat());
  disjoint_kL.assign(Ne * NL, Mat());
  rs_tei.assign(Ne * NL, Mat());
  rs_ktei.assign(Ne * NL, Mat());
  parallel_for(Ne * NL, [&](size_t idx) {
    const size_t L = idx / Ne, iel = idx % Ne;
    disjoint_iL[idx] = bessel_il_integral((int)L, lambda, iel);
    disjoint_kL[idx] = bessel_kl_integral((int)L, lambda, iel);
    rs_tei[idx] = yukawa_integral(fem.element_begin(iel), fem.element_end(iel), xq, wq, fem.get_basis(iel), (int)L, lambda);
    size_t Ni = fem.nprim(iel);
    rs_ktei[idx] = diatomic::exchange_tei(rs_tei[idx], Ni, Ni, Ni, Ni);
  });
}

void TwoDBasis::compute_erfc(double mu) {
  // TwoDBasis.cpp:780-815
  rs_kind = 2;
  rs_lambda = mu;
  const size_t Ne = Nel(), NL = (size_t)N_L();
  disjoint_iL.clear();
  disjoint_kL.clear();
  rs_tei.assign(Ne * Ne * NL, Mat());
  rs_ktei.assign(Ne * Ne * NL, Mat());
  parallel_for(Ne * Ne * NL, [&](size_t idx) {
    const size_t L = idx / (Ne * Ne), iel = (idx / Ne) % Ne, kel = idx % Ne;
    rs_tei[idx] = erfc_integral((int)L, mu, iel, kel);
    rs_ktei[idx] = diatomic::exchange_tei(rs_tei[idx], fem.nprim(iel), fem.nprim(iel), fem.nprim(kel), fem.nprim(kel));
  });
}

}  // namespace atomic
}  // namespace helfem

