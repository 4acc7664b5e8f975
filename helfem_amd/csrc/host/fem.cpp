#include "fem.h"
#include <algorithm>
#include <cmath>
#include <sstream>

namespace helfem {

void chebyshev_rule(int n, Vec &x, Vec &w) {
  // reference: libhelfem/src/chebyshev.cpp:22-53
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  double oonpp = 1.0 / (n + 1.0);
  for (int i = 1; i <= n; i++) {
    double sine = sin(i * M_PI * oonpp);
    double sinesq = sine * sine;
    double cosine = cos(i * M_PI * oonpp);
    w[i - 1] = 16.0 / 3.0 / (n + 1.0) * sinesq * sinesq;
    x[i - 1] = 1.0 - 2.0 * i * oonpp + M_2_PI * (1.0 + 2.0 / 3.0 * sinesq) * cosine * sine;
  }
  std::reverse(x.begin(), x.end());
  std::reverse(w.begin(), w.end());
}

Vec lobatto_nodes(int n) {
  // The reference tabulates the nodes to 30 digits for n<20 and Newton-iterates in double for
  // n>=20 (libhelfem/src/lobatto.cpp:588-743).  Here the same Newton iteration
  //   x <- x - (x P_{n-1}(x) - P_{n-2}(x)) / (n P_{n-1}(x))
  // is run in long double for every n, which reproduces the tabulated values to the last bit
  // of a double, and the result is symmetrised exactly.
  if (n < 2) throw std::runtime_error("Lobatto rule needs n>=2");
  std::vector<long double> x(n);
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 0; i < n; i++) x[i] = cosl(pi * i / (n - 1));
  for (int it = 0; it < 200; it++) {
    long double err = 0.0L;
    for (int i = 0; i < n; i++) {
      long double pm2 = 1.0L, pm1 = x[i];  // P_0, P_1
      for (int j = 2; j <= n - 1; j++) {
        long double pj = ((2 * j - 1) * x[i] * pm1 - (j - 1) * pm2) / j;
        pm2 = pm1;
        pm1 = pj;
      }
      // now pm1 = P_{n-1}, pm2 = P_{n-2}   (n=2: P_1, P_0)
      long double xn = x[i] - (x[i] * pm1 - pm2) / (n * pm1);
      err = std::max(err, fabsl(xn - x[i]));
      x[i] = xn;
    }
    if (err < 1e-19L) break;
  }
  Vec r(n);
  for (int i = 0; i < n; i++) r[i] = (double)x[n - 1 - i];  // ascending
  for (int i = 0; i < n / 2; i++) {
    double a = 0.5 * (r[n - 1 - i] - r[i]);
    r[i] = -a;
    r[n - 1 - i] = a;
  }
  if (n % 2) r[n / 2] = 0.0;
  r[0] = -1.0;
  r[n - 1] = 1.0;
  return r;
}

double arcosh(double x) { return log(x + sqrt(x * x - 1.0)); }

Vec get_grid(double rmax, int num_el, int igrid, double zexp) {
  // reference: libhelfem/src/grid.cpp:18-87
  Vec bval(num_el + 1, 0.0);
  switch (igrid) {
    case 1:
      for (int i = 0; i <= num_el; i++) bval[i] = rmax * i / num_el;
      break;
    case 2:
      for (int i = 0; i <= num_el; i++) bval[i] = i * i * rmax / (num_el * num_el);
      break;
    case 3:
      for (int i = 0; i <= num_el; i++) bval[i] = rmax * std::pow(i * 1.0 / num_el, zexp);
      break;
    case 4: {
      double top = std::pow(log(rmax + 1), 1.0 / zexp);
      for (int i = 0; i <= num_el; i++) {
        double t = top * i / num_el;  // linspace(0,top,num_el+1)
        bval[i] = exp(std::pow(t, zexp)) - 1.0;
      }
    } break;
    case 5: {
      if (zexp <= 0.0 || zexp >= 1.0) throw std::logic_error("Invalid value for s parameter!\n");
      Vec hk(num_el);
      hk[num_el - 1] = (1.0 - zexp) / (1.0 - std::pow(zexp, num_el)) * rmax;
      for (int iel = num_el - 2; iel >= 0; iel--) hk[iel] = zexp * hk[iel + 1];
      for (int iel = 0; iel < num_el; iel++) bval[iel + 1] = bval[iel] + hk[iel];
    } break;
    default:
      throw std::logic_error("Invalid choice for grid\n");
  }
  bval[0] = 0.0;
  bval[num_el] = rmax;
  return bval;
}

LIPBasis::LIPBasis(const Vec &nodes) : x0(nodes) {
  std::sort(x0.begin(), x0.end());
  enabled.resize(x0.size());
  for (size_t i = 0; i < x0.size(); i++) enabled[i] = (int)i;
}

Mat LIPBasis::eval_dnf(const Vec &x, int n, double element_length) const {
  // reference: libhelfem/src/LIPBasis_eval.cpp (cases 0,1,2) + PolynomialBasis.cpp:175-179
  const size_t np = x0.size();
  Mat full(x.size(), np);
  for (size_t ix = 0; ix < x.size(); ix++) {
    for (size_t fi = 0; fi < np; fi++) {
      double val = 0.0;
      if (n == 0) {
        val = 1.0;
        for (size_t ip = 0; ip < np; ip++) {
          if (ip == fi) continue;
          val *= (x[ix] - x0[ip]) / (x0[fi] - x0[ip]);
        }
      } else if (n == 1) {
        for (size_t d1 = 0; d1 < np; d1++) {
          if (d1 == fi) continue;
          double t = 1.0;
          for (size_t ip = 0; ip < np; ip++) {
            if (ip == d1 || ip == fi) continue;
            t *= (x[ix] - x0[ip]) / (x0[fi] - x0[ip]);
          }
          t /= (x0[fi] - x0[d1]);
          val += t;
        }
      } else if (n == 2) {
        for (size_t d1 = 0; d1 < np; d1++) {
          if (d1 == fi) continue;
          for (size_t d2 = 0; d2 < d1; d2++) {
            if (d2 == fi) continue;
            double t = 1.0;
            for (size_t ip = 0; ip < np; ip++) {
              if (ip == d1 || ip == d2 || ip == fi) continue;
              t *= (x[ix] - x0[ip]) / (x0[fi] - x0[ip]);
            }
            t /= (x0[fi] - x0[d1]);
            t /= (x0[fi] - x0[d2]);
            val += 2 * t;
          }
        }
      } else
        throw std::logic_error("LIP derivative order not implemented");
      full(ix, fi) = val;
    }
  }
  Mat r(x.size(), enabled.size());
  double scale = std::pow(element_length, n);
  for (size_t j = 0; j < enabled.size(); j++)
    for (size_t ix = 0; ix < x.size(); ix++) r(ix, j) = full(ix, enabled[j]) / scale;
  return r;
}

FEMBasis::FEMBasis(const LIPBasis &poly_, const Vec &bval_, bool zfl, bool zfr)
    : poly(poly_), bval(bval_), zero_func_left(zfl), zero_func_right(zfr) {
  if (bval.size() < 2) throw std::logic_error("Can't update basis function list since there are no elements!\n");
  // reference: FiniteElementBasis.cpp:37-50
  size_t ne = bval.size() - 1;
  first.assign(ne, 0);
  last.assign(ne, 0);
  for (size_t iel = 0; iel < ne; iel++) {
    first[iel] = (iel == 0) ? 0 : last[iel - 1] - poly.noverlap() + 1;
    last[iel] = first[iel] + get_basis(iel).nbf() - 1;
  }
}

LIPBasis FEMBasis::get_basis(size_t iel) const {
  // reference: FiniteElementBasis.cpp:253-262
  LIPBasis p(poly);
  if (iel == 0 && zero_func_left) p.drop_first();
  if (iel == bval.size() - 2 && zero_func_right) p.drop_last();
  return p;
}

Vec FEMBasis::eval_coord(const Vec &x, size_t iel) const {
  Vec r(x.size());
  double mid = element_midpoint(iel), len = scaling_factor(iel);
  for (size_t i = 0; i < x.size(); i++) r[i] = mid + len * x[i];
  return r;
}

Mat FEMBasis::matrix_element(size_t iel, int lhder, int rhder, const Vec &xq, const Vec &wq,
                             const std::function<double(double)> &f) const {
  // reference: FiniteElementBasis.cpp:387-415 (x_left=-1, x_right=1)
  Vec r(eval_coord(xq, iel));
  Vec wp(wq.size());
  for (size_t i = 0; i < wq.size(); i++) {
    wp[i] = wq[i] * scaling_factor(iel);
    if (f) wp[i] *= f(r[i]);
  }
  Mat lh = eval_dnf(xq, lhder, iel);
  Mat rh = eval_dnf(xq, rhder, iel);
  for (size_t j = 0; j < lh.n_cols; j++)
    for (size_t i = 0; i < lh.n_rows; i++) lh(i, j) *= wp[i];
  return matmul(lh, true, rh, false);
}

Mat FEMBasis::matrix_element(int lhder, int rhder, const Vec &xq, const Vec &wq,
                             const std::function<double(double)> &f) const {
  Mat M(nbf(), nbf());
  for (size_t iel = 0; iel < nelem(); iel++) {
    Mat m = matrix_element(iel, lhder, rhder, xq, wq, f);
    size_t i0 = first[iel];
    for (size_t j = 0; j < m.n_cols; j++)
      for (size_t i = 0; i < m.n_rows; i++) M(i0 + i, i0 + j) += m(i, j);
  }
  return M;
}

}  // namespace helfem
