// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Dense symmetric eigensolver (Householder tridiagonalisation + implicit-shift QL, the classic
// EISPACK tred2/tql2 pair) standing in for arma::eig_sym (LAPACK dsyevd) which the reference
// calls in scf::eig_gsym (src/general/scf_helpers.cpp:135) and utils::invh
// (libhelfem/src/utils.cpp:172), plus the reference's thin wrappers around it.
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <stdexcept>

namespace oracle {

void eig_sym(Vec &E, Mat &C, const Mat &A) {
  const size_t n = A.n_rows;
  if (A.n_cols != n) throw std::logic_error("eig_sym: matrix not square");
  E.assign(n, 0.0);
  C.zeros(n, n);
  if (n == 0) return;
  // row-major working copy a[i*n+k]
  std::vector<double> a(n * n);
  for (size_t i = 0; i < n; i++)
    for (size_t k = 0; k < n; k++) a[i * n + k] = 0.5 * (A(i, k) + A(k, i));
  std::vector<double> d(n, 0.0), e(n, 0.0);

  // ---- Householder reduction to tridiagonal form (tred2) ----
  for (size_t i = n - 1; i >= 1; i--) {
    size_t l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (size_t k = 0; k <= l; k++) scale += fabs(a[i * n + k]);
      if (scale == 0.0)
        e[i] = a[i * n + l];
      else {
        for (size_t k = 0; k <= l; k++) {
          a[i * n + k] /= scale;
          h += a[i * n + k] * a[i * n + k];
        }
        double f = a[i * n + l];
        double g = (f >= 0.0 ? -sqrt(h) : sqrt(h));
        e[i] = scale * g;
        h -= f * g;
        a[i * n + l] = f - g;
        f = 0.0;
        for (size_t j = 0; j <= l; j++) {
          a[j * n + i] = a[i * n + j] / h;
          g = 0.0;
          for (size_t k = 0; k <= j; k++) g += a[j * n + k] * a[i * n + k];
          for (size_t k = j + 1; k <= l; k++) g += a[k * n + j] * a[i * n + k];
          e[j] = g / h;
          f += e[j] * a[i * n + j];
        }
        double hh = f / (h + h);
        for (size_t j = 0; j <= l; j++) {
          f = a[i * n + j];
          e[j] = g = e[j] - hh * f;
          for (size_t k = 0; k <= j; k++) a[j * n + k] -= (f * e[k] + g * a[i * n + k]);
        }
      }
    } else
      e[i] = a[i * n + l];
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (size_t i = 0; i < n; i++) {
    if (d[i] != 0.0 && i > 0) {
      size_t l = i - 1;
      for (size_t j = 0; j <= l; j++) {
        double g = 0.0;
        for (size_t k = 0; k <= l; k++) g += a[i * n + k] * a[k * n + j];
        for (size_t k = 0; k <= l; k++) a[k * n + j] -= g * a[k * n + i];
      }
    }
    d[i] = a[i * n + i];
    a[i * n + i] = 1.0;
    for (size_t j = 0; j < i; j++) a[j * n + i] = a[i * n + j] = 0.0;
  }
  // z^T so that plane rotations act on contiguous rows: zt[i*n+k] = z(k,i)
  std::vector<double> zt(n * n);
  for (size_t i = 0; i < n; i++)
    for (size_t k = 0; k < n; k++) zt[i * n + k] = a[k * n + i];
  a.clear();
  a.shrink_to_fit();

  // ---- implicit QL (tql2) ----
  for (size_t i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (size_t l = 0; l < n; l++) {
    int iter = 0;
    size_t m;
    do {
      for (m = l; m + 1 < n; m++) {
        double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.2e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 200) throw std::logic_error("Eigendecomposition failed!\n");
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        bool underflow = false;
        for (size_t ii = m; ii-- > l;) {
          size_t i = ii;
          double f = s * e[i], b = c * e[i];
          e[i + 1] = (r = hypot(f, g));
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            underflow = true;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          double *zi = &zt[i * n], *zi1 = &zt[(i + 1) * n];
          for (size_t k = 0; k < n; k++) {
            double fz = zi1[k];
            zi1[k] = s * zi[k] + c * fz;
            zi[k] = c * zi[k] - s * fz;
          }
        }
        if (underflow) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  // sort ascending
  std::vector<size_t> ord(n);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return d[x] < d[y]; });
  for (size_t j = 0; j < n; j++) {
    E[j] = d[ord[j]];
    for (size_t k = 0; k < n; k++) C(k, j) = zt[ord[j] * n + k];
  }
}

void eig_gsym(Vec &E, Mat &C, const Mat &F, const Mat &Sinvh) {
  Mat Forth(helfem::matmul(helfem::matmul(Sinvh, true, F, false), false, Sinvh, false));
  Mat Co;
  eig_sym(E, Co, Forth);
  C = helfem::matmul(Sinvh, false, Co, false);
}

void eig_gsym_sub(Vec &E, Mat &C, const Mat &F, const Mat &Sinvh, const std::vector<std::vector<size_t> > &m_idx) {
  const size_t N = F.n_rows;
  E.assign(N, 0.0);
  C.zeros(N, N);
  size_t iidx = 0;
  for (size_t isym = 0; isym < m_idx.size(); isym++) {
    // columns of Sinvh with support on this block's rows (scf_helpers.cpp:150-157)
    std::vector<size_t> Sind;
    for (size_t c = 0; c < Sinvh.n_cols; c++) {
      double nrm = 0.0;
      for (size_t r : m_idx[isym]) nrm += Sinvh(r, c) * Sinvh(r, c);
      if (nrm != 0.0) Sind.push_back(c);
    }
    Mat Ssub(N, Sind.size());
    for (size_t c = 0; c < Sind.size(); c++)
      for (size_t r = 0; r < N; r++) Ssub(r, c) = Sinvh(r, Sind[c]);
    Vec Esub;
    Mat Csub;
    eig_gsym(Esub, Csub, F, Ssub);
    for (size_t c = 0; c < Esub.size(); c++) {
      E[iidx + c] = Esub[c];
      for (size_t r = 0; r < N; r++) C(r, iidx + c) = Csub(r, c);
    }
    iidx += Esub.size();
  }
  if (iidx != N) throw std::logic_error("Symmetry mismatch in eig_gsym_sub\n");
  std::vector<size_t> ord(N);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return E[x] < E[y]; });
  Vec Es(N);
  Mat Cs(N, N);
  for (size_t j = 0; j < N; j++) {
    Es[j] = E[ord[j]];
    for (size_t r = 0; r < N; r++) Cs(r, j) = C(r, ord[j]);
  }
  E = Es;
  C = Cs;
}

Mat invh(Mat S, bool chol) {
  const size_t n = S.n_rows;
  Vec nrm(n);
  for (size_t i = 0; i < n; i++) nrm[i] = 1.0 / sqrt(S(i, i));
  for (size_t j = 0; j < n; j++)
    for (size_t i = 0; i < n; i++) S(i, j) *= nrm[i] * nrm[j];
  Mat X(n, n);
  if (chol) {
    // X = inv(chol(S)), chol upper-triangular R with S = R^T R
    Mat R(n, n);
    for (size_t j = 0; j < n; j++) {
      for (size_t i = 0; i <= j; i++) {
        double s = S(i, j);
        for (size_t k = 0; k < i; k++) s -= R(k, i) * R(k, j);
        if (i == j) {
          if (s <= 0.0) throw std::logic_error("Cholesky failed\n");
          R(i, j) = sqrt(s);
        } else
          R(i, j) = s / R(i, i);
      }
    }
    for (size_t j = 0; j < n; j++) {  // solve R X = I column by column (upper triangular)
      for (size_t ii = j + 1; ii-- > 0;) {
        double s = (ii == j) ? 1.0 : 0.0;
        for (size_t k = ii + 1; k <= j; k++) s -= R(ii, k) * X(k, j);
        X(ii, j) = s / R(ii, ii);
      }
    }
  } else {
    Vec val;
    Mat vec;
    eig_sym(val, vec, S);
    Mat tmp(vec);
    for (size_t j = 0; j < n; j++) {
      double f = 1.0 / sqrt(val[j]);
      for (size_t i = 0; i < n; i++) tmp(i, j) *= f;
    }
    X = helfem::matmul(tmp, false, vec, true);
  }
  for (size_t j = 0; j < n; j++)
    for (size_t i = 0; i < n; i++) X(i, j) *= nrm[i];
  return X;
}

Mat form_Sinvh(const Mat &S, bool chol, const std::vector<std::vector<size_t> > &sym) {
  const size_t N = S.n_rows;
  if (sym.size() <= 1) return invh(S, chol);
  Mat Sinvh(N, N);
  size_t ioff = 0;
  for (const auto &idx : sym) {
    if (idx.empty()) continue;
    Mat Ss(idx.size(), idx.size());
    for (size_t j = 0; j < idx.size(); j++)
      for (size_t i = 0; i < idx.size(); i++) Ss(i, j) = S(idx[i], idx[j]);
    Mat X = invh(Ss, chol);
    for (size_t j = 0; j < idx.size(); j++)
      for (size_t i = 0; i < idx.size(); i++) Sinvh(idx[i], ioff + j) = X(i, j);
    ioff += idx.size();
  }
  return Sinvh;
}

Mat form_density(const Mat &C, size_t nocc) {
  if (C.n_cols < nocc) throw std::logic_error("Not enough orbitals!\n");
  Mat P(C.n_rows, C.n_rows);
  for (size_t o = 0; o < nocc; o++)
    for (size_t j = 0; j < C.n_rows; j++) {
      double cj = C(j, o);
      for (size_t i = 0; i < C.n_rows; i++) P(i, j) += C(i, o) * cj;
    }
  return P;
}

Mat enforce_fock_symmetry(const Mat &F, const std::vector<std::vector<size_t> > &m_idx) {
  Mat out(F.n_rows, F.n_rows);
  for (const auto &idx : m_idx)
    for (size_t j : idx)
      for (size_t i : idx) out(i, j) = F(i, j);
  return out;
}

}  // namespace oracle
