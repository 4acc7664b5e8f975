// Minimal column-major dense matrix used by the host-side setup code.
//
// Memory layout is identical to arma::mat (contiguous, column-major, memptr()), which is the
// type every reference entry point on the hot path takes and returns
// (/root/reference/src/diatomic/basis.h:247-249, src/general/scf_helpers.h:24-36), so the C ABI
// in include/helfem_gpu.h can be wrapped around real Armadillo objects without a copy.
#pragma once
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace helfem {

struct Mat {
  size_t n_rows = 0, n_cols = 0;
  std::vector<double> d;

  Mat() {}
  Mat(size_t r, size_t c) : n_rows(r), n_cols(c), d(r * c, 0.0) {}
  void zeros(size_t r, size_t c) {
    n_rows = r;
    n_cols = c;
    d.assign(r * c, 0.0);
  }
  double &operator()(size_t i, size_t j) { return d[j * n_rows + i]; }
  double operator()(size_t i, size_t j) const { return d[j * n_rows + i]; }
  double *memptr() { return d.data(); }
  const double *memptr() const { return d.data(); }
  size_t n_elem() const { return d.size(); }

  Mat t() const {
    Mat r(n_cols, n_rows);
    for (size_t j = 0; j < n_cols; j++)
      for (size_t i = 0; i < n_rows; i++) r(j, i) = (*this)(i, j);
    return r;
  }
  Mat &operator+=(const Mat &o) {
    if (o.n_rows != n_rows || o.n_cols != n_cols) throw std::logic_error("Mat += shape mismatch");
    for (size_t i = 0; i < d.size(); i++) d[i] += o.d[i];
    return *this;
  }
  Mat &operator-=(const Mat &o) {
    if (o.n_rows != n_rows || o.n_cols != n_cols) throw std::logic_error("Mat -= shape mismatch");
    for (size_t i = 0; i < d.size(); i++) d[i] -= o.d[i];
    return *this;
  }
  Mat &operator*=(double s) {
    for (auto &x : d) x *= s;
    return *this;
  }
};

inline Mat operator+(Mat a, const Mat &b) { return a += b; }
inline Mat operator-(Mat a, const Mat &b) { return a -= b; }
inline Mat operator*(Mat a, double s) { return a *= s; }
inline Mat operator*(double s, Mat a) { return a *= s; }

// C = op(A) * op(B), plain triple loop; setup-time use only (never on the hot path)
inline Mat matmul(const Mat &A, bool tA, const Mat &B, bool tB) {
  size_t m = tA ? A.n_cols : A.n_rows, k = tA ? A.n_rows : A.n_cols;
  size_t k2 = tB ? B.n_cols : B.n_rows, n = tB ? B.n_rows : B.n_cols;
  if (k != k2) throw std::logic_error("matmul shape mismatch");
  Mat C(m, n);
  for (size_t j = 0; j < n; j++)
    for (size_t l = 0; l < k; l++) {
      double b = tB ? B(j, l) : B(l, j);
      if (b == 0.0) continue;
      for (size_t i = 0; i < m; i++) C(i, j) += (tA ? A(l, i) : A(i, l)) * b;
    }
  return C;
}

inline double trace_prod(const Mat &A, const Mat &B) {
  // trace(A*B)
  if (A.n_cols != B.n_rows || A.n_rows != B.n_cols) throw std::logic_error("trace_prod shape mismatch");
  double t = 0.0;
  for (size_t j = 0; j < A.n_cols; j++)
    for (size_t i = 0; i < A.n_rows; i++) t += A(i, j) * B(j, i);
  return t;
}

typedef std::vector<double> Vec;
typedef std::vector<int> IVec;

}  // namespace helfem
