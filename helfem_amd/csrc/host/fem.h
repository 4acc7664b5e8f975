// Quadrature rules, Lagrange interpolating polynomial (LIP) shape functions and the 1-D finite
// element bookkeeping that generate the tables the Fock-build kernels consume.
//
// Follows the behaviour of (reference, read-only):
//   libhelfem/src/chebyshev.cpp:22-53        modified Gauss-Chebyshev rule of the 2nd kind
//   libhelfem/src/lobatto.cpp:588-743        Gauss-Lobatto nodes (LIP node positions)
//   libhelfem/src/LIPBasis.cpp:21-50         LIP: nprim=nnodes, noverlap=1, drop_first/last
//   libhelfem/src/LIPBasis_eval.cpp:8-61     values / first derivative as explicit Lagrange products
//   libhelfem/src/PolynomialBasis.cpp:175-179 n-th derivative divided by (len/2)^n
//   libhelfem/src/FiniteElementBasis.cpp:37-50,141-144,203-211,253-262,327-415
//   libhelfem/src/grid.cpp:18-87             element grids (igrid 1..5)
#pragma once
#include "linalg.h"
#include <functional>

namespace helfem {

/// Modified Gauss-Chebyshev (2nd kind) rule for \int_{-1}^{1} f(x) dx, ascending nodes
void chebyshev_rule(int n, Vec &x, Vec &w);
/// Gauss-Lobatto nodes on [-1,1], ascending (weights are not used by the FEM path)
Vec lobatto_nodes(int n);
/// Element boundaries: igrid 1 linear, 2 quadratic, 3 polynomial, 4 exponential, 5 geometric
Vec get_grid(double rmax, int num_el, int igrid, double zexp);
double arcosh(double x);

/// LIP primitive basis on [-1,1]
struct LIPBasis {
  Vec x0;                // nodes
  std::vector<int> enabled;  // indices of enabled primitive functions
  LIPBasis() {}
  explicit LIPBasis(const Vec &nodes);
  int nprim() const { return (int)x0.size(); }
  int nbf() const { return (int)enabled.size(); }
  int noverlap() const { return 1; }
  void drop_first() { enabled.erase(enabled.begin()); }
  void drop_last() { enabled.pop_back(); }
  /// n-th derivative (n=0,1,2) of the enabled functions at points x; rows = points.
  /// element_length is the scaling factor len/2 (derivative divided by its n-th power).
  Mat eval_dnf(const Vec &x, int n, double element_length) const;
};

/// 1-D finite element basis: elements [bval(i),bval(i+1)], LIP shape functions
struct FEMBasis {
  LIPBasis poly;
  Vec bval;
  bool zero_func_left = false, zero_func_right = true;
  std::vector<size_t> first, last;

  FEMBasis() {}
  FEMBasis(const LIPBasis &poly, const Vec &bval, bool zero_func_left, bool zero_func_right);

  size_t nelem() const { return bval.size() - 1; }
  size_t nbf() const { return last.back() + 1; }
  size_t max_nprim() const { return poly.nprim(); }
  LIPBasis get_basis(size_t iel) const;
  size_t nprim(size_t iel) const { return get_basis(iel).nbf(); }
  void get_idx(size_t iel, size_t &ifirst, size_t &ilast) const {
    ifirst = first[iel];
    ilast = last[iel];
  }
  double element_begin(size_t iel) const { return bval[iel]; }
  double element_end(size_t iel) const { return bval[iel + 1]; }
  double element_midpoint(size_t iel) const { return 0.5 * (bval[iel] + bval[iel + 1]); }
  double scaling_factor(size_t iel) const { return (bval[iel + 1] - bval[iel]) / 2; }
  Vec eval_coord(const Vec &x, size_t iel) const;
  Mat eval_dnf(const Vec &x, int n, size_t iel) const {
    return get_basis(iel).eval_dnf(x, n, scaling_factor(iel));
  }
  /// \int_el lh_i^(lhder) rh_j^(rhder) f(r) dr in element iel
  Mat matrix_element(size_t iel, int lhder, int rhder, const Vec &xq, const Vec &wq,
                     const std::function<double(double)> &f) const;
  /// global matrix (sum over elements into overlapping blocks)
  Mat matrix_element(int lhder, int rhder, const Vec &xq, const Vec &wq,
                     const std::function<double(double)> &f) const;
};

}  // namespace helfem
