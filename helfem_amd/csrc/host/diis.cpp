#include "diis.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <deque>
#include <stdexcept>
#include <utility>

namespace helfem {

namespace {
constexpr double COOLTHR = 0.1;  // diis.cpp:26

typedef std::vector<double> vec;

double dot(const vec &a, const vec &b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
  return s;
}

void print_row(const char *title, const vec &w) {
  printf("%s\n", title);
  for (double v : w) printf(" % .4e", v);
  printf("\n");
}

// eigen-decomposition of a small symmetric matrix (cyclic Jacobi): A = V diag(lam) V^T, columns of V in v[col*n+row]
void jacobi_eig(size_t n, std::vector<double> A, vec &lam, std::vector<double> &V) {
  V.assign(n * n, 0.0);
  for (size_t i = 0; i < n; i++) V[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0, diag = 0.0;
    for (size_t i = 0; i < n; i++)
      for (size_t j = 0; j < n; j++) (i == j ? diag : off) += A[i * n + j] * A[i * n + j];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (size_t p = 0; p + 1 < n; p++)
      for (size_t q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (size_t k = 0; k < n; k++) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (size_t k = 0; k < n; k++) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (size_t k = 0; k < n; k++) {
          const double vkp = V[p * n + k], vkq = V[q * n + k];
          V[p * n + k] = c * vkp - s * vkq;
          V[q * n + k] = s * vkp + c * vkq;
        }
      }
  }
  lam.resize(n);
  for (size_t i = 0; i < n; i++) lam[i] = A[i * n + i];
}

// src/general/lbfgs.cpp (Nocedal & Wright, algorithm 9.1), history of at most nmax = 10 pairs
struct LBFGS {
  size_t nmax = 10;
  std::deque<vec> xk, gk;
  void update(const vec &x, const vec &g) {
    xk.push_back(x);
    gk.push_back(g);
    if (xk.size() > nmax) {
      xk.pop_front();
      gk.pop_front();
    }
  }
  vec solve() const {
    const size_t k = gk.size() - 1;
    vec q(gk[k]);
    std::vector<vec> sk(k), yk(k);
    for (size_t i = 0; i < k; i++) {
      sk[i].resize(q.size());
      yk[i].resize(q.size());
      for (size_t a = 0; a < q.size(); a++) {
        sk[i][a] = xk[i + 1][a] - xk[i][a];
        yk[i][a] = gk[i + 1][a] - gk[i][a];
      }
    }
    vec alphai(k);
    for (size_t i = k; i-- > 0;) {
      alphai[i] = dot(sk[i], q) / dot(yk[i], sk[i]);
      for (size_t a = 0; a < q.size(); a++) q[a] -= alphai[i] * yk[i][a];
    }
    vec r(q);
    if (xk.size() >= 2) {  // apply_diagonal_hessian
      const vec &s = sk[k - 1], &y = yk[k - 1];
      const double f = dot(s, y) / dot(y, y);
      for (double &v : r) v *= f;
    }
    for (size_t i = 0; i < k; i++) {
      const double beta = dot(yk[i], r) / dot(yk[i], sk[i]);
      for (size_t a = 0; a < r.size(); a++) r[a] += sk[i][a] * (alphai[i] - beta);
    }
    return r;
  }
};

vec compute_c(const vec &x) {  // diis.cpp:458-461
  const double xn = dot(x, x);
  vec c(x.size());
  for (size_t i = 0; i < x.size(); i++) c[i] = x[i] * x[i] / xn;
  return c;
}
}  // namespace

DiisMixer::DiisMixer(bool usediis, double diiseps, double diisthr, bool useadiis, bool verbose, size_t imax)
    : usediis_(usediis), useadiis_(useadiis), verbose_(verbose), diiseps_(diiseps), diisthr_(diisthr), imax_(imax) {
  if (imax_ == 0) throw std::logic_error("DIIS history length must be positive\n");
  B_.assign(imax_ * imax_, 0.0);
  T_.assign(imax_ * imax_, 0.0);
}

void DiisMixer::pop_oldest() {
  if (E_.empty()) return;
  const size_t n = E_.size();
  for (size_t i = 1; i < n; i++)
    for (size_t j = 1; j < n; j++) {
      B_[(i - 1) * imax_ + (j - 1)] = B_[i * imax_ + j];
      T_[(i - 1) * imax_ + (j - 1)] = T_[i * imax_ + j];
    }
  E_.erase(E_.begin());
  err_.erase(err_.begin());
}

void DiisMixer::push(double E, double maxerr) {
  if (full()) throw std::logic_error("DiisMixer::push on a full history: pop_oldest first\n");
  E_.push_back(E);
  err_.push_back(maxerr);
}

// PiF(i) = Tr (P_i - P_n) F_n,  PiFj(i,j) = Tr (P_i - P_n)(F_j - F_n)     (diis.cpp:116-127, 170-187)
void DiisMixer::adiis_terms(vec &PiF, vec &PiFj) const {
  const size_t N = E_.size(), n = N - 1;
  auto T = [this](size_t i, size_t j) { return T_[i * imax_ + j]; };
  PiF.assign(N, 0.0);
  PiFj.assign(N * N, 0.0);
  for (size_t i = 0; i < N; i++) PiF[i] = T(i, n) - T(n, n);
  for (size_t i = 0; i < N; i++)
    for (size_t j = 0; j < N; j++) PiFj[i * N + j] = T(i, j) - T(i, n) - T(n, j) + T(n, n);
}

// get_w_diis_wrk (diis.cpp:297-372): B w = 1 through the singular value decomposition, zero singular values left out,
// weights normalised to sum one.  B is symmetric: singular triplets (|lam|, u, sign(lam) u), so
// sum_i (u_i . 1) / s_i v_i = sum_i (u_i . 1) / lam_i u_i.
vec DiisMixer::weights_cdiis() const {
  const size_t N = E_.size();
  std::vector<double> B(N * N);
  for (size_t i = 0; i < N; i++)
    for (size_t j = 0; j < N; j++) B[i * N + j] = B_[i * imax_ + j];
  vec lam;
  std::vector<double> V;
  jacobi_eig(N, B, lam, V);
  vec sol(N, 0.0);
  for (size_t k = 0; k < N; k++) {
    if (lam[k] == 0.0) continue;
    double u1 = 0.0;
    for (size_t a = 0; a < N; a++) u1 += V[k * N + a];
    for (size_t a = 0; a < N; a++) sol[a] += u1 / lam[k] * V[k * N + a];
  }
  double s = 0.0;
  for (double v : sol) s += v;
  if (s == 0.0) {
    sol.assign(N, 1.0);
    s = (double)N;
  }
  for (double &v : sol) v /= s;
  return sol;
}

double DiisMixer::adiis_energy(const vec &x) const {  // get_E_adiis, diis.cpp:602-616
  vec PiF, PiFj;
  adiis_terms(PiF, PiFj);
  const size_t N = PiF.size();
  if (x.size() != N) throw std::domain_error("Incorrect number of parameters.\n");
  const vec c = compute_c(x);
  double E = 2.0 * dot(c, PiF);
  for (size_t i = 0; i < N; i++)
    for (size_t j = 0; j < N; j++) E += c[i] * PiFj[i * N + j] * c[j];
  return E;
}

// get_w_adiis (diis.cpp:492-600): minimise E(c) = 2 c.PiF + c^T PiFj c over c_i = x_i^2 / x.x with L-BFGS directions and a
// bracketing line search with parabolic interpolation
vec DiisMixer::weights_adiis() const {
  vec PiF, PiFj;
  adiis_terms(PiF, PiFj);
  const size_t N = PiF.size();
  if (N == 1) return vec(1, 1.0);
  auto energy = [&](const vec &x) {
    const vec c = compute_c(x);
    double E = 2.0 * dot(c, PiF);
    for (size_t i = 0; i < N; i++)
      for (size_t j = 0; j < N; j++) E += c[i] * PiFj[i * N + j] * c[j];
    return E;
  };
  auto gradient = [&](const vec &x) {  // get_dEdx_adiis with compute_jac (diis.cpp:463-489, 618-648)
    const vec c = compute_c(x);
    vec dEdc(N);
    for (size_t i = 0; i < N; i++) {
      double s = 2.0 * PiF[i];
      for (size_t j = 0; j < N; j++) s += (PiFj[i * N + j] + PiFj[j * N + i]) * c[j];
      dEdc[i] = s;
    }
    const double xn = dot(x, x);
    vec g(N, 0.0);  // g_j = sum_i jac(i,j) dEdc_i,  jac(i,j) = -2 c_i x_j / xn + delta_ij 2 x_i / xn
    double cd = dot(c, dEdc);
    for (size_t j = 0; j < N; j++) g[j] = -2.0 * x[j] / xn * cd + 2.0 * x[j] / xn * dEdc[j];
    return g;
  };
  auto step_to = [&](const vec &x, const vec &sd, double len) {
    vec y(x);
    for (size_t i = 0; i < N; i++) y[i] += sd[i] * len;
    return y;
  };
  vec x(N, 1.0 / (double)N);
  LBFGS bfgs;
  double steplen = 0.01;
  const double fac = 2.0;
  typedef std::pair<double, double> step_t;  // (length, energy)
  for (size_t iiter = 0; iiter < 1000; iiter++) {
    const vec g = gradient(x);
    if (sqrt(dot(g, g)) <= 1e-7) break;
    bfgs.update(x, g);
    vec sd = bfgs.solve();
    for (double &v : sd) v = -v;
    std::vector<step_t> steps;
    steps.push_back(step_t(steplen / fac, energy(step_to(x, sd, steplen / fac))));
    steps.push_back(step_t(steplen, energy(step_to(x, sd, steplen))));
    double Emin = 0.0;
    size_t imin = 0;
    auto find_min = [&]() {
      Emin = steps[0].second;
      imin = 0;
      for (size_t i = 1; i < steps.size(); i++)
        if (steps[i].second < Emin) {
          Emin = steps[i].second;
          imin = i;
        }
    };
    while (true) {
      std::sort(steps.begin(), steps.end());
      find_min();
      if (imin == 0 || imin == steps.size() - 1) {
        step_t p;
        if (imin == 0) {
          p.first = steps[imin].first / fac;
          if (steps[imin].first < DBL_EPSILON) break;
        } else
          p.first = steps[imin].first * fac;
        p.second = energy(step_to(x, sd, p.first));
        steps.push_back(p);
      } else
        break;
    }
    if (imin != 0 && imin != steps.size() - 1) {
      // parabola through the three points around the minimum: y = b0 + b1 t + b2 t^2
      const double t0 = steps[imin - 1].first, t1 = steps[imin].first, t2 = steps[imin + 1].first;
      const double y0 = steps[imin - 1].second, y1 = steps[imin].second, y2 = steps[imin + 1].second;
      const double d01 = (y1 - y0) / (t1 - t0), d12 = (y2 - y1) / (t2 - t1);
      const double b2 = (d12 - d01) / (t2 - t0);
      const double b1 = d01 - b2 * (t0 + t1);
      if (std::isfinite(b2) && b2 > sqrt(DBL_EPSILON)) {
        const double x0 = -b1 / (2.0 * b2);
        if (t0 < x0 && x0 < t2) {
          steps.push_back(step_t(x0, energy(step_to(x, sd, x0))));
          find_min();
        }
      }
    }
    if (steps[imin].first < DBL_EPSILON) break;
    x = step_to(x, sd, steps[imin].first);
    steplen = steps[imin].first;
  }
  return compute_c(x);
}

// DIIS::get_w (diis.cpp:214-290)
vec DiisMixer::get_w() {
  const size_t N = E_.size();
  const double err = err_.back();
  vec w;
  if (useadiis_ && !usediis_) {
    w = weights_adiis();
    if (verbose_) print_row("ADIIS weights", w);
  } else if (!useadiis_ && usediis_) {
    if (err > diisthr_) throw std::runtime_error("DIIS error too large for only DIIS to converge wave function.\n");
    w = weights_cdiis();
    if (verbose_) print_row("DIIS weights", w);
  } else if (useadiis_ && usediis_) {
    double diisw = std::max(std::min(1.0 - (err - diisthr_) / (diiseps_ - diisthr_), 1.0), 0.0);
    const double adiisw = 1.0 - diisw;  // as in the reference: NOT recomputed when the cool-off below zeroes diisw
    if (cooloff_ > 0) {
      diisw = 0.0;
      cooloff_--;
    } else if (N > 1 && E_[N - 1] - E_[N - 2] > COOLTHR) {
      cooloff_ = 2;
      diisw = 0.0;
    }
    w.assign(N, 0.0);
    vec wd, wa;
    if (diisw != 0.0) {
      wd = weights_cdiis();
      for (size_t i = 0; i < N; i++) w[i] += diisw * wd[i];
    }
    if (adiisw != 0.0) {
      wa = weights_adiis();
      for (size_t i = 0; i < N; i++) w[i] += adiisw * wa[i];
    }
    if (verbose_) {
      if (adiisw != 0.0) print_row("ADIIS weights", wa);
      if (diisw != 0.0) print_row("CDIIS weights", wd);
      if (adiisw != 0.0 && diisw != 0.0) print_row(" DIIS weights", w);
    }
  } else
    throw std::runtime_error("Nor DIIS or ADIIS has been turned on.\n");
  return w;
}

vec DiisMixer::solve(size_t &dropped) {
  dropped = 0;
  if (E_.empty()) throw std::logic_error("DiisMixer::solve on an empty history\n");
  vec sol;
  while (true) {
    sol = get_w();
    if (E_.size() == 1) {
      // (the reference would erase its only entry here and fail; a one-entry history is that entry)
      sol.assign(1, 1.0);
      break;
    }
    if (fabs(sol.back()) <= sqrt(DBL_EPSILON)) {
      if (verbose_) printf("Weight on last matrix too small, reducing to %i matrices.\n", (int)E_.size() - 1);
      pop_oldest();
      dropped++;
    } else
      break;
  }
  return sol;
}

}  // namespace helfem
