#include "special.h"
#include "fem.h"
#include <cfloat>
#include <cmath>
#include <stdexcept>
#include <cstdlib>
#include <algorithm>

namespace helfem {

// ---------------------------------------------------------------------------------------------
// Normalised associated Legendre functions
// ---------------------------------------------------------------------------------------------
template <typename T>
static T theta_lm_t(int l, int m, T x) {
  // Theta_lm = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) P_l^m(x), Condon-Shortley phase in P_l^m,
  // i.e. the value gsl_sf_legendre_sphPlm returns (spherical_harmonics.cpp:38).
  int am = std::abs(m);
  if (am > l) return T(0);
  const T pi = T(3.141592653589793238462643383279502884L);
  T sth = std::sqrt((T(1) - x) * (T(1) + x));
  T pmm = std::sqrt(T(1) / (T(4) * pi));
  for (int k = 1; k <= am; k++) pmm *= -std::sqrt(T(2 * k + 1) / T(2 * k)) * sth;
  T res;
  if (l == am)
    res = pmm;
  else {
    T pm1 = std::sqrt(T(2 * am + 3)) * x * pmm;  // Theta_{m+1,m}
    T pm2 = pmm;
    for (int ll = am + 2; ll <= l; ll++) {
      T a = std::sqrt(T(4 * ll * ll - 1) / T(ll * ll - am * am));
      T b = std::sqrt(T((ll - 1) * (ll - 1) - am * am) / T(4 * (ll - 1) * (ll - 1) - 1));
      T p = a * (x * pm1 - b * pm2);
      pm2 = pm1;
      pm1 = p;
    }
    res = pm1;
  }
  // Y_l^{-m} = (-1)^m conj(Y_l^m)
  if (m < 0 && (am % 2)) res = -res;
  return res;
}

double theta_lm(int l, int m, double x) { return theta_lm_t<double>(l, m, x); }

double dtheta_lm(int l, int m, double x) {
  // d/dtheta Y_l^m = m cot(theta) Y_l^m + sqrt((l-m)(l+m+1)) e^{-i phi} Y_l^{m+1}
  // (reference: src/diatomic/basis.cpp:1914-1926); the phi phases cancel to e^{i m phi}.
  double cotth = x / sqrt(1.0 - x * x);
  double r = m * cotth * theta_lm(l, m, x);
  if (m < l) r += sqrt((double)((l - m) * (l + m + 1))) * theta_lm(l, m + 1, x);
  return r;
}

std::complex<double> spherical_harmonics(int l, int m, double cth, double phi) {
  return theta_lm(l, m, cth) * std::exp(std::complex<double>(0.0, m * phi));
}

// ---------------------------------------------------------------------------------------------
// Gaunt coefficients by exact Gauss-Legendre quadrature in extended precision
// ---------------------------------------------------------------------------------------------
namespace {
struct GLRule {
  std::vector<long double> x, w;
};

GLRule gauss_legendre(int n) {
  GLRule r;
  r.x.resize(n);
  r.w.resize(n);
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 0; i < n; i++) {
    long double x = cosl(pi * (i + 0.75L) / (n + 0.5L));
    long double dp = 1.0L;
    for (int it = 0; it < 100; it++) {
      long double p0 = 1.0L, p1 = x;
      for (int j = 2; j <= n; j++) {
        long double pj = ((2 * j - 1) * x * p1 - (j - 1) * p0) / j;
        p0 = p1;
        p1 = pj;
      }
      if (n == 1) { p1 = x; p0 = 1.0L; }
      dp = n * (x * p1 - p0) / (x * x - 1.0L);
      long double dx = p1 / dp;
      x -= dx;
      if (fabsl(dx) < 1e-20L) break;
    }
    // recompute derivative at converged x
    long double p0 = 1.0L, p1 = x;
    for (int j = 2; j <= n; j++) {
      long double pj = ((2 * j - 1) * x * p1 - (j - 1) * p0) / j;
      p0 = p1;
      p1 = pj;
    }
    dp = n * (x * p1 - p0) / (x * x - 1.0L);
    r.x[i] = x;
    r.w[i] = 2.0L / ((1.0L - x * x) * dp * dp);
  }
  return r;
}

long double gaunt_quad(int L, int M, int l, int m, int lp, int mp) {
  if (M != m + mp) return 0.0L;
  if (L < std::abs(l - lp) || L > l + lp) return 0.0L;
  if (std::abs(M) > L || std::abs(m) > l || std::abs(mp) > lp) return 0.0L;
  if ((L + l + lp) % 2) return 0.0L;  // (L l lp; 0 0 0) vanishes for odd sum
  int n = (L + l + lp) / 2 + 1;
  static thread_local std::map<int, GLRule> rules;
  auto it = rules.find(n);
  if (it == rules.end()) it = rules.emplace(n, gauss_legendre(n)).first;
  const GLRule &r = it->second;
  // Theta_l^m at the nodes of the n-point rule, tabulated once per (n, l, m): a basis with l up to 20 asks for ~70 000
  // coefficients, and evaluating the three recurrences at every node of every coefficient took 0.6 s of its set-up
  static thread_local std::map<long long, std::vector<long double> > tables;
  auto table = [&](int ll, int mm) -> const std::vector<long double> & {
    const long long key = ((long long)n << 40) | ((long long)ll << 20) | (long long)(mm + 4096);
    auto jt = tables.find(key);
    if (jt == tables.end()) {
      std::vector<long double> v(n);
      for (int i = 0; i < n; i++) v[i] = theta_lm_t<long double>(ll, mm, r.x[i]);
      jt = tables.emplace(key, std::move(v)).first;
    }
    return jt->second;
  };
  const std::vector<long double> &tL = table(L, M), &tl = table(l, m), &tp = table(lp, mp);
  long double s = 0.0L;
  for (int i = 0; i < n; i++) s += r.w[i] * tL[i] * tl[i] * tp[i];
  // phi integral gives 2 pi (M = m + mp); conj(Y_L^M) has the same Theta
  return 2.0L * 3.141592653589793238462643383279502884L * s;
}
}  // namespace

double gaunt_coefficient(int L, int M, int l, int m, int lp, int mp) {
  return (double)gaunt_quad(L, M, l, m, lp, mp);
}

double Gaunt::coeff(int L, int M, int l, int m, int lp, int mp) const {
  if (std::abs(M) > L) return 0.0;
  if (std::abs(m) > l) return 0.0;
  if (std::abs(mp) > lp) return 0.0;
  if (M != m + mp) return 0.0;
  // pack (L,l,lp < 1024; M,m in [-512,511])
  long long key = ((long long)L << 50) | ((long long)l << 40) | ((long long)lp << 30) |
                  ((long long)(M + 512) << 20) | ((long long)(m + 512) << 10);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  double v = gaunt_coefficient(L, M, l, m, lp, mp);
  cache[key] = v;
  return v;
}

double Gaunt::mod_coeff(int lj, int mj, int L, int M, int li, int mi) const {
  // reference: gaunt.cpp:167-180
  static const double const0(2.0 / 3.0 * sqrt(M_PI));
  static const double const2(4.0 / 15.0 * sqrt(5.0 * M_PI));
  double cpl0(coeff(L, M, 0, 0, L, M) * coeff(lj, mj, li, mi, L, M));
  double cpl2 = 0.0;
  for (int Lp = std::max(std::max(L - 2, 0), std::abs(M)); Lp <= L + 2; Lp++)
    cpl2 += coeff(Lp, M, 2, 0, L, M) * coeff(lj, mj, li, mi, Lp, M);
  return const0 * cpl0 + const2 * cpl2;
}

double Gaunt::cosine_coupling(int lj, int mj, int li, int mi) const {
  static const double const1(2.0 * sqrt(M_PI / 3.0));
  return const1 * coeff(lj, mj, 1, 0, li, mi);
}
double Gaunt::cosine2_coupling(int lj, int mj, int li, int mi) const {
  static const double const0(2.0 / 3.0 * sqrt(M_PI));
  static const double const2(4.0 / 15.0 * sqrt(5.0 * M_PI));
  return const0 * coeff(lj, mj, 0, 0, li, mi) + const2 * coeff(lj, mj, 2, 0, li, mi);
}
double Gaunt::cosine3_coupling(int lj, int mj, int li, int mi) const {
  static const double const1(2.0 / 5.0 * sqrt(3.0 * M_PI));
  static const double const3(4.0 / 35.0 * sqrt(7.0 * M_PI));
  return const1 * coeff(lj, mj, 1, 0, li, mi) + const3 * coeff(lj, mj, 3, 0, li, mi);
}
double Gaunt::cosine4_coupling(int lj, int mj, int li, int mi) const {
  static const double const0(2.0 / 5.0 * sqrt(M_PI));
  static const double const2(8.0 / 35.0 * sqrt(5.0 * M_PI));
  static const double const4(16.0 / 105.0 * sqrt(M_PI));
  return const0 * coeff(lj, mj, 0, 0, li, mi) + const2 * coeff(lj, mj, 2, 0, li, mi) +
         const4 * coeff(lj, mj, 4, 0, li, mi);
}
double Gaunt::cosine5_coupling(int lj, int mj, int li, int mi) const {
  static const double const1(2.0 / 7.0 * sqrt(3.0 * M_PI));
  static const double const3(8.0 / 63.0 * sqrt(7.0 * M_PI));
  static const double const5(16.0 / 693.0 * sqrt(11.0 * M_PI));
  return const1 * coeff(lj, mj, 1, 0, li, mi) + const3 * coeff(lj, mj, 3, 0, li, mi) +
         const5 * coeff(lj, mj, 5, 0, li, mi);
}
double Gaunt::sine2_coupling(int lj, int mj, int li, int mi) const {
  static const double const0(4.0 / 3.0 * sqrt(M_PI));
  static const double const2(-4.0 / 15.0 * sqrt(5.0 * M_PI));
  return const0 * coeff(lj, mj, 0, 0, li, mi) + const2 * coeff(lj, mj, 2, 0, li, mi);
}
double Gaunt::cosine2_sine2_coupling(int lj, int mj, int li, int mi) const {
  static const double const0(4.0 / 15.0 * sqrt(M_PI));
  static const double const2(4.0 / 105.0 * sqrt(5.0 * M_PI));
  static const double const4(-16.0 / 105.0 * sqrt(M_PI));
  return const0 * coeff(lj, mj, 0, 0, li, mi) + const2 * coeff(lj, mj, 2, 0, li, mi) +
         const4 * coeff(lj, mj, 4, 0, li, mi);
}

// ---------------------------------------------------------------------------------------------
// Legendre functions of the first and second kind outside the cut, xi > 1
// ---------------------------------------------------------------------------------------------
void legendre_PQ(int Lmax, int Mmax, double xi, double *P, double *Q) {
  const int ld = Lmax + 1;
  for (int i = 0; i < ld * (Mmax + 1); i++) {
    P[i] = 0.0;
    Q[i] = 0.0;
  }
  if (!(xi > 1.0)) return;  // xi==1: table entry stays zero (legendretable.cpp:73)

  typedef long double real;
  const real x = xi;
  const real xm1 = x - 1.0L, xp1 = x + 1.0L;
  const real s2 = xm1 * xp1;  // xi^2-1
  const real s = sqrtl(s2);
  const real mu = logl(x + s);  // xi = cosh(mu)

  // ----- P_L^M: upward recurrence in L (dominant solution, stable) -----
  //   P_M^M = (2M-1)!! (xi^2-1)^{M/2},  (L-M+1) P_{L+1}^M = (2L+1) xi P_L^M - (L+M) P_{L-1}^M
  {
    real pmm = 1.0L;
    for (int M = 0; M <= Mmax && M <= Lmax; M++) {
      if (M > 0) pmm *= (2 * M - 1) * s;
      real pm1 = pmm, pm2 = 0.0L;
      P[M * ld + M] = (double)pm1;
      for (int L = M; L < Lmax; L++) {
        real pn = ((2 * L + 1) * x * pm1 - (L + M) * pm2) / (L - M + 1);
        pm2 = pm1;
        pm1 = pn;
        P[M * ld + L + 1] = (double)pn;
      }
    }
  }

  // ----- Q_L^M -----
  // closed forms at L=0:  Q_0^0 = 1/2 ln((xi+1)/(xi-1)),
  //   d^k/dxi^k Q_0 = 1/2 (-1)^{k-1} (k-1)! [ (xi+1)^{-k} - (xi-1)^{-k} ],  Q_0^M = (xi^2-1)^{M/2} Q_0^{(M)}
  std::vector<real> dQ0(Mmax + 2);
  dQ0[0] = 0.5L * logl(xp1 / xm1);
  {
    real fact = 1.0L;  // (k-1)!
    for (int k = 1; k <= Mmax + 1; k++) {
      if (k > 1) fact *= (k - 1);
      real sign = ((k - 1) % 2) ? -1.0L : 1.0L;
      dQ0[k] = 0.5L * sign * fact * (powl(xp1, -k) - powl(xm1, -k));
    }
  }
  // The three-term recurrence in L loses a factor e^{2 mu L} upwards; it is used upwards only
  // while that is harmless (rows M=0,1, then the M-raising relation
  //   Q_L^{M+1} = -2M xi/sqrt(xi^2-1) Q_L^M + (L+M)(L-M+1) Q_L^{M-1},
  // both terms of equal sign), otherwise Miller's downward recurrence from far above Lmax,
  // normalised with the closed-form Q_0^M.
  const bool upward = (2.0L * mu * Lmax < 1.0L);
  std::vector<std::vector<real> > q(Mmax + 1, std::vector<real>(Lmax + 2, 0.0L));
  if (upward) {
    for (int M = 0; M <= std::min(1, Mmax); M++) {
      // Q_1 = xi Q_0 - 1  =>  Q_1^{(1)} = xi Q_0^{(1)} + Q_0
      q[M][0] = (M == 0) ? dQ0[0] : s * dQ0[1];
      if (Lmax >= 1) q[M][1] = (M == 0) ? (x * dQ0[0] - 1.0L) : s * (x * dQ0[1] + dQ0[0]);
      for (int L = 1; L < Lmax; L++)
        q[M][L + 1] = ((2 * L + 1) * x * q[M][L] - (L + M) * q[M][L - 1]) / (L - M + 1);
    }
    for (int M = 1; M < Mmax; M++)
      for (int L = 0; L <= Lmax; L++)
        q[M + 1][L] = -2.0L * M * x / s * q[M][L] + (real)(L + M) * (real)(L - M + 1) * q[M - 1][L];
  } else {
    real sM = 1.0L;  // (xi^2-1)^{M/2}
    for (int M = 0; M <= Mmax; M++) {
      if (M > 0) sM *= s;
      const real q0 = sM * dQ0[M];
      // start far enough above Lmax that the P-like contamination e^{-2 mu pad} is < 1e-22
      int pad = (int)ceill(52.0L / (2.0L * mu)) + 4;
      int Ls = Lmax + pad;
      real qp1 = 0.0L, qc = 1e-200L;  // Q_{Ls+1}, Q_{Ls} (unnormalised)
      std::vector<real> tmp(Lmax + 1, 0.0L);
      for (int L = Ls; L >= 1; L--) {
        // (L-M+1) Q_{L+1} = (2L+1) xi Q_L - (L+M) Q_{L-1}
        real qm1 = ((2 * L + 1) * x * qc - (L - M + 1) * qp1) / (L + M);
        qp1 = qc;
        qc = qm1;
        if (L - 1 <= Lmax) tmp[L - 1] = qc;
        if (fabsl(qc) > 1e3000L) {  // rescale to avoid overflow
          const real sc = 1e-3000L;
          qc *= sc;
          qp1 *= sc;
          for (int k = L - 1; k <= Lmax; k++) tmp[k] *= sc;
        }
      }
      real norm = q0 / tmp[0];
      for (int L = 0; L <= Lmax; L++) q[M][L] = tmp[L] * norm;
    }
  }
  for (int M = 0; M <= Mmax; M++)
    for (int L = M; L <= Lmax; L++) Q[M * ld + L] = (double)q[M][L];
}

// -------------------------------------------------------------------------------------------------
// Radial kernels of the range-separated exchange (atomic program)
// -------------------------------------------------------------------------------------------------
double bessel_il(double x, int L) {
  x = fabs(x);
  if (L < 0) throw std::logic_error("bessel_il: negative order");
  if (x == 0.0) return L == 0 ? 1.0 : 0.0;
  if (x > 30.0 && x > 4.0 * L) {
    // upward recurrence i_{n+1} = i_{n-1} - (2n+1)/x i_n, harmless for x >> n
    double im = sinh(x) / x;
    if (L == 0) return im;
    double ic = (x * cosh(x) - sinh(x)) / (x * x);
    for (int n = 1; n < L; n++) {
      double ip = im - (2 * n + 1) / x * ic;
      im = ic;
      ic = ip;
    }
    return ic;
  }
  // ascending series: x^L/(2L+1)!! sum_k (x^2/2)^k / (k! (2L+3)(2L+5)...(2L+2k+1)), all terms positive
  double pref = 1.0;
  for (int n = 1; n <= L; n++) pref *= x / (2 * n + 1);
  const double h = 0.5 * x * x;
  double term = 1.0, sum = 1.0;
  for (int k = 1; k < 2000; k++) {
    term *= h / ((double)k * (2 * L + 2 * k + 1));
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return pref * sum;
}

double bessel_kl(double x, int L) {
  if (L < 0) throw std::logic_error("bessel_kl: negative order");
  // k_0 = e^{-x}/x, k_1 = e^{-x}(1 + 1/x)/x, k_{n+1} = k_{n-1} + (2n+1)/x k_n  (dominant solution: upward is stable)
  const double ex = exp(-x);
  double km = ex / x;
  if (L == 0) return km;
  double kc = ex * (1.0 + 1.0 / x) / x;
  for (int n = 1; n < L; n++) {
    double kp = km + (2 * n + 1) / x * kc;
    km = kc;
    kc = kp;
  }
  return kc;
}

namespace {
int g_erfc_binomial_mode = 0;

double factorial_d(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; i++) f *= i;
  return f;
}
double double_factorial_d(int n) {
  double f = 1.0;
  for (int i = n; i > 1; i -= 2) f *= i;
  return f;
}
// binomial coefficient with any integer upper argument: C(n,m) = n (n-1) ... (n-m+1) / m!
double gen_binomial(int n, int m) {
  if (m < 0) return 0.0;
  double c = 1.0;
  for (int i = 0; i < m; i++) c = c * (n - i) / (i + 1);
  return std::round(c);
}
// the helper of erfc_expn.cpp:46-70, kept only for the test hook
double ref_binomial(int n, int m) {
  if (n == -1) return (m % 2) ? -1.0 : 1.0;
  if (n == 0) return m == 0 ? 1.0 : 0.0;
  if (m == 0) return 1.0;
  if (m == 1) return n;
  if (n > 0 && m > 0 && m > n) return 0.0;
  if (n < 0) return ref_binomial(n + m - 1, m) * ((m % 2) ? -1.0 : 1.0);
  return gen_binomial(n, m);
}

// eq 22
double erfc_F(int n, double Xi, double xi) {
  const double ep = exp(-(Xi + xi) * (Xi + xi)), em = exp(-(Xi - xi) * (Xi - xi));
  const double q = -1.0 / (4.0 * Xi * xi);
  double s = 0.0, qp = q;
  for (int p = 0; p <= n; p++) {
    double sg = ((n - p) % 2) ? -1.0 : 1.0;
    s += qp * (factorial_d(n + p) / (factorial_d(p) * factorial_d(n - p))) * (sg * ep - em);
    qp *= q;
  }
  return 2.0 / sqrt(M_PI) * s;
}
// eq 24
double erfc_H(int n, double Xi, double xi) {
  const double A = std::pow(Xi, 2 * n + 1), a = std::pow(xi, 2 * n + 1);
  return ((A + a) * erfc(Xi + xi) - (A - a) * erfc(Xi - xi)) / (2.0 * std::pow(xi * Xi, n + 1));
}
// eq 21
double erfc_Phi_general(int n, double Xi, double xi) {
  double s = erfc_F(n, Xi, xi) + erfc_H(n, Xi, xi);
  for (int m = 1; m <= n; m++) {
    double Am = std::pow(Xi, m), am = std::pow(xi, m);
    s += erfc_F(n - m, Xi, xi) * ((Am * Am + am * am) / (Am * am));
  }
  return s;
}
// eqs 28, 29
double erfc_D(int n, int k, double Xi) {
  const double pref = exp(-Xi * Xi) / sqrt(M_PI) * std::pow(2.0, n + 1) * std::pow(Xi, 2 * n + 1);
  if (k == 0) {
    double s = 0.0;
    for (int m = 1; m <= n; m++) s += 1.0 / (double_factorial_d(2 * (n - m) + 1) * std::pow(2 * Xi * Xi, m));
    return erfc(Xi) + pref * s;
  }
  double s = 0.0;
  for (int m = 1; m <= k; m++) {
    double c = g_erfc_binomial_mode ? ref_binomial(m - k - 1, m - 1) : gen_binomial(m - k - 1, m - 1);
    s += c * std::pow(2 * Xi * Xi, k - m) / double_factorial_d(2 * (n + k - m) + 1);
  }
  return pref * (2.0 * n + 1.0) / (factorial_d(k) * (2.0 * (n + k) + 1.0)) * s;
}
// eq 30, summed until the (paired) terms drop below machine precision, at most 32 terms (erfc_expn.cpp:150-178)
double erfc_Phi_short(int n, double Xi, double xi) {
  if (xi == 0.0 && n > 0) return 0.0;
  if (n == 0 && xi == 0.0 && Xi == 0.0) return 1.0;
  double phi = 0.0;
  for (int k = 0; k <= 30; k += 2) {
    double d = erfc_D(n, k, Xi) * std::pow(xi, n + 2 * k) + erfc_D(n, k + 1, Xi) * std::pow(xi, n + 2 * (k + 1));
    phi += d;
    if (fabs(d) < DBL_EPSILON * fabs(phi)) break;
  }
  return phi / std::pow(Xi, n + 1);
}
}  // namespace

void set_erfc_binomial_mode(int mode) { g_erfc_binomial_mode = mode; }

double erfc_Phi(int n, double Xi, double xi) {
  if (n < 0) throw std::logic_error("erfc_Phi: negative order");
  if (Xi < xi) std::swap(Xi, xi);
  // p. 8624 of the paper: the closed form loses its digits to cancellation for small arguments
  if (xi < 0.4 || (Xi < 0.5 && xi < 2 * Xi)) return erfc_Phi_short(n, Xi, xi);
  return erfc_Phi_general(n, Xi, xi);
}

static legendre_provider_t g_legendre_provider = nullptr;
void set_legendre_provider(legendre_provider_t fn) { g_legendre_provider = fn; }
legendre_provider_t get_legendre_provider() { return g_legendre_provider; }

void angular_chebyshev(int ltheta, int nphi, Vec &cth, Vec &phi, Vec &w) {
  // reference: angular.cpp:22-45 (compound_rule), 64-71
  Vec xl, wl;
  chebyshev_rule(ltheta, xl, wl);
  cth.assign(xl.size() * nphi, 0.0);
  phi.assign(xl.size() * nphi, 0.0);
  w.assign(xl.size() * nphi, 0.0);
  double dphi = 2.0 * M_PI / nphi;
  for (size_t i = 0; i < xl.size(); i++)
    for (int j = 0; j < nphi; j++) {
      size_t idx = i * nphi + j;
      cth[idx] = xl[i];
      phi[idx] = j * dphi;
      w[idx] = wl[i] * dphi;
    }
}

}  // namespace helfem
