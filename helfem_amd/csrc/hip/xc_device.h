// Exchange-correlation functionals (spin-unpolarised and spin-polarised) evaluated on the GPU grid (reference: the
// libxc calls of DFTGridWorker::compute_xc, src/diatomic/dftgrid.cpp:343-458).  libxc is a
// third-party dependency of the reference that is not vendored and not present here; the energy
// densities below are the published closed forms and the derivatives vrho = d(rho exc)/d rho,
// vsigma = d(rho exc)/d sigma come from forward-mode automatic differentiation (two-component
// dual numbers), deliberately a different route from the oracle's hand-derived formulas.
#pragma once
#include <hip/hip_runtime.h>

namespace hfg {
namespace xc {

struct Dual {
  double v, dr, ds;  // value, d/d rho, d/d sigma
};

__host__ __device__ inline Dual mk(double v, double dr = 0.0, double ds = 0.0) {
  Dual d;
  d.v = v;
  d.dr = dr;
  d.ds = ds;
  return d;
}
__host__ __device__ inline Dual operator+(Dual a, Dual b) { return mk(a.v + b.v, a.dr + b.dr, a.ds + b.ds); }
__host__ __device__ inline Dual operator-(Dual a, Dual b) { return mk(a.v - b.v, a.dr - b.dr, a.ds - b.ds); }
__host__ __device__ inline Dual operator-(Dual a) { return mk(-a.v, -a.dr, -a.ds); }
__host__ __device__ inline Dual operator*(Dual a, Dual b) {
  return mk(a.v * b.v, a.dr * b.v + a.v * b.dr, a.ds * b.v + a.v * b.ds);
}
__host__ __device__ inline Dual operator/(Dual a, Dual b) {
  double inv = 1.0 / b.v;
  double q = a.v * inv;
  return mk(q, (a.dr - q * b.dr) * inv, (a.ds - q * b.ds) * inv);
}
__host__ __device__ inline Dual operator+(Dual a, double c) { return mk(a.v + c, a.dr, a.ds); }
__host__ __device__ inline Dual operator+(double c, Dual a) { return mk(a.v + c, a.dr, a.ds); }
__host__ __device__ inline Dual operator-(Dual a, double c) { return mk(a.v - c, a.dr, a.ds); }
__host__ __device__ inline Dual operator-(double c, Dual a) { return mk(c - a.v, -a.dr, -a.ds); }
__host__ __device__ inline Dual operator*(Dual a, double c) { return mk(a.v * c, a.dr * c, a.ds * c); }
__host__ __device__ inline Dual operator*(double c, Dual a) { return mk(a.v * c, a.dr * c, a.ds * c); }
__host__ __device__ inline Dual operator/(Dual a, double c) { return a * (1.0 / c); }
__host__ __device__ inline Dual operator/(double c, Dual a) { return mk(c) / a; }
__host__ __device__ inline Dual dsqrt(Dual a) {
  double s = sqrt(a.v);
  double f = 0.5 / s;
  return mk(s, a.dr * f, a.ds * f);
}
__host__ __device__ inline Dual dcbrt(Dual a) {
  double c = cbrt(a.v);
  double f = c / (3.0 * a.v);
  return mk(c, a.dr * f, a.ds * f);
}
__host__ __device__ inline Dual dlog(Dual a) { return mk(log(a.v), a.dr / a.v, a.ds / a.v); }
// log(1 + a) and exp(a) - 1 without the cancellation: at densities below ~1e-28 (reached in the far field when the
// density threshold is lowered to zero) log(1 + 1/den) of PW92 rounds to 0, exp(-ec/gamma) - 1 with it, and PBE's A = inf
__host__ __device__ inline Dual dlog1p(Dual a) { return mk(log1p(a.v), a.dr / (1.0 + a.v), a.ds / (1.0 + a.v)); }
__host__ __device__ inline Dual dexpm1(Dual a) {
  double e = exp(a.v);
  return mk(expm1(a.v), a.dr * e, a.ds * e);
}
__host__ __device__ inline Dual dexp(Dual a) {
  double e = exp(a.v);
  return mk(e, a.dr * e, a.ds * e);
}
__host__ __device__ inline Dual datan(Dual a) {
  double f = 1.0 / (1.0 + a.v * a.v);
  return mk(atan(a.v), a.dr * f, a.ds * f);
}

__host__ __device__ inline Dual derf(Dual a) {
  double f = 1.1283791670955126 * exp(-a.v * a.v);  // 2/sqrt(pi)
  return mk(erf(a.v), a.dr * f, a.ds * f);
}

#define HFG_PI 3.14159265358979323846

// energy per particle of each functional as a Dual in (rho, sigma)
// External functional parameters (libxc's xc_func_set_ext_params as the reference calls it for --x_pars / --c_pars,
// dftgrid.cpp:405-410), in libxc's order: lda_x {alpha = 1}, gga_x_pbe {kappa = 0.8040, mu = beta pi^2/3},
// gga_c_pbe {beta = 0.06672455060314922, gamma = (1 - ln 2)/pi^2, BB = 1}.  One constant-memory copy per translation
// unit (this header is included by fock.hip only); set_xc_params() there fills it before the grid kernels are launched.
struct XCPar {
  double x_alpha, x_kappa, x_mu, c_beta, c_gamma, c_BB;
};
#define HFG_XCPAR_DEFAULTS \
  { 1.0, 0.8040, 0.06672455060314922 * HFG_PI * HFG_PI / 3.0, 0.06672455060314922, (1.0 - 0.6931471805599453) / (HFG_PI * HFG_PI), 1.0 }
static __constant__ XCPar c_xcpar = HFG_XCPAR_DEFAULTS;

__host__ __device__ inline Dual eps_lda_x(Dual rho) { return (-0.75 * cbrt(3.0 / HFG_PI)) * dcbrt(rho); }

// Short-range LDA exchange of the range-separated hybrids: eps_x^sr = eps_x^LDA(rho) F(a), a = omega/(2 k_F).
// erfc(omega r)/r (libxc lda_x_erf; Toulouse, Savin, Flad, Int. J. Quantum Chem. 100, 1047 (2004), eq 18-19):
//   F = 1 - (8/3) a [sqrt(pi) erf(1/(2a)) + (2a - 4a^3) exp(-1/(4a^2)) - 3a + 4a^3]
// exp(-omega r)/r (libxc lda_x_yukawa; Savin, Flad, Int. J. Quantum Chem. 56, 327 (1995)):
//   F = 1 - (8/3) a [atan(1/a) + a/4 - (a/4)(a^2 + 3) ln(1 + 1/a^2)]
// Both closed forms cancel like a^4 for large a (low density); there the expansions in 1/a^2 are summed:
//   erf: sum_k (-1)^{k+1} 2/(4^k k! (2k+1)(k+1)(k+2)) a^{-2k},  Yukawa: sum_k (-1)^{k+1} 2/((2k+1)(k+1)(k+2)) a^{-2k}.
__host__ __device__ inline Dual att_erf(Dual a) {
  if (a.v > 0.75) {
    Dual u = 1.0 / (a * a), t = mk(1.0), sum = mk(0.0);
    for (int k = 1; k <= 24; k++) {
      t = t * u * (-0.25 / k);
      sum = sum - t * (2.0 / ((2.0 * k + 1.0) * (k + 1.0) * (k + 2.0)));
    }
    return sum;
  }
  Dual a2 = a * a;
  Dual e = dexp(-0.25 / a2);
  return 1.0 - (8.0 / 3.0) * a * (1.7724538509055159 * derf(0.5 / a) + (2.0 * a - 4.0 * a2 * a) * e - 3.0 * a + 4.0 * a2 * a);
}
__host__ __device__ inline Dual att_yukawa(Dual a) {
  if (a.v > 2.0) {
    Dual u = 1.0 / (a * a), t = mk(-1.0), sum = mk(0.0);
    for (int k = 1; k <= 40; k++) {
      t = -1.0 * t * u;
      sum = sum + t * (2.0 / ((2.0 * k + 1.0) * (k + 1.0) * (k + 2.0)));
    }
    return sum;
  }
  Dual a2 = a * a;
  return 1.0 - (8.0 / 3.0) * a * (datan(1.0 / a) + 0.25 * a - 0.25 * a * (a2 + 3.0) * dlog(1.0 + 1.0 / a2));
}
/// kind 1: Yukawa, 2: erfc
__host__ __device__ inline Dual eps_lda_x_sr(Dual rho, double omega, int kind) {
  Dual kf = dcbrt((3.0 * HFG_PI * HFG_PI) * rho);
  Dual a = (0.5 * omega) / kf;
  return eps_lda_x(rho) * (kind == 1 ? att_yukawa(a) : att_erf(a));
}
// hyb_lda_xc_cam_lda0 (libxc id 178; Mosquera, Borca, Ratner, Schatz, J. Phys. Chem. A 120, 1605 (2016)): omega = 1/3,
// exact exchange alpha = 1/2 over the full range plus beta = -1/4 of the short range (1/4 at short, 1/2 at long range);
// DFT part (1 - alpha) lda_x - beta lda_x_erf + lda_c_pw_mod
#define HFG_CAM_LDA0_OMEGA (1.0 / 3.0)
__host__ __device__ inline Dual eps_cam_lda0_x(Dual rho) {
  return 0.5 * eps_lda_x(rho) + 0.25 * eps_lda_x_sr(rho, HFG_CAM_LDA0_OMEGA, 2);
}

__host__ __device__ inline Dual eps_vwn_fit(Dual rho, double A, double b, double c, double x0) {
  Dual rs = dcbrt(3.0 / (4.0 * HFG_PI) / rho);
  Dual x = dsqrt(rs);
  Dual X = x * x + b * x + c;
  const double X0 = x0 * x0 + b * x0 + c;
  const double Q = sqrt(4.0 * c - b * b);
  Dual at = datan(Q / (2.0 * x + b));
  Dual xm = x - x0;
  return A * (dlog(x * x / X) + (2.0 * b / Q) * at -
              (b * x0 / X0) * (dlog(xm * xm / X) + (2.0 * (b + 2.0 * x0) / Q) * at));
}
// VWN5 (libxc lda_c_vwn = 7: fit to the Ceperley-Alder data) and the RPA fit of the same paper (lda_c_vwn_rpa = 8, the
// "VWN" inside B3LYP as libxc and Gaussian define it); Vosko, Wilk, Nusair, Can. J. Phys. 58, 1200 (1980), paramagnetic sets
__host__ __device__ inline Dual eps_lda_c_vwn(Dual rho) { return eps_vwn_fit(rho, 0.0310907, 3.72744, 12.9352, -0.10498); }
__host__ __device__ inline Dual eps_lda_c_vwn_rpa(Dual rho) { return eps_vwn_fit(rho, 0.0310907, 13.0720, 42.7198, -0.409286); }

// Becke 88 exchange (libxc gga_x_b88 = 106; Becke, Phys. Rev. A 38, 3098 (1988)), per spin channel s:
//   e_s = -Cx rho_s^{4/3} - beta rho_s^{4/3} x^2 / (1 + 6 beta x asinh x),  x = |grad rho_s| / rho_s^{4/3},  beta = 0.0042
// written in t = x^2: x asinh x is smooth in t (no square-root singularity where the gradient vanishes)
__host__ __device__ inline Dual d_x_asinh_x(Dual t) {
  const double sq = sqrt(t.v), as = asinh(sq);
  const double df = (sq > 1e-8 ? as / (2.0 * sq) : 0.5) + 0.5 / sqrt(1.0 + t.v);
  return mk(sq * as, t.dr * df, t.ds * df);
}
__host__ __device__ inline Dual eps_gga_x_b88(Dual rho, Dual sigma) {
  const double beta = 0.0042, Cx = 0.9305257363491000;  // (3/2) (3/(4 pi))^{1/3}
  Dual rs = 0.5 * rho;  // one spin channel of the unpolarised density, gradient invariant sigma/4
  Dual r43 = rs * dcbrt(rs);
  Dual t = (0.25 * sigma) / (r43 * r43);
  Dual es = (-1.0 * r43) * (Cx + beta * t / (1.0 + (6.0 * beta) * d_x_asinh_x(t)));
  return 2.0 * es / rho;
}

// Lee-Yang-Parr correlation (libxc gga_c_lyp = 131; Lee, Yang, Parr, Phys. Rev. B 37, 785 (1988)) in the gradient-only
// form of Miehlich, Savin, Stoll, Preuss, Chem. Phys. Lett. 157, 200 (1989).  Closed shell (their eq 2 at rho_a = rho_b):
//   E = -a rho/(1 + d rho^{-1/3}) - a b omega [C_F rho^{14/3} - rho^2 sigma (1/24 + 7 delta/72)],
//   omega = exp(-c rho^{-1/3}) rho^{-11/3} / (1 + d rho^{-1/3}),  delta = c rho^{-1/3} + d rho^{-1/3}/(1 + d rho^{-1/3})
#define HFG_LYP_A 0.04918
#define HFG_LYP_B 0.132
#define HFG_LYP_C 0.2533
#define HFG_LYP_D 0.349
__host__ __device__ inline Dual eps_gga_c_lyp(Dual rho, Dual sigma) {
  const double CF = 0.3 * 9.570780000627305;  // (3/10) (3 pi^2)^{2/3}
  Dual rm13 = 1.0 / dcbrt(rho);
  Dual den = 1.0 + HFG_LYP_D * rm13;
  Dual delta = HFG_LYP_C * rm13 + HFG_LYP_D * rm13 / den;
  Dual rm23 = rm13 * rm13;
  Dual rm83 = rm23 * rm23 * rm23 * rm23;
  return (-HFG_LYP_A) / den -
         (HFG_LYP_A * HFG_LYP_B) * dexp(-HFG_LYP_C * rm13) / den * (CF - rm83 * sigma * (1.0 / 24.0 + (7.0 / 72.0) * delta));
}
// hyb_gga_xc_b3lyp (libxc 402; Stephens, Devlin, Chabalowski, Frisch, J. Phys. Chem. 98, 11623 (1994)), DFT part:
// 0.08 lda_x + 0.72 gga_x_b88 + 0.19 lda_c_vwn_rpa + 0.81 gga_c_lyp; 0.20 exact exchange
__host__ __device__ inline Dual eps_b3lyp(Dual r, Dual s) {
  return 0.08 * eps_lda_x(r) + 0.72 * eps_gga_x_b88(r, s) + 0.19 * eps_lda_c_vwn_rpa(r) + 0.81 * eps_gga_c_lyp(r, s);
}

__host__ __device__ inline Dual eps_pw92(Dual rs, bool mod) {
  const double a = mod ? 0.0310907 : 0.031091;  // libxc lda_c_pw.c: par_pw / par_pw_mod
  const double a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
  Dual srs = dsqrt(rs);
  Dual den = (2.0 * a) * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
  return (-2.0 * a) * (1.0 + a1 * rs) * dlog1p(1.0 / den);
}

__host__ __device__ inline Dual eps_lda_c_pw(Dual rho) {
  return eps_pw92(dcbrt(3.0 / (4.0 * HFG_PI) / rho), false);
}

__host__ __device__ inline Dual eps_gga_x_pbe(Dual rho, Dual sigma) {
  const double kappa = c_xcpar.x_kappa;
  const double mu = c_xcpar.x_mu;
  Dual exu = eps_lda_x(rho);
  Dual kf = dcbrt((3.0 * HFG_PI * HFG_PI) * rho);
  Dual s2 = sigma / (4.0 * kf * kf * rho * rho);
  Dual Fx = 1.0 + kappa - kappa / (1.0 + (mu / kappa) * s2);
  return exu * Fx;
}

__host__ __device__ inline Dual eps_gga_c_pbe(Dual rho, Dual sigma) {
  const double beta = c_xcpar.c_beta;
  const double gamma = c_xcpar.c_gamma;
  const double B = beta / gamma;
  const double BB = c_xcpar.c_BB;
  Dual rs = dcbrt(3.0 / (4.0 * HFG_PI) / rho);
  Dual ec = eps_pw92(rs, true);
  Dual kf = dcbrt((3.0 * HFG_PI * HFG_PI) * rho);
  Dual ks2 = (4.0 / HFG_PI) * kf;
  Dual t2 = sigma / (4.0 * ks2 * rho * rho);
  Dual Aa = B / dexpm1(-ec / gamma);
  Dual At2 = Aa * t2;
  // libxc: f1 = t^2 + BB A t^4, H = gamma log(1 + (beta/gamma) f1 / (1 + A f1)); BB = 1 is PBE
  Dual f1 = t2 * (1.0 + BB * At2);
  Dual H = gamma * dlog1p(B * f1 / (1.0 + Aa * f1));
  return ec + H;
}

__host__ __device__ inline bool is_gga(int id) {
  return id == 101 || id == 130 || id == 406 || id == 202 || id == 231 || id == 106 || id == 131 || id == 402;
}
__host__ __device__ inline bool is_supported(int id) {
  return id == 1 || id == 7 || id == 8 || id == 12 || id == 13 || id == 101 || id == 130 || id == 406 || id == 202 || id == 231 ||
         id == 546 || id == 641 || id == 178 || id == 106 || id == 131 || id == 402;
}

__host__ __device__ inline bool is_exchange(int id) { return id == 1 || id == 101 || id == 546 || id == 641 || id == 202 || id == 106; }

/// adds functional id's exc (per particle), vrho, vsigma at one point; rho >= threshold assumed.
/// live: the density of one spin channel, rho/2, reaches the threshold.  Exchange is a sum over the spin channels and
/// libxc (>= 5) leaves a channel below the threshold out of it, in the unpolarised evaluation too — which is what makes
/// the restricted and the unrestricted build agree on a closed shell.
__host__ __device__ inline void eval_add(int id, double rho, double sigma, bool live, double &exc, double &vrho,
                                         double &vsigma) {
  Dual r = mk(rho, 1.0, 0.0), s = mk(sigma, 0.0, 1.0);
  Dual e;
  if (!live) {
    if (is_exchange(id)) return;
    if (id == 178) id = 13;   // the hybrids keep their correlation part
    if (id == 406) id = 130;
    if (id == 402) id = -402;  // 0.19 lda_c_vwn_rpa + 0.81 gga_c_lyp
  }
  switch (id) {
    case 1: e = c_xcpar.x_alpha * eps_lda_x(r); break;
    case 7: e = eps_lda_c_vwn(r); break;
    case 12: e = eps_lda_c_pw(r); break;
    case 13: e = eps_pw92(dcbrt(3.0 / (4.0 * HFG_PI) / r), true); break;  // lda_c_pw_mod
    case 546: e = eps_lda_x_sr(r, 0.3, 2); break;                          // lda_x_erf, libxc default omega
    case 641: e = eps_lda_x_sr(r, 0.3, 1); break;                          // lda_x_yukawa, libxc default omega
    case 178: e = eps_cam_lda0_x(r) + eps_pw92(dcbrt(3.0 / (4.0 * HFG_PI) / r), true); break;  // hyb_lda_xc_cam_lda0, DFT part
    case 101: e = eps_gga_x_pbe(r, s); break;
    case 130: e = eps_gga_c_pbe(r, s); break;
    case 406: e = 0.75 * eps_gga_x_pbe(r, s) + eps_gga_c_pbe(r, s); break;  // hyb_gga_xc_pbeh (PBE0), DFT part
    case 8: e = eps_lda_c_vwn_rpa(r); break;
    case 106: e = eps_gga_x_b88(r, s); break;
    case 131: e = eps_gga_c_lyp(r, s); break;
    case 402: e = eps_b3lyp(r, s); break;
    case -402: e = 0.19 * eps_lda_c_vwn_rpa(r) + 0.81 * eps_gga_c_lyp(r, s); break;
    default: return;
  }
  Dual en = r * e;  // energy per volume
  exc += e.v;
  vrho += en.dr;
  vsigma += en.ds;
}

// ---------------------------------------------------------------------------------------------------------
// Spin-polarised evaluation (xc_*_exc_vxc with XC_POLARIZED; dftgrid.cpp:343-458 with rho 2 x N, sigma 3 x N).
// Exchange: spin scaling E_x[ra,rb] = (E_x[2ra] + E_x[2rb])/2 through the two-component duals above.
// Correlation: three-component duals in (rho_a, rho_b, sigma_total).
// ---------------------------------------------------------------------------------------------------------
struct T3 {
  double v, a, b, s;
};
__host__ __device__ inline T3 t3(double v, double a = 0.0, double b = 0.0, double s = 0.0) {
  T3 r;
  r.v = v;
  r.a = a;
  r.b = b;
  r.s = s;
  return r;
}
__host__ __device__ inline T3 t3f(T3 x, double f, double df) { return t3(f, df * x.a, df * x.b, df * x.s); }
__host__ __device__ inline T3 operator+(T3 x, T3 y) { return t3(x.v + y.v, x.a + y.a, x.b + y.b, x.s + y.s); }
__host__ __device__ inline T3 operator-(T3 x, T3 y) { return t3(x.v - y.v, x.a - y.a, x.b - y.b, x.s - y.s); }
__host__ __device__ inline T3 operator-(T3 x) { return t3(-x.v, -x.a, -x.b, -x.s); }
__host__ __device__ inline T3 operator*(T3 x, T3 y) {
  return t3(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.s * y.v + x.v * y.s);
}
__host__ __device__ inline T3 operator/(T3 x, T3 y) {
  double inv = 1.0 / y.v, q = x.v * inv;
  return t3(q, (x.a - q * y.a) * inv, (x.b - q * y.b) * inv, (x.s - q * y.s) * inv);
}
__host__ __device__ inline T3 operator+(T3 x, double c) { return t3(x.v + c, x.a, x.b, x.s); }
__host__ __device__ inline T3 operator+(double c, T3 x) { return t3(x.v + c, x.a, x.b, x.s); }
__host__ __device__ inline T3 operator-(T3 x, double c) { return t3(x.v - c, x.a, x.b, x.s); }
__host__ __device__ inline T3 operator-(double c, T3 x) { return t3(c - x.v, -x.a, -x.b, -x.s); }
__host__ __device__ inline T3 operator*(T3 x, double c) { return t3(x.v * c, x.a * c, x.b * c, x.s * c); }
__host__ __device__ inline T3 operator*(double c, T3 x) { return t3(x.v * c, x.a * c, x.b * c, x.s * c); }
__host__ __device__ inline T3 operator/(T3 x, double c) { return x * (1.0 / c); }
__host__ __device__ inline T3 operator/(double c, T3 x) { return t3(c) / x; }
__host__ __device__ inline T3 tsqrt(T3 x) {
  double r = sqrt(x.v);
  return t3f(x, r, 0.5 / r);
}
__host__ __device__ inline T3 tcbrt(T3 x) {
  double c = cbrt(x.v);
  return t3f(x, c, c / (3.0 * x.v));
}
__host__ __device__ inline T3 tlog(T3 x) { return t3f(x, log(x.v), 1.0 / x.v); }
__host__ __device__ inline T3 tlog1p(T3 x) { return t3f(x, log1p(x.v), 1.0 / (1.0 + x.v)); }
__host__ __device__ inline T3 texpm1(T3 x) { return t3f(x, expm1(x.v), exp(x.v)); }
__host__ __device__ inline T3 texp(T3 x) {
  double e = exp(x.v);
  return t3f(x, e, e);
}
__host__ __device__ inline T3 tatan(T3 x) { return t3f(x, atan(x.v), 1.0 / (1.0 + x.v * x.v)); }
__host__ __device__ inline T3 tpow43(T3 x) {
  double c = cbrt(x.v);
  return t3f(x, x.v * c, (4.0 / 3.0) * c);
}
__host__ __device__ inline T3 tpow23(T3 x) {
  double c = cbrt(x.v);
  return t3f(x, c * c, 2.0 / (3.0 * c));
}

// ---- seven-slot duals: derivatives with respect to (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb, tau_a, tau_b), for
// the spin-polarised meta-GGA correlation; the correlation building blocks below are templates over the dual type ----
struct T7 {
  double v, d[7];
};
__host__ __device__ inline T7 t7(double v) {
  T7 r;
  r.v = v;
#pragma unroll
  for (int k = 0; k < 7; k++) r.d[k] = 0.0;
  return r;
}
__host__ __device__ inline T7 t7var(double v, int k) {
  T7 r = t7(v);
  r.d[k] = 1.0;
  return r;
}
__host__ __device__ inline T7 t7f(T7 x, double f, double df) {
  T7 r;
  r.v = f;
#pragma unroll
  for (int k = 0; k < 7; k++) r.d[k] = df * x.d[k];
  return r;
}
__host__ __device__ inline T7 operator+(T7 x, T7 y) {
  T7 r;
  r.v = x.v + y.v;
#pragma unroll
  for (int k = 0; k < 7; k++) r.d[k] = x.d[k] + y.d[k];
  return r;
}
__host__ __device__ inline T7 operator-(T7 x, T7 y) {
  T7 r;
  r.v = x.v - y.v;
#pragma unroll
  for (int k = 0; k < 7; k++) r.d[k] = x.d[k] - y.d[k];
  return r;
}
__host__ __device__ inline T7 operator-(T7 x) { return t7f(x, -x.v, -1.0); }
__host__ __device__ inline T7 operator*(T7 x, T7 y) {
  T7 r;
  r.v = x.v * y.v;
#pragma unroll
  for (int k = 0; k < 7; k++) r.d[k] = x.d[k] * y.v + x.v * y.d[k];
  return r;
}
__host__ __device__ inline T7 operator/(T7 x, T7 y) {
  const double inv = 1.0 / y.v, q = x.v * inv;
  T7 r;
  r.v = q;
#pragma unroll
  for (int k = 0; k < 7; k++) r.d[k] = (x.d[k] - q * y.d[k]) * inv;
  return r;
}
__host__ __device__ inline T7 operator+(T7 x, double c) { x.v += c; return x; }
__host__ __device__ inline T7 operator+(double c, T7 x) { x.v += c; return x; }
__host__ __device__ inline T7 operator-(T7 x, double c) { x.v -= c; return x; }
__host__ __device__ inline T7 operator-(double c, T7 x) { return t7f(x, c - x.v, -1.0); }
__host__ __device__ inline T7 operator*(T7 x, double c) { return t7f(x, x.v * c, c); }
__host__ __device__ inline T7 operator*(double c, T7 x) { return t7f(x, x.v * c, c); }
__host__ __device__ inline T7 operator/(T7 x, double c) { return x * (1.0 / c); }
__host__ __device__ inline T7 operator/(double c, T7 x) { return t7f(x, c / x.v, -c / (x.v * x.v)); }
__host__ __device__ inline T7 tsqrt(T7 x) {
  double r = sqrt(x.v);
  return t7f(x, r, 0.5 / r);
}
__host__ __device__ inline T7 tcbrt(T7 x) {
  double c = cbrt(x.v);
  return t7f(x, c, c / (3.0 * x.v));
}
__host__ __device__ inline T7 tlog(T7 x) { return t7f(x, log(x.v), 1.0 / x.v); }
__host__ __device__ inline T7 tlog1p(T7 x) { return t7f(x, log1p(x.v), 1.0 / (1.0 + x.v)); }
__host__ __device__ inline T7 texpm1(T7 x) { return t7f(x, expm1(x.v), exp(x.v)); }
__host__ __device__ inline T7 texp(T7 x) {
  double e = exp(x.v);
  return t7f(x, e, e);
}
__host__ __device__ inline T7 tpow43(T7 x) {
  double c = cbrt(x.v);
  return t7f(x, x.v * c, (4.0 / 3.0) * c);
}
__host__ __device__ inline T7 tpow23(T7 x) {
  double c = cbrt(x.v);
  return t7f(x, c * c, 2.0 / (3.0 * c));
}
__host__ __device__ inline T7 tmaxv(T7 x, T7 y) { return (x.v >= y.v) ? x : y; }

template <class T>
__host__ __device__ inline T pol_fzeta(T z) {
  return (tpow43(1.0 + z) + tpow43(1.0 - z) - 2.0) / (2.0 * 1.2599210498948732 - 2.0);
}
__host__ __device__ inline T3 pol_vwn_fit(T3 x, double A, double b, double c, double x0) {
  T3 X = x * x + b * x + c;
  const double X0 = x0 * x0 + b * x0 + c, Q = sqrt(4.0 * c - b * b);
  T3 at = tatan(Q / (2.0 * x + b));
  T3 xm = x - x0;
  return A * (tlog(x * x / X) + (2.0 * b / Q) * at - (b * x0 / X0) * (tlog(xm * xm / X) + (2.0 * (b + 2.0 * x0) / Q) * at));
}
__host__ __device__ inline T3 pol_eps_vwn(T3 rs, T3 z) {
  T3 x = tsqrt(rs);
  T3 eP = pol_vwn_fit(x, 0.0310907, 3.72744, 12.9352, -0.10498);
  T3 eF = pol_vwn_fit(x, 0.01554535, 7.06042, 18.0578, -0.32500);
  T3 al = pol_vwn_fit(x, -1.0 / (6.0 * HFG_PI * HFG_PI), 1.13107, 13.0045, -0.0047584);
  const double fpp = 4.0 / (9.0 * (1.2599210498948732 - 1.0));
  T3 f = pol_fzeta(z), z2 = z * z;
  T3 z4 = z2 * z2;
  return eP + al * f * (1.0 - z4) / fpp + (eF - eP) * f * z4;
}
template <class T>
__host__ __device__ inline T pol_pw_G(T rs, double A, double a1, double b1, double b2, double b3, double b4) {
  T s = tsqrt(rs);
  T den = (2.0 * A) * (b1 * s + b2 * rs + b3 * rs * s + b4 * rs * rs);
  return (-2.0 * A) * (1.0 + a1 * rs) * tlog1p(1.0 / den);
}
template <class T>
__host__ __device__ inline T pol_eps_pw(T rs, T z, bool mod) {
  T e0 = pol_pw_G(rs, mod ? 0.0310907 : 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
  T e1 = pol_pw_G(rs, mod ? 0.01554535 : 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
  T mac = pol_pw_G(rs, mod ? 0.0168869 : 0.016887, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);
  const double fz20 = mod ? 1.709920934161365617563962776245 : 1.709921;
  T f = pol_fzeta(z), z2 = z * z;
  T z4 = z2 * z2;
  return e0 - mac * f * (1.0 - z4) / fz20 + (e1 - e0) * f * z4;
}
template <class T>
__host__ __device__ inline T pol_eps_pbe_c(T n, T rs, T z, T sig, bool ext = false) {
  // ext: the stand-alone gga_c_pbe takes the external parameters; TPSS's inner PBE keeps the published constants
  const double beta = ext ? c_xcpar.c_beta : 0.06672455060314922;
  const double gamma = ext ? c_xcpar.c_gamma : (1.0 - 0.6931471805599453) / (HFG_PI * HFG_PI);
  const double B = beta / gamma;
  const double BB = ext ? c_xcpar.c_BB : 1.0;
  T ec = pol_eps_pw(rs, z, true);
  T phi = 0.5 * (tpow23(1.0 + z) + tpow23(1.0 - z));
  T phi3 = phi * phi * phi;
  T kf = tcbrt((3.0 * HFG_PI * HFG_PI) * n);
  T ks2 = (4.0 / HFG_PI) * kf;
  T t2 = sig / (4.0 * phi * phi * ks2 * n * n);
  T Aa = B / texpm1(-ec / (gamma * phi3));
  T At2 = Aa * t2;
  T f1 = t2 * (1.0 + BB * At2);
  return ec + gamma * phi3 * tlog1p(B * f1 / (1.0 + Aa * f1));
}

// lda_c_vwn_rpa, spin-polarised: libxc interpolates the paramagnetic and the ferromagnetic RPA fits with f(zeta) alone
__host__ __device__ inline T3 pol_eps_vwn_rpa(T3 rs, T3 z) {
  T3 x = tsqrt(rs);
  T3 eP = pol_vwn_fit(x, 0.0310907, 13.0720, 42.7198, -0.409286);
  T3 eF = pol_vwn_fit(x, 0.01554535, 20.1231, 101.578, -0.743294);
  return eP + (eF - eP) * pol_fzeta(z);
}
// gga_c_lyp for a spin-polarised density, Miehlich et al. eq 2 (energy per particle):
//   n e = -a 4 ra rb / (n (1 + d n^{-1/3}))
//         - a b omega { ra rb [2^{11/3} C_F (ra^{8/3} + rb^{8/3}) + (47/18 - 7 delta/18) s_t - (5/2 - delta/18)(s_aa + s_bb)
//                              - (delta - 11)/9 (ra s_aa + rb s_bb)/n] - (2/3) n^2 s_t + ((2/3) n^2 - ra^2) s_bb + ((2/3) n^2 - rb^2) s_aa }
// (vanishes identically for a fully polarised density: no self-correlation)
template <class T>
__host__ __device__ inline T pol_eps_lyp(T ra, T rb, T saa, T sab, T sbb) {
  const double CF = 0.3 * 9.570780000627305, c11 = 12.699208415745595;  // 2^{11/3}
  T n = ra + rb;
  T rm13 = 1.0 / tcbrt(n);
  T den = 1.0 + HFG_LYP_D * rm13;
  T delta = HFG_LYP_C * rm13 + HFG_LYP_D * rm13 / den;
  T rm23 = rm13 * rm13;
  T rm113 = rm23 * rm23 * rm23 * rm23 * rm23 * rm13;  // n^{-11/3}
  T omega = texp((-HFG_LYP_C) * rm13) / den * rm113;
  T st = saa + 2.0 * sab + sbb;
  T rab = ra * rb, n2 = n * n;
  T ra83 = ra * ra * tpow23(ra), rb83 = rb * rb * tpow23(rb);
  T t1 = (c11 * CF) * (ra83 + rb83) + (47.0 / 18.0 - (7.0 / 18.0) * delta) * st - (2.5 - delta / 18.0) * (saa + sbb) -
         ((delta - 11.0) / 9.0) * (ra * saa + rb * sbb) / n;
  // the last three terms of the published brace, -(2/3) n^2 s_t + ((2/3) n^2 - ra^2) s_bb + ((2/3) n^2 - rb^2) s_aa, collapse to
  // -(4/3) n^2 s_ab - ra^2 s_bb - rb^2 s_aa: summed as published they cancel to 1e-8 of their size in strongly polarised
  // regions (the checker keeps the published arrangement)
  T E = (-4.0 * HFG_LYP_A) * rab / (n * den) -
        (HFG_LYP_A * HFG_LYP_B) * omega * (rab * t1 - (4.0 / 3.0) * n2 * sab - ra * ra * sbb - rb * rb * saa);
  (void)st;
  return E / n;
}

/// adds functional id's exc (per particle of ra+rb), vrho[2], vsigma[3] (aa, ab, bb); ra + rb >= threshold assumed,
/// ra, rb already raised to the threshold; live_a, live_b: the channel's own density reached the threshold (a channel
/// below it is left out of the exchange sum, as libxc >= 5 does)
__host__ __device__ inline void eval_add_pol_basic(int id, double ra, double rb, double saa, double sab, double sbb, bool live_a,
                                             bool live_b, double &exc, double &va, double &vb, double &vsaa, double &vsab,
                                             double &vsbb) {
  const double rt = ra + rb;
  if (id == 131) {  // gga_c_lyp: depends on the three gradient invariants separately
    T7 A = t7var(ra, 0), B = t7var(rb, 1), Saa = t7var(saa, 2), Sab = t7var(sab, 3), Sbb = t7var(sbb, 4);
    T7 e = pol_eps_lyp(A, B, Saa, Sab, Sbb);
    T7 en = (A + B) * e;
    exc += e.v;
    va += en.d[0];
    vb += en.d[1];
    vsaa += en.d[2];
    vsab += en.d[3];
    vsbb += en.d[4];
    return;
  }
  if (id == 1 || id == 101 || id == 546 || id == 641 || id == -178 || id == 106) {
    Dual a = mk(2.0 * ra, 1.0, 0.0), b = mk(2.0 * rb, 1.0, 0.0);
    Dual sa = mk(4.0 * saa, 0.0, 1.0), sb = mk(4.0 * sbb, 0.0, 1.0);
    Dual ea, eb;
    switch (id) {
      case 1: ea = c_xcpar.x_alpha * eps_lda_x(a); eb = c_xcpar.x_alpha * eps_lda_x(b); break;
      case 546: ea = eps_lda_x_sr(a, 0.3, 2); eb = eps_lda_x_sr(b, 0.3, 2); break;
      case 641: ea = eps_lda_x_sr(a, 0.3, 1); eb = eps_lda_x_sr(b, 0.3, 1); break;
      case -178: ea = eps_cam_lda0_x(a); eb = eps_cam_lda0_x(b); break;
      case 106: ea = eps_gga_x_b88(a, sa); eb = eps_gga_x_b88(b, sb); break;
      default: ea = eps_gga_x_pbe(a, sa); eb = eps_gga_x_pbe(b, sb); break;
    }
    Dual na = a * ea, nb = b * eb;  // energy per volume of the doubled densities
    if (!live_a) na = mk(0.0, 0.0, 0.0);
    if (!live_b) nb = mk(0.0, 0.0, 0.0);
    exc += 0.5 * (na.v + nb.v) / rt;
    va += na.dr;          // d/d ra [ (1/2) n(2 ra) ] = n'(2 ra)
    vb += nb.dr;
    vsaa += 2.0 * na.ds;  // d/d saa [ (1/2) n(.., 4 saa) ]
    vsbb += 2.0 * nb.ds;
    return;
  }
  T3 a = t3(ra, 1.0, 0.0, 0.0), b = t3(rb, 0.0, 1.0, 0.0), st = t3(saa + 2.0 * sab + sbb, 0.0, 0.0, 1.0);
  T3 n = a + b;
  T3 rs = tcbrt((3.0 / (4.0 * HFG_PI)) / n);
  T3 z = (a - b) / n;
  T3 e;
  switch (id) {
    case 7: e = pol_eps_vwn(rs, z); break;
    case 8: e = pol_eps_vwn_rpa(rs, z); break;
    case 12: e = pol_eps_pw(rs, z, false); break;
    case 13: e = pol_eps_pw(rs, z, true); break;
    case 130: e = pol_eps_pbe_c(n, rs, z, st, true); break;
    default: return;
  }
  T3 en = n * e;
  exc += e.v;
  va += en.a;
  vb += en.b;
  vsaa += en.s;
  vsab += 2.0 * en.s;
  vsbb += en.s;
}

/// eval_add_pol_basic plus the composite functionals (no recursion: a recursive device function needs a dynamic stack)
__host__ __device__ inline void eval_add_pol(int id, double ra, double rb, double saa, double sab, double sbb, bool live_a,
                                             bool live_b, double &exc, double &va, double &vb, double &vsaa, double &vsab,
                                             double &vsbb) {
#define HFG_ADD_SCALED(ID, W)                                                                          \
  {                                                                                                    \
    double e_ = 0.0, a_ = 0.0, b_ = 0.0, x_ = 0.0, y_ = 0.0, z_ = 0.0;                                 \
    eval_add_pol_basic(ID, ra, rb, saa, sab, sbb, live_a, live_b, e_, a_, b_, x_, y_, z_);             \
    exc += (W)*e_;                                                                                     \
    va += (W)*a_;                                                                                      \
    vb += (W)*b_;                                                                                      \
    vsaa += (W)*x_;                                                                                    \
    vsab += (W)*y_;                                                                                    \
    vsbb += (W)*z_;                                                                                    \
  }
  if (id == 406) {  // hyb_gga_xc_pbeh (PBE0), DFT part: 0.75 gga_x_pbe + gga_c_pbe
    HFG_ADD_SCALED(101, 0.75)
    HFG_ADD_SCALED(130, 1.0)
  } else if (id == 178) {  // hyb_lda_xc_cam_lda0, DFT part: spin-scaled exchange mixture + lda_c_pw_mod
    HFG_ADD_SCALED(-178, 1.0)
    HFG_ADD_SCALED(13, 1.0)
  } else if (id == 402) {  // hyb_gga_xc_b3lyp, DFT part
    HFG_ADD_SCALED(1, 0.08)
    HFG_ADD_SCALED(106, 0.72)
    HFG_ADD_SCALED(8, 0.19)
    HFG_ADD_SCALED(131, 0.81)
  } else
    eval_add_pol_basic(id, ra, rb, saa, sab, sbb, live_a, live_b, exc, va, vb, vsaa, vsab, vsbb);
#undef HFG_ADD_SCALED
}

// ---------------------------------------------------------------------------------------------------------
// meta-GGA (tau-dependent), spin-unpolarised: TPSS exchange (libxc id 202) and correlation (231).  The T3 slots
// (a, b, s) carry the derivatives with respect to (rho, sigma, tau) here.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline T3 tmaxv(T3 x, T3 y) { return (x.v >= y.v) ? x : y; }

__host__ __device__ inline T3 mg_eps_tpss_x(T3 rho, T3 sig, T3 tau) {
  const double b = 0.40, c = 1.59096, e = 1.537, kappa = 0.804, mu = 0.21951, muge = 10.0 / 81.0;
  const double c32 = 9.570780000627305;  // (3 pi^2)^{2/3}
  T3 exu = (-0.75 * cbrt(3.0 / HFG_PI)) * tcbrt(rho);
  T3 rho23 = tpow23(rho);
  T3 pp = sig / ((4.0 * c32) * rho * rho * rho23);
  T3 tauw = sig / (8.0 * rho);
  T3 tt = tmaxv(tau, tauw);
  T3 z = tauw / tt;
  T3 tunif = (0.3 * c32) * rho * rho23;
  T3 alpha = (tt - tauw) / tunif;
  T3 qb = (9.0 / 20.0) * (alpha - 1.0) / tsqrt(1.0 + b * alpha * (alpha - 1.0)) + (2.0 / 3.0) * pp;
  T3 z2 = z * z;
  T3 opz2 = 1.0 + z2;
  T3 num = (muge + c * z2 / (opz2 * opz2)) * pp + (146.0 / 2025.0) * qb * qb -
           (73.0 / 405.0) * qb * tsqrt(0.5 * (9.0 / 25.0) * z2 + 0.5 * pp * pp) + (muge * muge / kappa) * pp * pp +
           (2.0 * sqrt(e) * muge * 9.0 / 25.0) * z2 + (e * mu) * pp * pp * pp;
  T3 den = 1.0 + sqrt(e) * pp;
  T3 x = num / (den * den);
  T3 F = 1.0 + kappa - kappa * kappa / (kappa + x);
  return exu * F;
}

template <class T>
__host__ __device__ inline T mg_eps_pbe_c_fullpol(T n, T sig) {
  const double beta = 0.06672455060314922;
  const double gamma = (1.0 - 0.6931471805599453) / (HFG_PI * HFG_PI);
  const double B = beta / gamma;
  T rs = tcbrt((3.0 / (4.0 * HFG_PI)) / n);
  T ec = pol_pw_G(rs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
  const double phi = 0.7937005259840998, phi3 = 0.5;  // 2^{-1/3}
  T kf = tcbrt((3.0 * HFG_PI * HFG_PI) * n);
  T ks2 = (4.0 / HFG_PI) * kf;
  T t2 = sig / ((4.0 * phi * phi) * ks2 * n * n);
  T Aa = B / texpm1(-ec / (gamma * phi3));
  T At2 = Aa * t2;
  return ec + (gamma * phi3) * tlog1p(B * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2));
}

// TPSS correlation for a spin-polarised density (Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003), eqs 11-14):
//   rev = e_PBE [1 + C(zeta,xi) z^2] - [1 + C(zeta,xi)] z^2 sum_s (n_s/n) max(e_PBE(n_s,0,grad n_s,0), e_PBE)
//   C(zeta,xi) = C(zeta,0) / {1 + xi^2 [(1+zeta)^{-4/3} + (1-zeta)^{-4/3}]/2}^4,  C(zeta,0) = 0.53 + 0.87 z^2 + 0.50 z^4 + 2.26 z^6
//   xi = |grad zeta| / (2 (3 pi^2 n)^{1/3}),  z = tau_W/tau,  e_c = rev [1 + d rev z^3]
__host__ __device__ inline T7 mg_eps_tpss_c_pol(T7 ra, T7 rb, T7 saa, T7 sab, T7 sbb, T7 ta, T7 tb) {
  const double d = 2.8;
  T7 n = ra + rb;
  T7 rs = tcbrt((3.0 / (4.0 * HFG_PI)) / n);
  T7 zeta = (ra - rb) / n;
  T7 st = saa + 2.0 * sab + sbb;
  T7 epbe = pol_eps_pbe_c(n, rs, zeta, st);
  T7 eta = tmaxv(mg_eps_pbe_c_fullpol(ra, saa), epbe);
  T7 etb = tmaxv(mg_eps_pbe_c_fullpol(rb, sbb), epbe);
  T7 tauw = st / (8.0 * n);
  T7 tt = tmaxv(ta + tb, tauw);
  T7 z = tauw / tt;
  T7 z2 = z * z;
  T7 omz = 1.0 - zeta, opz = 1.0 + zeta;
  T7 gz2 = (omz * omz * saa - 2.0 * omz * opz * sab + opz * opz * sbb) / (n * n);  // |grad zeta|^2
  T7 kf = tcbrt((3.0 * HFG_PI * HFG_PI) * n);
  T7 xi2 = gz2 / (4.0 * kf * kf);
  T7 zz = zeta * zeta;
  T7 C0 = 0.53 + 0.87 * zz + 0.50 * zz * zz + 2.26 * zz * zz * zz;
  T7 den = 1.0 + 0.5 * xi2 * (1.0 / tpow43(opz) + 1.0 / tpow43(omz));
  T7 den2 = den * den;
  T7 Cz = C0 / (den2 * den2);
  T7 rev = epbe * (1.0 + Cz * z2) - (1.0 + Cz) * z2 * (ra * eta + rb * etb) / n;
  return rev * (1.0 + d * rev * z2 * z);
}

__host__ __device__ inline T3 mg_eps_tpss_c(T3 rho, T3 sig, T3 tau) {
  const double d = 2.8, C0 = 0.53;
  T3 rs = tcbrt((3.0 / (4.0 * HFG_PI)) / rho);
  T3 epbe = pol_eps_pbe_c(rho, rs, t3(0.0), sig);
  T3 esig = mg_eps_pbe_c_fullpol(0.5 * rho, 0.25 * sig);
  T3 etil = tmaxv(esig, epbe);
  T3 tauw = sig / (8.0 * rho);
  T3 tt = tmaxv(tau, tauw);
  T3 z = tauw / tt;
  T3 z2 = z * z;
  T3 rev = epbe * (1.0 + C0 * z2) - (1.0 + C0) * z2 * etil;
  return rev * (1.0 + d * rev * z2 * z);
}

__host__ __device__ inline bool is_mgga(int id) { return id == 202 || id == 231; }

/// adds a meta-GGA's exc, vrho, vsigma, vtau at one point (rho >= threshold assumed)
__host__ __device__ inline void eval_add_mgga(int id, double rho, double sigma, double tau, bool live, double &exc,
                                              double &vrho, double &vsigma, double &vtau) {
  if (id == 202 && !live) return;  // exchange channel rho/2 below the threshold, see eval_add
  T3 r = t3(rho, 1.0, 0.0, 0.0), s = t3(fmax(sigma, 1e-40), 0.0, 1.0, 0.0), t = t3(fmax(tau, 1e-40), 0.0, 0.0, 1.0);
  T3 e = (id == 202) ? mg_eps_tpss_x(r, s, t) : mg_eps_tpss_c(r, s, t);
  T3 en = r * e;
  exc += e.v;
  vrho += en.a;
  vsigma += en.b;
  vtau += en.s;
}

/// spin-polarised meta-GGA: adds exc (per particle of ra + rb), v_rho[2], v_sigma[3] (aa, ab, bb), v_tau[2]; ra, rb
/// already raised to the threshold.  Exchange by spin scaling, E_x[a,b] = (E_x[2a] + E_x[2b])/2.
/// live_a, live_b: the channel's own density reached the threshold.  A channel below it is left out of the exchange sum
/// (libxc screens it the same way): its v_tau = d(n e)/d tau would be of order 1/tau_unif(threshold) and enters the Fock
/// matrix unweighted, unlike v_rho and v_sigma which multiply the channel's density or gradient.
__host__ __device__ inline void eval_add_mgga_pol(int id, double ra, double rb, double saa, double sab, double sbb, double ta,
                                                  double tb, bool live_a, bool live_b, double &exc, double &va, double &vb,
                                                  double &vsaa, double &vsab, double &vsbb, double &vta, double &vtb) {
  const double rt = ra + rb;
  if (id == 202) {
    T3 na = t3(0.0), nb = t3(0.0);
    if (live_a) {
      T3 Ra = t3(2.0 * ra, 1.0, 0.0, 0.0), Sa = t3(fmax(4.0 * saa, 1e-40), 0.0, 1.0, 0.0), Ta = t3(fmax(2.0 * ta, 1e-40), 0.0, 0.0, 1.0);
      na = Ra * mg_eps_tpss_x(Ra, Sa, Ta);
    }
    if (live_b) {
      T3 Rb = t3(2.0 * rb, 1.0, 0.0, 0.0), Sb = t3(fmax(4.0 * sbb, 1e-40), 0.0, 1.0, 0.0), Tb = t3(fmax(2.0 * tb, 1e-40), 0.0, 0.0, 1.0);
      nb = Rb * mg_eps_tpss_x(Rb, Sb, Tb);
    }
    exc += 0.5 * (na.v + nb.v) / rt;
    va += na.a;  // d/d ra [ n(2 ra, 4 saa, 2 ta)/2 ]
    vb += nb.a;
    vsaa += 2.0 * na.b;
    vsbb += 2.0 * nb.b;
    vta += na.s;
    vtb += nb.s;
    return;
  }
  if (id != 231) return;
  T7 A = t7var(ra, 0), B = t7var(rb, 1);
  T7 Saa = t7var(fmax(saa, 1e-40), 2), Sab = t7var(sab, 3), Sbb = t7var(fmax(sbb, 1e-40), 4);
  T7 Ta = t7var(fmax(ta, 1e-40), 5), Tb = t7var(fmax(tb, 1e-40), 6);
  T7 e = mg_eps_tpss_c_pol(A, B, Saa, Sab, Sbb, Ta, Tb);
  T7 en = (A + B) * e;
  exc += e.v;
  va += en.d[0];
  vb += en.d[1];
  vsaa += en.d[2];
  vsab += en.d[3];
  vsbb += en.d[4];
  vta += en.d[5];
  vtb += en.d[6];
}

}  // namespace xc
}  // namespace hfg
