// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Spin-unpolarised LDA / PBE functionals with hand-derived analytic first derivatives, in the
// conventions of the libxc calls the reference makes (src/diatomic/dftgrid.cpp:343-458:
// xc_lda_exc_vxc / xc_gga_exc_vxc with XC_UNPOLARIZED, exc per particle, vrho = d(rho exc)/d rho,
// vsigma = d(rho exc)/d sigma, points with rho below the density threshold give zero).
// libxc itself is not vendored by the reference and is absent from this image; the formulas
// are the published ones (Slater/Dirac; Vosko-Wilk-Nusair 1980 fit V; Perdew-Wang 1992 with
// libxc's "pw_mod" constants as used inside gga_c_pbe; Perdew-Burke-Ernzerhof 1996).
// PARITY UNPINNED with respect to a libxc binary.
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <sstream>
#include <strings.h>

namespace oracle {

namespace {
const double PI = 3.14159265358979323846;

// ---- Slater exchange: exc = -3/4 (3/pi)^{1/3} rho^{1/3} ----
// external functional parameters (libxc's ext_params of lda_x, gga_x_pbe, gga_c_pbe; orc_set_xc_params)
struct XCParams {
  double x_alpha = 1.0, x_kappa = 0.8040, x_mu = 0.06672455060314922 * PI * PI / 3.0;
  double c_beta = 0.06672455060314922, c_gamma = (1.0 - 0.6931471805599453094) / (PI * PI), c_BB = 1.0;
};
XCParams g_xcp;

void lda_x(double rho, double &exc, double &vrho) {
  double cx = -0.75 * cbrt(3.0 / PI);
  exc = cx * cbrt(rho);
  vrho = 4.0 / 3.0 * exc;
}

// ---- short-range LDA exchange (libxc lda_x_erf / lda_x_yukawa): exc = exc_LDA F(a), a = omega/(2 kF) ----
// F and dF/da written out by hand (closed forms for moderate a, the 1/a^2 expansions beyond), see xc_device.h for
// the formulas and sources.  kind 1: Yukawa, 2: erfc.
void attenuation(double a, int kind, double &F, double &dF) {
  if (kind == 2) {
    if (a > 0.75) {
      double u = 1.0 / (a * a), t = 1.0;
      F = 0.0;
      dF = 0.0;
      for (int k = 1; k <= 24; k++) {
        t *= -0.25 * u / k;
        double c = -t * 2.0 / ((2.0 * k + 1.0) * (k + 1.0) * (k + 2.0));
        F += c;
        dF += -2.0 * k * c / a;
      }
      return;
    }
    double e = exp(-0.25 / (a * a)), a3 = a * a * a;
    double G = a * (sqrt(PI) * erf(0.5 / a) + (2.0 * a - 4.0 * a3) * e - 3.0 * a + 4.0 * a3);
    double dG = sqrt(PI) * erf(0.5 / a) + (2.0 * a - 16.0 * a3) * e - 6.0 * a + 16.0 * a3;
    F = 1.0 - 8.0 / 3.0 * G;
    dF = -8.0 / 3.0 * dG;
  } else {
    if (a > 2.0) {
      double u = 1.0 / (a * a), t = -1.0;
      F = 0.0;
      dF = 0.0;
      for (int k = 1; k <= 40; k++) {
        t *= -u;
        double c = t * 2.0 / ((2.0 * k + 1.0) * (k + 1.0) * (k + 2.0));
        F += c;
        dF += -2.0 * k * c / a;
      }
      return;
    }
    double a2 = a * a, lg = log(1.0 + 1.0 / a2);
    double G = a * (atan(1.0 / a) + 0.25 * a - 0.25 * a * (a2 + 3.0) * lg);
    double dG = atan(1.0 / a) - a / (1.0 + a2) + 0.5 * a - (a2 * a + 1.5 * a) * lg + a * (a2 + 3.0) / (2.0 * (a2 + 1.0));
    F = 1.0 - 8.0 / 3.0 * G;
    dF = -8.0 / 3.0 * dG;
  }
}
void lda_x_sr(double rho, double omega, int kind, double &exc, double &vrho) {
  double ex, vx;
  lda_x(rho, ex, vx);
  double kf = cbrt(3.0 * PI * PI * rho), a = 0.5 * omega / kf, F, dF;
  attenuation(a, kind, F, dF);
  exc = ex * F;
  vrho = ex * (4.0 / 3.0 * F - a / 3.0 * dF);  // da/drho = -a/(3 rho)
}
// exchange part of hyb_lda_xc_cam_lda0: 1/2 lda_x + 1/4 lda_x_erf(omega = 1/3)
void cam_lda0_x(double rho, double &exc, double &vrho) {
  double e1, v1, e2, v2;
  lda_x(rho, e1, v1);
  lda_x_sr(rho, 1.0 / 3.0, 2, e2, v2);
  exc = 0.5 * e1 + 0.25 * e2;
  vrho = 0.5 * v1 + 0.25 * v2;
}

// ---- VWN paramagnetic correlation: the Ceperley-Alder fit (VWN5, lda_c_vwn) and the RPA fit (lda_c_vwn_rpa) ----
void lda_c_vwn_par(double rho, double A, double b, double c, double x0, double &exc, double &vrho) {
  double rs = cbrt(3.0 / (4.0 * PI * rho));
  double x = sqrt(rs);
  double X = x * x + b * x + c;
  double X0 = x0 * x0 + b * x0 + c;
  double Q = sqrt(4.0 * c - b * b);
  double at = atan(Q / (2.0 * x + b));
  exc = A * (log(x * x / X) + 2.0 * b / Q * at -
             b * x0 / X0 * (log((x - x0) * (x - x0) / X) + 2.0 * (b + 2.0 * x0) / Q * at));
  double den = Q * Q + (2.0 * x + b) * (2.0 * x + b);
  double dedx = A * (2.0 / x - (2.0 * x + b) / X - 4.0 * b / den -
                     b * x0 / X0 * (2.0 / (x - x0) - (2.0 * x + b) / X - 4.0 * (2.0 * x0 + b) / den));
  // rho d exc/d rho = -(rs/3) d exc/d rs = -(x/6) d exc/dx
  vrho = exc - x / 6.0 * dedx;
}
void lda_c_vwn(double rho, double &exc, double &vrho) { lda_c_vwn_par(rho, 0.0310907, 3.72744, 12.9352, -0.10498, exc, vrho); }
void lda_c_vwn_rpa(double rho, double &exc, double &vrho) { lda_c_vwn_par(rho, 0.0310907, 13.0720, 42.7198, -0.409286, exc, vrho); }

// ---- Becke 88 exchange (Phys. Rev. A 38, 3098): hand-derived.  Per spin channel (density r = rho/2, gradient invariant
// s = sigma/4):  e_s = -r^{4/3} [Cx + beta g(x)],  g = x^2/(1 + 6 beta x asinh x),  x = sqrt(s)/r^{4/3};
// unpolarised energy per volume 2 e_s; d/d rho and d/d sigma through x^2 = s r^{-8/3}
void gga_x_b88(double rho, double sigma, double &exc, double &vrho, double &vsigma) {
  const double beta = 0.0042, Cx = 1.5 * cbrt(3.0 / (4.0 * PI));
  const double r = 0.5 * rho, sg = 0.25 * sigma;
  const double r13 = cbrt(r), r43 = r * r13;
  const double t = sg / (r43 * r43);  // x^2
  const double x = sqrt(t), as = asinh(x);
  const double D = 1.0 + 6.0 * beta * x * as;
  const double g = t / D;
  // dg/dt = 1/D - t/D^2 * 6 beta d(x asinh x)/dt,  d(x asinh x)/dt = asinh(x)/(2x) + 1/(2 sqrt(1+t))
  const double dh = (x > 1e-8 ? as / (2.0 * x) : 0.5) + 0.5 / sqrt(1.0 + t);
  const double dg = 1.0 / D - t / (D * D) * 6.0 * beta * dh;
  const double es = -r43 * (Cx + beta * g);  // per volume, one channel
  // d es/d r = -(4/3) r^{1/3} (Cx + beta g) - r^{4/3} beta dg dt/dr,  dt/dr = -(8/3) t/r
  const double des_dr = -(4.0 / 3.0) * r13 * (Cx + beta * g) + r43 * beta * dg * (8.0 / 3.0) * t / r;
  const double des_ds = -r43 * beta * dg / (r43 * r43);  // dt/ds = r^{-8/3}
  exc = 2.0 * es / rho;
  vrho = des_dr;           // d(2 es)/d rho = 2 des/dr * (1/2)
  vsigma = 0.5 * des_ds;   // d(2 es)/d sigma = 2 des/ds * (1/4)
}

// ---- PW92 paramagnetic correlation; mod=true uses the higher-precision constants of pw_mod ----
void pw92(double rs, bool mod, double &ec, double &decdrs) {
  const double a = mod ? 0.0310907 : 0.031091;  // libxc lda_c_pw.c: par_pw / par_pw_mod
  const double a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
  double srs = sqrt(rs);
  double q0 = -2.0 * a * (1.0 + a1 * rs);
  double q1 = 2.0 * a * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
  double q1p = a * (b1 / srs + 2.0 * b2 + 3.0 * b3 * srs + 4.0 * b4 * rs);
  double lg = log1p(1.0 / q1);  // log(1 + 1/q1) rounds to 0 for rs > ~1e9 (rho < 1e-28) and PBE's A = beta/gamma/(exp(-ec/gamma) - 1) to inf
  ec = q0 * lg;
  decdrs = -2.0 * a * a1 * lg - q0 * q1p / (q1 * q1 + q1);
}

void lda_c_pw(double rho, double &exc, double &vrho) {
  double rs = cbrt(3.0 / (4.0 * PI * rho));
  double ec, dec;
  pw92(rs, false, ec, dec);
  exc = ec;
  vrho = ec - rs / 3.0 * dec;
}
void lda_c_pw_mod(double rho, double &exc, double &vrho) {
  double rs = cbrt(3.0 / (4.0 * PI * rho));
  double ec, dec;
  pw92(rs, true, ec, dec);
  exc = ec;
  vrho = ec - rs / 3.0 * dec;
}

// ---- PBE exchange ----
void gga_x_pbe(double rho, double sigma, double &exc, double &vrho, double &vsigma) {
  const double kappa = g_xcp.x_kappa;
  const double mu = g_xcp.x_mu;
  double exu = -0.75 * cbrt(3.0 / PI) * cbrt(rho);
  double kf2 = pow(3.0 * PI * PI * rho, 2.0 / 3.0);
  double s2 = sigma / (4.0 * kf2 * rho * rho);
  double d = 1.0 + mu * s2 / kappa;
  double Fx = 1.0 + kappa - kappa / d;
  double dF = mu / (d * d);  // dFx/d(s^2)
  exc = exu * Fx;
  vrho = exu * (4.0 / 3.0 * Fx - 8.0 / 3.0 * s2 * dF);
  vsigma = rho * exu * dF / (4.0 * kf2 * rho * rho);
}

// ---- PBE correlation ----
void gga_c_pbe(double rho, double sigma, double &exc, double &vrho, double &vsigma) {
  const double beta = g_xcp.c_beta;
  const double gamma = g_xcp.c_gamma;
  const double B = beta / gamma, BB = g_xcp.c_BB;  // libxc: f1 = t^2 + BB A t^4, H = gamma log(1 + B f1/(1 + A f1))
  double rs = cbrt(3.0 / (4.0 * PI * rho));
  double ec, dec;
  pw92(rs, true, ec, dec);
  double kf = cbrt(3.0 * PI * PI * rho);
  double ks2 = 4.0 * kf / PI;
  double u = sigma / (4.0 * ks2 * rho * rho);  // t^2
  double E = exp(-ec / gamma);
  double A = B / expm1(-ec / gamma);
  double N = B * u * (1.0 + BB * A * u);
  double D = 1.0 + A * u + BB * A * A * u * u;
  double arg = 1.0 + N / D;
  double H = gamma * log1p(N / D);
  double dN_du = B * (1.0 + 2.0 * BB * A * u), dD_du = A + 2.0 * BB * A * A * u;
  double dN_dA = B * BB * u * u, dD_dA = u + 2.0 * BB * A * u * u;
  double dH_du = gamma * (dN_du * D - N * dD_du) / (D * D * arg);
  double dH_dA = gamma * (dN_dA * D - N * dD_dA) / (D * D * arg);
  double dA_dec = A * A * E / (B * gamma);
  double rho_dec_drho = -rs / 3.0 * dec;  // rho * d ec / d rho
  exc = ec + H;
  vrho = ec + rho_dec_drho + H + dH_dA * dA_dec * rho_dec_drho - 7.0 / 3.0 * u * dH_du;
  vsigma = rho * dH_du / (4.0 * ks2 * rho * rho);
}
}  // namespace

// Lee-Yang-Parr correlation, defined behind the differentiation type below
void gga_c_lyp(double rho, double sigma, double &exc, double &vrho, double &vsigma);

bool xc_is_gga(int id) {  // needs the gradient
  return id == 101 || id == 130 || id == 406 || id == 202 || id == 231 || id == 106 || id == 131 || id == 402;
}

void xc_unpolarized(int id, size_t N, const double *rho, const double *sigma, double *exc, double *vrho,
                    double *vsigma, double thr) {
  for (size_t i = 0; i < N; i++) {
    exc[i] = 0.0;
    vrho[i] = 0.0;
    if (vsigma) vsigma[i] = 0.0;
    double r = rho[i];
    if (!(r >= thr) || r <= 0.0) continue;
    double e = 0, v = 0, vs = 0;
    // Exchange is a sum over the two spin channels, each carrying r/2; libxc (>= 5) leaves a channel whose density is
    // below the threshold out of the sum, in the unpolarised evaluation too (its generated code tests
    // rho/2 <= dens_threshold), so that restricted and unrestricted builds agree on a closed shell.
    const bool live = 0.5 * r >= thr;
    int idl = id;
    if (!live) {
      if (id == 1 || id == 101 || id == 546 || id == 641 || id == 106) continue;
      if (id == 178) idl = 13;  // the hybrids keep their correlation part
      if (id == 406) idl = 130;
      if (id == 402) idl = -402;
    }
    switch (idl) {
      case 1:
        lda_x(r, e, v);
        e *= g_xcp.x_alpha;
        v *= g_xcp.x_alpha;
        break;
      case 7: lda_c_vwn(r, e, v); break;
      case 12: lda_c_pw(r, e, v); break;
      case 13: lda_c_pw_mod(r, e, v); break;
      case 546: lda_x_sr(r, 0.3, 2, e, v); break;
      case 641: lda_x_sr(r, 0.3, 1, e, v); break;
      case 178: {  // hyb_lda_xc_cam_lda0, DFT part
        double e2, v2;
        cam_lda0_x(r, e, v);
        lda_c_pw_mod(r, e2, v2);
        e += e2;
        v += v2;
        break;
      }
      case 101: gga_x_pbe(r, sigma[i], e, v, vs); break;
      case 130: gga_c_pbe(r, sigma[i], e, v, vs); break;
      case 8: lda_c_vwn_rpa(r, e, v); break;
      case 106: gga_x_b88(r, sigma[i], e, v, vs); break;
      case 131: gga_c_lyp(r, sigma[i], e, v, vs); break;
      case 402:     // hyb_gga_xc_b3lyp, DFT part: 0.08 lda_x + 0.72 gga_x_b88 + 0.19 lda_c_vwn_rpa + 0.81 gga_c_lyp
      case -402: {  // ... its correlation part alone (exchange channel below the threshold)
        double e1 = 0, v1 = 0, e2 = 0, v2 = 0, vs2 = 0, e3, v3, e4, v4, vs4;
        if (idl == 402) {
          lda_x(r, e1, v1);
          gga_x_b88(r, sigma[i], e2, v2, vs2);
        }
        lda_c_vwn_rpa(r, e3, v3);
        gga_c_lyp(r, sigma[i], e4, v4, vs4);
        e = 0.08 * e1 + 0.72 * e2 + 0.19 * e3 + 0.81 * e4;
        v = 0.08 * v1 + 0.72 * v2 + 0.19 * v3 + 0.81 * v4;
        vs = 0.72 * vs2 + 0.81 * vs4;
        break;
      }
      case 406: {  // hyb_gga_xc_pbeh (PBE0), DFT part
        double e2, v2, vs2;
        gga_x_pbe(r, sigma[i], e, v, vs);
        gga_c_pbe(r, sigma[i], e2, v2, vs2);
        e = 0.75 * e + e2;
        v = 0.75 * v + v2;
        vs = 0.75 * vs + vs2;
        break;
      }
      default: {
        std::ostringstream oss;
        oss << "Functional " << id << " not found!";
        throw std::runtime_error(oss.str());
      }
    }
    exc[i] = e;
    vrho[i] = v;
    if (vsigma) vsigma[i] = vs;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Spin-polarised functionals.  Exchange follows from the spin-scaling relation
//   E_x[rho_a, rho_b] = (E_x[2 rho_a] + E_x[2 rho_b]) / 2
// applied to the hand-derived unpolarised formulas above.  Correlation depends on (rs, zeta, sigma_total);
// its derivatives are taken with a small forward-mode differentiation type over (rho_a, rho_b, sigma_tot).
// Densities of one spin channel below the threshold are raised to it (libxc >= 5 does the same before
// evaluating a polarised functional), which keeps (1 -+ zeta) away from the non-analytic end points; such a
// channel is left out of the exchange sum (the generated exchange code tests rho_s <= dens_threshold).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int ND = 7;  // derivative slots: (rho_a, rho_b, sigma_tot) for the GGAs, (rho, sigma, tau) for the unpolarised
                       // meta-GGAs, (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb, tau_a, tau_b) for the polarised ones
struct D3 {
  double v, d[ND];
};
inline D3 C(double v) {
  D3 r;
  r.v = v;
  for (int k = 0; k < ND; k++) r.d[k] = 0.0;
  return r;
}
inline D3 var(double v, int k) {
  D3 r = C(v);
  r.d[k] = 1.0;
  return r;
}
inline D3 operator+(D3 a, D3 b) {
  D3 r;
  r.v = a.v + b.v;
  for (int k = 0; k < ND; k++) r.d[k] = a.d[k] + b.d[k];
  return r;
}
inline D3 operator-(D3 a, D3 b) {
  D3 r;
  r.v = a.v - b.v;
  for (int k = 0; k < ND; k++) r.d[k] = a.d[k] - b.d[k];
  return r;
}
inline D3 operator-(D3 a) {
  D3 r;
  r.v = -a.v;
  for (int k = 0; k < ND; k++) r.d[k] = -a.d[k];
  return r;
}
inline D3 operator*(D3 a, D3 b) {
  D3 r;
  r.v = a.v * b.v;
  for (int k = 0; k < ND; k++) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
inline D3 chain(D3 a, double f, double df) {
  D3 r;
  r.v = f;
  for (int k = 0; k < ND; k++) r.d[k] = df * a.d[k];
  return r;
}
inline D3 operator/(D3 a, D3 b) { return a * chain(b, 1.0 / b.v, -1.0 / (b.v * b.v)); }
inline D3 operator+(D3 a, double c) { return a + C(c); }
inline D3 operator+(double c, D3 a) { return a + C(c); }
inline D3 operator-(D3 a, double c) { return a - C(c); }
inline D3 operator-(double c, D3 a) { return C(c) - a; }
inline D3 operator*(double c, D3 a) { return C(c) * a; }
inline D3 operator*(D3 a, double c) { return C(c) * a; }
inline D3 operator/(D3 a, double c) { return a * (1.0 / c); }
inline D3 operator/(double c, D3 a) { return C(c) / a; }
inline D3 Dsqrt(D3 a) { double s = sqrt(a.v); return chain(a, s, 0.5 / s); }
inline D3 Dcbrt(D3 a) { double c = cbrt(a.v); return chain(a, c, c / (3.0 * a.v)); }
inline D3 Dlog(D3 a) { return chain(a, log(a.v), 1.0 / a.v); }
inline D3 Dexp(D3 a) { double e = exp(a.v); return chain(a, e, e); }
inline D3 Dlog1p(D3 a) { return chain(a, log1p(a.v), 1.0 / (1.0 + a.v)); }
inline D3 Dexpm1(D3 a) { return chain(a, expm1(a.v), exp(a.v)); }
inline D3 Datan(D3 a) { return chain(a, atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
inline D3 Dpow43(D3 a) { double c = cbrt(a.v); return chain(a, a.v * c, 4.0 / 3.0 * c); }
inline D3 Dpow23(D3 a) { double c = cbrt(a.v); return chain(a, c * c, 2.0 / (3.0 * c)); }

// f(zeta) = ((1+z)^{4/3} + (1-z)^{4/3} - 2) / (2^{4/3} - 2)
D3 fzeta(D3 z) { return (Dpow43(1.0 + z) + Dpow43(1.0 - z) - 2.0) / (2.0 * cbrt(2.0) - 2.0); }

// one VWN fit: A [ ln(x^2/X) + 2b/Q atan(Q/(2x+b)) - b x0/X0 ( ln((x-x0)^2/X) + 2(b+2x0)/Q atan(Q/(2x+b)) ) ]
D3 vwn_fit(D3 x, double A, double b, double c, double x0) {
  D3 X = x * x + b * x + c;
  double X0 = x0 * x0 + b * x0 + c, Q = sqrt(4.0 * c - b * b);
  D3 at = Datan(Q / (2.0 * x + b));
  D3 xm = x - x0;
  return A * (Dlog(x * x / X) + (2.0 * b / Q) * at - (b * x0 / X0) * (Dlog(xm * xm / X) + (2.0 * (b + 2.0 * x0) / Q) * at));
}

// VWN5 (libxc lda_c_vwn): e_P + alpha f(z)(1-z^4)/f''(0) + (e_F - e_P) f(z) z^4
D3 eps_vwn(D3 rs, D3 z) {
  D3 x = Dsqrt(rs);
  D3 eP = vwn_fit(x, 0.0310907, 3.72744, 12.9352, -0.10498);
  D3 eF = vwn_fit(x, 0.01554535, 7.06042, 18.0578, -0.32500);
  D3 al = vwn_fit(x, -1.0 / (6.0 * PI * PI), 1.13107, 13.0045, -0.0047584);
  const double fpp = 4.0 / (9.0 * (cbrt(2.0) - 1.0));
  D3 f = fzeta(z), z4 = z * z * z * z;
  return eP + al * f * (1.0 - z4) / fpp + (eF - eP) * f * z4;
}

D3 pw_G(D3 rs, double A, double a1, double b1, double b2, double b3, double b4) {
  D3 s = Dsqrt(rs);
  D3 den = (2.0 * A) * (b1 * s + b2 * rs + b3 * rs * s + b4 * rs * rs);
  return (-2.0 * A) * (1.0 + a1 * rs) * Dlog1p(1.0 / den);
}

// PW92 (libxc lda_c_pw / lda_c_pw_mod): e0 + alpha_c f(z)(1-z^4)/f''(0) + (e1-e0) f(z) z^4, alpha_c = -G(third set)
D3 eps_pw(D3 rs, D3 z, bool mod) {
  D3 e0 = pw_G(rs, mod ? 0.0310907 : 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
  D3 e1 = pw_G(rs, mod ? 0.01554535 : 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
  D3 mac = pw_G(rs, mod ? 0.0168869 : 0.016887, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);
  const double fz20 = mod ? 1.709920934161365617563962776245 : 1.709921;
  D3 f = fzeta(z), z4 = z * z * z * z;
  return e0 - mac * f * (1.0 - z4) / fz20 + (e1 - e0) * f * z4;
}

// PBE correlation (libxc gga_c_pbe): e_pw_mod(rs,z) + gamma phi^3 ln(1 + beta/gamma t^2 (1+A t^2)/(1+A t^2+A^2 t^4))
D3 eps_pbe_c(D3 rho, D3 rs, D3 z, D3 sig) {
  const double beta = g_xcp.c_beta, gamma = g_xcp.c_gamma, B = beta / gamma, BB = g_xcp.c_BB;
  D3 ec = eps_pw(rs, z, true);
  D3 phi = 0.5 * (Dpow23(1.0 + z) + Dpow23(1.0 - z));
  D3 phi3 = phi * phi * phi;
  D3 kf = Dcbrt((3.0 * PI * PI) * rho);
  D3 ks2 = (4.0 / PI) * kf;
  D3 t2 = sig / (4.0 * phi * phi * ks2 * rho * rho);
  D3 A = B / Dexpm1(-ec / (gamma * phi3));
  D3 At2 = A * t2;
  return ec + gamma * phi3 * Dlog1p(B * t2 * (1.0 + BB * At2) / (1.0 + At2 + BB * At2 * At2));
}
// lda_c_vwn_rpa, spin-polarised: the paramagnetic and ferromagnetic RPA fits interpolated with f(zeta) alone (libxc)
D3 eps_vwn_rpa(D3 rs, D3 z) {
  D3 x = Dsqrt(rs);
  D3 eP = vwn_fit(x, 0.0310907, 13.0720, 42.7198, -0.409286);
  D3 eF = vwn_fit(x, 0.01554535, 20.1231, 101.578, -0.743294);
  return eP + (eF - eP) * fzeta(z);
}

// Lee-Yang-Parr correlation energy per VOLUME in the form of Miehlich, Savin, Stoll, Preuss (Chem. Phys. Lett. 157, 200
// (1989), eq 2) as a function of (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb)
D3 lyp_energy(D3 ra, D3 rb, D3 saa, D3 sab, D3 sbb) {
  const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349, CF = 0.3 * pow(3.0 * PI * PI, 2.0 / 3.0);
  D3 n = ra + rb;
  D3 rm13 = 1.0 / Dcbrt(n);
  D3 den = 1.0 + d * rm13;
  D3 delta = c * rm13 + d * rm13 / den;
  D3 n113 = n * n * n * Dpow23(n);  // n^{11/3}
  D3 omega = Dexp(-c * rm13) / (den * n113);
  D3 st = saa + 2.0 * sab + sbb;
  D3 ra83 = ra * ra * Dpow23(ra), rb83 = rb * rb * Dpow23(rb);
  D3 brace = ra * rb * (pow(2.0, 11.0 / 3.0) * CF * (ra83 + rb83) + (47.0 / 18.0 - 7.0 / 18.0 * delta) * st -
                        (2.5 - delta / 18.0) * (saa + sbb) - (delta - 11.0) / 9.0 * (ra * saa + rb * sbb) / n) -
             2.0 / 3.0 * n * n * st + (2.0 / 3.0 * n * n - ra * ra) * sbb + (2.0 / 3.0 * n * n - rb * rb) * saa;
  return -a * 4.0 / den * ra * rb / n - a * b * omega * brace;
}
}  // namespace

// spin-unpolarised LYP from the general form at rho_a = rho_b = rho/2, sigma_aa = sigma_ab = sigma_bb = sigma/4 (the
// kernels use the closed-shell reduction of the formula instead: two different algebraic routes)
void gga_c_lyp(double rho, double sigma, double &exc, double &vrho, double &vsigma) {
  D3 E = lyp_energy(var(0.5 * rho, 0), var(0.5 * rho, 1), var(0.25 * sigma, 2), var(0.25 * sigma, 3), var(0.25 * sigma, 4));
  exc = E.v / rho;
  vrho = 0.5 * (E.d[0] + E.d[1]);
  vsigma = 0.25 * (E.d[2] + E.d[3] + E.d[4]);
}

void xc_polarized(int id, size_t N, const double *rho, const double *sigma, double *exc, double *vrho, double *vsigma,
                  double thr) {
  if (id == 178) {  // hyb_lda_xc_cam_lda0, DFT part: spin-scaled exchange mixture + lda_c_pw_mod
    Vec e(N), v(2 * N);
    xc_polarized(-178, N, rho, sigma, exc, vrho, nullptr, thr);
    xc_polarized(13, N, rho, sigma, e.data(), v.data(), nullptr, thr);
    for (size_t i = 0; i < N; i++) exc[i] += e[i];
    for (size_t i = 0; i < 2 * N; i++) vrho[i] += v[i];
    if (vsigma) std::fill(vsigma, vsigma + 3 * N, 0.0);
    return;
  }
  if (id == 402) {  // hyb_gga_xc_b3lyp, DFT part
    const int ids[4] = {1, 106, 8, 131};
    const double wts[4] = {0.08, 0.72, 0.19, 0.81};
    std::fill(exc, exc + N, 0.0);
    std::fill(vrho, vrho + 2 * N, 0.0);
    std::fill(vsigma, vsigma + 3 * N, 0.0);
    Vec e(N), v(2 * N), vs(3 * N);
    for (int q = 0; q < 4; q++) {
      std::fill(vs.begin(), vs.end(), 0.0);
      xc_polarized(ids[q], N, rho, sigma, e.data(), v.data(), vs.data(), thr);
      for (size_t i = 0; i < N; i++) exc[i] += wts[q] * e[i];
      for (size_t i = 0; i < 2 * N; i++) vrho[i] += wts[q] * v[i];
      if (xc_is_gga(ids[q]))
        for (size_t i = 0; i < 3 * N; i++) vsigma[i] += wts[q] * vs[i];
    }
    return;
  }
  if (id == 406) {  // hyb_gga_xc_pbeh (PBE0), DFT part: 0.75 gga_x_pbe + gga_c_pbe
    Vec e(N), v(2 * N), vs(3 * N);
    xc_polarized(101, N, rho, sigma, exc, vrho, vsigma, thr);
    xc_polarized(130, N, rho, sigma, e.data(), v.data(), vs.data(), thr);
    for (size_t i = 0; i < N; i++) exc[i] = 0.75 * exc[i] + e[i];
    for (size_t i = 0; i < 2 * N; i++) vrho[i] = 0.75 * vrho[i] + v[i];
    for (size_t i = 0; i < 3 * N; i++) vsigma[i] = 0.75 * vsigma[i] + vs[i];
    return;
  }
  for (size_t i = 0; i < N; i++) {
    exc[i] = 0.0;
    vrho[2 * i] = vrho[2 * i + 1] = 0.0;
    if (vsigma) vsigma[3 * i] = vsigma[3 * i + 1] = vsigma[3 * i + 2] = 0.0;
    double ra = rho[2 * i], rb = rho[2 * i + 1];
    if (!(ra + rb >= thr) || ra + rb <= 0.0) continue;
    const bool live_a = ra >= thr, live_b = rb >= thr;  // exchange: a channel below the threshold is left out
    ra = std::max(ra, thr);
    rb = std::max(rb, thr);
    const double rt = ra + rb;
    const bool gga = xc_is_gga(id);
    double saa = 0, sab = 0, sbb = 0;
    if (gga) {
      saa = sigma[3 * i];
      sab = sigma[3 * i + 1];
      sbb = sigma[3 * i + 2];
    }
    switch (id) {
      case 1:
      case 546:
      case 641:
      case -178:
      case 106:
      case 101: {  // spin-scaled exchange
        double ea, va, vsa = 0, eb, vb, vsb = 0;
        if (id == 1) {
          lda_x(2.0 * ra, ea, va);
          lda_x(2.0 * rb, eb, vb);
          ea *= g_xcp.x_alpha;
          va *= g_xcp.x_alpha;
          eb *= g_xcp.x_alpha;
          vb *= g_xcp.x_alpha;
        } else if (id == 546 || id == 641) {
          lda_x_sr(2.0 * ra, 0.3, id == 546 ? 2 : 1, ea, va);
          lda_x_sr(2.0 * rb, 0.3, id == 546 ? 2 : 1, eb, vb);
        } else if (id == -178) {
          cam_lda0_x(2.0 * ra, ea, va);
          cam_lda0_x(2.0 * rb, eb, vb);
        } else if (id == 106) {
          gga_x_b88(2.0 * ra, 4.0 * saa, ea, va, vsa);
          gga_x_b88(2.0 * rb, 4.0 * sbb, eb, vb, vsb);
        } else {
          gga_x_pbe(2.0 * ra, 4.0 * saa, ea, va, vsa);
          gga_x_pbe(2.0 * rb, 4.0 * sbb, eb, vb, vsb);
        }
        if (!live_a) ea = va = vsa = 0.0;
        if (!live_b) eb = vb = vsb = 0.0;
        exc[i] = (ra * ea + rb * eb) / rt;  // (1/2)(2 ra ea + 2 rb eb) per particle of the total density
        vrho[2 * i] = va;
        vrho[2 * i + 1] = vb;
        if (gga) {
          vsigma[3 * i] = 2.0 * vsa;
          vsigma[3 * i + 2] = 2.0 * vsb;
        }
        break;
      }
      case 131: {  // LYP: the three gradient invariants enter separately
        D3 E = lyp_energy(var(ra, 0), var(rb, 1), var(saa, 2), var(sab, 3), var(sbb, 4));
        exc[i] = E.v / rt;
        vrho[2 * i] = E.d[0];
        vrho[2 * i + 1] = E.d[1];
        vsigma[3 * i] = E.d[2];
        vsigma[3 * i + 1] = E.d[3];
        vsigma[3 * i + 2] = E.d[4];
        break;
      }
      case 7:
      case 8:
      case 12:
      case 13:
      case 130: {
        D3 a = var(ra, 0), b = var(rb, 1), st = var(saa + 2.0 * sab + sbb, 2);
        D3 n = a + b;
        D3 rs = Dcbrt((3.0 / (4.0 * PI)) / n);
        D3 z = (a - b) / n;
        D3 e = (id == 7) ? eps_vwn(rs, z) : (id == 8) ? eps_vwn_rpa(rs, z) : (id == 12) ? eps_pw(rs, z, false) : (id == 13) ? eps_pw(rs, z, true) : eps_pbe_c(n, rs, z, st);
        D3 en = n * e;
        exc[i] = e.v;
        vrho[2 * i] = en.d[0];
        vrho[2 * i + 1] = en.d[1];
        if (gga) {
          vsigma[3 * i] = en.d[2];
          vsigma[3 * i + 1] = 2.0 * en.d[2];
          vsigma[3 * i + 2] = en.d[2];
        }
        break;
      }
      default: {
        std::ostringstream oss;
        oss << "Functional " << id << " not found!";
        throw std::runtime_error(oss.str());
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// meta-GGA: TPSS exchange and correlation, spin-unpolarised (libxc mgga_x_tpss = 202, mgga_c_tpss = 231;
// Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401).  Variables of the D3 type here: (rho, sigma, tau).
// ---------------------------------------------------------------------------------------------------------
namespace {
inline D3 Dmax(D3 a, D3 b) { return (a.v >= b.v) ? a : b; }

D3 eps_tpss_x(D3 rho, D3 sig, D3 tau) {
  const double b = 0.40, c = 1.59096, e = 1.537, kappa = 0.804, mu = 0.21951, muge = 10.0 / 81.0;
  D3 exu = (-0.75 * cbrt(3.0 / PI)) * Dcbrt(rho);
  D3 rho83 = rho * rho * Dpow23(rho);
  D3 pp = sig / ((4.0 * pow(3.0 * PI * PI, 2.0 / 3.0)) * rho83);  // p = s^2
  D3 tauw = sig / (8.0 * rho);
  D3 tt = Dmax(tau, tauw);                                         // tau >= tau_W
  D3 z = tauw / tt;
  D3 tunif = (0.3 * pow(3.0 * PI * PI, 2.0 / 3.0)) * rho * Dpow23(rho);
  D3 alpha = (tt - tauw) / tunif;
  D3 qb = (9.0 / 20.0) * (alpha - 1.0) / Dsqrt(1.0 + b * alpha * (alpha - 1.0)) + (2.0 / 3.0) * pp;
  D3 z2 = z * z;
  D3 opz2 = 1.0 + z2;
  D3 num = (muge + c * z2 / (opz2 * opz2)) * pp + (146.0 / 2025.0) * qb * qb -
           (73.0 / 405.0) * qb * Dsqrt(0.5 * (9.0 / 25.0) * z2 + 0.5 * pp * pp) + (muge * muge / kappa) * pp * pp +
           (2.0 * sqrt(e) * muge * 9.0 / 25.0) * z2 + (e * mu) * pp * pp * pp;
  D3 den = 1.0 + sqrt(e) * pp;
  D3 x = num / (den * den);
  D3 F = 1.0 + kappa - kappa * kappa / (kappa + x);
  return exu * F;
}

// PBE correlation of a fully spin-polarised density n (zeta = 1), gradient invariant sig
D3 eps_pbe_c_fullpol(D3 n, D3 sig) {
  const double beta = 0.06672455060314922, gamma = (1.0 - log(2.0)) / (PI * PI), B = beta / gamma;
  D3 rs = Dcbrt((3.0 / (4.0 * PI)) / n);
  D3 ec = pw_G(rs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);  // pw_mod, ferromagnetic set
  const double phi = cbrt(0.5), phi3 = 0.5;                                 // ((1+1)^{2/3} + 0)/2 = 2^{-1/3}
  D3 kf = Dcbrt((3.0 * PI * PI) * n);
  D3 ks2 = (4.0 / PI) * kf;
  D3 t2 = sig / ((4.0 * phi * phi) * ks2 * n * n);
  D3 A = B / Dexpm1(-ec / (gamma * phi3));
  D3 At2 = A * t2;
  return ec + (gamma * phi3) * Dlog1p(B * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2));
}

D3 eps_tpss_c(D3 rho, D3 sig, D3 tau) {
  const double d = 2.8, C0 = 0.53;
  D3 rs = Dcbrt((3.0 / (4.0 * PI)) / rho);
  D3 epbe = eps_pbe_c(rho, rs, C(0.0), sig);
  D3 esig = eps_pbe_c_fullpol(0.5 * rho, 0.25 * sig);  // each spin channel on its own
  D3 etil = Dmax(esig, epbe);
  D3 tauw = sig / (8.0 * rho);
  D3 tt = Dmax(tau, tauw);
  D3 z = tauw / tt;
  D3 z2 = z * z;
  D3 rev = epbe * (1.0 + C0 * z2) - (1.0 + C0) * z2 * etil;
  return rev * (1.0 + d * rev * z2 * z);
}
}  // namespace

namespace {
// TPSS correlation, spin-polarised (PRL 91, 146401 (2003), eqs 11-14; see xc_device.h for the formulas)
D3 eps_tpss_c_pol(D3 ra, D3 rb, D3 saa, D3 sab, D3 sbb, D3 ta, D3 tb) {
  const double d = 2.8;
  D3 n = ra + rb;
  D3 rs = Dcbrt((3.0 / (4.0 * PI)) / n);
  D3 zeta = (ra - rb) / n;
  D3 st = saa + 2.0 * sab + sbb;
  D3 epbe = eps_pbe_c(n, rs, zeta, st);
  D3 eta = Dmax(eps_pbe_c_fullpol(ra, saa), epbe), etb = Dmax(eps_pbe_c_fullpol(rb, sbb), epbe);
  D3 tauw = st / (8.0 * n);
  D3 z = tauw / Dmax(ta + tb, tauw);
  D3 z2 = z * z;
  D3 omz = 1.0 - zeta, opz = 1.0 + zeta;
  D3 gz2 = (omz * omz * saa - 2.0 * omz * opz * sab + opz * opz * sbb) / (n * n);
  D3 kf = Dcbrt((3.0 * PI * PI) * n);
  D3 xi2 = gz2 / (4.0 * kf * kf);
  D3 zz = zeta * zeta;
  D3 C0 = 0.53 + 0.87 * zz + 0.50 * zz * zz + 2.26 * zz * zz * zz;
  D3 den = 1.0 + 0.5 * xi2 * (1.0 / Dpow43(opz) + 1.0 / Dpow43(omz));
  D3 Cz = C0 / (den * den * den * den);
  D3 rev = epbe * (1.0 + Cz * z2) - (1.0 + Cz) * z2 * (ra * eta + rb * etb) / n;
  return rev * (1.0 + d * rev * z2 * z);
}
}  // namespace

/// spin-polarised meta-GGA: rho[2N], sigma[3N], tau[2N] point-major; exc[N], vrho[2N], vsigma[3N], vtau[2N]
void xc_polarized_mgga(int id, size_t N, const double *rho, const double *sigma, const double *tau, double *exc, double *vrho,
                       double *vsigma, double *vtau, double thr) {
  for (size_t i = 0; i < N; i++) {
    exc[i] = 0.0;
    vrho[2 * i] = vrho[2 * i + 1] = 0.0;
    vsigma[3 * i] = vsigma[3 * i + 1] = vsigma[3 * i + 2] = 0.0;
    vtau[2 * i] = vtau[2 * i + 1] = 0.0;
    double ra = rho[2 * i], rb = rho[2 * i + 1];
    if (!(ra + rb >= thr) || ra + rb <= 0.0) continue;
    // Exchange is a sum over the spin channels: a channel whose density is below the threshold contributes nothing
    // (libxc screens it the same way).  Raising it to the threshold instead, as the GGAs above do, is harmless for
    // v_rho and v_sigma (they multiply the channel's own density or gradient) but not for v_tau = d(n e)/d tau, which
    // enters the Fock matrix unweighted and is of order 1/tau_unif(threshold) there.
    const bool live_a = ra >= thr, live_b = rb >= thr;
    ra = std::max(ra, thr);
    rb = std::max(rb, thr);
    const double saa = sigma[3 * i], sab = sigma[3 * i + 1], sbb = sigma[3 * i + 2];
    const double ta = tau[2 * i], tb = tau[2 * i + 1];
    if (id == 202) {  // exchange: spin scaling of the unpolarised functional
      D3 na = C(0.0), nb = C(0.0);
      if (live_a) {
        D3 Ra = var(2.0 * ra, 0), Sa = var(std::max(4.0 * saa, 1e-40), 1), Ta = var(std::max(2.0 * ta, 1e-40), 2);
        na = Ra * eps_tpss_x(Ra, Sa, Ta);
      }
      if (live_b) {
        D3 Rb = var(2.0 * rb, 0), Sb = var(std::max(4.0 * sbb, 1e-40), 1), Tb = var(std::max(2.0 * tb, 1e-40), 2);
        nb = Rb * eps_tpss_x(Rb, Sb, Tb);
      }
      exc[i] = 0.5 * (na.v + nb.v) / (ra + rb);
      vrho[2 * i] = na.d[0];
      vrho[2 * i + 1] = nb.d[0];
      vsigma[3 * i] = 2.0 * na.d[1];
      vsigma[3 * i + 2] = 2.0 * nb.d[1];
      vtau[2 * i] = na.d[2];
      vtau[2 * i + 1] = nb.d[2];
    } else if (id == 231) {
      D3 e = eps_tpss_c_pol(var(ra, 0), var(rb, 1), var(std::max(saa, 1e-40), 2), var(sab, 3), var(std::max(sbb, 1e-40), 4),
                            var(std::max(ta, 1e-40), 5), var(std::max(tb, 1e-40), 6));
      D3 en = (var(ra, 0) + var(rb, 1)) * e;
      exc[i] = e.v;
      vrho[2 * i] = en.d[0];
      vrho[2 * i + 1] = en.d[1];
      vsigma[3 * i] = en.d[2];
      vsigma[3 * i + 1] = en.d[3];
      vsigma[3 * i + 2] = en.d[4];
      vtau[2 * i] = en.d[5];
      vtau[2 * i + 1] = en.d[6];
    } else {
      std::ostringstream oss;
      oss << "Functional " << id << " not found!";
      throw std::runtime_error(oss.str());
    }
  }
}

bool xc_is_mgga(int id) { return id == 202 || id == 231; }

void xc_unpolarized_mgga(int id, size_t N, const double *rho, const double *sigma, const double *tau, double *exc,
                         double *vrho, double *vsigma, double *vtau, double thr) {
  for (size_t i = 0; i < N; i++) {
    exc[i] = vrho[i] = vsigma[i] = vtau[i] = 0.0;
    double r = rho[i];
    if (!(r >= thr) || r <= 0.0) continue;
    if (id == 202 && !(0.5 * r >= thr)) continue;  // exchange channel r/2 below the threshold, see xc_unpolarized
    D3 R = var(r, 0), S = var(std::max(sigma[i], 1e-40), 1), T = var(std::max(tau[i], 1e-40), 2);
    D3 e = (id == 202) ? eps_tpss_x(R, S, T) : eps_tpss_c(R, S, T);
    D3 en = R * e;
    exc[i] = e.v;
    vrho[i] = en.d[0];
    vsigma[i] = en.d[1];
    vtau[i] = en.d[2];
  }
}

static int find_func(const std::string &name) {
  if (isdigit(name[0])) return atoi(name.c_str());
  if (!strcasecmp(name.c_str(), "none")) return 0;
  if (!strcasecmp(name.c_str(), "hyb_x_hf") || !strcasecmp(name.c_str(), "HF")) return -1;
  if (!strcasecmp(name.c_str(), "lda_x")) return 1;
  if (!strcasecmp(name.c_str(), "lda_c_vwn")) return 7;
  if (!strcasecmp(name.c_str(), "lda_c_pw")) return 12;
  if (!strcasecmp(name.c_str(), "lda_c_pw_mod")) return 13;
  if (!strcasecmp(name.c_str(), "lda_x_erf")) return 546;
  if (!strcasecmp(name.c_str(), "lda_x_yukawa")) return 641;
  if (!strcasecmp(name.c_str(), "hyb_lda_xc_cam_lda0")) return 178;
  if (!strcasecmp(name.c_str(), "gga_x_pbe")) return 101;
  if (!strcasecmp(name.c_str(), "gga_c_pbe")) return 130;
  if (!strcasecmp(name.c_str(), "hyb_gga_xc_pbeh")) return 406;
  if (!strcasecmp(name.c_str(), "lda_c_vwn_rpa")) return 8;
  if (!strcasecmp(name.c_str(), "gga_x_b88")) return 106;
  if (!strcasecmp(name.c_str(), "gga_c_lyp")) return 131;
  if (!strcasecmp(name.c_str(), "hyb_gga_xc_b3lyp")) return 402;
  if (!strcasecmp(name.c_str(), "mgga_x_tpss")) return 202;
  if (!strcasecmp(name.c_str(), "mgga_c_tpss")) return 231;
  std::ostringstream oss;
  oss << "\nError: functional " << name << " is not available in this build!\n";
  throw std::runtime_error(oss.str());
}

void set_xc_params(const double *x_pars, int nx, int x_func, const double *c_pars, int nc, int c_func) {
  g_xcp = XCParams();
  if (nx > 0) {
    if (x_func == 1 && nx == 1) g_xcp.x_alpha = x_pars[0];
    else if (x_func == 101 && nx == 2) {
      g_xcp.x_kappa = x_pars[0];
      g_xcp.x_mu = x_pars[1];
    } else
      throw std::runtime_error("oracle: external parameters not supported for this exchange functional\n");
  }
  if (nc > 0) {
    if (c_func == 130 && nc == 3) {
      g_xcp.c_beta = c_pars[0];
      g_xcp.c_gamma = c_pars[1];
      g_xcp.c_BB = c_pars[2];
    } else
      throw std::runtime_error("oracle: external parameters not supported for this correlation functional\n");
  }
}

void parse_xc_func(int &x_func, int &c_func, const std::string &xc) {
  x_func = 0;
  c_func = 0;
  size_t dpos = xc.find('-', 0);
  if (dpos != std::string::npos) {
    x_func = find_func(xc.substr(0, dpos));
    c_func = find_func(xc.substr(dpos + 1));
  } else
    x_func = find_func(xc);
}

}  // namespace oracle
