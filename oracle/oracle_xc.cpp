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
void lda_x(double rho, double &exc, double &vrho) {
  double cx = -0.75 * cbrt(3.0 / PI);
  exc = cx * cbrt(rho);
  vrho = 4.0 / 3.0 * exc;
}

// ---- VWN5 paramagnetic correlation ----
void lda_c_vwn(double rho, double &exc, double &vrho) {
  const double A = 0.0310907, b = 3.72744, c = 12.9352, x0 = -0.10498;
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

// ---- PW92 paramagnetic correlation; mod=true uses the higher-precision constants of pw_mod ----
void pw92(double rs, bool mod, double &ec, double &decdrs) {
  const double a = mod ? 0.0310906908696549 : 0.031091;
  const double a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
  double srs = sqrt(rs);
  double q0 = -2.0 * a * (1.0 + a1 * rs);
  double q1 = 2.0 * a * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
  double q1p = a * (b1 / srs + 2.0 * b2 + 3.0 * b3 * srs + 4.0 * b4 * rs);
  double lg = log(1.0 + 1.0 / q1);
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

// ---- PBE exchange ----
void gga_x_pbe(double rho, double sigma, double &exc, double &vrho, double &vsigma) {
  const double kappa = 0.8040;
  const double mu = 0.06672455060314922 * PI * PI / 3.0;
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
  const double beta = 0.06672455060314922;
  const double gamma = (1.0 - log(2.0)) / (PI * PI);
  const double B = beta / gamma;
  double rs = cbrt(3.0 / (4.0 * PI * rho));
  double ec, dec;
  pw92(rs, true, ec, dec);
  double kf = cbrt(3.0 * PI * PI * rho);
  double ks2 = 4.0 * kf / PI;
  double u = sigma / (4.0 * ks2 * rho * rho);  // t^2
  double E = exp(-ec / gamma);
  double A = B / (E - 1.0);
  double N = B * u * (1.0 + A * u);
  double D = 1.0 + A * u + A * A * u * u;
  double arg = 1.0 + N / D;
  double H = gamma * log(arg);
  double dN_du = B * (1.0 + 2.0 * A * u), dD_du = A + 2.0 * A * A * u;
  double dN_dA = B * u * u, dD_dA = u + 2.0 * A * u * u;
  double dH_du = gamma * (dN_du * D - N * dD_du) / (D * D * arg);
  double dH_dA = gamma * (dN_dA * D - N * dD_dA) / (D * D * arg);
  double dA_dec = A * A * E / (B * gamma);
  double rho_dec_drho = -rs / 3.0 * dec;  // rho * d ec / d rho
  exc = ec + H;
  vrho = ec + rho_dec_drho + H + dH_dA * dA_dec * rho_dec_drho - 7.0 / 3.0 * u * dH_du;
  vsigma = rho * dH_du / (4.0 * ks2 * rho * rho);
}
}  // namespace

bool xc_is_gga(int id) { return id == 101 || id == 130; }

void xc_unpolarized(int id, size_t N, const double *rho, const double *sigma, double *exc, double *vrho,
                    double *vsigma, double thr) {
  for (size_t i = 0; i < N; i++) {
    exc[i] = 0.0;
    vrho[i] = 0.0;
    if (vsigma) vsigma[i] = 0.0;
    double r = rho[i];
    if (!(r >= thr) || r <= 0.0) continue;
    double e = 0, v = 0, vs = 0;
    switch (id) {
      case 1: lda_x(r, e, v); break;
      case 7: lda_c_vwn(r, e, v); break;
      case 12: lda_c_pw(r, e, v); break;
      case 101: gga_x_pbe(r, sigma[i], e, v, vs); break;
      case 130: gga_c_pbe(r, sigma[i], e, v, vs); break;
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

static int find_func(const std::string &name) {
  if (isdigit(name[0])) return atoi(name.c_str());
  if (!strcasecmp(name.c_str(), "none")) return 0;
  if (!strcasecmp(name.c_str(), "hyb_x_hf") || !strcasecmp(name.c_str(), "HF")) return -1;
  if (!strcasecmp(name.c_str(), "lda_x")) return 1;
  if (!strcasecmp(name.c_str(), "lda_c_vwn")) return 7;
  if (!strcasecmp(name.c_str(), "lda_c_pw")) return 12;
  if (!strcasecmp(name.c_str(), "gga_x_pbe")) return 101;
  if (!strcasecmp(name.c_str(), "gga_c_pbe")) return 130;
  std::ostringstream oss;
  oss << "\nError: functional " << name << " is not available in this build!\n";
  throw std::runtime_error(oss.str());
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
