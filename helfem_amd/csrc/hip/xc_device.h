// Spin-unpolarised exchange-correlation functionals evaluated on the GPU grid (reference: the
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
__host__ __device__ inline Dual dexp(Dual a) {
  double e = exp(a.v);
  return mk(e, a.dr * e, a.ds * e);
}
__host__ __device__ inline Dual datan(Dual a) {
  double f = 1.0 / (1.0 + a.v * a.v);
  return mk(atan(a.v), a.dr * f, a.ds * f);
}

#define HFG_PI 3.14159265358979323846

// energy per particle of each functional as a Dual in (rho, sigma)
__host__ __device__ inline Dual eps_lda_x(Dual rho) { return (-0.75 * cbrt(3.0 / HFG_PI)) * dcbrt(rho); }

__host__ __device__ inline Dual eps_lda_c_vwn(Dual rho) {
  const double A = 0.0310907, b = 3.72744, c = 12.9352, x0 = -0.10498;
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

__host__ __device__ inline Dual eps_pw92(Dual rs, bool mod) {
  const double a = mod ? 0.0310906908696549 : 0.031091;
  const double a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
  Dual srs = dsqrt(rs);
  Dual den = (2.0 * a) * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
  return (-2.0 * a) * (1.0 + a1 * rs) * dlog(1.0 + 1.0 / den);
}

__host__ __device__ inline Dual eps_lda_c_pw(Dual rho) {
  return eps_pw92(dcbrt(3.0 / (4.0 * HFG_PI) / rho), false);
}

__host__ __device__ inline Dual eps_gga_x_pbe(Dual rho, Dual sigma) {
  const double kappa = 0.8040;
  const double mu = 0.06672455060314922 * HFG_PI * HFG_PI / 3.0;
  Dual exu = eps_lda_x(rho);
  Dual kf = dcbrt((3.0 * HFG_PI * HFG_PI) * rho);
  Dual s2 = sigma / (4.0 * kf * kf * rho * rho);
  Dual Fx = 1.0 + kappa - kappa / (1.0 + (mu / kappa) * s2);
  return exu * Fx;
}

__host__ __device__ inline Dual eps_gga_c_pbe(Dual rho, Dual sigma) {
  const double beta = 0.06672455060314922;
  const double gamma = (1.0 - 0.6931471805599453) / (HFG_PI * HFG_PI);
  const double B = beta / gamma;
  Dual rs = dcbrt(3.0 / (4.0 * HFG_PI) / rho);
  Dual ec = eps_pw92(rs, true);
  Dual kf = dcbrt((3.0 * HFG_PI * HFG_PI) * rho);
  Dual ks2 = (4.0 / HFG_PI) * kf;
  Dual t2 = sigma / (4.0 * ks2 * rho * rho);
  Dual Aa = B / (dexp(-ec / gamma) - 1.0);
  Dual At2 = Aa * t2;
  Dual H = gamma * dlog(1.0 + B * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2));
  return ec + H;
}

__host__ __device__ inline bool is_gga(int id) { return id == 101 || id == 130; }
__host__ __device__ inline bool is_supported(int id) {
  return id == 1 || id == 7 || id == 12 || id == 101 || id == 130;
}

/// adds functional id's exc (per particle), vrho, vsigma at one point; rho >= threshold assumed
__host__ __device__ inline void eval_add(int id, double rho, double sigma, double &exc, double &vrho, double &vsigma) {
  Dual r = mk(rho, 1.0, 0.0), s = mk(sigma, 0.0, 1.0);
  Dual e;
  switch (id) {
    case 1: e = eps_lda_x(r); break;
    case 7: e = eps_lda_c_vwn(r); break;
    case 12: e = eps_lda_c_pw(r); break;
    case 101: e = eps_gga_x_pbe(r, s); break;
    case 130: e = eps_gga_c_pbe(r, s); break;
    default: return;
  }
  Dual en = r * e;  // energy per volume
  exc += e.v;
  vrho += en.dr;
  vsigma += en.ds;
}

}  // namespace xc
}  // namespace hfg
