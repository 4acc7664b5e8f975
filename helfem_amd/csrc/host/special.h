// Angular / special functions that feed the Fock-build tables.
//
// Reference behaviour followed (read-only tree):
//   src/general/gaunt.cpp:35-53      Gaunt coefficient (GSL 3j there; exact quadrature here)
//   src/general/gaunt.cpp:55-69,167-180 "modified" coefficient (cos^2 inserted)
//   src/general/gaunt.cpp:154-217    cos^n / sin^2 couplings
//   src/general/spherical_harmonics.cpp:25-41  Y_lm with Condon-Shortley phase (GSL sphPlm)
//   src/general/angular.cpp:22-45,64-71        (cos theta Chebyshev) x (uniform phi) rule
//   src/general/legendretable.cpp:60-97 + src/legendre/*.f90: P_L^M(xi), Q_L^M(xi), xi>1,
//       Hobson ("type 3") convention, no Condon-Shortley phase: Q_L^M carries (-1)^M.
// GSL and the Fortran library are absent from the target image, so these are own
// implementations; they are pinned by the reference's gaunt_test values, by the Fortran
// library compiled into oracle/_ref, and by mpmath (tests/test_special.py).
#pragma once
#include "linalg.h"
#include <complex>
#include <map>

namespace helfem {

/// Normalised associated Legendre functions Theta_lm(x) = N_lm P_l^m(x) (with CS phase),
/// Y_lm = Theta_lm(cos th) e^{i m phi}; valid for any sign of m, |m|<=l. x = cos(theta).
double theta_lm(int l, int m, double x);
/// d Theta_lm / d theta (reference: basis.cpp:1914-1926, m cot(th) Y_l^m + sqrt((l-m)(l+m+1)) e^{-i phi} Y_l^{m+1})
double dtheta_lm(int l, int m, double x);
std::complex<double> spherical_harmonics(int l, int m, double cth, double phi);

/// G^{M m m'}_{L l l'}: Y_l^m Y_l'^m' = sum_LM G Y_L^M   (gaunt.cpp:35-53)
double gaunt_coefficient(int L, int M, int l, int m, int lp, int mp);

/// Memoising table with the reference's Gaunt class interface
class Gaunt {
  mutable std::map<long long, double> cache;

 public:
  double coeff(int L, int M, int l, int m, int lp, int mp) const;
  double mod_coeff(int lj, int mj, int L, int M, int li, int mi) const;
  double cosine_coupling(int lj, int mj, int li, int mi) const;
  double cosine2_coupling(int lj, int mj, int li, int mi) const;
  double cosine3_coupling(int lj, int mj, int li, int mi) const;
  double cosine4_coupling(int lj, int mj, int li, int mi) const;
  double cosine5_coupling(int lj, int mj, int li, int mi) const;
  double sine2_coupling(int lj, int mj, int li, int mi) const;
  double cosine2_sine2_coupling(int lj, int mj, int li, int mi) const;
};

/// P_L^M(xi) and Q_L^M(xi) for xi>1, L=0..Lmax, M=0..Mmax; out arrays are (Lmax+1) x (Mmax+1)
/// column-major (index M*(Lmax+1)+L), same layout as the Fortran wrapper's calc_Plm_arr.
/// Entries with L<M are zero.  xi==1 gives all zeros (legendretable.cpp:73 skips it).
void legendre_PQ(int Lmax, int Mmax, double xi, double *P, double *Q);
/// Test hook: replace the Legendre evaluation used by compute_tei (signature of legendre_PQ plus lpad).
/// Only the oracle's test API sets it, to measure how the limited accuracy of the reference's Fortran
/// library (compiled into oracle/_ref) propagates into integrals and energies; the product never does.
typedef void (*legendre_provider_t)(int Lmax, int Mmax, int lpad, double xi, double *P, double *Q);
void set_legendre_provider(legendre_provider_t fn);
legendre_provider_t get_legendre_provider();

/// Modified spherical Bessel functions of the range-separated (Yukawa) kernel, exp(-lambda r12)/r12 =
/// 4 pi lambda sum_LM i_L(lambda r<) k_L(lambda r>) Y_LM^* Y_LM.  Conventions of libhelfem/src/utils.cpp:47-70:
/// i_L(x) -> x^L/(2L+1)!! (i_0 = sinh x / x); k_L is GSL's k_l divided by pi/2 (k_0 = exp(-x)/x).  GSL is absent:
/// ascending series / stable upward recurrences here, pinned by mpmath (tests/golden/rs_special.json).
double bessel_il(double x, int L);
double bessel_kl(double x, int L);
/// Phi_n(Xi,xi) of the Legendre expansion erfc(mu r12)/r12 = mu sum_n Phi_n(mu r>, mu r<) P_n(cos gamma)
/// (Angyan, Gerber, Marsman, J. Phys. A 39, 8613 (2006), eqs 21-30; libhelfem/src/erfc_expn.cpp:181-195: short-range
/// power series for xi < 0.4 or (Xi < 0.5 and xi < 2 Xi), closed form otherwise).  Argument order is free.
double erfc_Phi(int n, double Xi, double xi);
/// Test hook.  The reference evaluates the generalised binomial coefficients C(m-k-1, m-1) of eq 29 with a helper
/// (erfc_expn.cpp:46-70) that is wrong for upper arguments <= -2 and lower arguments >= 2 (C(-2,2) = 1 instead of 3),
/// which puts relative errors of up to 1e-5 into the short-range series; mode 1 reproduces that helper so the tests
/// can measure what it does to integrals and energies.  Default 0 = exact binomials.  The product never sets it.
void set_erfc_binomial_mode(int mode);

/// Angular product rule: cos(theta) Chebyshev nodes (ltheta of them) x nphi uniform phi
void angular_chebyshev(int ltheta, int nphi, Vec &cth, Vec &phi, Vec &w);

}  // namespace helfem
