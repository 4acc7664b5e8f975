// Spherical-coordinate two-dimensional basis (B_n(r)/r) Y_l^m and the tables of the atomic Fock build.
// Host-side (setup) counterpart of helfem::atomic::basis::TwoDBasis (/root/reference/src/atomic/TwoDBasis.cpp:38
// ctor, :202 get_sym_idx, :304-420 one-electron matrices, :666-739 compute_tei) and of
// helfem::atomic::basis::RadialBasis (libhelfem/src/RadialBasis.cpp:190 radial_integral, :484 twoe_integral,
// :649/:676 get_bf/get_df) with libhelfem/src/quadrature.cpp:22-130 (in-element two-electron integrals).
//
// Range-separated exchange tables: :741-778 compute_yukawa, :780-815 compute_erfc with RadialBasis.cpp:201-209
// bessel_il/kl_integral, :491 yukawa_integral, :502-558 erfc_integral and libhelfem/src/quadrature.cpp:128-222.
//
// Point nucleus at the origin only (finite nuclei, off-centre charges, confinement are outside this round's scope
// and rejected by the driver).
//
// B(r)/r near the origin: the reference switches to a Taylor series of order nprim-1 below a numerically
// chosen cutoff (RadialBasis.cpp:59-133, 575-631).  For LIPs whose first function is dropped, B_i(r)/r is a
// polynomial and that series is exact; here it is evaluated directly as the Lagrange product with the
// (x - x_0) factor removed, which is the same polynomial without the cancellation the series avoids.
#pragma once
#include "diatomic_basis.h"
#include "fem.h"
#include "special.h"
#include "model_potential.h"

namespace helfem {
namespace atomic {

/// atomic::basis::angular_basis (src/atomic/basis.cpp:174): |m|=0..mmax, l=|m|..lmax, (l,+|m|),(l,-|m|)
void angular_basis(int lmax, int mmax, IVec &lval, IVec &mval);

/// In-element two-electron integral table of one radial element, int int B_i(r1)B_j(r1) r_<^L/r_>^{L+1} B_k(r2)B_l(r2),
/// (ij) x (kl), without the 4 pi/(2L+1) factor (quadrature::twoe_integral, libhelfem/src/quadrature.cpp:22-130)
Mat twoe_integral(double rmin, double rmax, const Vec &xq, const Vec &wq, const LIPBasis &poly, int L);
/// Same with the Yukawa kernel i_L(lambda r<) k_L(lambda r>), without the 4 pi lambda factor
/// (quadrature::yukawa_integral, libhelfem/src/quadrature.cpp:128-169)
Mat yukawa_integral(double rmin, double rmax, const Vec &xq, const Vec &wq, const LIPBasis &poly, int L, double lambda);

struct TwoDBasis {
  int Z = 0;
  int nnodes = 0;
  FEMBasis fem;
  Vec xq, wq;
  IVec lval, mval;
  Gaunt gaunt;

  std::vector<Mat> disjoint_L, disjoint_m1L;  // [L*Nel+iel]
  std::vector<Mat> prim_tei, prim_ktei;      // [L*Nel+iel]
  bool have_tei = false, have_ktei = false;

  // range-separated exchange (TwoDBasis.cpp:741-815).  rs_kind 1: Yukawa, exp(-lambda r12)/r12 -- disjoint_iL/kL and
  // the in-element tables rs_tei[L*Nel+iel]; rs_kind 2: erfc(mu r12)/r12 -- no factorisation, one table per element
  // pair rs_tei[(L*Nel+iel)*Nel+kel] ((ij) x (kl), i,j in iel, k,l in kel).  rs_ktei: the exchange-ordered copies
  // (utils::exchange_tei) the reference stores; kept for the oracle only.
  int rs_kind = 0;
  double rs_lambda = 0.0;
  std::vector<Mat> disjoint_iL, disjoint_kL;  // [L*Nel+iel]
  std::vector<Mat> rs_tei, rs_ktei;

  TwoDBasis() {}
  TwoDBasis(int Z, int nnodes, int n_quad, const Vec &bval, const IVec &lval, const IVec &mval);

  size_t Nel() const { return fem.nelem(); }
  size_t Nrad() const { return fem.nbf(); }
  size_t Nang() const { return lval.size(); }
  size_t Nbf() const { return Nang() * Nrad(); }
  size_t Ndummy() const { return Nbf(); }
  size_t max_Nprim() const { return fem.max_nprim(); }
  int nquad() const { return (int)xq.size(); }
  int N_L() const;  // 2*max(l)+1
  int Mmax() const;  // max(m)-min(m)

  std::vector<size_t> m_indices(int m) const;
  std::vector<size_t> lm_indices(int l, int m) const;
  std::vector<std::vector<size_t> > get_sym_idx(int symm) const;

  /// B_n(r)/r and d/dr (B_n(r)/r) at the quadrature points of element iel (nq x Nprim(iel))
  Mat get_bf(size_t iel) const;
  Mat get_df(size_t iel) const;
  Vec get_wrad(size_t iel) const;
  Vec get_r(size_t iel) const { return fem.eval_coord(xq, iel); }
  /// per-element operands of the in-element two-electron integrals for the GPU (hip/tei_dev.hip), in the layout of
  /// diatomic::TwoDBasis::TeiElementTables with ONE operand type and one "channel" per L:
  /// wP[L][s] = w_s r_s^L at the sub-interval points, wQ[L][q] = w_q r_q^{-L-1} at the main points -- the prefix form of
  /// quadrature::twoe_inner_integral (libhelfem/src/quadrature.cpp:22-75), which carries the inner integral from point
  /// to point with the ratio (r_{q-1}/r_q)^{L+1} instead; all terms are positive, the two forms agree to rounding
  void tei_element_tables(size_t iel, diatomic::TwoDBasis::TeiElementTables &t) const;
  /// interbasis overlap <this | rh>, Nbf() x rh.Nbf() (atomic/TwoDBasis.cpp:330-344 with RadialBasis::overlap =
  /// radial_integral(rh, 0), RadialBasis.cpp:211-300: one quadrature rule per intersection of elements): what projects the
  /// orbitals of a checkpoint made in another basis (--load)
  Mat overlap(const TwoDBasis &rh) const;
  /// the disjoint (cross-element) integrals only (the cheap part of compute_tei)
  void compute_disjoint();
  bool have_disjoint = false;
  /// \int (B_i/r)(B_j/r) r^{Rexp+2} dr over element iel   (RadialBasis::radial_integral)
  Mat radial_integral(int Rexp, size_t iel) const;

  Mat overlap() const;
  Mat kinetic() const;
  Mat nuclear() const;
  /// TwoDBasis::model_potential (src/atomic/TwoDBasis.cpp:458): int B_i B_j V(r) dr on every shell's diagonal block
  Mat model_potential(const ModelPotential &pot) const;
  void compute_tei(bool exchange);
  /// int B_i B_j f(r) dr over element iel   (RadialBasis::bessel_il_integral / bessel_kl_integral)
  Mat bessel_il_integral(int L, double lambda, size_t iel) const;
  Mat bessel_kl_integral(int L, double lambda, size_t iel) const;
  /// int int B_i(r)B_j(r) Phi_L(mu r, mu r') B_k(r')B_l(r') dr dr', i,j in iel, k,l in kel   (RadialBasis::erfc_integral)
  Mat erfc_integral(int L, double mu, size_t iel, size_t kel) const;
  void compute_yukawa(double lambda);
  void compute_erfc(double mu);
};

}  // namespace atomic
}  // namespace helfem
