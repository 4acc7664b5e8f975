// Prolate-spheroidal two-dimensional basis B_n(mu) Y_l^m(nu,phi) and every table the diatomic
// Fock build consumes.  Host-side (setup) counterpart of the reference class
// helfem::diatomic::basis::TwoDBasis (/root/reference/src/diatomic/basis.h:118-305,
// basis.cpp:307-412 ctor, :482 pure_indices, :561 get_sym_idx, :677-953 one-electron matrices,
// :1166-1302 compute_tei, :1735/:1754 remove/expand_boundaries) and of
// helfem::diatomic::quadrature::twoe_integral (src/diatomic/quadrature.cpp:22-123).
//
// The per-iteration entry points coulomb / exchange / eval_Fxc are NOT here: the product runs
// them on the GPU (helfem_amd/csrc/hip), the CPU restatement used as checker lives in oracle/.
#pragma once
#include "fem.h"
#include "special.h"
#include <utility>

namespace helfem {
namespace diatomic {

typedef std::pair<int, int> lmidx_t;

/// (|m|=0: l=0..lmax[0]), (|m|=1: (l,+1),(l,-1), ...)   reference: basis.cpp:287-302
void lm_to_l_m(const IVec &lmmax, IVec &lval, IVec &mval);

struct TwoDBasis {
  int Z1 = 0, Z2 = 0;
  double Rhalf = 0.0;
  int lpad = 0;
  int primbas = 4, nnodes = 0;
  FEMBasis fem;
  Vec xq, wq;  // radial quadrature rule on [-1,1]
  IVec lval, mval;
  std::vector<lmidx_t> lm_map;  // (L,|M|), sorted
  std::vector<lmidx_t> LM_map;  // (L,M), sorted
  int Lmax = 0, Mmax = 0;
  Gaunt gaunt;

  // primitive integrals (compute_tei)
  std::vector<Mat> disjoint_P0, disjoint_P2, disjoint_Q0, disjoint_Q2;  // [ilm*Nel+iel], Ni x Ni
  std::vector<Mat> prim_tei00, prim_tei02, prim_tei20, prim_tei22;      // [ilm*Nel+iel], Ni^2 x Ni^2
  std::vector<Mat> prim_ktei00, prim_ktei02, prim_ktei20, prim_ktei22;  // exchange-ordered copies
  bool have_tei = false, have_ktei = false;

  TwoDBasis() {}
  /// ctor arguments of the reference (basis.cpp:307); poly = LIP on nnodes Lobatto nodes (primbas 4)
  TwoDBasis(int Z1, int Z2, double Rhalf, int nnodes, int n_quad, const Vec &bval, const IVec &lval,
            const IVec &mval, int lpad);

  size_t Nel() const { return fem.nelem(); }
  size_t Nrad() const { return fem.nbf(); }
  size_t Nang() const { return lval.size(); }
  size_t Ndummy() const { return Nang() * Nrad(); }
  size_t Nbf() const;
  size_t max_Nprim() const { return fem.max_nprim(); }
  int nquad() const { return (int)xq.size(); }

  std::vector<size_t> pure_indices() const;
  std::vector<size_t> m_indices(int m) const;
  std::vector<size_t> m_indices(int m, bool odd) const;
  std::vector<std::vector<size_t> > get_sym_idx(int symm) const;

  size_t lmind(int L, int M) const;  // index of (L,|M|) in lm_map (throws if absent)
  size_t LMind(int L, int M) const;  // index of (L,M) in LM_map

  Mat radial_integral(int m, int n) const;  // \int B_i B_j sinh^m cosh^n
  Mat overlap() const;
  /// radial overlap with ANOTHER basis, \int B_i B'_j sinh(mu) cosh^n(mu) dmu (RadialBasis::overlap(rh, n), basis.cpp:104-200)
  Mat radial_overlap(const TwoDBasis &rh, int n) const;
  /// interbasis overlap <this | rh> in the boundary-cleaned index spaces, Nbf() x rh.Nbf() (basis.cpp:713-750): what
  /// projects the orbitals of a checkpoint made in another basis (--load)
  Mat overlap(const TwoDBasis &rh) const;
  Mat kinetic() const;
  Mat nuclear() const;
  Mat dipole_z() const;
  Mat quadrupole_zz() const;

  Mat remove_boundaries(const Mat &Fnob) const;
  Mat expand_boundaries(const Mat &Ppure) const;

  /// fill the disjoint_* and prim_tei* tables (basis.cpp:1166); exchange also fills prim_ktei*
  void compute_tei(bool exchange);

  /// Per-element tables of compute_tei in a device-friendly form: everything the in-element integrals need that is
  /// cheap on the host (quadrature points, LIP products, Legendre values); the O(Nlm nq p^4) sums are then done on
  /// the GPU (hip/tei_dev.hip).  Arrays over points are contiguous per (L,|M|) channel.
  struct TeiElementTables {
    size_t Ni = 0, Np = 0, nq = 0, Nlm = 0;
    Mat bb0;                 // Np x nq      products B_i B_j at the main points
    Mat bbs;                 // Np x nq^2    products at the sub-interval points (point s = isub*nq + q)
    std::vector<double> wQ;  // [2][Nlm][nq]    w_q sinh cosh^{2k} Q_L^M at the main points, k = 0,1
    std::vector<double> wP;  // [2][Nlm][nq^2]  w_s sinh cosh^{2l} P_L^M at the sub-interval points, l = 0,1
  };
  void tei_element_tables(size_t iel, TeiElementTables &t) const;
  /// the disjoint (cross-element) integrals only (the cheap part of compute_tei)
  void compute_disjoint();
  bool have_disjoint = false;

  /// radial functions / derivatives / weights / mu at the quadrature points of element iel
  Mat get_bf(size_t iel) const { return fem.eval_dnf(xq, 0, iel); }
  Mat get_df(size_t iel) const { return fem.eval_dnf(xq, 1, iel); }
  Vec get_wrad(size_t iel) const;
  Vec get_r(size_t iel) const { return fem.eval_coord(xq, iel); }

  /// 4 pi Rh^5 (-1)^M (L-|M|)!/(L+|M|)!   (basis.cpp:1421)
  double LMfac(int L, int M) const;
};

/// utils::exchange_tei (libhelfem/src/utils.cpp:130-155): ktei(k*Nj+j, l*Ni+i) = tei(j*Ni+i, l*Nk+k)
Mat exchange_tei(const Mat &tei, size_t Ni, size_t Nj, size_t Nk, size_t Nl);

}  // namespace diatomic
}  // namespace helfem
