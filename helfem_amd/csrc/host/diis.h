// Weights of the Fock-matrix extrapolation: the reference's combined ADIIS + CDIIS scheme
// (/root/reference/src/general/diis.cpp: DIIS::get_w :214-290, get_w_diis_wrk :297-372, solve_F :392-412,
// get_w_adiis :492-600 with get_E_adiis / get_dEdx_adiis :602-640, the L-BFGS helper src/general/lbfgs.cpp, COOLTHR :26).
//
// Only the small problem lives here.  The matrices of the history (Fock, density, error) stay where the driver keeps
// them -- host memory in the host loop, HBM in the device-resident loop -- and the driver hands over inner products:
//   B(i,j) = err_i . err_j                                   (CDIIS)
//   T(i,j) = Tr(Pa_i Fa_j) + Tr(Pb_i Fb_j)                   (ADIIS: PiF(i) = T(i,n) - T(n,n),
//                                                              PiFj(i,j) = T(i,j) - T(i,n) - T(n,j) + T(n,n), diis.cpp:170-187)
// Entries are numbered oldest first; n is the newest.
#pragma once
#include <cstddef>
#include <vector>

namespace helfem {

class DiisMixer {
 public:
  DiisMixer(bool usediis, double diiseps, double diisthr, bool useadiis, bool verbose, size_t imax);

  size_t size() const { return E_.size(); }
  size_t capacity() const { return imax_; }
  /// true when the next push needs the oldest entry to go first (diis.cpp:104-107); the driver then drops its oldest
  /// matrices and calls pop_oldest()
  bool full() const { return E_.size() == imax_; }
  void pop_oldest();
  /// new (newest) entry with energy E and its maximum absolute error; its inner products follow through set_B / set_T
  void push(double E, double maxerr);
  void set_B(size_t i, size_t j, double v) { B_[i * imax_ + j] = B_[j * imax_ + i] = v; }  // symmetric
  void set_T(size_t i, size_t j, double v) { T_[i * imax_ + j] = v; }                       // not symmetric

  /// uDIIS::solve_F: the weights (oldest first) of the extrapolated Fock matrix.  While the weight of the newest entry is
  /// below sqrt(DBL_EPSILON) the OLDEST entry is dropped and the weights recomputed; `dropped` tells the driver how many of
  /// its oldest matrices to drop as well.
  std::vector<double> solve(size_t &dropped);

  // the pieces, public for the tests
  std::vector<double> weights_cdiis() const;
  std::vector<double> weights_adiis() const;
  double adiis_energy(const std::vector<double> &x) const;

 private:
  std::vector<double> get_w();
  void adiis_terms(std::vector<double> &PiF, std::vector<double> &PiFj) const;
  bool usediis_, useadiis_, verbose_;
  double diiseps_, diisthr_;
  size_t imax_;
  int cooloff_ = 0;
  std::vector<double> E_, err_;
  std::vector<double> B_, T_;  // imax x imax, row-major, entries [0, size) in use
};

}  // namespace helfem
