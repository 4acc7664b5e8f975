// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  The checker's own SCF drivers (oracle_scf.cpp).
#pragma once
#include "oracle.h"
#include <string>
#include <vector>

namespace oracle {

struct ScfIn {
  // diatomic: Z1, Z2, Rbond, lmmax, lpad;  atomic: Z1 = Z, lmax, mmax
  int Z1 = 1, Z2 = 1, Q = 0;
  double Rbond = 1.4;
  helfem::IVec lmmax;
  int lmax = 0, mmax = 0;
  int lpad = 10;
  double Rmax = 40.0;
  int igrid = 4;
  double zexp = 1.0;
  int nelem = 3, nnodes = 15, nquad = 0;
  int maxit = 50;
  double convthr = 1e-7;
  bool diag = true;
  int x_func = -1, c_func = 0;
  double kfrac = 1.0, kshort = 0.0, omega = 0.0;
  int rs_kind = 0;
  int ldft = 0, mdft = 0;
  double dftthr = 1e-12;
  int symmetry = 1;
  int multiplicity = 1;
  int restricted = -1;
  double diiseps = 1e-2, diisthr = 1e-3;
  int diisorder = 5;
  int iguess = 0;
  double gsz_d1 = 0.0, gsz_d2 = 0.0;
  bool maverage = false;
  double dampfock = 1.0, dampthr = 0.1;  // atomic program: 0.7 / 0.1 by default (atomic/main.cpp:111-112)
  // --readocc: rows (nalpha, nbeta, m [, parity | l, m]) of occs.dat; enforced while iteration < readocc (negative: always)
  int readocc = 0;
  std::vector<std::vector<int> > occs;
  bool verbose = false;
};

struct ScfOut {
  double Ekin = 0, Epot = 0, Enucr = 0, Ecoul = 0, Exx = 0, Exc = 0, Etot = 0;
  int iterations = 0;
  bool converged = false;
  int nela = 0, nelb = 0;
  size_t Nbf = 0;
  Vec Ea, Eb;
};

ScfOut scf_diatomic(const ScfIn &in);
ScfOut scf_atomic(const ScfIn &in);

}  // namespace oracle
