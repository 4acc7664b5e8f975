// SCF driver loops (diatomic and atomic, restricted / unrestricted / restricted open shell) written against an abstract
// backend whose compute methods are the GPU entry points (hip/scf_gpu.cpp, hfg_scf_*): the host-pointer variant of the
// product's SCF, kept as the checker of the device-resident loop (hip/scf_device.hip).  The CPU oracle has its OWN
// driver (oracle/oracle_scf.cpp) and does not compile this file.  Mirrors the control flow, energy expression and printed lines
// of the reference driver (/root/reference/src/diatomic/main.cpp:402-1009):
//   S,T,Vnuc -> Sinvh -> guess (core Hamiltonian, --iguess 0) -> compute_tei ->
//   loop { P = C_occ C_occ^T; J; K; XC; F; E; DIIS; eig_gsym_sub } -> energy table.
// External fields, finite nuclei and the SAP guess are outside the hot-path scope and are
// rejected loudly; checkpoints (host/checkpoint.cpp) and the GSZ / Thomas-Fermi guesses are supported.
#pragma once
#include "atomic_basis.h"
#include "diatomic_basis.h"
#include <map>
#include <memory>
#include <string>

namespace helfem {
namespace scf {

struct Backend {
  virtual ~Backend() {}
  virtual const char *name() const = 0;
  /// upload / prepare tables after compute_tei
  virtual void prepare(const diatomic::TwoDBasis &basis, bool exchange, int ldft, int mdft) = 0;
  /// same for the atomic program's basis (src/atomic/main.cpp)
  virtual void prepare_atomic(const atomic::TwoDBasis &basis, bool exchange, int ldft, int mdft) = 0;
  virtual Mat coulomb(const Mat &P) = 0;
  virtual Mat exchange(const Mat &P) = 0;
  /// atomic::basis::TwoDBasis::rs_exchange (src/atomic/TwoDBasis.cpp:1142), tables of compute_yukawa / compute_erfc
  virtual Mat rs_exchange(const Mat &P) = 0;
  /// initial-guess model potential: TwoDGrid::model_potential(p1, p2) (src/diatomic/twodquadrature.cpp:351) on the
  /// quadrature grid given to prepare(), or atomic TwoDBasis::model_potential(p1) (src/atomic/TwoDBasis.cpp:458)
  virtual Mat model_potential(const ModelPotential &p1, const ModelPotential &p2) = 0;
  virtual void eval_Fxc(int x_func, int c_func, const Mat &P, Mat &H, double &Exc, double &Nel, double &Ekin,
                        double thr) = 0;
  /// unrestricted: both spin matrices
  virtual void eval_Fxc_pol(int x_func, int c_func, const Mat &Pa, const Mat &Pb, Mat &Ha, Mat &Hb, double &Exc,
                            double &Nel, double &Ekin, double thr) = 0;
  virtual void eig_gsym_sub(Vec &E, Mat &C, const Mat &F, const Mat &Sinvh,
                            const std::vector<std::vector<size_t> > &sym) = 0;
  virtual Mat Sinvh(const Mat &S, bool chol, const std::vector<std::vector<size_t> > &sym) = 0;
  /// C = op(A) op(B)
  virtual Mat gemm(const Mat &A, bool tA, const Mat &B, bool tB) = 0;
  /// arma::eig_sym of a dense symmetric matrix, eigenvalues ascending (natural orbitals of ROHF_update)
  virtual void eig_sym(Vec &E, Mat &C, const Mat &A) = 0;
};

struct Options {
  int Z1 = 1, Z2 = 1;
  double Rbond = 1.4;
  IVec lmmax;  // per |m|
  int lpad = 10;
  double Rmax = 40.0;
  int igrid = 4;
  double zexp = 1.0;
  int nelem = 3, nnodes = 15, nquad = 0;
  int maxit = 50;
  double convthr = 1e-7;
  bool diag = true;
  std::string method = "HF";
  int x_func = -1, c_func = 0;  // filled by the caller from method
  double kfrac = 1.0;
  // range-separated hybrids (atomic/main.cpp:358-372): K = kfrac K[1/r12] + kshort K[screened], omega != 0 switches the
  // second term on; rs_kind 1 Yukawa, 2 erfc.  Filled by the caller from the functional (range_separation()).
  double kshort = 0.0, omega = 0.0;
  int rs_kind = 0;
  int ldft = 0, mdft = 0;
  double dftthr = 1e-12;
  int symmetry = 1;
  int Q = 0;              // --Q: charge state (number of electrons = Z1 + Z2 - Q, atomic: Z - Q)
  int nela = 0, nelb = 0; // --nela / --nelb: explicit occupations; both zero: from Q and M (scf::parse_nela_nelb)
  int multiplicity = 1;   // --M: spin multiplicity 2S+1; nela - nelb = M - 1
  int restricted = -1;    // --restricted: -1 auto (restricted iff M == 1), 0 unrestricted, 1 restricted; with M > 1 that
                          // is the constrained-UHF form of ROHF (scf::ROHF_update, scf_helpers.cpp:470)
  // ADIIS / CDIIS mixing of the Fock extrapolation (diatomic/main.cpp:119-121, diis.cpp:214-290)
  double diiseps = 1e-2;  // --diiseps: DIIS error below which CDIIS starts to be mixed in
  double diisthr = 1e-3;  // --diisthr: DIIS error below which the extrapolation is pure CDIIS
  int diisorder = 5;
  // atomic program: the occupied-virtual blocks of the extrapolated Fock matrix in the basis of the current orbitals are
  // scaled by dampfock while the DIIS error is at least dampthr (atomic/main.cpp:917-936); 1.0 = none (the diatomic program)
  double dampfock = 1.0, dampthr = 0.1;
  int iguess = 0;  // --iguess: 0 core Hamiltonian, 1 GSZ (needs gsz_d), 3 Thomas-Fermi; 2 (SAP) is not available
  double gsz_d1 = 0.0, gsz_d2 = 0.0;  // screening lengths of the GSZ guess for the two centres (atomic: gsz_d1)
  // --readocc (diatomic/main.cpp:215-222, 338-382; atomic/main.cpp:209-221, 317-345): rows of occs.dat = occupied alpha
  // orbitals, occupied beta orbitals, then m (diatomic, atomic --symmetry 1), m and parity +-1 (homonuclear diatomic with
  // --symmetry 2) or l and m (atomic --symmetry 2); enforced after the guess and after the eigensolves of the iterations
  // i < readocc (negative: always)
  int readocc = 0;
  std::vector<std::vector<int> > occs;
  // --load (main.cpp:552-648, the "project lowest orbitals" branch): orbitals, overlap matrix and basis of a previous run.
  // Same basis (the stored S equals this run's): the projection is the identity.  Another diatomic basis (guess_basis,
  // read from the checkpoint): C = S^-1 S12 C_old with the interbasis overlap S12 (basis.cpp:713-750; atomic/TwoDBasis.cpp:330-344 for the atomic program), S^-1 = Sinvh Sinvh^T.
  // The occupied orbitals are re-orthonormalised by Gram-Schmidt as in main.cpp:630-640.
  bool have_guess = false;
  Mat guessS, guessCa, guessCb;
  Vec guessEa, guessEb;
  std::shared_ptr<diatomic::TwoDBasis> guess_basis;
  std::shared_ptr<atomic::TwoDBasis> guess_basis_atomic;
  bool keep_matrices = false;  // fill Result::mats with what the reference's drivers write to their checkpoint
  bool verbose = true;
};

struct Result {
  double Ekin = 0, Epot = 0, Enucr = 0, Ecoul = 0, Exx = 0, Exc = 0, Etot = 0;
  int iterations = 0;
  bool converged = false;
  double tJ = 0, tK = 0, tXC = 0, tdiag = 0;  // seconds of the last iteration
  Vec E;       // orbital energies (alpha)
  Mat C, P, F;  // final orbitals (alpha), total density, Fock (alpha)
  Vec Eb;      // unrestricted: beta orbital energies / orbitals / Fock
  Mat Cb, Fb;
  int nela = 0, nelb = 0;
  size_t Nbf = 0;
  // keep_matrices: S, T, Vnuc, H0, Sinvh, P, Pa, Pb, J, Ka, Kb, XCa, XCb, Fa, Fb, Ca, Cb of the last iteration
  // (diatomic/main.cpp:406-537, 790-963)
  std::map<std::string, Mat> mats;
};

/// options of the atomic program on top of the common ones (Z1/Z2/Rbond/lmmax/lpad of `common` are unused)
struct AtomicOptions {
  Options common;
  int Z = 2, Q = 0;
  int lmax = 0, mmax = 0;
  bool maverage = false;  // --maverage: average the Fock matrices over m for every l (scf::fock_symmetry_average)
};

/// scf::parse_nela_nelb (src/general/scf_helpers.cpp:558-603): occupations from the charge state and the multiplicity, or
/// charge and multiplicity from explicit occupations; Ztot = total nuclear charge.  Throws std::runtime_error like the
/// reference.
void parse_nela_nelb(int &nela, int &nelb, int &Q, int &M, int Ztot);

/// forced occupations in the form scf::enforce_occupations takes them (scf_helpers.cpp:31): per row of occs.dat the number
/// of occupied alpha / beta orbitals and the basis-function indices of the symmetry
struct OccupationPlan {
  int until = 0;  // enforce while iteration < until (INT_MAX: always; 0: never)
  std::vector<int> na, nb;
  std::vector<std::vector<size_t> > sym;
  bool active(int iteration) const { return iteration < until; }
};
/// the drivers' parsing and checks of occs.dat, with their error texts; nela / nelb: the wanted spin state
OccupationPlan occupation_plan(const Options &opt, const diatomic::TwoDBasis &basis, int nela, int nelb);
OccupationPlan occupation_plan(const Options &opt, const atomic::TwoDBasis &basis, int nela, int nelb);
/// the orbital order scf::enforce_occupations produces (scf_helpers.cpp:52-128): per symmetry the first nocc orbitals
/// with weight in that symmetry (w[isym][orbital] = diag(Csub^T S_sub Csub), counted when > 10 DBL_EPSILON) are occupied;
/// occupied orbitals first, each group by ascending energy.  Throws like the reference on duplicates.
std::vector<size_t> occupation_order(const Vec &E, const std::vector<std::vector<double> > &w, const std::vector<int> &nocc);
/// C, E reordered accordingly (host matrices; the device loop gathers on the device)
void enforce_occupations(Mat &C, Vec &E, const Mat &S, const std::vector<int> &nocc, const std::vector<std::vector<size_t> > &sym);
/// --load: checks that the stored overlap is this basis' and returns the stored orbitals with the first nela / nelb columns
/// S-orthonormalised (main.cpp:618-646); throws std::logic_error for a checkpoint of another basis
/// S12: interbasis overlap (this basis x checkpoint basis) or an empty matrix when none can be formed; Sinvh: this run's
void guess_from_checkpoint(const Options &opt, const Mat &S, const Mat &Sinvh, const Mat &S12, size_t nela, size_t nelb, Mat &Ca, Mat &Cb,
                           Vec &Ea, Vec &Eb);

std::vector<std::vector<std::vector<size_t> > > atomic_average_groups(const atomic::TwoDBasis &basis);
Result run_diatomic(const Options &opt, Backend &be);
/// src/atomic/main.cpp:100-1010, restricted closed shell, point nucleus, core guess
Result run_atomic(const AtomicOptions &opt, Backend &be);

}  // namespace scf
}  // namespace helfem
