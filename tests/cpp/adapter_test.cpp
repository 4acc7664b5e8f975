// Calls the SCF hot path THROUGH include/helfem_gpu_arma.hpp with the reference's signatures (basis.h:205,247,249,
// dftgrid.h:179,181, scf_helpers.h:24,34,36), instantiated for the Armadillo-compatible matrix type of
// helfem_amd/csrc/host/linalg.h (Armadillo itself is not in the build image).  Prints checksums that
// tests/test_gpu_adapter.py compares with the same quantities obtained through the ctypes binding (which the parity tests
// tie to the oracle), and exercises the error translation (std::logic_error / std::runtime_error).
//   adapter_test compile-only | run
#include "../../include/helfem_gpu_arma.hpp"
#include "../../helfem_amd/csrc/host/linalg.h"
#include <cmath>
#include <cstdio>
#include <cstring>

using helfem::Mat;
typedef std::vector<double> Vec;
typedef std::vector<int> IVec;
namespace hg = helfem::gpu;

static double frob(const Mat &M) {
  double s = 0;
  for (double v : M.d) s += v * v;
  return std::sqrt(s);
}
static double tr(const Mat &A, const Mat &B) { return helfem::trace_prod(A, B); }

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "compile-only")) {
    printf("adapter compiled\n");
    return 0;
  }
  try {
    auto ctx = std::make_shared<hg::Context>(0);
    // N2-like sigma + pi basis: 2 elements of 5-node LIPs, the basis of the parity tests
    const double Rhalf = 1.034, Rmax = 40.0;
    const int nelem = 2, nnodes = 5;
    Vec bval(nelem + 1);
    if (hfg_radial_grid(std::acosh(Rmax / Rhalf), nelem, 4, 1.0, bval.data())) throw std::runtime_error(hfg_last_error());
    int lmmax[2] = {3, 2};
    IVec lval(64), mval(64);
    int nang = 64;
    if (hfg_lm_list(lmmax, 2, lval.data(), mval.data(), &nang)) throw std::runtime_error(hfg_last_error());
    lval.resize(nang);
    mval.resize(nang);
    hg::diatomic::TwoDBasis<Mat> basis(ctx, 7, 7, Rhalf, nnodes, 5 * nnodes, bval, lval, mval, 10);
    const size_t N = basis.Nbf();
    printf("Nbf %zu\n", N);

    // misuse before compute_tei: the reference throws std::logic_error("Primitive teis have not been computed!")
    Mat P0(N, N);
    bool logic = false;
    try {
      basis.coulomb(P0);
    } catch (const std::logic_error &e) {
      logic = std::string(e.what()).find("Primitive teis have not been computed") != std::string::npos;
    }
    printf("logic_error_before_compute_tei %d\n", logic ? 1 : 0);

    basis.compute_tei(true);
    Mat S = basis.overlap(), T = basis.kinetic(), V = basis.nuclear();
    auto sym = basis.get_sym_idx(1);
    Mat Sinvh = hg::scf::form_Sinvh(*ctx, S, false, sym);
    Mat H0 = T + V;
    Vec E;
    Mat C;
    hg::scf::eig_gsym_sub(*ctx, E, C, H0, Sinvh, sym);
    printf("E0 %.12f %.12f %.12f\n", E[0], E[1], E[2]);
    Mat Pa = hg::scf::form_density(*ctx, C, 7);
    Mat P = 2.0 * Pa;
    printf("TrPS %.12f\n", tr(P, S));
    Mat J = basis.coulomb(P);
    Mat K = basis.exchange(Pa);
    printf("Ecoul %.12f\n", 0.5 * tr(P, J));
    printf("Exx %.12f\n", tr(Pa, K));
    printf("Jnorm %.12f Knorm %.12f\n", frob(J), frob(K));

    hg::DFTGrid<Mat> grid(&basis, 4 * 3 + 12, 4 * 2 + 5);
    Mat H;
    double Exc, Nel, Ekin;
    Vec nopar;
    grid.eval_Fxc(101, nopar, 130, nopar, P, H, Exc, Nel, Ekin, 1e-12);
    printf("Exc %.12f Nel %.10f Hnorm %.12f\n", Exc, Nel, frob(H));
    // external parameters: the defaults spelled out reproduce the default build
    Vec xp = {0.8040, 0.06672455060314922 * M_PI * M_PI / 3.0}, cp = {0.06672455060314922, (1.0 - std::log(2.0)) / (M_PI * M_PI), 1.0};
    Mat H2;
    double Exc2, Nel2, Ekin2;
    grid.eval_Fxc(101, xp, 130, cp, P, H2, Exc2, Nel2, Ekin2, 1e-12);
    printf("Exc_default_pars_diff %.3e\n", std::fabs(Exc2 - Exc));
    xp[0] = 1.245;  // revPBE's kappa
    grid.eval_Fxc(101, xp, 130, cp, P, H2, Exc2, Nel2, Ekin2, 1e-12);
    printf("Exc_revPBE %.12f\n", Exc2);
    Mat Ha, Hb;
    double Excp, Nelp, Ekinp;
    grid.eval_Fxc(101, nopar, 130, nopar, Pa, Pa, Ha, Hb, Excp, Nelp, Ekinp, true, 1e-12);
    printf("Exc_pol_diff %.3e\n", std::fabs(Excp - Exc));

    // full generalized eigenproblem and the runtime_error translation (unsupported functional id)
    Vec E2;
    Mat C2;
    hg::scf::eig_gsym(*ctx, E2, C2, H0, Sinvh);
    printf("E0_full %.12f\n", E2[0]);
    bool runtime = false;
    try {
      Vec three = {1.0, 2.0, 3.0};
      grid.eval_Fxc(101, three, 130, nopar, P, H, Exc, Nel, Ekin, 1e-12);
    } catch (const std::logic_error &) {
    } catch (const std::runtime_error &) {
      runtime = true;
    }
    printf("runtime_error_on_bad_parameters %d\n", runtime ? 1 : 0);
    // symmetry blocks that do not cover the basis: std::logic_error as in scf_helpers.cpp:178-181
    bool mismatch = false;
    try {
      auto bad = sym;
      bad.pop_back();
      hg::scf::eig_gsym_sub(*ctx, E, C, H0, Sinvh, bad);
    } catch (const std::logic_error &) {
      mismatch = true;
    }
    printf("logic_error_on_symmetry_mismatch %d\n", mismatch ? 1 : 0);
    printf("adapter ok\n");
  } catch (const std::exception &e) {
    printf("FAILED: %s\n", e.what());
    return 1;
  }
  return 0;
}
