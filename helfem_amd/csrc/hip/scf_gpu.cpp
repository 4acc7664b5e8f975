// GPU backend of the SCF driver (host/scf.h): every per-iteration operation goes through the
// C ABI entry points of include/helfem_gpu.h, exactly as an Armadillo-based caller would.
#include "common.h"
#include "../host/dftfuncs.h"
#include "../host/scf.h"
#include <cstring>

namespace {
using helfem::Mat;
using helfem::Vec;

void chk(int rc) {
  if (rc) throw std::runtime_error(hfg_last_error());
}

struct GPUBackend : public helfem::scf::Backend {
  hfg_ctx *ctx = nullptr;
  hfg_basis *hb = nullptr;  // owns a copy of the basis for the device tables
  explicit GPUBackend(hfg_ctx *c) : ctx(c) {}
  ~GPUBackend() {
    if (hb) hfg_basis_destroy(hb);
  }
  const char *name() const override { return "gfx950"; }

  static void blocks(const std::vector<std::vector<size_t> > &sym, std::vector<int64_t> &ptr, std::vector<int64_t> &idx) {
    ptr.assign(1, 0);
    idx.clear();
    for (const auto &b : sym) {
      for (size_t v : b) idx.push_back((int64_t)v);
      ptr.push_back((int64_t)idx.size());
    }
  }

  void prepare(const helfem::diatomic::TwoDBasis &basis, bool, int ldft, int mdft) override {
    if (hb) hfg_basis_destroy(hb);
    hb = new hfg_basis();
    hb->b = basis;  // tables already computed by the driver
    chk(hfg_basis_upload(ctx, hb, ldft, mdft));
  }
  void prepare_atomic(const helfem::atomic::TwoDBasis &basis, bool, int ldft, int mdft) override {
    if (hb) hfg_basis_destroy(hb);
    hb = new hfg_basis();
    hb->kind = 1;
    hb->ab = basis;
    chk(hfg_basis_upload(ctx, hb, ldft, mdft));
  }
  Mat coulomb(const Mat &P) override {
    Mat J(P.n_rows, P.n_cols);
    chk(hfg_coulomb(ctx, hb, P.memptr(), J.memptr()));
    return J;
  }
  Mat exchange(const Mat &P) override {
    Mat K(P.n_rows, P.n_cols);
    chk(hfg_exchange(ctx, hb, P.memptr(), K.memptr()));
    return K;
  }
  void eval_Fxc(int x, int c, const Mat &P, Mat &H, double &Exc, double &Nel, double &Ekin, double thr) override {
    H.zeros(P.n_rows, P.n_cols);
    chk(hfg_xc_fock(ctx, hb, x, c, P.memptr(), H.memptr(), &Exc, &Nel, &Ekin, thr));
  }
  void eval_Fxc_pol(int x, int c, const Mat &Pa, const Mat &Pb, Mat &Ha, Mat &Hb, double &Exc, double &Nel, double &Ekin,
                    double thr) override {
    Ha.zeros(Pa.n_rows, Pa.n_cols);
    Hb.zeros(Pa.n_rows, Pa.n_cols);
    chk(hfg_xc_fock_pol(ctx, hb, x, c, Pa.memptr(), Pb.memptr(), Ha.memptr(), Hb.memptr(), &Exc, &Nel, &Ekin, thr));
  }
  void eig_gsym_sub(Vec &E, Mat &C, const Mat &F, const Mat &Sinvh, const std::vector<std::vector<size_t> > &sym) override {
    std::vector<int64_t> ptr, idx;
    blocks(sym, ptr, idx);
    size_t N = F.n_rows;
    E.assign(N, 0.0);
    C.zeros(N, N);
    chk(hfg_eig_gsym_sub(ctx, (int64_t)N, F.memptr(), Sinvh.memptr(), (int)sym.size(), ptr.data(), idx.data(),
                         E.data(), C.memptr()));
  }
  Mat Sinvh(const Mat &S, bool chol, const std::vector<std::vector<size_t> > &sym) override {
    std::vector<int64_t> ptr, idx;
    blocks(sym, ptr, idx);
    Mat X(S.n_rows, S.n_cols);
    chk(hfg_form_sinvh(ctx, (int64_t)S.n_rows, S.memptr(), chol ? 1 : 0, (int)sym.size(), ptr.data(), idx.data(),
                       X.memptr()));
    return X;
  }
  Mat gemm(const Mat &A, bool tA, const Mat &B, bool tB) override {
    size_t m = tA ? A.n_cols : A.n_rows, k = tA ? A.n_rows : A.n_cols, n = tB ? B.n_rows : B.n_cols;
    Mat C(m, n);
    chk(hfg_gemm(ctx, tA, tB, (int64_t)m, (int64_t)n, (int64_t)k, A.memptr(), (int64_t)A.n_rows, B.memptr(),
                 (int64_t)B.n_rows, C.memptr(), (int64_t)m));
    return C;
  }
};
}  // namespace

extern "C" {

/// Restricted closed-shell diatomic SCF on the GPU (driver loop of src/diatomic/main.cpp:780-995).
/// out[0..7] = Etot, Ekin, Epot, Ecoul, Exx, Exc, Enucr, iterations (+0.5 if converged);
/// out[8..11] = seconds of the last iteration's J, K, XC, diagonalisation
int hfg_scf_diatomic(hfg_ctx *ctx, int Z1, int Z2, double Rbond, const int *lmmax, int nlm, int nelem, int nnodes,
                     int nquad, double Rmax, int igrid, double zexp, int lpad, const char *method, int ldft, int mdft,
                     int symmetry, int multiplicity, int maxit, double convthr, int verbose, double *out) {
  try {
    helfem::scf::Options o;
    o.multiplicity = multiplicity;
    o.Z1 = Z1;
    o.Z2 = Z2;
    o.Rbond = Rbond;
    o.lmmax.assign(lmmax, lmmax + nlm);
    o.nelem = nelem;
    o.nnodes = nnodes;
    o.nquad = nquad;
    o.Rmax = Rmax;
    o.igrid = igrid;
    o.zexp = zexp;
    o.lpad = lpad;
    o.method = method;
    helfem::parse_xc_func(o.x_func, o.c_func, o.method);
    o.kfrac = helfem::exact_exchange(o.x_func);
    o.ldft = ldft;
    o.mdft = mdft;
    o.symmetry = symmetry;
    o.maxit = maxit;
    o.convthr = convthr;
    o.verbose = verbose != 0;
    GPUBackend be(ctx);
    helfem::scf::Result r = helfem::scf::run_diatomic(o, be);
    out[0] = r.Etot;
    out[1] = r.Ekin;
    out[2] = r.Epot;
    out[3] = r.Ecoul;
    out[4] = r.Exx;
    out[5] = r.Exc;
    out[6] = r.Enucr;
    out[7] = r.iterations + (r.converged ? 0.5 : 0.0);
    out[8] = r.tJ;
    out[9] = r.tK;
    out[10] = r.tXC;
    out[11] = r.tdiag;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 1;
  }
  return 0;
}

int hfg_scf_atomic(hfg_ctx *ctx, int Z, int Q, int lmax, int mmax, int nelem, int nnodes, int nquad, double Rmax,
                   int igrid, double zexp, const char *method, int ldft, int mdft, int symmetry, int multiplicity, int maxit,
                   double convthr, int verbose, double *out) {
  try {
    helfem::scf::AtomicOptions a;
    a.common.multiplicity = multiplicity;
    a.Z = Z;
    a.Q = Q;
    a.lmax = lmax;
    a.mmax = mmax;
    helfem::scf::Options &o = a.common;
    o.nelem = nelem;
    o.nnodes = nnodes;
    o.nquad = nquad;
    o.Rmax = Rmax;
    o.igrid = igrid;
    o.zexp = zexp;
    o.method = method;
    helfem::parse_xc_func(o.x_func, o.c_func, o.method);
    o.kfrac = helfem::exact_exchange(o.x_func);
    o.ldft = ldft;
    o.mdft = mdft;
    o.symmetry = symmetry;
    o.maxit = maxit;
    o.convthr = convthr;
    o.verbose = verbose != 0;
    GPUBackend be(ctx);
    helfem::scf::Result r = helfem::scf::run_atomic(a, be);
    out[0] = r.Etot;
    out[1] = r.Ekin;
    out[2] = r.Epot;
    out[3] = r.Ecoul;
    out[4] = r.Exx;
    out[5] = r.Exc;
    out[6] = r.Enucr;
    out[7] = r.iterations + (r.converged ? 0.5 : 0.0);
    out[8] = r.tJ;
    out[9] = r.tK;
    out[10] = r.tXC;
    out[11] = r.tdiag;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 1;
  }
  return 0;
}
}
