// GPU backend of the SCF driver (host/scf.h): every per-iteration operation goes through the
// C ABI entry points of include/helfem_gpu.h, exactly as an Armadillo-based caller would.
#include "common.h"
#include "../host/dftfuncs.h"
#include "../host/checkpoint.h"
#include "../host/scf.h"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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
  Mat model_potential(const helfem::ModelPotential &p1, const helfem::ModelPotential &p2) override {
    Mat H(hb->Nbf(), hb->Nbf());
    hfg_model_pot a{p1.kind, p1.Z, p1.d, p1.H}, b{p2.kind, p2.Z, p2.d, p2.H};
    chk(hfg_model_potential(ctx, hb, &a, &b, H.memptr()));
    return H;
  }
  Mat rs_exchange(const Mat &P) override {
    Mat K(P.n_rows, P.n_cols);
    chk(hfg_rs_exchange(ctx, hb, P.memptr(), K.memptr()));
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
  void eig_sym(Vec &E, Mat &C, const Mat &A) override {
    size_t n = A.n_rows;
    E.assign(n, 0.0);
    C.zeros(n, n);
    chk(hfg_eig_sym(ctx, (int64_t)n, A.memptr(), E.data(), C.memptr()));
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

namespace hfg {
void set_xc_params(hfg_ctx *ctx, int x_func, const double *x_pars, int nx, int c_func, const double *c_pars, int nc);  // fock.hip
helfem::scf::Result scf_device_loop(hfg_ctx *ctx, hfg_basis *hb, const helfem::scf::Options &opt, int nel, double Enucr,
                                    int symm, const std::vector<std::vector<size_t> > &dsym, int ldft, int mdft,
                                    const std::vector<std::vector<std::vector<size_t> > > &avg_idx =
                                        std::vector<std::vector<std::vector<size_t> > >());
}

namespace {
thread_local int g_readocc = 0;  // --readocc of the following hfg_scf_diatomic / hfg_scf_atomic calls (hfg_scf_set_occupations)
thread_local std::vector<std::vector<int> > g_occs;
thread_local int g_iguess = 0;  // --iguess of the following hfg_scf_* calls of this thread (hfg_scf_set_iguess)

bool host_driver() {
  const char *e = getenv("HELFEM_SCF");
  return e && std::string(e) == "host";
}

// set-up of src/diatomic/main.cpp:245-430 (basis, quadrature defaults, symmetry), then the device-resident loop
helfem::scf::Result run_diatomic_device(hfg_ctx *ctx, const helfem::scf::Options &opt) {
  int nel = opt.Z1 + opt.Z2 - opt.Q;
  int Nquad = opt.nquad;
  if (Nquad == 0) Nquad = 5 * opt.nnodes;
  else if (Nquad < 2 * opt.nnodes) throw std::logic_error("Insufficient radial quadrature.\n");
  helfem::IVec lval, mval;
  helfem::diatomic::lm_to_l_m(opt.lmmax, lval, mval);
  const double Rhalf = 0.5 * opt.Rbond;
  helfem::Vec bval = helfem::get_grid(helfem::arcosh(opt.Rmax / Rhalf), opt.nelem, opt.igrid, opt.zexp);
  hfg_basis *hb = new hfg_basis();
  helfem::scf::Result r;
  try {
    hb->kind = 0;
    hb->b = helfem::diatomic::TwoDBasis(opt.Z1, opt.Z2, Rhalf, opt.nnodes, Nquad, bval, lval, mval, opt.lpad);
    if (opt.verbose)
      printf("Basis set consists of %i angular shells composed of %i radial functions, totaling %i basis functions\n",
             (int)hb->b.Nang(), (int)hb->b.Nrad(), (int)hb->b.Nbf());
    const bool dft = (opt.x_func > 0 || opt.c_func > 0);
    int ldft = opt.ldft, mdft = opt.mdft;
    if (dft || opt.iguess != 0) {  // the model-potential guess is evaluated on the same product grid
      int lmaxmax = 0;
      for (int l : opt.lmmax) lmaxmax = std::max(lmaxmax, l);
      if (ldft == 0) ldft = 4 * lmaxmax + 12;
      if (ldft < 2 * lmaxmax + 2) throw std::logic_error("Increase ldft to guarantee accuracy of quadrature!\n");
      if (mdft == 0) mdft = 4 * (int)opt.lmmax.size() + 5;
      if (mdft < 2 * (int)opt.lmmax.size()) throw std::logic_error("Increase mdft to guarantee accuracy of quadrature!\n");
    } else
      ldft = mdft = 0;
    int symm = opt.symmetry;
    if (symm == 2 && opt.Z1 != opt.Z2) symm = 1;
    r = hfg::scf_device_loop(ctx, hb, opt, nel, opt.Z1 * opt.Z2 / opt.Rbond, symm, hb->b.get_sym_idx(symm), ldft, mdft);
  } catch (...) {
    hfg_basis_destroy(hb);
    throw;
  }
  hfg_basis_destroy(hb);
  return r;
}

helfem::scf::Result run_atomic_device(hfg_ctx *ctx, const helfem::scf::AtomicOptions &a) {
  const helfem::scf::Options &opt = a.common;
  int nel = a.Z - a.Q;
  if (nel <= 0) throw std::logic_error("No electrons.\n");
  int Nquad = opt.nquad;
  if (Nquad == 0) Nquad = 5 * opt.nnodes;
  else if (Nquad < 2 * opt.nnodes) throw std::logic_error("Insufficient radial quadrature.\n");
  helfem::IVec lval, mval;
  helfem::atomic::angular_basis(a.lmax, a.mmax, lval, mval);
  helfem::Vec bval = helfem::get_grid(opt.Rmax, opt.nelem, opt.igrid, opt.zexp);
  hfg_basis *hb = new hfg_basis();
  helfem::scf::Result r;
  try {
    hb->kind = 1;
    hb->ab = helfem::atomic::TwoDBasis(a.Z, opt.nnodes, Nquad, bval, lval, mval);
    if (opt.verbose)
      printf("Basis set consists of %i angular shells composed of %i radial functions, totaling %i basis functions\n",
             (int)hb->ab.Nang(), (int)hb->ab.Nrad(), (int)hb->ab.Nbf());
    const bool dft = (opt.x_func > 0 || opt.c_func > 0);
    int ldft = opt.ldft, mdft = opt.mdft;
    if (dft) {
      if (ldft == 0) ldft = 4 * a.lmax + 10;
      if (ldft < 2 * a.lmax) throw std::logic_error("Increase ldft to guarantee accuracy of quadrature!\n");
      if (mdft == 0) mdft = 4 * a.mmax + 5;
      if (mdft < 2 * a.mmax) throw std::logic_error("Increase mdft to guarantee accuracy of quadrature!\n");
    } else
      ldft = mdft = 0;
    std::vector<std::vector<std::vector<size_t> > > avg;
    if (a.maverage) avg = helfem::scf::atomic_average_groups(hb->ab);
    r = hfg::scf_device_loop(ctx, hb, opt, nel, 0.0, opt.symmetry, hb->ab.get_sym_idx(opt.symmetry), ldft, mdft, avg);
  } catch (...) {
    hfg_basis_destroy(hb);
    throw;
  }
  hfg_basis_destroy(hb);
  return r;
}
}  // namespace

extern "C" {
int hfg_scf_set_occupations(int readocc, int nrows, int ncols, const int *rows) {
  g_readocc = readocc;
  g_occs.clear();
  for (int r = 0; r < nrows; r++) g_occs.push_back(std::vector<int>(rows + (size_t)r * ncols, rows + (size_t)(r + 1) * ncols));
  return 0;
}
int hfg_scf_set_iguess(int iguess) {
  if (iguess != 0 && iguess != 3) {
    hfg::set_error(iguess == 2 ? "Unsupported guess (SAP needs the reference's tabulated potentials)\n"
                               : "Unsupported guess (GSZ needs per-element parameters: use hfg_model_potential)\n");
    return 1;
  }
  g_iguess = iguess;
  return 0;
}

/// Restricted closed-shell diatomic SCF on the GPU (driver loop of src/diatomic/main.cpp:780-995).
/// out[0..7] = Etot, Ekin, Epot, Ecoul, Exx, Exc, Enucr, iterations (+0.5 if converged);
/// out[8..11] = seconds of the last iteration's J, K, XC, diagonalisation
int hfg_scf_diatomic(hfg_ctx *ctx, int Z1, int Z2, double Rbond, const int *lmmax, int nlm, int nelem, int nnodes,
                     int nquad, double Rmax, int igrid, double zexp, int lpad, const char *method, int ldft, int mdft,
                     int symmetry, int multiplicity, int maxit, double convthr, int verbose, double *out) {
  try {
    helfem::scf::Options o;
    o.multiplicity = multiplicity < 0 ? -multiplicity : multiplicity;  // negative: restricted open shell (ROHF)
    if (multiplicity < 0) o.restricted = 1;
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
    o.iguess = g_iguess;
    o.readocc = g_readocc;
    o.occs = g_occs;
    helfem::range_separation(o.x_func, o.omega, o.kfrac, o.kshort);
    {
      bool erf, yuk;
      helfem::is_range_separated(o.x_func, erf, yuk);
      o.rs_kind = yuk ? 1 : (erf ? 2 : 0);
    }
    o.ldft = ldft;
    o.mdft = mdft;
    o.symmetry = symmetry;
    o.maxit = maxit;
    o.convthr = convthr;
    o.verbose = verbose != 0;
    helfem::scf::Result r;
    if (host_driver()) {
      GPUBackend be(ctx);
      r = helfem::scf::run_diatomic(o, be);
    } else
      r = run_diatomic_device(ctx, o);
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
  } catch (const std::logic_error &e) {
    hfg::set_error(e.what());
    return 1;
  } catch (const std::runtime_error &e) {
    hfg::set_error(e.what());
    return 2;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 3;
  }
  return 0;
}

int hfg_scf_atomic(hfg_ctx *ctx, int Z, int Q, int lmax, int mmax, int nelem, int nnodes, int nquad, double Rmax,
                   int igrid, double zexp, const char *method, int ldft, int mdft, int symmetry, int multiplicity,
                   int maverage, int maxit, double convthr, int verbose, double *out) {
  try {
    helfem::scf::AtomicOptions a;
    a.maverage = maverage != 0;
    a.common.multiplicity = multiplicity < 0 ? -multiplicity : multiplicity;
    if (multiplicity < 0) a.common.restricted = 1;
    a.Z = Z;
    a.Q = Q;
    a.lmax = lmax;
    a.mmax = mmax;
    helfem::scf::Options &o = a.common;
    o.dampfock = 0.7;  // defaults of the atomic program (atomic/main.cpp:111-112)
    o.dampthr = 0.1;
    o.nelem = nelem;
    o.nnodes = nnodes;
    o.nquad = nquad;
    o.Rmax = Rmax;
    o.igrid = igrid;
    o.zexp = zexp;
    o.method = method;
    helfem::parse_xc_func(o.x_func, o.c_func, o.method);
    o.iguess = g_iguess;
    o.readocc = g_readocc;
    o.occs = g_occs;
    helfem::range_separation(o.x_func, o.omega, o.kfrac, o.kshort);
    {
      bool erf, yuk;
      helfem::is_range_separated(o.x_func, erf, yuk);
      o.rs_kind = yuk ? 1 : (erf ? 2 : 0);
    }
    o.ldft = ldft;
    o.mdft = mdft;
    o.symmetry = symmetry;
    o.maxit = maxit;
    o.convthr = convthr;
    o.verbose = verbose != 0;
    helfem::scf::Result r;
    if (host_driver()) {
      GPUBackend be(ctx);
      r = helfem::scf::run_atomic(a, be);
    } else
      r = run_atomic_device(ctx, a);
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
  } catch (const std::logic_error &e) {
    hfg::set_error(e.what());
    return 1;
  } catch (const std::runtime_error &e) {
    hfg::set_error(e.what());
    return 2;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 3;
  }
  return 0;
}

// ---- complete runs behind one options structure (the reference's command lines) --------------------------------------
static const char *const k_elements[] = {
    "",   "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",  "Cl", "Ar", "K",
    "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",
    "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr",
    "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au",
    "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es",
    "Fm", "Md", "No", "Lr", "Rf", "Db", "Sg", "Bh", "Hs", "Mt", "Ds", "Rg", "Cn", "Nh", "Fl", "Mc", "Lv", "Ts", "Og"};

int hfg_get_Z(const char *el) {
  if (!el || !el[0]) return 0;  // no nucleus (elements.cpp:27-29)
  if (!isalpha((unsigned char)el[0])) return atoi(el);
  for (int Z = 1; Z < (int)(sizeof(k_elements) / sizeof(k_elements[0])); Z++) {
    const char *a = el, *b = k_elements[Z];
    while (*a && *b && tolower((unsigned char)*a) == tolower((unsigned char)*b)) a++, b++;
    if (!*a && !*b) return Z;
  }
  hfg::set_error(std::string("Element \"") + el + "\" not found in table of elements!\n");
  return -1;
}

int hfg_parse_xc_params(const char *input, double *pars, int *n) {
  try {
    const int cap = *n;
    *n = 0;
    if (!input || !input[0]) return 0;
    std::string text;
    {
      FILE *f = fopen(input, "r");
      if (f) {  // a file name: one or more numbers, whitespace separated (arma raw_ascii)
        char buf[4096];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
        fclose(f);
      } else
        text = input;  // a string of numbers (arma::vec(std::string))
    }
    const char *ptr = text.c_str();
    for (;;) {
      while (*ptr && (isspace((unsigned char)*ptr) || *ptr == ',' || *ptr == ';')) ptr++;
      if (!*ptr) break;
      char *end = nullptr;
      double v = strtod(ptr, &end);
      if (end == ptr) throw std::runtime_error(std::string("Cannot parse functional parameters from \"") + input + "\"\n");
      if (*n >= cap) throw std::logic_error("hfg_parse_xc_params: capacity too small\n");
      pars[(*n)++] = v;
      ptr = end;
    }
  } catch (const std::logic_error &e) {
    hfg::set_error(e.what());
    return 1;
  } catch (const std::runtime_error &e) {
    hfg::set_error(e.what());
    return 2;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 3;
  }
  return 0;
}

int hfg_scf_options_default(hfg_scf_options *o, int program) {
  if (!o) return 1;
  memset(o, 0, sizeof(*o));
  o->program = program;
  o->Rbond = 0.0;
  o->M = 0;
  o->mmax = program ? 0 : -1;
  o->lpad = 10;
  o->Rmax = 40.0;
  o->grid = 4;
  o->zexp = program ? 2.0 : 1.0;
  o->nnodes = 15;
  o->maxit = 50;
  o->convthr = 1e-7;
  o->diag = 1;
  strcpy(o->method, "HF");
  o->dftthr = 1e-12;
  o->restricted = -1;
  o->symmetry = 1;
  o->primbas = 4;
  o->diiseps = 1e-2;
  o->diisthr = 1e-3;
  o->diisorder = 5;
  o->dampfock = program ? 0.7 : 1.0;
  o->dampthr = 0.1;
  o->iguess = 0;  // the reference default (2, SAP) needs its tabulated potentials: the drivers say so when it is asked for
  strcpy(o->save, "helfem.chk");
  o->verbose = 1;
  return 0;
}

namespace hfg {
// what the reference's drivers leave in their checkpoint (diatomic/main.cpp:236-537, 790-963), in its HDF5 layout
void write_checkpoint(const hfg_scf_options &p, const helfem::scf::Options &o, const helfem::scf::Result &r) {
  helfem::Checkpoint chk(p.save, true);
  chk.write("nela", r.nela);
  chk.write("nelb", r.nelb);
  int Nquad = o.nquad ? o.nquad : 5 * o.nnodes;
  if (p.program == 0) {
    helfem::IVec lval, mval;
    helfem::diatomic::lm_to_l_m(o.lmmax, lval, mval);
    const double Rhalf = 0.5 * o.Rbond;
    helfem::Vec bval = helfem::get_grid(helfem::arcosh(o.Rmax / Rhalf), o.nelem, o.igrid, o.zexp);
    chk.write(helfem::diatomic::TwoDBasis(o.Z1, o.Z2, Rhalf, o.nnodes, Nquad, bval, lval, mval, o.lpad));
  } else {
    helfem::IVec lval, mval;
    helfem::atomic::angular_basis(p.lmax, p.mmax, lval, mval);
    chk.write(helfem::atomic::TwoDBasis(p.Z1, o.nnodes, Nquad, helfem::get_grid(o.Rmax, o.nelem, o.igrid, o.zexp), lval, mval));
  }
  chk.write("Enucr", r.Enucr);
  for (const char *name : {"S", "T", "Sinvh", "Vnuc", "H0", "P", "Pa", "Pb", "J", "Ka", "Kb", "XCa", "XCb", "Fa", "Fb", "Ca", "Cb"}) {
    auto it = r.mats.find(name);
    if (it != r.mats.end()) chk.write(name, it->second);
  }
  chk.write("Ekin", r.Ekin);
  chk.write("Epot", r.Epot);
  chk.write("Eefield", 0.0);
  chk.write("Emfield", 0.0);
  chk.write("Ecoul", r.Ecoul);
  chk.write("Exx", r.Exx);
  chk.write("Exc", r.Exc);
  chk.write("Etot", r.Etot);
  chk.write("Ea", r.E);
  chk.write("Eb", r.Eb.empty() ? r.E : r.Eb);
  chk.write("Converged", r.converged ? 1 : 0);
}
}  // namespace hfg

static void check_options(const hfg_scf_options &p) {
    // features outside the hot-path scope: refuse loudly rather than compute something else
  if (p.Ez != 0.0 || p.Qzz != 0.0 || p.Bz != 0.0) throw std::logic_error("External electric / magnetic fields are not supported by this build.\n");
  if (p.finitenuc != 0) throw std::logic_error("Finite nuclear models are not supported by this build.\n");
  if (p.readocc != 0 && (!p.occs || p.occ_rows <= 0 || p.occ_cols < 3))
    throw std::logic_error(p.program == 0 ? "Must have at least three columns in occupation data.\n"
                                          : "Must have three columns in occupation data to use axial symmetry.\n");
  if (p.perturb != 0.0) throw std::logic_error("Random perturbation of the guess (--perturb) is not supported by this build.\n");
  if (p.primbas != 4) throw std::logic_error("Only the LIP primitive basis (--primbas 4) is supported by this build.\n");
  if (p.iconf != 0) throw std::logic_error("Confinement potentials (--iconf) are not supported by this build.\n");
  if (p.zeroder != 0) throw std::logic_error("--zeroder is not supported by this build.\n");
  if (p.iguess == 2) throw std::logic_error("Unsupported guess (SAP needs the reference's tabulated potentials)\n");
  if (p.iguess == 1) throw std::logic_error("Unsupported guess (GSZ needs the reference's per-element parameters)\n");
  if (p.iguess != 0 && p.iguess != 3) throw std::logic_error("Unsupported guess\n");

  if (p.nelem <= 0) throw std::logic_error("need option: --nelem\n");
  if (p.program == 0) {
    if (p.nlm <= 0 || p.nlm > HFG_MAX_LMMAX) throw std::logic_error("need option: --lmax\n");
    if (!(p.Rbond > 0.0)) throw std::logic_error("need option: --Rbond\n");
    if (p.maverage) throw std::logic_error("--maverage is implemented for the atomic program only in this build.\n");
  }
  int x_func, c_func;
  helfem::parse_xc_func(x_func, c_func, p.method);  // throws on unknown functionals
  bool erf, yuk;
  helfem::is_range_separated(x_func, erf, yuk);
  if (p.program == 0 && (erf || yuk)) throw std::logic_error("Range separated functionals are not supported.\n");  // diatomic/main.cpp:393
  int nela = p.nela, nelb = p.nelb, Q = p.Q, M = p.M > 0 ? p.M : 1;
  helfem::scf::parse_nela_nelb(nela, nelb, Q, M, p.program == 0 ? p.Z1 + p.Z2 : p.Z1);
  if (nela + nelb <= 0) throw std::logic_error("No electrons.\n");
  if (p.readocc) {  // diatomic/main.cpp:369-380
    int sa = 0, sb = 0;
    for (int r = 0; r < p.occ_rows; r++) {
      sa += p.occs[(size_t)r * p.occ_cols];
      sb += p.occs[(size_t)r * p.occ_cols + 1];
    }
    if (sa != nela)
      throw std::logic_error("Specified alpha occupations don't match wanted spin state.\nOccupying " + std::to_string(sa) +
                             " orbitals but should have " + std::to_string(nela) + " orbitals.\n");
    if (sb != nelb)
      throw std::logic_error("Specified alpha occupations don't match wanted spin state.\nOccupying " + std::to_string(sb) +
                             " orbitals but should have " + std::to_string(nelb) + " orbitals.\n");
  }
}

/* validates an options structure the way hfg_scf_run does before it touches the device (usable without a GPU) */
int hfg_scf_options_check(const hfg_scf_options *opt) {
  try {
    if (!opt) throw std::logic_error("hfg_scf_options_check: null argument\n");
    check_options(*opt);
  } catch (const std::logic_error &e) {
    hfg::set_error(e.what());
    return 1;
  } catch (const std::runtime_error &e) {
    hfg::set_error(e.what());
    return 2;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 3;
  }
  return 0;
}

int hfg_scf_run(hfg_ctx *ctx, const hfg_scf_options *in, hfg_scf_result *res, double *E, double *C) {
  try {
    if (!ctx || !in || !res) throw std::logic_error("hfg_scf_run: null argument\n");
    const hfg_scf_options &p = *in;
    check_options(p);

    helfem::scf::AtomicOptions a;
    helfem::scf::Options &o = a.common;
    o.nela = p.nela;
    o.nelb = p.nelb;
    o.Q = p.Q;
    o.multiplicity = p.M > 0 ? p.M : 1;
    o.restricted = p.restricted;
    o.lpad = p.lpad;
    o.Rmax = p.Rmax;
    o.igrid = p.grid;
    o.zexp = p.zexp;
    o.nelem = p.nelem;
    o.nnodes = p.nnodes;
    o.nquad = p.nquad;
    o.maxit = p.maxit;
    o.convthr = p.convthr;
    o.diag = p.diag != 0;
    o.method = p.method;
    helfem::parse_xc_func(o.x_func, o.c_func, o.method);
    helfem::range_separation(o.x_func, o.omega, o.kfrac, o.kshort);
    {
      bool erf, yuk;
      helfem::is_range_separated(o.x_func, erf, yuk);
      o.rs_kind = yuk ? 1 : (erf ? 2 : 0);
    }
    o.ldft = p.ldft;
    o.mdft = p.mdft;
    o.dftthr = p.dftthr;
    o.symmetry = p.symmetry;
    o.diiseps = p.diiseps;
    o.diisthr = p.diisthr;
    o.diisorder = p.diisorder;
    o.iguess = p.iguess;
    o.readocc = p.readocc;
    if (p.readocc)
      for (int r = 0; r < p.occ_rows; r++) o.occs.push_back(std::vector<int>(p.occs + (size_t)r * p.occ_cols, p.occs + (size_t)(r + 1) * p.occ_cols));
    o.dampfock = p.program ? p.dampfock : 1.0;
    o.dampthr = p.dampthr;
    o.verbose = p.verbose != 0;
    o.keep_matrices = p.save[0] != 0;
    if (o.keep_matrices) {
      std::string why;
      if (!helfem::hdf5_available(&why))  // before the run, not after it
        throw std::runtime_error("Checkpoint: no usable HDF5 library in this process (" + why + "set HELFEM_HDF5_LIB, or run with --save \"\")\n");
    }
    // --x_pars / --c_pars (main.cpp:150-157, xc_func_set_ext_params at dftgrid.cpp:399-417): in force for this run only
    struct ResetPars {
      hfg_ctx *c;
      bool on;
      ~ResetPars() {
        if (on) try {
            hfg::set_xc_params(c, 0, nullptr, 0, 0, nullptr, 0);
          } catch (...) {
          }
      }
    } reset_pars{ctx, p.n_x_pars > 0 || p.n_c_pars > 0};
    if (reset_pars.on) hfg::set_xc_params(ctx, o.x_func, p.x_pars, p.n_x_pars, o.c_func, p.c_pars, p.n_c_pars);
    if (p.load[0]) {  // main.cpp:552-648
      helfem::Checkpoint chk(p.load, false);
      chk.read("S", o.guessS);
      chk.read("Ca", o.guessCa);
      chk.read("Ea", o.guessEa);
      if (chk.exist("Cb")) chk.read("Cb", o.guessCb);
      if (chk.exist("Eb")) chk.read("Eb", o.guessEb);
      if (p.program == 0) o.guess_basis = std::make_shared<helfem::diatomic::TwoDBasis>(chk.read_diatomic_basis(p.lpad));
      else o.guess_basis_atomic = std::make_shared<helfem::atomic::TwoDBasis>(chk.read_atomic_basis());
      o.have_guess = true;
    }
    helfem::scf::Result r;
    if (p.program == 0) {
      o.Z1 = p.Z1;
      o.Z2 = p.Z2;
      o.Rbond = p.Rbond;
      o.lmmax.assign(p.lmmax, p.lmmax + p.nlm);
      if (host_driver()) {
        GPUBackend be(ctx);
        r = helfem::scf::run_diatomic(o, be);
      } else
        r = run_diatomic_device(ctx, o);
    } else {
      a.Z = p.Z1;
      a.Q = p.Q;
      a.lmax = p.lmax;
      a.mmax = p.mmax;
      a.maverage = p.maverage != 0;
      if (host_driver()) {
        GPUBackend be(ctx);
        r = helfem::scf::run_atomic(a, be);
      } else
        r = run_atomic_device(ctx, a);
    }
    res->Etot = r.Etot;
    res->Ekin = r.Ekin;
    res->Epot = r.Epot;
    res->Enucr = r.Enucr;
    res->Ecoul = r.Ecoul;
    res->Exx = r.Exx;
    res->Exc = r.Exc;
    res->iterations = r.iterations;
    res->converged = r.converged ? 1 : 0;
    res->nela = r.nela;
    res->nelb = r.nelb;
    res->Nbf = (int64_t)r.Nbf;
    res->tJ = r.tJ;
    res->tK = r.tK;
    res->tXC = r.tXC;
    res->tdiag = r.tdiag;
    if (E && r.E.size()) memcpy(E, r.E.data(), sizeof(double) * r.E.size());
    if (C && r.C.n_elem()) memcpy(C, r.C.memptr(), sizeof(double) * r.C.n_elem());
    if (p.save[0]) hfg::write_checkpoint(p, o, r);
  } catch (const std::logic_error &e) {
    hfg::set_error(e.what());
    return 1;
  } catch (const std::runtime_error &e) {
    hfg::set_error(e.what());
    return 2;
  } catch (const std::exception &e) {
    hfg::set_error(e.what());
    return 3;
  }
  return 0;
}
}
