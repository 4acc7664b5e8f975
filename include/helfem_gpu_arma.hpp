// helfem_gpu_arma.hpp — header-only C++ adapter: the reference's own class and function signatures for the SCF hot path
// on top of the C ABI of helfem_gpu.h, so that the reference's drivers (src/diatomic/main.cpp, src/atomic/main.cpp)
// compile against the MI355X implementation by swapping a namespace.
//
//   reference (paths under /root/reference)                                   here (namespace helfem::gpu)
//   diatomic::basis::TwoDBasis::TwoDBasis(...)          src/diatomic/basis.cpp:307      diatomic::TwoDBasis<Mat>(...)
//   void TwoDBasis::compute_tei(bool exchange)          src/diatomic/basis.h:205        same
//   arma::mat TwoDBasis::coulomb(const arma::mat &P)    src/diatomic/basis.h:247        same
//   arma::mat TwoDBasis::exchange(const arma::mat &P)   src/diatomic/basis.h:249        same
//   arma::mat TwoDBasis::overlap/kinetic/nuclear()      src/diatomic/basis.h:227-233    same
//   std::vector<arma::uvec> TwoDBasis::get_sym_idx(int) src/diatomic/basis.h:303        same (std::vector<std::vector<unsigned long long>>)
//   void DFTGrid::eval_Fxc(x_func, x_pars, c_func, c_pars, P, H, Exc, Nel, Ekin, thr)             dftgrid.h:179   same
//   void DFTGrid::eval_Fxc(x_func, x_pars, c_func, c_pars, Pa, Pb, Ha, Hb, Exc, Nel, Ekin, beta, thr)  dftgrid.h:181   same
//   arma::mat scf::form_density(const arma::mat &C, size_t nocc)                        scf_helpers.h:24  same
//   void scf::eig_gsym(arma::vec &E, arma::mat &C, const arma::mat &F, const arma::mat &Sinvh)          scf_helpers.h:34  same
//   void scf::eig_gsym_sub(E, C, F, Sinvh, const std::vector<arma::uvec> &m_idx, bool verbose = true)   scf_helpers.h:36  same
//   atomic::basis::TwoDBasis (coulomb / exchange / rs_exchange / compute_tei / compute_yukawa / compute_erfc),
//   atomic::dftgrid::DFTGrid::eval_Fxc                  src/atomic/TwoDBasis.h:180-190  atomic::TwoDBasis<Mat>, DFTGrid
//
// Matrix types.  Everything is a template over the matrix / vector types, which need what arma::mat / arma::vec offer:
//   Mat(rows, cols) constructor, memptr(), n_rows, n_cols;   Vec: memptr() + set_size(n), or data() + resize(n).
// With Armadillo included first (ARMA_VERSION_MAJOR defined) the aliases at the end give the reference's exact names for
// arma::mat; helfem_amd/csrc/host/linalg.h's Mat satisfies the same requirements (that is what tests/cpp/adapter_test.cpp
// compiles against in the Armadillo-free build image).
//
// Errors.  A non-zero status of the C ABI becomes the exception class the reference throws in that situation: 1 ->
// std::logic_error ("Primitive teis have not been computed!", basis.cpp:1361), 2 and 3 -> std::runtime_error.
//
// Data movement.  These calls take host matrices, as the reference's do: every call stages its operands through pinned
// memory over PCIe (about 12 ms per Fock build + eigensolve at Nbf = 4230).  A driver that wants the device-resident SCF
// loop (no PCIe traffic per iteration) calls run_scf() below, which is hfg_scf_run(): the whole loop of main.cpp:780-995 in HBM.
#pragma once
#include "helfem_gpu.h"

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace helfem {
namespace gpu {

inline void check(int rc) {
  if (rc == 0) return;
  const std::string msg = hfg_last_error();
  if (rc == 1) throw std::logic_error(msg);
  throw std::runtime_error(msg);
}

namespace detail {
template <class V>
auto data_of(V &v, int) -> decltype(v.memptr()) {
  return v.memptr();
}
template <class V>
auto data_of(V &v, long) -> decltype(v.data()) {
  return v.data();
}
template <class V>
auto set_len(V &v, size_t n, int) -> decltype(v.set_size(n), void()) {
  v.set_size(n);
}
template <class V>
auto set_len(V &v, size_t n, long) -> decltype(v.resize(n), void()) {
  v.resize(n);
}
template <class V>
auto len_of(const V &v, int) -> decltype(v.n_elem, size_t()) {
  return (size_t)v.n_elem;
}
template <class V>
auto len_of(const V &v, long) -> decltype(v.size(), size_t()) {
  return (size_t)v.size();
}
template <class Mat>
void need_square(const Mat &M, size_t N, const char *what) {
  if ((size_t)M.n_rows != N || (size_t)M.n_cols != N) throw std::logic_error(std::string(what) + ": matrix does not have the dimension of the basis!\n");
}
}  // namespace detail

/// one device + one stream; shared by the objects created from it (one context per host thread)
class Context {
 public:
  explicit Context(int device = 0) { check(hfg_ctx_create(&ctx_, device, nullptr)); }
  ~Context() {
    if (ctx_) hfg_ctx_destroy(ctx_);
  }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  hfg_ctx *handle() const { return ctx_; }

 private:
  hfg_ctx *ctx_ = nullptr;
};

/// common part of the two basis classes: tables, one-electron matrices, J and K
template <class Mat>
class BasisBase {
 public:
  typedef std::vector<unsigned long long> uvec;

  size_t Nbf() const { return N_; }
  size_t Nrad() const { return Nrad_; }
  size_t Nang() const { return Nang_; }
  size_t Nel() const { return Nel_; }

  /// TwoDBasis::compute_tei (basis.cpp:1166 / atomic TwoDBasis.cpp:666)
  void compute_tei(bool exchange) {
    check(hfg_compute_tei(b_, exchange ? 1 : 0));
    uploaded_ = false;
  }
  Mat overlap() const { return one_body(hfg_basis_overlap); }
  Mat kinetic() const { return one_body(hfg_basis_kinetic); }
  Mat nuclear() const { return one_body(hfg_basis_nuclear); }
  /// TwoDBasis::coulomb (basis.cpp:1359): throws std::logic_error before compute_tei, like the reference
  Mat coulomb(const Mat &P) const { return two_body(hfg_coulomb, P, "coulomb"); }
  /// TwoDBasis::exchange (basis.cpp:1532)
  Mat exchange(const Mat &P) const { return two_body(hfg_exchange, P, "exchange"); }
  /// TwoDBasis::get_sym_idx (basis.cpp:561 / atomic TwoDBasis.cpp:202)
  std::vector<uvec> get_sym_idx(int symm) const {
    int nblk = 0;
    check(hfg_basis_sym_blocks(b_, symm, &nblk, nullptr, nullptr));
    std::vector<int64_t> ptr(nblk + 1), idx(N_);
    check(hfg_basis_sym_blocks(b_, symm, &nblk, ptr.data(), idx.data()));
    std::vector<uvec> out(nblk);
    for (int i = 0; i < nblk; i++)
      for (int64_t k = ptr[i]; k < ptr[i + 1]; k++) out[i].push_back((unsigned long long)idx[k]);
    return out;
  }

  hfg_basis *handle() const { return b_; }
  hfg_ctx *context() const { return ctx_->handle(); }
  /// tables -> HBM with the XC grid (lang, mang); done lazily by the first call that needs it
  void upload(int lang, int mang) const {
    if (uploaded_ && lang == lang_ && mang == mang_) return;
    check(hfg_basis_upload(ctx_->handle(), b_, lang, mang));
    uploaded_ = true;
    lang_ = lang;
    mang_ = mang;
  }

 protected:
  explicit BasisBase(const std::shared_ptr<Context> &ctx) : ctx_(ctx) {}
  ~BasisBase() {
    if (b_) hfg_basis_destroy(b_);
  }
  BasisBase(const BasisBase &) = delete;
  BasisBase &operator=(const BasisBase &) = delete;
  void read_dims() {
    int64_t N, Nd, Nr, Na, Ne;
    check(hfg_basis_dims(b_, &N, &Nd, &Nr, &Na, &Ne));
    N_ = (size_t)N;
    Nrad_ = (size_t)Nr;
    Nang_ = (size_t)Na;
    Nel_ = (size_t)Ne;
  }
  Mat one_body(int (*fn)(const hfg_basis *, double *)) const {
    Mat M(N_, N_);
    check(fn(b_, detail::data_of(M, 0)));
    return M;
  }
  Mat two_body(int (*fn)(hfg_ctx *, hfg_basis *, const double *, double *), const Mat &P, const char *what) const {
    detail::need_square(P, N_, what);
    if (!uploaded_) upload(lang_, mang_);
    Mat out(N_, N_);
    check(fn(ctx_->handle(), b_, detail::data_of(P, 0), detail::data_of(out, 0)));
    return out;
  }
  std::shared_ptr<Context> ctx_;
  hfg_basis *b_ = nullptr;
  size_t N_ = 0, Nrad_ = 0, Nang_ = 0, Nel_ = 0;
  mutable bool uploaded_ = false;
  mutable int lang_ = 0, mang_ = 0;
};

namespace diatomic {
/// diatomic::basis::TwoDBasis with the constructor arguments of basis.cpp:307 (poly = LIP on nnodes Lobatto nodes)
template <class Mat>
class TwoDBasis : public BasisBase<Mat> {
 public:
  template <class Vec, class IVec>
  TwoDBasis(const std::shared_ptr<Context> &ctx, int Z1, int Z2, double Rhalf, int nnodes, int n_quad, const Vec &bval, const IVec &lval,
            const IVec &mval, int lpad)
      : BasisBase<Mat>(ctx) {
    std::vector<double> b(detail::len_of(bval, 0));
    for (size_t i = 0; i < b.size(); i++) b[i] = detail::data_of(bval, 0)[i];
    std::vector<int> l(detail::len_of(lval, 0)), m(l.size());
    for (size_t i = 0; i < l.size(); i++) {
      l[i] = (int)detail::data_of(lval, 0)[i];
      m[i] = (int)detail::data_of(mval, 0)[i];
    }
    hfg_diatomic_desc d;
    d.Z1 = Z1;
    d.Z2 = Z2;
    d.Rhalf = Rhalf;
    d.primbas = 4;
    d.nnodes = nnodes;
    d.nquad = n_quad;
    d.bval = b.data();
    d.nbval = (int)b.size();
    d.lval = l.data();
    d.mval = m.data();
    d.nang = (int)l.size();
    d.lpad = lpad;
    check(hfg_diatomic_basis_create(&d, &this->b_));
    this->read_dims();
  }
};
}  // namespace diatomic

namespace atomic {
/// atomic::basis::TwoDBasis (src/atomic/TwoDBasis.cpp:38), point nucleus
template <class Mat>
class TwoDBasis : public BasisBase<Mat> {
 public:
  template <class Vec, class IVec>
  TwoDBasis(const std::shared_ptr<Context> &ctx, int Z, int nnodes, int n_quad, const Vec &bval, const IVec &lval, const IVec &mval)
      : BasisBase<Mat>(ctx) {
    std::vector<double> b(detail::len_of(bval, 0));
    for (size_t i = 0; i < b.size(); i++) b[i] = detail::data_of(bval, 0)[i];
    std::vector<int> l(detail::len_of(lval, 0)), m(l.size());
    for (size_t i = 0; i < l.size(); i++) {
      l[i] = (int)detail::data_of(lval, 0)[i];
      m[i] = (int)detail::data_of(mval, 0)[i];
    }
    hfg_atomic_desc d;
    d.Z = Z;
    d.primbas = 4;
    d.nnodes = nnodes;
    d.nquad = n_quad;
    d.bval = b.data();
    d.nbval = (int)b.size();
    d.lval = l.data();
    d.mval = m.data();
    d.nang = (int)l.size();
    check(hfg_atomic_basis_create(&d, &this->b_));
    this->read_dims();
  }
  /// TwoDBasis::compute_yukawa / compute_erfc (TwoDBasis.cpp:741 / :780) and rs_exchange (:1142)
  void compute_yukawa(double lambda) {
    check(hfg_compute_rs_tei(this->b_, 1, lambda));
    this->uploaded_ = false;
  }
  void compute_erfc(double mu) {
    check(hfg_compute_rs_tei(this->b_, 2, mu));
    this->uploaded_ = false;
  }
  Mat rs_exchange(const Mat &P) const { return this->two_body(hfg_rs_exchange, P, "rs_exchange"); }
};
}  // namespace atomic

/// diatomic::dftgrid::DFTGrid / atomic::dftgrid::DFTGrid: basis pointer + angular rule (dftgrid.h:160-181)
template <class Mat>
class DFTGrid {
 public:
  DFTGrid() {}
  DFTGrid(const BasisBase<Mat> *bas, int lang, int mang) : basp_(bas), lang_(lang), mang_(mang) {}

  /// restricted: dftgrid.h:179
  template <class Vec>
  void eval_Fxc(int x_func, const Vec &x_pars, int c_func, const Vec &c_pars, const Mat &P, Mat &H, double &Exc, double &Nel, double &Ekin,
                double thr) const {
    need();
    detail::need_square(P, basp_->Nbf(), "eval_Fxc");
    basp_->upload(lang_, mang_);
    H = Mat(basp_->Nbf(), basp_->Nbf());
    const int nx = (int)detail::len_of(x_pars, 0), nc = (int)detail::len_of(c_pars, 0);
    check(hfg_xc_fock_ext(basp_->context(), basp_->handle(), x_func, nx ? detail::data_of(x_pars, 0) : nullptr, nx, c_func,
                          nc ? detail::data_of(c_pars, 0) : nullptr, nc, detail::data_of(P, 0), detail::data_of(H, 0), &Exc, &Nel, &Ekin, thr));
  }
  /// unrestricted: dftgrid.h:181 (beta = false: no beta electrons -- Hb is still returned, as zeros of the functional's
  /// response to an empty channel, the reference leaves it unset)
  template <class Vec>
  void eval_Fxc(int x_func, const Vec &x_pars, int c_func, const Vec &c_pars, const Mat &Pa, const Mat &Pb, Mat &Ha, Mat &Hb, double &Exc,
                double &Nel, double &Ekin, bool beta, double thr) const {
    (void)beta;
    need();
    detail::need_square(Pa, basp_->Nbf(), "eval_Fxc");
    detail::need_square(Pb, basp_->Nbf(), "eval_Fxc");
    basp_->upload(lang_, mang_);
    Ha = Mat(basp_->Nbf(), basp_->Nbf());
    Hb = Mat(basp_->Nbf(), basp_->Nbf());
    const int nx = (int)detail::len_of(x_pars, 0), nc = (int)detail::len_of(c_pars, 0);
    check(hfg_xc_fock_pol_ext(basp_->context(), basp_->handle(), x_func, nx ? detail::data_of(x_pars, 0) : nullptr, nx, c_func,
                              nc ? detail::data_of(c_pars, 0) : nullptr, nc, detail::data_of(Pa, 0), detail::data_of(Pb, 0), detail::data_of(Ha, 0),
                              detail::data_of(Hb, 0), &Exc, &Nel, &Ekin, thr));
  }

 private:
  void need() const {
    if (!basp_) throw std::logic_error("DFTGrid has no basis!\n");
  }
  const BasisBase<Mat> *basp_ = nullptr;
  int lang_ = 0, mang_ = 0;
};

namespace scf {
/// scf::form_density (scf_helpers.cpp:22): P = C(:, 0:nocc-1) C(:, 0:nocc-1)^T
template <class Mat>
Mat form_density(const Context &ctx, const Mat &C, size_t nocc) {
  if ((size_t)C.n_cols < nocc) throw std::logic_error("Not enough orbitals!\n");
  Mat P(C.n_rows, C.n_rows);
  check(hfg_form_density(ctx.handle(), (int64_t)C.n_rows, (int64_t)C.n_cols, detail::data_of(C, 0), (int64_t)nocc, detail::data_of(P, 0)));
  return P;
}
/// scf::eig_gsym (scf_helpers.cpp:131)
template <class Vec, class Mat>
void eig_gsym(const Context &ctx, Vec &E, Mat &C, const Mat &F, const Mat &Sinvh) {
  const size_t N = F.n_rows, n = Sinvh.n_cols;
  if ((size_t)F.n_cols != N || (size_t)Sinvh.n_rows != N) throw std::logic_error("eig_gsym: incompatible dimensions\n");
  detail::set_len(E, n, 0);
  C = Mat(N, n);
  check(hfg_eig_gsym(ctx.handle(), (int64_t)N, (int64_t)n, detail::data_of(F, 0), detail::data_of(Sinvh, 0), detail::data_of(E, 0), detail::data_of(C, 0)));
}
/// scf::eig_gsym_sub (scf_helpers.cpp:142): one generalized eigenproblem per symmetry block, levels sorted globally;
/// throws std::logic_error("Symmetry mismatch in eig_gsym_sub") when the blocks do not cover the basis, like the reference
template <class Vec, class Mat, class UVec>
void eig_gsym_sub(const Context &ctx, Vec &E, Mat &C, const Mat &F, const Mat &Sinvh, const std::vector<UVec> &m_idx, bool verbose = true) {
  (void)verbose;
  const size_t N = F.n_rows;
  if ((size_t)F.n_cols != N || (size_t)Sinvh.n_rows != N || (size_t)Sinvh.n_cols != N) throw std::logic_error("eig_gsym_sub: incompatible dimensions\n");
  std::vector<int64_t> ptr(1, 0), idx;
  for (const UVec &b : m_idx) {
    for (size_t k = 0; k < detail::len_of(b, 0); k++) idx.push_back((int64_t)detail::data_of(b, 0)[k]);
    ptr.push_back((int64_t)idx.size());
  }
  detail::set_len(E, N, 0);
  C = Mat(N, N);
  check(hfg_eig_gsym_sub(ctx.handle(), (int64_t)N, detail::data_of(F, 0), detail::data_of(Sinvh, 0), (int)m_idx.size(), ptr.data(), idx.data(),
                         detail::data_of(E, 0), detail::data_of(C, 0)));
}
/// the two eig_gsym_sub calls of an unrestricted iteration (diatomic/main.cpp:936-958) as one batch on the device
/// (hfg_eig_gsym_sub_pair): in the reference's loop, replace
///     scf::eig_gsym_sub(Ea,Ca,Fa,Sinvh,dsym); ... scf::eig_gsym_sub(Eb,Cb,Fb,Sinvh,dsym);
/// by one call of this function -- the results are the same, the time is that of ONE call plus the matrix products
template <class Vec, class Mat, class UVec>
void eig_gsym_sub_pair(const Context &ctx, Vec &Ea, Mat &Ca, Vec &Eb, Mat &Cb, const Mat &Fa, const Mat &Fb, const Mat &Sinvh,
                       const std::vector<UVec> &m_idx) {
  const size_t N = Fa.n_rows;
  if ((size_t)Fa.n_cols != N || (size_t)Fb.n_rows != N || (size_t)Fb.n_cols != N || (size_t)Sinvh.n_rows != N || (size_t)Sinvh.n_cols != N)
    throw std::logic_error("eig_gsym_sub_pair: incompatible dimensions\n");
  std::vector<int64_t> ptr(1, 0), idx;
  for (const UVec &b : m_idx) {
    for (size_t k = 0; k < detail::len_of(b, 0); k++) idx.push_back((int64_t)detail::data_of(b, 0)[k]);
    ptr.push_back((int64_t)idx.size());
  }
  detail::set_len(Ea, N, 0);
  detail::set_len(Eb, N, 0);
  Ca = Mat(N, N);
  Cb = Mat(N, N);
  check(hfg_eig_gsym_sub_pair(ctx.handle(), (int64_t)N, detail::data_of(Fa, 0), detail::data_of(Fb, 0), detail::data_of(Sinvh, 0), (int)m_idx.size(),
                              ptr.data(), idx.data(), detail::data_of(Ea, 0), detail::data_of(Ca, 0), detail::data_of(Eb, 0), detail::data_of(Cb, 0)));
}
/// TwoDBasis::Sinvh (basis.cpp:627) -> utils::invh per symmetry block
template <class Mat, class UVec>
Mat form_Sinvh(const Context &ctx, const Mat &S, bool chol, const std::vector<UVec> &m_idx) {
  const size_t N = S.n_rows;
  std::vector<int64_t> ptr(1, 0), idx;
  for (const UVec &b : m_idx) {
    for (size_t k = 0; k < detail::len_of(b, 0); k++) idx.push_back((int64_t)detail::data_of(b, 0)[k]);
    ptr.push_back((int64_t)idx.size());
  }
  Mat X(N, N);
  check(hfg_form_sinvh(ctx.handle(), (int64_t)N, detail::data_of(S, 0), chol ? 1 : 0, (int)m_idx.size(), ptr.data(), idx.data(), detail::data_of(X, 0)));
  return X;
}
}  // namespace scf

/// the whole calculation with every matrix resident in HBM (hfg_scf_run): what a driver calls instead of looping over
/// the host-matrix entry points above when it does not need the intermediate matrices
inline hfg_scf_result run_scf(const Context &ctx, const hfg_scf_options &opt) {
  hfg_scf_result r;
  check(hfg_scf_run(ctx.handle(), &opt, &r, nullptr, nullptr));
  return r;
}

}  // namespace gpu
}  // namespace helfem

#ifdef ARMA_VERSION_MAJOR
// the reference's names for the reference's matrix type
namespace helfem {
namespace gpu {
namespace diatomic {
namespace basis {
typedef ::helfem::gpu::diatomic::TwoDBasis<arma::mat> TwoDBasis;
}
namespace dftgrid {
typedef ::helfem::gpu::DFTGrid<arma::mat> DFTGrid;
}
}  // namespace diatomic
namespace atomic {
namespace basis {
typedef ::helfem::gpu::atomic::TwoDBasis<arma::mat> TwoDBasis;
}
namespace dftgrid {
typedef ::helfem::gpu::DFTGrid<arma::mat> DFTGrid;
}
}  // namespace atomic
}  // namespace gpu
}  // namespace helfem
#endif
