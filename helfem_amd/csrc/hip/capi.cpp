// C ABI (include/helfem_gpu.h): contexts, host-side basis API, host-pointer and device-pointer
// entry points.  No torch types, no exceptions across the boundary.
#include "common.h"
#include "../host/checkpoint.h"
#include "../host/diis.h"
#include "tables.h"
#include <cstring>
#include <mutex>

namespace hfg {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// implemented in the .hip translation units
void coulomb_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dJ);
void xc_fock_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dH, double *dScal,
                 double thr);
void exchange_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dK, bool rs = false, const double *Lknown = nullptr,
                  int rknown = 0);
void set_xc_params(hfg_ctx *ctx, int x_func, const double *x_pars, int nx, int c_func, const double *c_pars, int nc);
void xc_fock_pol_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dPa, const double *dPb,
                     double *dHa, double *dHb, double *dScal, double thr);
void model_potential_dev(hfg_ctx *ctx, hfg_basis *basis, int kind1, int Z1, double d1, double H1, int kind2, int Z2,
                         double d2, double H2, double *dH);
void fock_release(hfg_dev_tables *t);
size_t fock_compact_size(hfg_basis *basis);
void fock_compact_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dFc,
                      double *dScal, double thr);
void fock_finish_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dFc, const double *dH0, const int *dBlockId,
                     double *dF);
void exchange_release(hfg_dev_tables *t);
void exchange_lr_release(hfg_dev_tables *t);
void compute_tei_dev(hfg_ctx *ctx, hfg_basis *basis);
void eig_release(hfg_ctx *ctx);
void dc_release(hfg_ctx *ctx);
void trd_release(hfg_ctx *ctx);
void trdp_release(hfg_ctx *ctx);
void trdp_check_status(hfg_ctx *ctx);
void trd_measure_gemv(hfg_ctx *ctx, double *ms, int64_t *launches);
void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);
void eig_sym_dev(hfg_ctx *ctx, int n, const double *dA, double *dE, double *dC);
void eig_gsym_dev(hfg_ctx *ctx, int N, int n, const double *dF, const double *dS, double *dE, double *dC);
void eig_gsym_sub_dev(hfg_ctx *ctx, int N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                      const int64_t *blk_idx, double *dE, double *dC);
size_t eig_block_buf_size(int nblk, const int64_t *blk_ptr);
void eig_gsym_sub_pair_dev(hfg_ctx *ctx, int N, const double *dFa, const double *dFb, const double *dS, int nblk, const int64_t *blk_ptr,
                           const int64_t *blk_idx, double *dEa, double *dCa, double *dEb, double *dCb);
void eig_blocks_dev(hfg_ctx *ctx, int N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                    const int64_t *blk_idx, double *dBlockBuf);
void eig_assemble_dev(hfg_ctx *ctx, int N, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                      const double *dBlockBuf, double *dE, double *dC);
void form_sinvh_dev(hfg_ctx *ctx, int N, const double *dS, bool chol, int nblk, const int64_t *blk_ptr,
                    const int64_t *blk_idx, double *dSinvh);
void form_density_dev(hfg_ctx *ctx, int N, int ncols, const double *dC, int nocc, double *dP);
}  // namespace hfg

using namespace hfg;

#define HFG_TRY try {
// status codes carry the exception class across the C boundary so that the C++ adapter (include/helfem_gpu_arma.hpp)
// can rethrow what the reference would have thrown: 1 std::logic_error (misuse: "Primitive teis have not been
// computed!", basis.cpp:1361), 2 std::runtime_error (functional / shape / device errors, dftgrid.cpp:54), 3 anything else
#define HFG_CATCH                       \
  }                                     \
  catch (const std::logic_error &e) {   \
    hfg::set_error(e.what());           \
    return 1;                           \
  }                                     \
  catch (const std::runtime_error &e) { \
    hfg::set_error(e.what());           \
    return 2;                           \
  }                                     \
  catch (const std::exception &e) {     \
    hfg::set_error(e.what());           \
    return 3;                           \
  }                                     \
  return 0;

// ---- context helpers ---------------------------------------------------------------------------
hipStream_t hfg_ctx::side() {
  if (!side_stream) {
    HFG_HIP_CHECK(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    HFG_HIP_CHECK(hipEventCreateWithFlags(&side_ev[0], hipEventDisableTiming));
    HFG_HIP_CHECK(hipEventCreateWithFlags(&side_ev[1], hipEventDisableTiming));
  }
  return side_stream;
}
void hfg_ctx::drop_side() {
  if (!side_stream) return;
  (void)hipStreamSynchronize(side_stream);
  (void)hipStreamDestroy(side_stream);
  (void)hipEventDestroy(side_ev[0]);
  (void)hipEventDestroy(side_ev[1]);
  side_stream = nullptr;
  side_ev[0] = side_ev[1] = nullptr;
}
void *hfg_ctx::pinned_buf(size_t bytes) {
  if (bytes > pinned_bytes) {
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    HFG_HIP_CHECK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
    pinned_bytes = bytes;
  }
  return pinned;
}
hipEvent_t hfg_ctx::get_event() {
  if (!event_pool.empty()) {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HFG_HIP_CHECK(hipEventCreate(&e));
  return e;
}
void hfg_ctx::prof_begin(const char *name) {
  hipEvent_t a = get_event(), b = get_event();
  HFG_HIP_CHECK(hipEventRecord(a, stream));
  prof[name].pending.push_back(std::make_pair(a, b));
}
void hfg_ctx::prof_end(const char *name) {
  ProfEntry &p = prof[name];
  HFG_HIP_CHECK(hipEventRecord(p.pending.back().second, stream));
  p.launches++;
}
void hfg_ctx::prof_collect() {
  for (auto &kv : prof) {
    for (auto &ev : kv.second.pending) {
      float ms = 0.f;
      HFG_HIP_CHECK(hipEventSynchronize(ev.second));
      HFG_HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
      kv.second.ms += ms;
      event_pool.push_back(ev.first);
      event_pool.push_back(ev.second);
    }
    kv.second.pending.clear();
  }
}

namespace {
// scoped device staging of host matrices
struct Stage {
  hfg_ctx *ctx;
  std::vector<double *> bufs;
  explicit Stage(hfg_ctx *c) : ctx(c) { HFG_HIP_CHECK(hipSetDevice(c->device)); }
  ~Stage() {
    for (double *p : bufs) (void)hipFree(p);
  }
  double *alloc(size_t n) {
    double *p = nullptr;
    HFG_HIP_CHECK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(double)));
    bufs.push_back(p);
    return p;
  }
  double *up(const double *h, size_t n) {
    double *p = alloc(n);
    HFG_HIP_CHECK(hipMemcpyAsync(p, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return p;
  }
  void down(double *h, const double *d, size_t n) {
    HFG_HIP_CHECK(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  void sync() { HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream)); }
};

void require_gpu() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw std::runtime_error("no usable HIP device: the helfem_amd hot path has no CPU fallback");
}
}  // namespace

extern "C" {

const char *hfg_last_error(void) { return g_err.c_str(); }
const char *hfg_version(void) { return "helfem_amd 0.1 (gfx950)"; }

int hfg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int hfg_ctx_create(hfg_ctx **out, int device, void *stream) {
  HFG_TRY
  require_gpu();
  HFG_HIP_CHECK(hipSetDevice(device));
  hfg_ctx *c = new hfg_ctx();
  c->device = device;
  if (stream == HFG_NULL_STREAM) {
    c->stream = nullptr;  // the device's default (null) stream: ordered with everything a framework enqueues there
    c->own_stream = false;
  } else if (stream) {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  } else {
    HFG_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  *out = c;
  HFG_CATCH
}

int hfg_ctx_destroy(hfg_ctx *c) {
  HFG_TRY
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  eig_release(c);
  dc_release(c);
  trd_release(c);
  trdp_release(c);
  for (auto &kv : c->prof)
    for (auto &ev : kv.second.pending) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->side_stream) {
    (void)hipStreamSynchronize(c->side_stream);
    (void)hipStreamDestroy(c->side_stream);
    (void)hipEventDestroy(c->side_ev[0]);
    (void)hipEventDestroy(c->side_ev[1]);
  }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  HFG_CATCH
}

int hfg_ctx_synchronize(hfg_ctx *c) {
  HFG_TRY
  HFG_HIP_CHECK(hipStreamSynchronize(c->stream));
  trdp_check_status(c);  // a persistent tridiagonalisation that ran into a spin limit fails here, loudly
  HFG_CATCH
}

int hfg_ctx_fix_sinvh(hfg_ctx *c, const double *dSinvh) {
  HFG_TRY
  c->fix_sinvh(dSinvh);
  HFG_CATCH
}

int hfg_ctx_set_shard(hfg_ctx *c, int rank, int nranks) {
  HFG_TRY
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::logic_error("invalid shard");
  c->shard_rank = rank;
  c->shard_n = nranks;
  HFG_CATCH
}

// ---- basis ---------------------------------------------------------------------------------------
int hfg_diatomic_basis_create(const hfg_diatomic_desc *d, hfg_basis **out) {
  HFG_TRY
  if (d->primbas != 4) throw std::logic_error("Unsupported primitive basis.\n");
  hfg_basis *b = new hfg_basis();
  b->b = helfem::diatomic::TwoDBasis(d->Z1, d->Z2, d->Rhalf, d->nnodes, d->nquad,
                                     helfem::Vec(d->bval, d->bval + d->nbval),
                                     helfem::IVec(d->lval, d->lval + d->nang),
                                     helfem::IVec(d->mval, d->mval + d->nang), d->lpad);
  *out = b;
  HFG_CATCH
}

int hfg_atomic_basis_create(const hfg_atomic_desc *d, hfg_basis **out) {
  HFG_TRY
  if (d->primbas != 4) throw std::logic_error("Unsupported primitive basis.\n");
  hfg_basis *b = new hfg_basis();
  b->kind = 1;
  b->ab = helfem::atomic::TwoDBasis(d->Z, d->nnodes, d->nquad, helfem::Vec(d->bval, d->bval + d->nbval),
                                    helfem::IVec(d->lval, d->lval + d->nang), helfem::IVec(d->mval, d->mval + d->nang));
  *out = b;
  HFG_CATCH
}

int hfg_angular_basis(int lmax, int mmax, int *lval, int *mval, int *nang) {
  HFG_TRY
  helfem::IVec l, m;
  helfem::atomic::angular_basis(lmax, mmax, l, m);
  if ((int)l.size() > *nang) {
    *nang = (int)l.size();
    throw std::logic_error("hfg_angular_basis: output capacity too small");
  }
  *nang = (int)l.size();
  for (size_t i = 0; i < l.size(); i++) {
    lval[i] = l[i];
    mval[i] = m[i];
  }
  HFG_CATCH
}

int hfg_basis_destroy(hfg_basis *b) {
  HFG_TRY
  if (!b) return 0;
  if (b->dev) {
    fock_release(b->dev);
    exchange_release(b->dev);
    exchange_lr_release(b->dev);
    delete b->dev;
  }
  if (b->dev_rs) {
    exchange_release(b->dev_rs);
    exchange_lr_release(b->dev_rs);
    delete b->dev_rs;
  }
  delete b;
  HFG_CATCH
}

int hfg_basis_dims(const hfg_basis *b, int64_t *Nbf, int64_t *Ndummy, int64_t *Nrad, int64_t *Nang, int64_t *Nel) {
  if (Nbf) *Nbf = b->Nbf();
  if (Ndummy) *Ndummy = b->Ndummy();
  if (Nrad) *Nrad = b->Nrad();
  if (Nang) *Nang = b->Nang();
  if (Nel) *Nel = b->Nel();
  return 0;
}

static int copy_out(const helfem::Mat &M, double *out) {
  memcpy(out, M.memptr(), sizeof(double) * M.n_elem());
  return 0;
}
int hfg_basis_overlap(const hfg_basis *b, double *S) {
  HFG_TRY copy_out(b->kind ? b->ab.overlap() : b->b.overlap(), S);
  HFG_CATCH
}
int hfg_basis_kinetic(const hfg_basis *b, double *T) {
  HFG_TRY copy_out(b->kind ? b->ab.kinetic() : b->b.kinetic(), T);
  HFG_CATCH
}
int hfg_basis_nuclear(const hfg_basis *b, double *V) {
  HFG_TRY copy_out(b->kind ? b->ab.nuclear() : b->b.nuclear(), V);
  HFG_CATCH
}

int hfg_basis_sym_blocks(const hfg_basis *b, int symm, int *nblk, int64_t *blk_ptr, int64_t *blk_idx) {
  HFG_TRY
  auto idx = b->kind ? b->ab.get_sym_idx(symm) : b->b.get_sym_idx(symm);
  *nblk = (int)idx.size();
  if (blk_ptr) {
    int64_t off = 0;
    for (size_t i = 0; i < idx.size(); i++) {
      blk_ptr[i] = off;
      for (size_t k = 0; k < idx[i].size(); k++) blk_idx[off + k] = (int64_t)idx[i][k];
      off += idx[i].size();
    }
    blk_ptr[idx.size()] = off;
  }
  HFG_CATCH
}

int hfg_compute_tei(hfg_basis *b, int exchange) {
  HFG_TRY
  if (b->kind) b->ab.compute_tei(exchange != 0);
  else b->b.compute_tei(exchange != 0);
  HFG_CATCH
}

int hfg_compute_rs_tei(hfg_basis *b, int rs_kind, double omega) {
  HFG_TRY
  if (b->kind == 0) throw std::logic_error("Range separated functionals are not supported.\n");  // diatomic/main.cpp:393
  if (rs_kind == 1) b->ab.compute_yukawa(omega);
  else if (rs_kind == 2) b->ab.compute_erfc(omega);
  else throw std::logic_error("unknown range-separation kernel (1 = Yukawa, 2 = erfc)\n");
  HFG_CATCH
}

int hfg_compute_tei_dev(hfg_ctx *ctx, hfg_basis *b, int exchange) {
  HFG_TRY
  (void)exchange;  // the exchange-ordered copies are made on the device when the exchange kernels first need them
  compute_tei_dev(ctx, b);  // diatomic and atomic bases (hip/tei_dev.hip)
  HFG_CATCH
}

// arma::mat TwoDBasis::overlap(const TwoDBasis &rh) (basis.cpp:713-750): <a | b>, Nbf(a) x Nbf(b), column-major
int hfg_basis_interbasis_overlap(const hfg_basis *a, const hfg_basis *b, double *S12) {
  HFG_TRY
  if (a->kind != b->kind) throw std::logic_error("hfg_basis_interbasis_overlap: the two bases must be of the same program\n");
  const helfem::Mat m = a->kind ? a->ab.overlap(b->ab) : a->b.overlap(b->b);
  std::copy(m.d.begin(), m.d.end(), S12);
  HFG_CATCH
}

int hfg_basis_lm_map(const hfg_basis *b, int *L, int *M, int *n) {
  HFG_TRY
  if (b->kind) throw std::logic_error("hfg_basis_lm_map: diatomic bases only\n");
  const int cap = *n, cnt = (int)b->b.lm_map.size();
  *n = cnt;
  if (cap < cnt) throw std::logic_error("hfg_basis_lm_map: capacity too small\n");
  for (int i = 0; i < cnt; i++) {
    L[i] = b->b.lm_map[i].first;
    M[i] = b->b.lm_map[i].second;
  }
  HFG_CATCH
}

// One primitive table of TwoDBasis::compute_tei (basis.cpp:1166): which 0-3 prim_tei00/02/20/22, 4-7 prim_ktei**,
// 8-11 disjoint_P0/P2/Q0/Q2, for channel ilm and element iel, column-major, in the reference's (unpadded) shape.
// Tables built by hfg_compute_tei_dev are read back from the device layout (hip/tables.h) and unpadded.
int hfg_basis_get_prim(hfg_ctx *ctx, const hfg_basis *b, int which, int ilm, int iel, double *out, int64_t *rows, int64_t *cols) {
  HFG_TRY
  if (b->kind) {
    // atomic basis: which 0 = prim_tei[L] (TwoDBasis.cpp:666-739), 8 = disjoint_L, 10 = disjoint_m1L; ilm = L
    const helfem::atomic::TwoDBasis &B = b->ab;
    const size_t E = B.Nel(), NL = (size_t)B.N_L();
    if (which != 4 && !(which >= 12 && which <= 15) && ((which != 0 && which != 8 && which != 10) || ilm < 0 || (size_t)ilm >= NL || iel < 0 || (size_t)iel >= E))
      throw std::logic_error("hfg_basis_get_prim: index out of range\n");
    if (which == 4 || (which >= 12 && which <= 15)) {
      // 4 = prim_ktei[L] (host tables); range-separated tables of compute_yukawa / compute_erfc (TwoDBasis.cpp:741-815):
      // 12 = disjoint_iL, 13 = disjoint_kL, 14 = rs_tei, 15 = rs_ktei -- Yukawa: one table per (L, iel); erfc: one per
      // (L, iel, kel), addressed with iel * Nel + kel in place of iel
      const std::vector<helfem::Mat> &t = which == 4 ? B.prim_ktei : which == 12 ? B.disjoint_iL : which == 13 ? B.disjoint_kL : which == 14 ? B.rs_tei : B.rs_ktei;
      const bool pairs = which >= 14 && B.rs_kind == 2;
      if (ilm < 0 || (size_t)ilm >= NL || iel < 0 || (size_t)iel >= (pairs ? E * E : E)) throw std::logic_error("hfg_basis_get_prim: index out of range\n");
      const size_t at = pairs ? (size_t)ilm * E * E + iel : (size_t)ilm * E + iel;
      if (at >= t.size()) throw std::logic_error("hfg_basis_get_prim: this table has not been computed\n");
      const helfem::Mat &m = t[at];
      *rows = (int64_t)m.n_rows;
      *cols = (int64_t)m.n_cols;
      if (out) std::copy(m.d.begin(), m.d.end(), out);
      return 0;
    }
    const size_t idx = (size_t)ilm * E + iel;
    if (which >= 8) {
      if (!B.have_tei && !B.have_disjoint) throw std::logic_error("Primitive teis have not been computed!\n");
      const helfem::Mat &m = (which == 8 ? B.disjoint_L : B.disjoint_m1L)[idx];
      *rows = (int64_t)m.n_rows;
      *cols = (int64_t)m.n_cols;
      if (out) std::copy(m.d.begin(), m.d.end(), out);
    } else if (b->tei_on_device) {
      if (!ctx) throw std::logic_error("hfg_basis_get_prim: the tables live on the device, a context is needed\n");
      const size_t p = B.max_Nprim(), pp = p * p, Ni = B.fem.nprim(iel), Np = Ni * Ni, lo = (iel == 0) ? 1 : 0;
      *rows = *cols = (int64_t)Np;
      if (out) {
        std::vector<double> pad(pp * pp);
        HFG_HIP_CHECK(hipSetDevice(ctx->device));
        HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HFG_HIP_CHECK(hipMemcpy(pad.data(), b->dev_tei.p + ((size_t)ilm * E + iel) * pp * pp, sizeof(double) * pp * pp, hipMemcpyDeviceToHost));
        for (size_t cj = 0; cj < Ni; cj++)
          for (size_t ci = 0; ci < Ni; ci++)
            for (size_t rj = 0; rj < Ni; rj++)
              for (size_t ri = 0; ri < Ni; ri++)
                out[(cj * Ni + ci) * Np + rj * Ni + ri] = pad[((cj + lo) * p + ci + lo) * pp + (rj + lo) * p + ri + lo];
      }
    } else {
      if (!B.have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
      const helfem::Mat &m = B.prim_tei[idx];
      *rows = (int64_t)m.n_rows;
      *cols = (int64_t)m.n_cols;
      if (out) std::copy(m.d.begin(), m.d.end(), out);
    }
    return 0;
  }
  const helfem::diatomic::TwoDBasis &B = b->b;
  const size_t E = B.Nel(), Nlm = B.lm_map.size();
  if (which < 0 || which > 11 || ilm < 0 || (size_t)ilm >= Nlm || iel < 0 || (size_t)iel >= E)
    throw std::logic_error("hfg_basis_get_prim: index out of range\n");
  const size_t idx = (size_t)ilm * E + iel;
  if (which >= 8) {
    if (!B.have_tei && !B.have_disjoint) throw std::logic_error("Primitive teis have not been computed!\n");
    const std::vector<helfem::Mat> *t[4] = {&B.disjoint_P0, &B.disjoint_P2, &B.disjoint_Q0, &B.disjoint_Q2};
    const helfem::Mat &m = (*t[which - 8])[idx];
    *rows = (int64_t)m.n_rows;
    *cols = (int64_t)m.n_cols;
    if (out) std::copy(m.d.begin(), m.d.end(), out);
  } else if (b->tei_on_device) {
    if (!ctx) throw std::logic_error("hfg_basis_get_prim: the tables live on the device, a context is needed\n");
    if (which >= 4) throw std::logic_error("hfg_basis_get_prim: exchange-ordered tables are formed inside the exchange kernels\n");
    const size_t p = B.max_Nprim(), pp = p * p, Ni = B.fem.nprim(iel), Np = Ni * Ni;
    *rows = *cols = (int64_t)Np;
    if (out) {
      std::vector<double> pad(pp * pp);
      HFG_HIP_CHECK(hipSetDevice(ctx->device));
      HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      HFG_HIP_CHECK(hipMemcpy(pad.data(), b->dev_tei.p + (((size_t)which * Nlm + ilm) * E + iel) * pp * pp, sizeof(double) * pp * pp,
                              hipMemcpyDeviceToHost));
      for (size_t cj = 0; cj < Ni; cj++)
        for (size_t ci = 0; ci < Ni; ci++)
          for (size_t rj = 0; rj < Ni; rj++)
            for (size_t ri = 0; ri < Ni; ri++) out[(cj * Ni + ci) * Np + rj * Ni + ri] = pad[(cj * p + ci) * pp + rj * p + ri];
    }
  } else {
    if (!B.have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
    if (which >= 4 && !B.have_ktei) throw std::logic_error("Primitive exchange teis have not been computed!\n");
    const std::vector<helfem::Mat> *t[8] = {&B.prim_tei00,  &B.prim_tei02,  &B.prim_tei20,  &B.prim_tei22,
                                            &B.prim_ktei00, &B.prim_ktei02, &B.prim_ktei20, &B.prim_ktei22};
    const helfem::Mat &m = (*t[which])[idx];
    *rows = (int64_t)m.n_rows;
    *cols = (int64_t)m.n_cols;
    if (out) std::copy(m.d.begin(), m.d.end(), out);
  }
  HFG_CATCH
}

// One evaluation of the reference's ADIIS + CDIIS weights (DIIS::get_w / solve_F, src/general/diis.cpp:214-290, 392-412)
// for a history of n entries, oldest first: B (n x n, err_i . err_j), T (n x n, T(i,j) = Tr P_i F_j, row-major), the
// energies E and the maximum absolute error of the newest entry.  mode 0 = mixed (the drivers' setting), 1 = CDIIS only,
// 2 = ADIIS only.  w receives n weights, the first *dropped of them zero (entries the extrapolation dropped).
int hfg_diis_weights(int n, const double *B, const double *T, const double *E, double maxerr, double diiseps, double diisthr,
                     int mode, double *w, int *dropped) {
  HFG_TRY
  if (n < 1) throw std::logic_error("hfg_diis_weights: empty history\n");
  helfem::DiisMixer mix(mode != 2, diiseps, diisthr, mode != 1, false, (size_t)n);
  for (int i = 0; i < n; i++) {
    mix.push(E[i], maxerr);
    for (int j = 0; j <= i; j++) {
      mix.set_B((size_t)i, (size_t)j, B[i * n + j]);
      mix.set_T((size_t)i, (size_t)j, T[i * n + j]);
      mix.set_T((size_t)j, (size_t)i, T[j * n + i]);
    }
  }
  size_t dr = 0;
  std::vector<double> sol = mix.solve(dr);
  for (int i = 0; i < n; i++) w[i] = 0.0;
  for (size_t i = 0; i < sol.size(); i++) w[dr + i] = sol[i];
  if (dropped) *dropped = (int)dr;
  HFG_CATCH
}

// ---- checkpoint files ------------------------------------------------------------------------------------------------
struct hfg_chk {
  helfem::Checkpoint c;
  hfg_chk(const char *path, bool write) : c(path, write) {}
};
int hfg_chk_available(void) { return helfem::hdf5_available() ? 1 : 0; }
int hfg_chk_open(const char *path, int write, hfg_chk **chk) {
  HFG_TRY
  *chk = new hfg_chk(path, write != 0);
  HFG_CATCH
}
int hfg_chk_close(hfg_chk *chk) {
  delete chk;
  return 0;
}
int hfg_chk_exist(hfg_chk *chk, const char *name) {
  try {
    return chk->c.exist(name) ? 1 : 0;
  } catch (...) {
    return 0;
  }
}
int hfg_chk_write_mat(hfg_chk *chk, const char *name, const double *m, int64_t rows, int64_t cols) {
  HFG_TRY
  helfem::Mat M((size_t)rows, (size_t)cols);
  if (rows * cols) memcpy(M.memptr(), m, sizeof(double) * rows * cols);
  chk->c.write(name, M);
  HFG_CATCH
}
int hfg_chk_write_ivec(hfg_chk *chk, const char *name, const int *v, int64_t n) {
  HFG_TRY
  chk->c.write(name, helfem::IVec(v, v + n));
  HFG_CATCH
}
int hfg_chk_write_double(hfg_chk *chk, const char *name, double v) {
  HFG_TRY
  chk->c.write(name, v);
  HFG_CATCH
}
int hfg_chk_write_int(hfg_chk *chk, const char *name, int v) {
  HFG_TRY
  chk->c.write(name, v);
  HFG_CATCH
}
int hfg_chk_write_basis(hfg_chk *chk, const hfg_basis *b) {
  HFG_TRY
  if (b->kind) chk->c.write(b->ab);
  else chk->c.write(b->b);
  HFG_CATCH
}
int hfg_chk_read_mat(hfg_chk *chk, const char *name, double *m, int64_t *rows, int64_t *cols) {
  HFG_TRY
  if (!m) {
    std::vector<long long> d = chk->c.dims(name);
    if (d.size() != 2) throw std::runtime_error(std::string("Error - ") + name + " should have dimension 2.\n");
    *rows = d[1];  // stored with swapped dimensions
    *cols = d[0];
  } else {
    helfem::Mat M;
    chk->c.read(name, M);
    *rows = (int64_t)M.n_rows;
    *cols = (int64_t)M.n_cols;
    if (M.n_elem()) memcpy(m, M.memptr(), sizeof(double) * M.n_elem());
  }
  HFG_CATCH
}
int hfg_chk_read_ivec(hfg_chk *chk, const char *name, int *v, int64_t *n) {
  HFG_TRY
  helfem::IVec iv;
  chk->c.read(name, iv);
  if (v) {
    if (*n < (int64_t)iv.size()) throw std::logic_error("hfg_chk_read_ivec: capacity too small\n");
    std::copy(iv.begin(), iv.end(), v);
  }
  *n = (int64_t)iv.size();
  HFG_CATCH
}
int hfg_chk_read_double(hfg_chk *chk, const char *name, double *v) {
  HFG_TRY
  chk->c.read(name, *v);
  HFG_CATCH
}
int hfg_chk_read_int(hfg_chk *chk, const char *name, int *v) {
  HFG_TRY
  chk->c.read(name, *v);
  HFG_CATCH
}
int hfg_chk_read_diatomic_basis(hfg_chk *chk, int lpad, hfg_basis **out) {
  HFG_TRY
  hfg_basis *b = new hfg_basis();
  try {
    b->kind = 0;
    b->b = chk->c.read_diatomic_basis(lpad);
  } catch (...) {
    delete b;
    throw;
  }
  *out = b;
  HFG_CATCH
}

int hfg_radial_grid(double mumax, int nelem, int igrid, double zexp, double *bval) {
  HFG_TRY
  helfem::Vec g = helfem::get_grid(mumax, nelem, igrid, zexp);
  memcpy(bval, g.data(), sizeof(double) * g.size());
  HFG_CATCH
}

int hfg_lm_list(const int *lmmax, int nlm, int *lval, int *mval, int *nang) {
  HFG_TRY
  helfem::IVec l, m;
  helfem::diatomic::lm_to_l_m(helfem::IVec(lmmax, lmmax + nlm), l, m);
  if ((int)l.size() > *nang) {
    *nang = (int)l.size();
    throw std::logic_error("hfg_lm_list: output capacity too small");
  }
  *nang = (int)l.size();
  for (size_t i = 0; i < l.size(); i++) {
    lval[i] = l[i];
    mval[i] = m[i];
  }
  HFG_CATCH
}

double hfg_gaunt_coefficient(int L, int M, int l, int m, int lp, int mp) {
  return helfem::gaunt_coefficient(L, M, l, m, lp, mp);
}
double hfg_modified_gaunt_coefficient(int lj, int mj, int L, int M, int li, int mi) {
  helfem::Gaunt g;
  return g.mod_coeff(lj, mj, L, M, li, mi);
}
void hfg_legendre_PQ(int Lmax, int Mmax, double xi, double *P, double *Q) { helfem::legendre_PQ(Lmax, Mmax, xi, P, Q); }
double hfg_theta_lm(int l, int m, double cth) { return helfem::theta_lm(l, m, cth); }
double hfg_bessel_il(double x, int L) { return helfem::bessel_il(x, L); }
double hfg_bessel_kl(double x, int L) { return helfem::bessel_kl(x, L); }
double hfg_erfc_phi(int n, double Xi, double xi) { return helfem::erfc_Phi(n, Xi, xi); }
void hfg_chebyshev_rule(int n, double *x, double *w) {
  helfem::Vec xv, wv;
  helfem::chebyshev_rule(n, xv, wv);
  memcpy(x, xv.data(), sizeof(double) * n);
  memcpy(w, wv.data(), sizeof(double) * n);
}
void hfg_lobatto_nodes(int n, double *x) {
  helfem::Vec xv = helfem::lobatto_nodes(n);
  memcpy(x, xv.data(), sizeof(double) * n);
}

int hfg_basis_upload(hfg_ctx *ctx, hfg_basis *b, int ldft, int mdft) {
  HFG_TRY
  if (b->dev) {
    fock_release(b->dev);
    exchange_release(b->dev);
    exchange_lr_release(b->dev);
  }
  upload_tables(ctx, b, ldft, mdft);
  if (b->dev_rs) {
    exchange_release(b->dev_rs);
    exchange_lr_release(b->dev_rs);
    delete b->dev_rs;
    b->dev_rs = nullptr;
  }
  if (b->kind == 1 && b->ab.rs_kind) upload_rs_tables(ctx, b);
  HFG_CATCH
}

// ---- device-pointer API --------------------------------------------------------------------------
int hfg_coulomb_dev(hfg_ctx *ctx, hfg_basis *b, const double *dP, double *dJ) {
  HFG_TRY coulomb_dev(ctx, b, dP, dJ);
  HFG_CATCH
}
int hfg_exchange_dev(hfg_ctx *ctx, hfg_basis *b, const double *dP, double *dK) {
  HFG_TRY exchange_dev(ctx, b, dP, dK);
  HFG_CATCH
}
int hfg_exchange_occ_dev(hfg_ctx *ctx, hfg_basis *b, const double *dP, const double *dC, int64_t nocc, double *dK) {
  HFG_TRY
  if (nocc < 1 || !dC) throw std::logic_error("hfg_exchange_occ_dev: occupied orbitals missing\n");
  exchange_dev(ctx, b, dP, dK, false, dC, (int)nocc);
  HFG_CATCH
}
int hfg_rs_exchange_dev(hfg_ctx *ctx, hfg_basis *b, const double *dP, double *dK) {
  HFG_TRY exchange_dev(ctx, b, dP, dK, true);
  HFG_CATCH
}
int hfg_xc_fock_dev(hfg_ctx *ctx, hfg_basis *b, int x_func, int c_func, const double *dP, double *dH, double *dScal,
                    double thr) {
  HFG_TRY xc_fock_dev(ctx, b, x_func, c_func, dP, dH, dScal, thr);
  HFG_CATCH
}
int hfg_xc_fock_pol_dev(hfg_ctx *ctx, hfg_basis *b, int x_func, int c_func, const double *dPa, const double *dPb,
                        double *dHa, double *dHb, double *dScal, double thr) {
  HFG_TRY xc_fock_pol_dev(ctx, b, x_func, c_func, dPa, dPb, dHa, dHb, dScal, thr);
  HFG_CATCH
}
int hfg_eig_gsym_sub_dev(hfg_ctx *ctx, int64_t N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                         const int64_t *blk_idx, double *dE, double *dC) {
  HFG_TRY eig_gsym_sub_dev(ctx, (int)N, dF, dS, nblk, blk_ptr, blk_idx, dE, dC);
  HFG_CATCH
}
int64_t hfg_eig_block_buf_size(int nblk, const int64_t *blk_ptr) { return (int64_t)eig_block_buf_size(nblk, blk_ptr); }
int hfg_eig_blocks_dev(hfg_ctx *ctx, int64_t N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                       const int64_t *blk_idx, double *dBlockBuf) {
  HFG_TRY eig_blocks_dev(ctx, (int)N, dF, dS, nblk, blk_ptr, blk_idx, dBlockBuf);
  HFG_CATCH
}
int hfg_eig_assemble_dev(hfg_ctx *ctx, int64_t N, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                         const double *dBlockBuf, double *dE, double *dC) {
  HFG_TRY eig_assemble_dev(ctx, (int)N, nblk, blk_ptr, blk_idx, dBlockBuf, dE, dC);
  HFG_CATCH
}
int hfg_form_density_dev(hfg_ctx *ctx, int64_t N, int64_t ncols, const double *dC, int64_t nocc, double *dP) {
  HFG_TRY form_density_dev(ctx, (int)N, (int)ncols, dC, (int)nocc, dP);
  HFG_CATCH
}
int hfg_gemm_dev(hfg_ctx *ctx, int tA, int tB, int64_t m, int64_t n, int64_t k, const double *dA, int64_t lda,
                 const double *dB, int64_t ldb, double *dC, int64_t ldc) {
  HFG_TRY gemm_dev(ctx, tA != 0, tB != 0, (int)m, (int)n, (int)k, 1.0, dA, (int)lda, dB, (int)ldb, 0.0, dC, (int)ldc);
  HFG_CATCH
}

int64_t hfg_fock_compact_size(hfg_basis *b) { return (int64_t)fock_compact_size(b); }
int hfg_fock_compact_dev(hfg_ctx *ctx, hfg_basis *b, int x_func, int c_func, const double *dP, double *dFc,
                         double *dScal, double thr) {
  HFG_TRY fock_compact_dev(ctx, b, x_func, c_func, dP, dFc, dScal, thr);
  HFG_CATCH
}
int hfg_fock_finish_dev(hfg_ctx *ctx, hfg_basis *b, const double *dFc, const double *dH0, const int *dBlockId,
                        double *dF) {
  HFG_TRY fock_finish_dev(ctx, b, dFc, dH0, dBlockId, dF);
  HFG_CATCH
}

// ---- host-pointer API ------------------------------------------------------------------------------
// The host-pointer entry points return COMPLETE matrices (they are what an arma::mat caller binds):
// a shard set on the context for the device-resident multi-GPU step (hfg_ctx_set_shard) must not leak into them, so
// they run with the shard (0, 1) and restore the caller's setting afterwards.  Partial (sharded) results are available
// from the *_dev entry points only.
namespace {
struct FullShard {
  hfg_ctx *c;
  int r, n;
  explicit FullShard(hfg_ctx *ctx) : c(ctx), r(ctx->shard_rank), n(ctx->shard_n) {
    c->shard_rank = 0;
    c->shard_n = 1;
  }
  ~FullShard() {
    c->shard_rank = r;
    c->shard_n = n;
  }
};
}  // namespace
int hfg_coulomb(hfg_ctx *ctx, hfg_basis *b, const double *P, double *J) {
  HFG_TRY
  FullShard full(ctx);
  size_t N = b->Nbf();
  Stage st(ctx);
  double *dP = st.up(P, N * N), *dJ = st.alloc(N * N);
  coulomb_dev(ctx, b, dP, dJ);
  st.down(J, dJ, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_exchange(hfg_ctx *ctx, hfg_basis *b, const double *P, double *K) {
  HFG_TRY
  FullShard full(ctx);
  size_t N = b->Nbf();
  Stage st(ctx);
  double *dP = st.up(P, N * N), *dK = st.alloc(N * N);
  exchange_dev(ctx, b, dP, dK);
  st.down(K, dK, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_rs_exchange(hfg_ctx *ctx, hfg_basis *b, const double *P, double *K) {
  HFG_TRY
  FullShard full(ctx);
  size_t N = b->Nbf();
  Stage st(ctx);
  double *dP = st.up(P, N * N), *dK = st.alloc(N * N);
  exchange_dev(ctx, b, dP, dK, true);
  st.down(K, dK, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_model_potential(hfg_ctx *ctx, hfg_basis *b, const hfg_model_pot *p1, const hfg_model_pot *p2, double *H) {
  HFG_TRY
  FullShard full(ctx);
  if (!p1) throw std::logic_error("hfg_model_potential: no potential given\n");
  size_t N = b->Nbf();
  if (b->kind == 1) {  // atomic: radial integrals on the host, as the reference does
    helfem::ModelPotential mp;
    mp.kind = p1->kind;
    mp.Z = p1->Z;
    mp.d = p1->d;
    mp.H = p1->H;
    helfem::Mat V = b->ab.model_potential(mp);
    memcpy(H, V.memptr(), sizeof(double) * N * N);
  } else {
    if (!p2) throw std::logic_error("hfg_model_potential: the diatomic basis needs both centres\n");
    Stage st(ctx);
    double *dH = st.alloc(N * N);
    model_potential_dev(ctx, b, p1->kind, p1->Z, p1->d, p1->H, p2->kind, p2->Z, p2->d, p2->H, dH);
    st.down(H, dH, N * N);
    st.sync();
  }
  HFG_CATCH
}
int hfg_xc_fock(hfg_ctx *ctx, hfg_basis *b, int x_func, int c_func, const double *P, double *H, double *Exc,
                double *Nel, double *Ekin, double thr) {
  HFG_TRY
  FullShard full(ctx);
  size_t N = b->Nbf();
  Stage st(ctx);
  double *dP = st.up(P, N * N), *dH = st.alloc(N * N), *dS = st.alloc(3);
  xc_fock_dev(ctx, b, x_func, c_func, dP, dH, dS, thr);
  double sc[3];
  st.down(H, dH, N * N);
  st.down(sc, dS, 3);
  st.sync();
  *Exc = sc[0];
  *Nel = sc[1];
  *Ekin = sc[2];
  HFG_CATCH
}
int hfg_xc_fock_pol(hfg_ctx *ctx, hfg_basis *b, int x_func, int c_func, const double *Pa, const double *Pb, double *Ha,
                    double *Hb, double *Exc, double *Nel, double *Ekin, double thr) {
  HFG_TRY
  FullShard full(ctx);
  size_t N = b->Nbf();
  Stage st(ctx);
  double *dPa = st.up(Pa, N * N), *dPb = st.up(Pb, N * N);
  double *dHa = st.alloc(N * N), *dHb = st.alloc(N * N), *dS = st.alloc(3);
  xc_fock_pol_dev(ctx, b, x_func, c_func, dPa, dPb, dHa, dHb, dS, thr);
  double sc[3];
  st.down(Ha, dHa, N * N);
  st.down(Hb, dHb, N * N);
  st.down(sc, dS, 3);
  st.sync();
  *Exc = sc[0];
  *Nel = sc[1];
  *Ekin = sc[2];
  HFG_CATCH
}
int hfg_xc_fock_ext(hfg_ctx *ctx, hfg_basis *b, int x_func, const double *x_pars, int n_x_pars, int c_func, const double *c_pars,
                    int n_c_pars, const double *P, double *H, double *Exc, double *Nel, double *Ekin, double thr) {
  HFG_TRY
  struct Reset {  // the defaults come back whatever happens
    hfg_ctx *c;
    ~Reset() {
      try {
        set_xc_params(c, 0, nullptr, 0, 0, nullptr, 0);
      } catch (...) {
      }
    }
  } reset{ctx};
  set_xc_params(ctx, x_func, x_pars, n_x_pars, c_func, c_pars, n_c_pars);
  int rc = hfg_xc_fock(ctx, b, x_func, c_func, P, H, Exc, Nel, Ekin, thr);
  if (rc) return rc;
  HFG_CATCH
}
int hfg_xc_fock_pol_ext(hfg_ctx *ctx, hfg_basis *b, int x_func, const double *x_pars, int n_x_pars, int c_func,
                        const double *c_pars, int n_c_pars, const double *Pa, const double *Pb, double *Ha, double *Hb, double *Exc,
                        double *Nel, double *Ekin, double thr) {
  HFG_TRY
  struct Reset {
    hfg_ctx *c;
    ~Reset() {
      try {
        set_xc_params(c, 0, nullptr, 0, 0, nullptr, 0);
      } catch (...) {
      }
    }
  } reset{ctx};
  set_xc_params(ctx, x_func, x_pars, n_x_pars, c_func, c_pars, n_c_pars);
  int rc = hfg_xc_fock_pol(ctx, b, x_func, c_func, Pa, Pb, Ha, Hb, Exc, Nel, Ekin, thr);
  if (rc) return rc;
  HFG_CATCH
}
int hfg_eig_sym(hfg_ctx *ctx, int64_t n, const double *A, double *E, double *C) {
  HFG_TRY
  Stage st(ctx);
  double *dA = st.up(A, n * n), *dE = st.alloc(n), *dC = st.alloc(n * n);
  eig_sym_dev(ctx, (int)n, dA, dE, dC);
  st.down(E, dE, n);
  st.down(C, dC, n * n);
  st.sync();
  HFG_CATCH
}
int hfg_eig_gsym(hfg_ctx *ctx, int64_t N, int64_t n, const double *F, const double *S, double *E, double *C) {
  HFG_TRY
  Stage st(ctx);
  double *dF = st.up(F, N * N), *dS = st.up(S, N * n), *dE = st.alloc(n), *dC = st.alloc(N * n);
  eig_gsym_dev(ctx, (int)N, (int)n, dF, dS, dE, dC);
  st.down(E, dE, n);
  st.down(C, dC, N * n);
  st.sync();
  HFG_CATCH
}
int hfg_eig_gsym_sub(hfg_ctx *ctx, int64_t N, const double *F, const double *S, int nblk, const int64_t *blk_ptr,
                     const int64_t *blk_idx, double *E, double *C) {
  HFG_TRY
  Stage st(ctx);
  double *dF = st.up(F, N * N), *dS = st.up(S, N * N), *dE = st.alloc(N), *dC = st.alloc(N * N);
  eig_gsym_sub_dev(ctx, (int)N, dF, dS, nblk, blk_ptr, blk_idx, dE, dC);
  st.down(E, dE, N);
  st.down(C, dC, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_eig_gsym_sub_pair(hfg_ctx *ctx, int64_t N, const double *Fa, const double *Fb, const double *S, int nblk, const int64_t *blk_ptr,
                          const int64_t *blk_idx, double *Ea, double *Ca, double *Eb, double *Cb) {
  HFG_TRY
  Stage st(ctx);
  double *dFa = st.up(Fa, N * N), *dFb = st.up(Fb, N * N), *dS = st.up(S, N * N);
  double *dEa = st.alloc(N), *dCa = st.alloc(N * N), *dEb = st.alloc(N), *dCb = st.alloc(N * N);
  eig_gsym_sub_pair_dev(ctx, (int)N, dFa, dFb, dS, nblk, blk_ptr, blk_idx, dEa, dCa, dEb, dCb);
  st.down(Ea, dEa, N);
  st.down(Ca, dCa, N * N);
  st.down(Eb, dEb, N);
  st.down(Cb, dCb, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_form_sinvh(hfg_ctx *ctx, int64_t N, const double *S, int chol, int nblk, const int64_t *blk_ptr,
                   const int64_t *blk_idx, double *Sinvh) {
  HFG_TRY
  Stage st(ctx);
  double *dS = st.up(S, N * N), *dX = st.alloc(N * N);
  form_sinvh_dev(ctx, (int)N, dS, chol != 0, nblk, blk_ptr, blk_idx, dX);
  st.down(Sinvh, dX, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_form_density(hfg_ctx *ctx, int64_t N, int64_t ncols, const double *C, int64_t nocc, double *P) {
  HFG_TRY
  Stage st(ctx);
  double *dC = st.up(C, N * ncols), *dP = st.alloc(N * N);
  form_density_dev(ctx, (int)N, (int)ncols, dC, (int)nocc, dP);
  st.down(P, dP, N * N);
  st.sync();
  HFG_CATCH
}
int hfg_gemm(hfg_ctx *ctx, int tA, int tB, int64_t m, int64_t n, int64_t k, const double *A, int64_t lda,
             const double *B, int64_t ldb, double *C, int64_t ldc) {
  HFG_TRY
  Stage st(ctx);
  size_t na = (size_t)lda * (tA ? m : k), nb = (size_t)ldb * (tB ? k : n);
  double *dA = st.up(A, na), *dB = st.up(B, nb), *dC = st.alloc((size_t)ldc * n);
  gemm_dev(ctx, tA != 0, tB != 0, (int)m, (int)n, (int)k, 1.0, dA, (int)lda, dB, (int)ldb, 0.0, dC, (int)ldc);
  st.down(C, dC, (size_t)ldc * n);
  st.sync();
  HFG_CATCH
}

// ---- measurement -----------------------------------------------------------------------------------
int hfg_profile_enable(hfg_ctx *ctx, int on) {
  ctx->profiling = on != 0;
  return 0;
}
int hfg_profile_reset(hfg_ctx *ctx) {
  HFG_TRY
  ctx->prof_collect();
  for (auto &kv : ctx->prof) {
    kv.second.ms = 0.0;
    kv.second.launches = 0;
  }
  HFG_CATCH
}
int hfg_profile_get(hfg_ctx *ctx, const char *name, double *ms, int64_t *launches) {
  HFG_TRY
  ctx->prof_collect();
  auto it = ctx->prof.find(name);
  if (it == ctx->prof.end()) {
    *ms = 0.0;
    *launches = 0;
  } else {
    *ms = it->second.ms;
    *launches = it->second.launches;
  }
  HFG_CATCH
}

int hfg_profile_names(hfg_ctx *ctx, char *buf, size_t cap) {
  HFG_TRY
  ctx->prof_collect();
  std::string all;
  for (const auto &kv : ctx->prof) {
    if (!all.empty()) all += "\n";
    all += kv.first;
  }
  if (cap == 0 || all.size() + 1 > cap) throw std::logic_error("hfg_profile_names: buffer too small");
  memcpy(buf, all.c_str(), all.size() + 1);
  HFG_CATCH
}

int hfg_measure_kernel(hfg_ctx *ctx, const char *name, double *ms, int64_t *launches) {
  HFG_TRY
  if (std::string(name) == "k_trdb_gemv" || std::string(name) == "k_trdf") trd_measure_gemv(ctx, ms, launches);
  else throw std::logic_error("hfg_measure_kernel: unknown kernel");
  HFG_CATCH
}

int hfg_pin(void *p, size_t bytes) {
  HFG_TRY HFG_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
  HFG_CATCH
}
int hfg_unpin(void *p) {
  HFG_TRY HFG_HIP_CHECK(hipHostUnregister(p));
  HFG_CATCH
}

}  // extern "C"
