// Shared declarations of the HIP side: context, error handling, device buffers, profiling.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/helfem_gpu.h"
#include "../host/atomic_basis.h"
#include "../host/diatomic_basis.h"

namespace hfg {

void set_error(const std::string &msg);

#define HFG_HIP_CHECK(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " in " + #expr + \
                               " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")");              \
  } while (0)

// RAII device buffer
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void resize(size_t count) {
    if (count <= n) return;
    release();
    HFG_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void upload(const std::vector<T> &h, hipStream_t s) {
    resize(h.size() ? h.size() : 1);
    if (h.size()) HFG_HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

// Uploads h into buf unless exactly these bytes are already there: the launch descriptors and GEMM task lists of the
// eigensolver are identical from one SCF iteration to the next, and each "upload + stream synchronisation" (the source
// used to live on the caller's stack) stalled the pipeline for 50-100 us.  The host copy `cache` is the source of the
// asynchronous copy, so nothing has to be waited for after queuing it; before it is overwritten the stream is drained.
template <typename T>
bool upload_cached(DevBuf<T> &buf, std::vector<T> &cache, const std::vector<T> &h, hipStream_t s) {
  if (buf.p && cache.size() == h.size() && buf.n >= h.size() &&
      (h.empty() || memcmp((const void *)cache.data(), (const void *)h.data(), h.size() * sizeof(T)) == 0))
    return false;
  HFG_HIP_CHECK(hipStreamSynchronize(s));  // an earlier copy may still be reading the old contents of `cache`
  cache = h;
  buf.upload(cache, s);
  return true;
}

struct ProfEntry {
  double ms = 0.0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t> > pending;
};

}  // namespace hfg

// Flattened device-side description of a diatomic basis (see tables.cpp for the layouts)
struct hfg_dev_tables;

struct hfg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // second, non-blocking stream for work that is independent of what runs on `stream` (the compact-WY set-up of the
  // back-transformation beside the divide-and-conquer stage); ordered with `stream` through the two events
  hipStream_t side_stream = nullptr;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
  hipStream_t side();
  // A second live stream costs every launch of this context about 1.2 us (measured: the launch chain of the
  // tridiagonalisation 12.05 -> 13.3 ms at 1380/1470/1380, 67.9 -> 72.9 ms for one 4230-matrix).  A batch that takes the
  // chain for a large matrix therefore gives the side stream up for good: drop_side() destroys it, avoid_side keeps the
  // Fock build and the eigensolve from creating it again (their overlaps are worth less than the chain loses).
  bool avoid_side = false;
  void drop_side();
  int shard_rank = 0, shard_n = 1;
  // hfg_ctx_fix_sinvh: the caller's promise that the device matrix at this address keeps its contents (S^{-1/2} of an
  // SCF run); eig_blocks_dev then derives the blocks' column supports from it once instead of in every iteration
  const double *fixed_sinvh = nullptr;
  unsigned long fixed_gen = 0;  // bumped by every declaration: a new matrix at a recycled address is a new matrix
  void fix_sinvh(const double *p) {
    fixed_sinvh = p;
    fixed_gen++;
  }
  bool profiling = false;
  std::map<std::string, hfg::ProfEntry> prof;
  std::vector<hipEvent_t> event_pool;
  // scratch
  hfg::DevBuf<double> ws[8];
  hfg::DevBuf<double> hstage;  // unused placeholder
  double *pinned = nullptr;
  size_t pinned_bytes = 0;

  void *pinned_buf(size_t bytes);
  hipEvent_t get_event();
  void prof_begin(const char *name);
  void prof_end(const char *name);
  void prof_collect();
};

// Either program's basis behind one handle: kind 0 = diatomic (prolate spheroidal), 1 = atomic (spherical)
struct hfg_basis {
  int kind = 0;
  helfem::diatomic::TwoDBasis b;
  helfem::atomic::TwoDBasis ab;
  hfg_dev_tables *dev = nullptr;
  hfg_dev_tables *dev_rs = nullptr;  // range-separated exchange kernel (atomic), see tables.h
  int dev_device = -1;
  // primitive in-element integral tables built on the device (hip/tei_dev.hip) instead of host Mats
  hfg::DevBuf<double> dev_tei;
  bool tei_on_device = false;

  size_t Nbf() const { return kind ? ab.Nbf() : b.Nbf(); }
  size_t Ndummy() const { return kind ? ab.Nbf() : b.Ndummy(); }
  size_t Nrad() const { return kind ? ab.Nrad() : b.Nrad(); }
  size_t Nang() const { return kind ? ab.Nang() : b.Nang(); }
  size_t Nel() const { return kind ? ab.Nel() : b.Nel(); }
  size_t max_Nprim() const { return kind ? ab.max_Nprim() : b.max_Nprim(); }
  bool have_tei() const { return tei_on_device || (kind ? ab.have_tei : b.have_tei); }
};

namespace hfg {
// one product C = A B (column-major) of a device-side task list (k_dgemm_tasks, dc.hip)
struct GemmTask {
  const double *A;
  const double *B;
  double *C;
  int M, N, K, lda, ldb, ldc;
  int tA = 0, tB = 0;  // op(A), op(B) transposed (k_dgemm_tasklist only)
  double alpha = 1.0, beta = 0.0;  // C = alpha op(A) op(B) + beta C (k_dgemm_tasklist only)
  int sym = 0;  // the product is known to be symmetric (M == N, e.g. X^T (F X)): only the tiles on and below the diagonal
                // are enumerated and computed (k_dgemm_tasklist with beta == 0), k_mirror_lower fills the rest.
                // 2 (accumulating task lists): only the lower triangle of C and a band of three tile rows above the
                // diagonal are read afterwards, the other tiles are skipped
  int over = 0;  // over-read permissions (k_dgemm_tasklist): bit 0 -- the rows of op(A) from M up to the edge of the last
                 // 128-row tile are readable memory (their products land in rows of C that are not stored), bit 1 -- likewise
                 // the columns of op(B) from N up to the tile edge; lets partial edge tiles use the 16-byte loads.
                 // (Also fills the struct: task lists are compared bytewise, upload_cached.)
};
static_assert(sizeof(GemmTask) == 80, "GemmTask is compared bytewise: keep it free of padding");

struct ProfScope {
  hfg_ctx *c;
  const char *n;
  ProfScope(hfg_ctx *ctx, const char *name) : c(ctx), n(name) {
    if (c->profiling) c->prof_begin(n);
  }
  ~ProfScope() {
    if (c->profiling) c->prof_end(n);
  }
};
}  // namespace hfg
