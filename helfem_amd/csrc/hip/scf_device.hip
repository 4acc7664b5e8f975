// Device-resident SCF loop: the per-iteration part of the reference drivers (src/diatomic/main.cpp:780-995,
// src/atomic/main.cpp:760-1005) with every matrix kept in HBM between the steps -- density, Fock build, DIIS
// (uDIIS::update / solve_F, src/general/diis.cpp:129-168, 392-412), generalized eigensolve.  Only a handful of
// scalars (energies, the DIIS error norm and the small B matrix) cross PCIe per iteration.  The host-pointer
// driver (host/scf.cpp with GPUBackend, scf_gpu.cpp) computes the same numbers and is kept as the checker of
// this one (HELFEM_SCF=host).
#include "tables.h"
#include "wave.h"
#include "../host/dftfuncs.h"
#include "../host/diis.h"
#include "../host/scf.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>

namespace hfg {

void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);
void coulomb_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dJ);
void exchange_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dK, bool rs = false, const double *Lknown = nullptr,
                  int rknown = 0);
void xc_fock_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dH, double *dScal,
                 double thr);
void xc_fock_pol_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dPa, const double *dPb,
                     double *dHa, double *dHb, double *dScal, double thr);
void form_sinvh_dev(hfg_ctx *ctx, int N, const double *dS, bool chol, int nblk, const int64_t *blk_ptr,
                    const int64_t *blk_idx, double *dSinvh);
void form_density_dev(hfg_ctx *ctx, int N, int ncols, const double *dC, int nocc, double *dP);
void eig_sym_dev(hfg_ctx *ctx, int n, const double *dA, double *dE, double *dC);
void eig_block_supports(hfg_ctx *ctx, int N, const double *dS, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                        std::vector<int64_t> &cols);
void gemm_tasklist_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void eig_gsym_sub_pair_dev(hfg_ctx *ctx, int N, const double *dFa, const double *dFb, const double *dS, int nblk, const int64_t *blk_ptr,
                           const int64_t *blk_idx, double *dEa, double *dCa, double *dEb, double *dCb);
void eig_gsym_sub_dev(hfg_ctx *ctx, int N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                      const int64_t *blk_idx, double *dE, double *dC);
void upload_tables(hfg_ctx *ctx, hfg_basis *basis, int ldft, int mdft);
void upload_rs_tables(hfg_ctx *ctx, hfg_basis *basis);
void model_potential_dev(hfg_ctx *ctx, hfg_basis *basis, int kind1, int Z1, double d1, double H1, int kind2, int Z2,
                         double d2, double H2, double *dH);
void compute_tei_dev(hfg_ctx *ctx, hfg_basis *basis);
void fock_release(hfg_dev_tables *t);
void exchange_release(hfg_dev_tables *t);
void exchange_lr_release(hfg_dev_tables *t);

namespace {

// y = a x + b y
__global__ void k_axpby(size_t n, double a, const double *__restrict__ x, double b, double *__restrict__ y) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) y[i] = a * x[i] + (b == 0.0 ? 0.0 : b * y[i]);
}
// y = sum_k c[k] x_k, the terms added in the order k = 0, 1, ... exactly as a chain of k_axpby launches adds them
// (one pass: every x_k is read once and y written once instead of read and written per term)
struct LinComb {
  const double *x[16];
  double c[16];
  int n;
};
__global__ void k_lincomb(size_t n, LinComb lc, double *__restrict__ y) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    double acc = lc.c[0] * lc.x[0][i] + 0.0;
    for (int k = 1; k < lc.n; k++) acc = lc.c[k] * lc.x[k][i] + 1.0 * acc;
    y[i] = acc;
  }
}
// E = X - X^T
__global__ void k_antisym(const double *__restrict__ X, int N, double *__restrict__ E) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= N) return;
  E[(size_t)j * N + i] = X[(size_t)j * N + i] - X[(size_t)i * N + j];
}
// out (n x n, column-major) = M(rows, cols) of an N x N matrix
__global__ void k_gather_rc(const double *__restrict__ M, int N, const int64_t *__restrict__ rows, const int64_t *__restrict__ cols, int n,
                            double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i < n) out[(size_t)j * n + i] = M[(size_t)cols[j] * N + rows[i]];
}
// out (n x nc, column-major, ld n) = M(rows, 0:nc) of a matrix with N rows
__global__ void k_gather_rows(const double *__restrict__ M, int N, const int64_t *__restrict__ rows, int n, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i < n) out[(size_t)j * n + i] = M[(size_t)j * N + rows[i]];
}
// E = X - X^T for a block of order n
__global__ void k_antisym_block(const double *__restrict__ X, int n, double *__restrict__ E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i < n) E[(size_t)j * n + i] = X[(size_t)j * n + i] - X[(size_t)i * n + j];
}

// F(i,j) = 0 when i and j belong to different symmetry blocks   (scf::enforce_fock_symmetry, scf_helpers.cpp:249)
__global__ void k_mask_blocks(double *__restrict__ F, int N, const int *__restrict__ blockid) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= N) return;
  if (blockid[i] != blockid[j]) F[(size_t)j * N + i] = 0.0;
}
// scf::fock_symmetry_average (scf_helpers.cpp:263-284) for one group: the diagonal blocks F(idx_c, idx_c), c < ng, are
// replaced by their mean; idx holds the ng index lists of length nn back to back
__global__ void k_fock_average(double *__restrict__ F, int N, const int *__restrict__ idx, int ng, int nn) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= nn) return;
  double m = 0.0;
  for (int c = 0; c < ng; c++) m += F[(size_t)idx[c * nn + j] * N + idx[c * nn + i]];
  m /= (double)ng;
  for (int c = 0; c < ng; c++) F[(size_t)idx[c * nn + j] * N + idx[c * nn + i]] = m;
}
// lambda of scf::ROHF_update in the natural-orbital basis (ascending occupations: virtual 0..Nv-1, core N-Nc..N-1):
// the core-virtual blocks of -Delta, everything else zero
__global__ void k_rohf_lambda(double *__restrict__ D, int N, int Nc, int Nv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= N) return;
  const bool ic = i >= N - Nc, iv = i < Nv, jc = j >= N - Nc, jv = j < Nv;
  const size_t o = (size_t)j * N + i;
  D[o] = ((ic && jv) || (iv && jc)) ? -D[o] : 0.0;
}
// two-stage deterministic reductions: partial[b] = sum / max over the block's grid-stride range
__global__ __launch_bounds__(256) void k_dot_partial(size_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                     double *__restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i] * y[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// K dot products against the same vector y in one pass: partial[k * gridDim.x + block]; per k the same sums in the same
// order as k_dot_partial
struct MultiDot {
  const double *x[16];
  int n;
};
__global__ __launch_bounds__(256) void k_multidot_partial(size_t n, MultiDot md, const double *__restrict__ y, double *__restrict__ partial) {
  __shared__ double sh[16][4];
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double yi = y[i];
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (k < md.n) s[k] += md.x[k][i] * yi;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    double v = s[k];  // (all 16: a loop with an exit is not unrolled and s[] would live in scratch memory)
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < md.n) partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}
// out[k] = sum of the nb partials of dot product k, in order (one thread per k)
__global__ void k_finish_multidot(const double *__restrict__ partial, int nb, int K, double *__restrict__ out) {
  const int k = threadIdx.x;
  if (k >= K || blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < nb; i++) s += partial[(size_t)k * nb + i];
  out[k] = s;
}
__global__ __launch_bounds__(256) void k_maxabs_partial(size_t n, const double *__restrict__ x, double *__restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s = fmax(s, fabs(x[i]));
  for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_down(s, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__global__ void k_finish_reduce(const double *__restrict__ partial, int nb, int is_max, double *__restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < nb; i++) s = is_max ? fmax(s, partial[i]) : s + partial[i];
  *out = s;
}

// f(i, j) *= fac on the occupied-virtual blocks (i < nocc <= j and j < nocc <= i) of an n x n matrix
__global__ void k_scale_offdiag_blocks(double *__restrict__ f, int n, int nocc, double fac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  if ((i < nocc) != (j < nocc)) f[(size_t)j * n + i] *= fac;
}

double wall() {
  using namespace std::chrono;
  return duration_cast<duration<double> >(steady_clock::now().time_since_epoch()).count();
}

constexpr int RED_BLOCKS = 512;

// forced occupations (scf::enforce_occupations, scf_helpers.cpp:61-75): weight of every orbital in one symmetry,
// w[o] = sum_{i in sym} C[i][o] (S C)[i][o]  -- S has no elements between a symmetry and the rest, so the rows of S C
// restricted to the symmetry are S_sub C_sub.  One wave per orbital.
__global__ __launch_bounds__(256) void k_sym_weight(const double *__restrict__ C, const double *__restrict__ SC, int n, const int *__restrict__ rows,
                                                    int nrows, double *__restrict__ w) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= n) return;
  double acc = 0.0;
  for (int k = lane; k < nrows; k += 64) {
    const size_t i = (size_t)o * n + rows[k];
    acc += C[i] * SC[i];
  }
  acc = wave_sum(acc);
  if (lane == 0) w[o] = acc;
}
// Cn[:, o] = C[:, order[o]], En[o] = E[order[o]]
__global__ __launch_bounds__(256) void k_gather_columns(const double *__restrict__ C, const double *__restrict__ E, int n, const int *__restrict__ order,
                                                        double *__restrict__ Cn, double *__restrict__ En) {
  const int o = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int src = order[o];
  if (i < n) Cn[(size_t)o * n + i] = C[(size_t)src * n + i];
  if (i == 0) En[o] = E[src];
}

struct DevSCF {
  hfg_ctx *ctx;
  hipStream_t s;
  size_t N = 0, NN = 0;
  DevBuf<double> S, T, V, H0, Sinvh, Ca, Cb, Ea, Eb, Pa, Pb, P, J, Ka, Kb, XCa, XCb, Fa, Fb, T1, T2, Err, scal, partial, res;
  DevBuf<int> blockid;
  std::unique_ptr<DevBuf<double>[]> histF, histE, histP;  // DIIS history (ring): Fock matrices, errors, densities
  std::vector<double> hres;

  explicit DevSCF(hfg_ctx *c) : ctx(c), s(c->stream) {}

  void up(DevBuf<double> &d, const helfem::Mat &M) {
    d.resize(M.n_elem());
    HFG_HIP_CHECK(hipMemcpyAsync(d.p, M.memptr(), sizeof(double) * M.n_elem(), hipMemcpyHostToDevice, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
  }
  void axpby(double a, const double *x, double b, double *y, size_t n) {
    hipLaunchKernelGGL(k_axpby, dim3(2048), dim3(256), 0, s, n, a, x, b, y);
  }
  // result slot k of the device scalar array <- x . y
  void dot(const double *x, const double *y, size_t n, int slot) {
    hipLaunchKernelGGL(k_dot_partial, dim3(RED_BLOCKS), dim3(256), 0, s, n, x, y, partial.p);
    hipLaunchKernelGGL(k_finish_reduce, dim3(1), dim3(64), 0, s, partial.p, RED_BLOCKS, 0, res.p + slot);
  }
  // result slots slot0 .. slot0 + K - 1 <- x_k . y   (K <= 16: one pass over y)
  void multidot(const std::vector<const double *> &xs, const double *y, size_t n, int slot0) {
    for (size_t k0 = 0; k0 < xs.size(); k0 += 16) {
      MultiDot md;
      md.n = (int)std::min<size_t>(16, xs.size() - k0);
      for (int k = 0; k < md.n; k++) md.x[k] = xs[k0 + k];
      hipLaunchKernelGGL(k_multidot_partial, dim3(RED_BLOCKS), dim3(256), 0, s, n, md, y, partial.p);
      hipLaunchKernelGGL(k_finish_multidot, dim3(1), dim3(64), 0, s, partial.p, RED_BLOCKS, md.n, res.p + slot0 + (int)k0);
    }
  }
  void maxabs(const double *x, size_t n, int slot) {
    hipLaunchKernelGGL(k_maxabs_partial, dim3(RED_BLOCKS), dim3(256), 0, s, n, x, partial.p);
    hipLaunchKernelGGL(k_finish_reduce, dim3(1), dim3(64), 0, s, partial.p, RED_BLOCKS, 1, res.p + slot);
  }
  void fetch(int nslots) {
    hres.resize(nslots);
    HFG_HIP_CHECK(hipMemcpyAsync(hres.data(), res.p, sizeof(double) * nslots, hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
  }
  // err = Sinvh^T (F P S - S P F) Sinvh    (diis.cpp:139-146)
  void diis_error(const double *F, const double *Ps, double *err) {
    const int n = (int)N;
    gemm_dev(ctx, false, false, n, n, n, 1.0, F, n, Ps, n, 0.0, T1.p, n);
    gemm_dev(ctx, false, false, n, n, n, 1.0, T1.p, n, S.p, n, 0.0, T2.p, n);
    hipLaunchKernelGGL(k_antisym, dim3((n + 255) / 256, n), dim3(256), 0, s, T2.p, n, T1.p);
    gemm_dev(ctx, true, false, n, n, n, 1.0, Sinvh.p, n, T1.p, n, 0.0, T2.p, n);
    gemm_dev(ctx, false, false, n, n, n, 1.0, T2.p, n, Sinvh.p, n, 0.0, err, n);
  }

  // The same error for block-diagonal F, P, S (symmetry-enforced Fock matrix, block-confined orbitals) and a
  // block-structured Sinvh: per symmetry block e_b = X_b^T (F_b P_b S_b - S_b P_b F_b) X_b with X_b = Sinvh(idx_b, cols_b).
  // The full product is these blocks in the ordering of Sinvh's columns and zeros elsewhere; the mixer only takes inner
  // products and the largest element of the error, so the blocks of both spins are stored one after the other at the head
  // of the error buffer and the inner products run over that part only.  Four task-list launches with nspin * nblk tasks each instead of four N^3 products:
  // 9 times fewer flops for three equal blocks.
  struct BlockedErr {
    bool on = false;
    int nblk = 0, nspin = 0, nmax = 0;
    std::vector<int> ns;
    std::vector<size_t> eoff;  // offset of block b in the error buffer
    size_t etot = 0;
    DevBuf<int64_t> rows, cols;
    std::vector<int64_t> ptr;
    DevBuf<double> Sb, Xb, Fb, Pb, T1b, T2b;  // [spin][block] slots of nmax^2 (Sb, Xb: [block])
    DevBuf<GemmTask> tasks;                   // 4 stages x nspin x nblk
  } be;
  void blocked_error_setup(int nspin, const std::vector<int64_t> &ptr, const std::vector<int64_t> &idx) {
    const int nblk = (int)ptr.size() - 1;
    std::vector<int64_t> cols;
    eig_block_supports(ctx, (int)N, Sinvh.p, nblk, ptr.data(), idx.data(), cols);
    be.nblk = nblk;
    be.nspin = nspin;
    be.ptr = ptr;
    be.ns.resize(nblk);
    be.eoff.resize(nblk);
    be.nmax = 0;
    be.etot = 0;
    for (int b = 0; b < nblk; b++) {
      be.ns[b] = (int)(ptr[b + 1] - ptr[b]);
      be.nmax = std::max(be.nmax, be.ns[b]);
      be.eoff[b] = be.etot;
      be.etot += (size_t)be.ns[b] * be.ns[b];
    }
    be.rows.upload(idx, s);
    be.cols.upload(cols, s);
    const size_t sl = (size_t)be.nmax * be.nmax;
    be.Sb.resize(sl * nblk);
    be.Xb.resize(sl * nblk);
    for (DevBuf<double> *q : {&be.Fb, &be.Pb, &be.T1b, &be.T2b}) q->resize(sl * nblk * nspin);
    HFG_HIP_CHECK(hipStreamSynchronize(s));  // idx / cols are the caller's vectors
    for (int b = 0; b < nblk; b++) {
      const int nb = be.ns[b];
      dim3 g((nb + 255) / 256, nb);
      hipLaunchKernelGGL(k_gather_rc, g, dim3(256), 0, s, S.p, (int)N, be.rows.p + ptr[b], be.rows.p + ptr[b], nb, be.Sb.p + sl * b);
      hipLaunchKernelGGL(k_gather_rc, g, dim3(256), 0, s, Sinvh.p, (int)N, be.rows.p + ptr[b], be.cols.p + ptr[b], nb, be.Xb.p + sl * b);
    }
    be.on = true;
  }
  // errs[sp]: where the error of spin sp goes (N*N doubles each); F, P: the spins' matrices
  void blocked_error(const double *const *F, const double *const *P, double *const *errs) {
    const size_t sl = (size_t)be.nmax * be.nmax;
    const int nt = be.nspin * be.nblk;
    std::vector<GemmTask> t((size_t)4 * nt);
    for (int sp = 0; sp < be.nspin; sp++) {
      for (int b = 0; b < be.nblk; b++) {
        const int nb = be.ns[b], k = sp * be.nblk + b;
        dim3 g((nb + 255) / 256, nb);
        hipLaunchKernelGGL(k_gather_rc, g, dim3(256), 0, s, F[sp], (int)N, be.rows.p + be.ptr[b], be.rows.p + be.ptr[b], nb, be.Fb.p + sl * k);
        hipLaunchKernelGGL(k_gather_rc, g, dim3(256), 0, s, P[sp], (int)N, be.rows.p + be.ptr[b], be.rows.p + be.ptr[b], nb, be.Pb.p + sl * k);
        GemmTask q;
        q.M = q.N = q.K = nb;
        q.lda = q.ldb = q.ldc = nb;
        q.A = be.Fb.p + sl * k;  // T1 = F P
        q.B = be.Pb.p + sl * k;
        q.C = be.T1b.p + sl * k;
        t[0 * nt + k] = q;
        q.A = be.T1b.p + sl * k;  // T2 = T1 S
        q.B = be.Sb.p + sl * b;
        q.C = be.T2b.p + sl * k;
        t[1 * nt + k] = q;
        q.A = be.Xb.p + sl * b;  // T2 = X^T A   (A = T2 - T2^T lives in T1 by then)
        q.tA = 1;
        q.B = be.T1b.p + sl * k;
        q.C = be.T2b.p + sl * k;
        t[2 * nt + k] = q;
        q.tA = 0;
        q.A = be.T2b.p + sl * k;  // e = T2 X, straight into the error buffer
        q.B = be.Xb.p + sl * b;
        q.C = errs[sp] + be.eoff[b];
        t[3 * nt + k] = q;
      }
    }
    be.tasks.upload(t, s);
    HFG_HIP_CHECK(hipStreamSynchronize(s));  // t lives on this stack frame
    gemm_tasklist_dev(ctx, be.tasks.p, nt, be.nmax, be.nmax);
    gemm_tasklist_dev(ctx, be.tasks.p + nt, nt, be.nmax, be.nmax);
    for (int sp = 0; sp < be.nspin; sp++)
      for (int b = 0; b < be.nblk; b++) {
        const int nb = be.ns[b], k = sp * be.nblk + b;
        hipLaunchKernelGGL(k_antisym_block, dim3((nb + 255) / 256, nb), dim3(256), 0, s, be.T2b.p + sl * k, nb, be.T1b.p + sl * k);
      }
    gemm_tasklist_dev(ctx, be.tasks.p + 2 * nt, nt, be.nmax, be.nmax);
    gemm_tasklist_dev(ctx, be.tasks.p + 3 * nt, nt, be.nmax, be.nmax);
  }
  // The same blocks from the occupied orbitals: P = C_o C_o^T (formed from these columns by the loop itself), so
  //   e_b = U Y^T - Y U^T,   U = X_b^T F_b C_o(idx_b, :),   Y = X_b^T S_b C_o(idx_b, :)
  // -- products with nocc columns instead of four n^3 products per block (orbitals of other blocks are zero rows here and
  // add nothing).  Three task-list launches: [F_b C | S_b C], [U | -Y] and [Y | U] (each product written where the last
  // one needs it), e_b = [U | -Y] [Y | U]^T.
  DevBuf<double> lrC, lrT, lrG;  // per (spin, block): C rows (nmax x nocc), F C and S C (2 x), [U | -Y] and [Y | U]
  void blocked_error_lowrank(const double *const *F, const double *const *C, const int *nocc, double *const *errs) {
    const size_t sl = (size_t)be.nmax * be.nmax;
    const int nt = be.nspin * be.nblk;
    int kmax = 0;
    for (int sp = 0; sp < be.nspin; sp++) kmax = std::max(kmax, nocc[sp]);
    const size_t cs = (size_t)be.nmax * std::max(kmax, 1);
    lrC.resize(cs * nt);
    lrT.resize(2 * cs * nt);
    lrG.resize(4 * cs * nt);
    std::vector<GemmTask> t;
    std::vector<GemmTask> t1, t2, t3;
    for (int sp = 0; sp < be.nspin; sp++)
      for (int b = 0; b < be.nblk; b++) {
        const int nb = be.ns[b], k = sp * be.nblk + b, no = nocc[sp];
        if (no == 0) {
          HFG_HIP_CHECK(hipMemsetAsync(errs[sp] + be.eoff[b], 0, sizeof(double) * (size_t)nb * nb, s));
          continue;
        }
        hipLaunchKernelGGL(k_gather_rc, dim3((nb + 255) / 256, nb), dim3(256), 0, s, F[sp], (int)N, be.rows.p + be.ptr[b],
                           be.rows.p + be.ptr[b], nb, be.Fb.p + sl * k);
        hipLaunchKernelGGL(k_gather_rows, dim3((nb + 255) / 256, no), dim3(256), 0, s, C[sp], (int)N, be.rows.p + be.ptr[b], nb,
                           lrC.p + cs * k);
        double *FC = lrT.p + 2 * cs * k, *SC = FC + cs;
        double *G1 = lrG.p + 4 * cs * k, *G2 = G1 + 2 * cs;  // [U | -Y], [Y | U], nb x 2 no each
        GemmTask q;
        q.M = nb;
        q.N = no;
        q.K = nb;
        q.lda = q.ldb = q.ldc = nb;
        q.B = lrC.p + cs * k;
        q.A = be.Fb.p + sl * k;
        q.C = FC;
        t1.push_back(q);
        q.A = be.Sb.p + sl * b;
        q.C = SC;
        t1.push_back(q);
        q.A = be.Xb.p + sl * b;
        q.tA = 1;
        q.B = FC;  // U
        q.C = G1;
        t2.push_back(q);
        q.C = G2 + (size_t)nb * no;
        t2.push_back(q);
        q.B = SC;  // Y
        q.C = G2;
        t2.push_back(q);
        q.alpha = -1.0;
        q.C = G1 + (size_t)nb * no;
        t2.push_back(q);
        GemmTask e;
        e.M = e.N = nb;
        e.K = 2 * no;
        e.lda = e.ldb = e.ldc = nb;
        e.A = G1;
        e.B = G2;
        e.tB = 1;
        e.C = errs[sp] + be.eoff[b];
        t3.push_back(e);
      }
    if (t3.empty()) return;
    t = t1;
    t.insert(t.end(), t2.begin(), t2.end());
    t.insert(t.end(), t3.begin(), t3.end());
    be.tasks.upload(t, s);
    HFG_HIP_CHECK(hipStreamSynchronize(s));  // t lives on this stack frame
    gemm_tasklist64_dev(ctx, be.tasks.p, (int)t1.size(), be.nmax, kmax);
    gemm_tasklist64_dev(ctx, be.tasks.p + t1.size(), (int)t2.size(), be.nmax, kmax);
    gemm_tasklist_dev(ctx, be.tasks.p + t1.size() + t2.size(), (int)t3.size(), be.nmax, be.nmax);
  }
};

}  // namespace

/// the shared part of both programs, everything on the device; `basis` already holds the host tables of S, T, V
helfem::scf::Result scf_device_loop(hfg_ctx *ctx, hfg_basis *hb, const helfem::scf::Options &opt, int nel, double Enucr,
                                    int symm, const std::vector<std::vector<size_t> > &dsym, int ldft, int mdft,
                                    const std::vector<std::vector<std::vector<size_t> > > &avg_idx) {
  using helfem::Mat;
  helfem::scf::Result res;
  res.Enucr = Enucr;
  const bool verbose = opt.verbose;
  const bool dft = (opt.x_func > 0 || opt.c_func > 0);
  int nela = opt.nela, nelb = opt.nelb;
  {
    int Qv = opt.Q, Mv = opt.multiplicity;
    helfem::scf::parse_nela_nelb(nela, nelb, Qv, Mv, nel + opt.Q);  // nel = Ztot - Q
  }
  const bool restr_req = (opt.restricted == -1) ? (nela == nelb) : (opt.restricted != 0);
  const bool rohf = restr_req && nela != nelb;  // restricted open shell: unrestricted machinery + CUHF constraint
  const bool restr = restr_req && !rohf;
  res.nela = nela;
  res.nelb = nelb;
  helfem::scf::Options oocc = opt;
  oocc.symmetry = symm;
  const helfem::scf::OccupationPlan focc = hb->kind ? helfem::scf::occupation_plan(oocc, hb->ab, nela, nelb) : helfem::scf::occupation_plan(oocc, hb->b, nela, nelb);

  DevSCF d(ctx);
  hipStream_t s = d.s;
  {
    Mat S = hb->kind ? hb->ab.overlap() : hb->b.overlap();
    Mat T = hb->kind ? hb->ab.kinetic() : hb->b.kinetic();
    Mat V = hb->kind ? hb->ab.nuclear() : hb->b.nuclear();
    d.N = S.n_rows;
    d.NN = d.N * d.N;
    d.up(d.S, S);
    d.up(d.T, T);
    d.up(d.V, V);
    Mat H0 = T + V;
    d.up(d.H0, H0);
  }
  const size_t N = d.N, NN = d.NN;
  const int n = (int)N;
  res.Nbf = N;
  // symmetry blocks
  std::vector<int64_t> ptr(1, 0), idx;
  std::vector<int> blockid(N, 0);
  for (size_t ib = 0; ib < dsym.size(); ib++) {
    for (size_t v : dsym[ib]) {
      idx.push_back((int64_t)v);
      blockid[v] = (int)ib;
    }
    ptr.push_back((int64_t)idx.size());
  }
  d.blockid.upload(blockid, s);
  for (DevBuf<double> *b : {&d.Sinvh, &d.Ca, &d.Pa, &d.P, &d.J, &d.Fa, &d.T1, &d.T2, &d.Err}) b->resize(NN);
  d.Ea.resize(N);
  d.scal.resize(4);
  d.partial.resize(16 * RED_BLOCKS);
  d.res.resize(8 + 3 * (size_t)std::max(1, opt.diisorder) + 8);
  if (!restr) {
    for (DevBuf<double> *b : {&d.Cb, &d.Pb, &d.Fb}) b->resize(NN);
    d.Eb.resize(N);
  }
  const bool anyK = (opt.kfrac != 0.0 || opt.kshort != 0.0);
  DevBuf<double> Krs;  // short-range exact exchange of the range-separated hybrids (atomic/main.cpp:768-769)
  if (anyK) {
    d.Ka.resize(NN);
    if (!restr) d.Kb.resize(NN);
    if (opt.omega != 0.0) Krs.resize(NN);
  }
  if (dft) {
    d.XCa.resize(NN);
    if (!restr) d.XCb.resize(NN);
  }
  const int nspin = restr ? 1 : 2;
  const int order = opt.diisorder;
  d.histF.reset(new DevBuf<double>[order]);
  d.histE.reset(new DevBuf<double>[order]);
  d.histP.reset(new DevBuf<double>[order]);
  for (int k = 0; k < order; k++) {
    d.histF[k].resize(nspin * NN);
    d.histE[k].resize(nspin * NN);
    d.histP[k].resize(nspin * NN);
  }

  std::unique_ptr<DevBuf<int>[]> avgdev(new DevBuf<int>[avg_idx.size() + 1]);
  for (size_t gi = 0; gi < avg_idx.size(); gi++) {
    std::vector<int> flat;
    for (const auto &l : avg_idx[gi])
      for (size_t v : l) flat.push_back((int)v);
    avgdev[gi].upload(flat, s);
  }
  HFG_HIP_CHECK(hipStreamSynchronize(s));
  double t0 = wall();
  form_sinvh_dev(ctx, n, d.S.p, !opt.diag, (int)dsym.size(), ptr.data(), idx.data(), d.Sinvh.p);
  // S^{-1/2} stays as it is until this run returns: the eigensolves derive the blocks' column supports once
  struct FixSinvh {
    hfg_ctx *c;
    ~FixSinvh() { c->fix_sinvh(nullptr); }
  } fix_sinvh_guard{ctx};
  ctx->fix_sinvh(d.Sinvh.p);
  HFG_HIP_CHECK(hipStreamSynchronize(s));
  if (verbose) printf("Half-inverse formed in %.6f\n", wall() - t0);
  DevBuf<double> Sh, Pvec, A2N, ShPv, occ;
  if (rohf) {  // partner of Sinvh: Sh^T Sinvh = 1 (S^{1/2} for the symmetric half-inverse)
    for (DevBuf<double> *b : {&Sh, &Pvec, &A2N, &ShPv}) b->resize(NN);
    occ.resize(N);
    gemm_dev(ctx, false, false, n, n, n, 1.0, d.S.p, n, d.Sinvh.p, n, 0.0, Sh.p, n);
  }
  auto prepare_tables = [&]() {
    if (verbose) printf("Computing two-electron integrals\n");
    t0 = wall();
    static const bool host_tei = getenv("HELFEM_TEI") && !strcmp(getenv("HELFEM_TEI"), "host");  // the checker of tei_dev.hip
    if (host_tei) {
      if (hb->kind) hb->ab.compute_tei(opt.kfrac != 0.0);
      else hb->b.compute_tei(opt.kfrac != 0.0);
      hb->tei_on_device = false;
    } else
      compute_tei_dev(ctx, hb);  // in-element tables on the device (tei_dev.hip)
    if (opt.omega != 0.0) {  // atomic/main.cpp:709-712
      if (!hb->kind) throw std::logic_error("Range separated functionals are not supported.\n");
      if (opt.rs_kind == 1) hb->ab.compute_yukawa(opt.omega);
      else hb->ab.compute_erfc(opt.omega);
    }
    if (hb->dev) {
      fock_release(hb->dev);
      exchange_release(hb->dev);
      exchange_lr_release(hb->dev);
    }
    upload_tables(ctx, hb, ldft, mdft);
    if (hb->dev_rs) {
      exchange_release(hb->dev_rs);
      exchange_lr_release(hb->dev_rs);
      delete hb->dev_rs;
      hb->dev_rs = nullptr;
    }
    if (opt.omega != 0.0) upload_rs_tables(ctx, hb);
    if (verbose) printf("Done in %.6f\n", wall() - t0);
  };
  // guess (main.cpp:650-712): core Hamiltonian, or T + the model potential of the screened nuclei by quadrature on the
  // device (diatomic) / radial integrals (atomic); the quadrature needs the tables, so they come first in that case
  const double *Hg = d.H0.p;
  if (opt.have_guess) {
    // --load: orbitals of a previous run (checked and re-orthonormalised on the host), nothing to evaluate
  } else if (opt.iguess != 0) {
    prepare_tables();
    if (verbose) printf("Guess orbitals from %s nucleus\n", opt.iguess == 3 ? "Thomas-Fermi" : "screened");
    if (opt.iguess != 3) throw std::logic_error("Unsupported guess\n");
    const int Za = hb->kind ? hb->ab.Z : hb->b.Z1, Zb = hb->kind ? 0 : hb->b.Z2;
    if (hb->kind) {
      helfem::ModelPotential mp;
      mp.kind = 3;
      mp.Z = Za;
      d.up(d.T1, hb->ab.model_potential(mp));
    } else
      model_potential_dev(ctx, hb, Za ? 3 : 0, Za, 0.0, 0.0, Zb ? 3 : 0, Zb, 0.0, 0.0, d.T1.p);
    d.axpby(1.0, d.T.p, 1.0, d.T1.p, NN);  // T1 = T + V_model
    Hg = d.T1.p;
  } else if (verbose)
    printf("Guess orbitals from core Hamiltonian\n");
  // forced occupations on the device: S C by one product, the symmetry weights of all orbitals by one small kernel per
  // occupied symmetry, the order on the host from the weights and the energies (the reference's rule), one gather
  DevBuf<double> FdB;  // the beta spin's extrapolated Fock matrix (the alpha one lives in d.T1)
  DevBuf<int> occ_rows, occ_order;
  DevBuf<double> occ_w;
  std::vector<int> occ_rowptr(1, 0);
  if (focc.until) {
    std::vector<char> seen(N, 0);
    std::vector<int> rows;
    for (const auto &ix : focc.sym) {
      for (size_t i : ix) {
        if (seen[i]) throw std::logic_error("Duplicate basis functions in symmetry list!\n");
        seen[i] = 1;
        rows.push_back((int)i);
      }
      occ_rowptr.push_back((int)rows.size());
    }
    occ_rows.resize(rows.size() + 1);
    HFG_HIP_CHECK(hipMemcpyAsync(occ_rows.p, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    occ_order.resize(N);
    occ_w.resize(focc.sym.size() * N);
  }
  auto enforce_occupations = [&](double *C, double *E, const std::vector<int> &nocc) {
    gemm_dev(ctx, false, false, n, n, n, 1.0, d.S.p, n, C, n, 0.0, d.T2.p, n);  // S C
    HFG_HIP_CHECK(hipMemsetAsync(occ_w.p, 0, sizeof(double) * focc.sym.size() * N, s));
    for (size_t isym = 0; isym < focc.sym.size(); isym++)
      if (nocc[isym])
        hipLaunchKernelGGL(k_sym_weight, dim3((n + 3) / 4), dim3(256), 0, s, C, d.T2.p, n, occ_rows.p + occ_rowptr[isym],
                           occ_rowptr[isym + 1] - occ_rowptr[isym], occ_w.p + isym * N);
    std::vector<double> hw(focc.sym.size() * N);
    helfem::Vec hE(N);
    HFG_HIP_CHECK(hipMemcpyAsync(hw.data(), occ_w.p, sizeof(double) * hw.size(), hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipMemcpyAsync(hE.data(), E, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<std::vector<double> > w(focc.sym.size());
    for (size_t isym = 0; isym < focc.sym.size(); isym++) w[isym].assign(hw.begin() + isym * N, hw.begin() + (isym + 1) * N);
    const std::vector<size_t> order = helfem::scf::occupation_order(hE, w, nocc);
    std::vector<int> io(order.begin(), order.end());
    HFG_HIP_CHECK(hipMemcpyAsync(occ_order.p, io.data(), sizeof(int) * N, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_columns, dim3((n + 255) / 256, n), dim3(256), 0, s, C, E, n, occ_order.p, d.T2.p, d.T1.p);
    HFG_HIP_CHECK(hipMemcpyAsync(C, d.T2.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
    HFG_HIP_CHECK(hipMemcpyAsync(E, d.T1.p, sizeof(double) * N, hipMemcpyDeviceToDevice, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));  // io is a local
  };
  if (opt.have_guess) {
    if (verbose) printf("Guess orbitals from checkpoint\nGuess orbitals from previous calculation\n");
    Mat Sh0 = hb->kind ? hb->ab.overlap() : hb->b.overlap();
    Mat gCa, gCb, S12, Sinvh_h;
    helfem::Vec gEa, gEb;
    if ((!hb->kind && opt.guess_basis) || (hb->kind && opt.guess_basis_atomic)) {
      // (possibly) another basis: interbasis overlap and this run's half-inverse on the host
      S12 = hb->kind ? hb->ab.overlap(*opt.guess_basis_atomic) : hb->b.overlap(*opt.guess_basis);
      Sinvh_h.zeros(N, N);
      HFG_HIP_CHECK(hipMemcpyAsync(Sinvh_h.memptr(), d.Sinvh.p, sizeof(double) * NN, hipMemcpyDeviceToHost, s));
      HFG_HIP_CHECK(hipStreamSynchronize(s));
    }
    helfem::scf::guess_from_checkpoint(opt, Sh0, Sinvh_h, S12, (size_t)nela, (size_t)nelb, gCa, gCb, gEa, gEb);
    auto put = [&](DevBuf<double> &dC, DevBuf<double> &dE, const Mat &C, const helfem::Vec &E) {
      // the checkpoint may hold fewer columns than basis functions (Cholesky-reduced runs): the rest stays zero
      Mat Cfull(N, N);
      for (size_t j = 0; j < std::min(C.n_cols, N); j++)
        for (size_t i = 0; i < N; i++) Cfull(i, j) = C(i, j);
      helfem::Vec Efull(N, 0.0);
      for (size_t j = 0; j < std::min(E.size(), N); j++) Efull[j] = E[j];
      HFG_HIP_CHECK(hipMemcpyAsync(dC.p, Cfull.memptr(), sizeof(double) * NN, hipMemcpyHostToDevice, s));
      HFG_HIP_CHECK(hipMemcpyAsync(dE.p, Efull.data(), sizeof(double) * N, hipMemcpyHostToDevice, s));
      HFG_HIP_CHECK(hipStreamSynchronize(s));
    };
    put(d.Ca, d.Ea, gCa, gEa);
    if (!restr) put(d.Cb, d.Eb, gCb, gEb);
  } else {
    eig_gsym_sub_dev(ctx, n, Hg, d.Sinvh.p, (int)dsym.size(), ptr.data(), idx.data(), d.Ea.p, d.Ca.p);
    if (!restr) {
      HFG_HIP_CHECK(hipMemcpyAsync(d.Cb.p, d.Ca.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
      HFG_HIP_CHECK(hipMemcpyAsync(d.Eb.p, d.Ea.p, sizeof(double) * N, hipMemcpyDeviceToDevice, s));
    }
  }
  if (focc.until && !opt.have_guess) {  // main.cpp:716-722
    enforce_occupations(d.Ca.p, d.Ea.p, focc.na);
    if (!restr) enforce_occupations(d.Cb.p, d.Eb.p, focc.nb);
  }
  if (opt.iguess == 0 || opt.have_guess) prepare_tables();

  // ADIIS + CDIIS weights of the reference (diis.cpp:214-290) on the host from inner products of the stored matrices;
  // the restricted driver counts its one spin twice, as the reference does by passing Fa = Fb, Pa = Pb to uDIIS
  // DIIS error by symmetry blocks when every matrix in it is block diagonal (not with the CUHF constraint, whose lambda
  // comes from natural orbitals of the whole density); HELFEM_DIIS_BLOCKS=0 keeps the four dense N^3 products (checker)
  static const bool blocks_off = getenv("HELFEM_DIIS_BLOCKS") && atoi(getenv("HELFEM_DIIS_BLOCKS")) == 0;
  bool blocked_err = symm != 0 && dsym.size() > 1 && !rohf && !blocks_off;
  if (blocked_err) d.blocked_error_setup(nspin, ptr, idx);
  helfem::DiisMixer mixer(true, opt.diiseps, opt.diisthr, true, verbose, (size_t)order);
  const double spinfac = restr ? 2.0 : 1.0;
  std::deque<int> slots;  // ring slots in age order
  double Eold = 0.0;
  for (int it = 1; it <= opt.maxit; it++) {
    if (verbose) printf("\n**** Iteration %i ****\n\n", it);
    form_density_dev(ctx, n, n, d.Ca.p, nela, d.Pa.p);
    const double *Pb = d.Pa.p;
    if (!restr) {
      if (nelb) form_density_dev(ctx, n, n, d.Cb.p, nelb, d.Pb.p);
      else HFG_HIP_CHECK(hipMemsetAsync(d.Pb.p, 0, sizeof(double) * NN, s));
      Pb = d.Pb.p;
    }
    HFG_HIP_CHECK(hipMemcpyAsync(d.P.p, d.Pa.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
    d.axpby(1.0, Pb, 1.0, d.P.p, NN);

    double tJ0 = wall();
    coulomb_dev(ctx, hb, d.P.p, d.J.p);
    if (verbose) HFG_HIP_CHECK(hipStreamSynchronize(s));
    res.tJ = wall() - tJ0;
    double tK0 = wall();
    if (anyK) {
      // Ps = C_occ C_occ^T was formed from the first nocc columns of Cs above: the exchange fast path takes them as its
      // factors instead of recovering them from Ps
      auto buildK = [&](const double *Ps, const double *Cs, int nocc, double *K) {  // K = kfrac K[1/r12] + kshort K[screened kernel]
        if (opt.kfrac != 0.0) {
          exchange_dev(ctx, hb, Ps, K, false, Cs, nocc);
          d.axpby(0.0, K, opt.kfrac, K, NN);
        } else
          HFG_HIP_CHECK(hipMemsetAsync(K, 0, sizeof(double) * NN, s));
        if (opt.omega != 0.0) {
          exchange_dev(ctx, hb, Ps, Krs.p, true, Cs, nocc);
          d.axpby(opt.kshort, Krs.p, 1.0, K, NN);
        }
      };
      buildK(d.Pa.p, d.Ca.p, nela, d.Ka.p);
      if (!restr) {
        if (nelb) buildK(d.Pb.p, d.Cb.p, nelb, d.Kb.p);
        else HFG_HIP_CHECK(hipMemsetAsync(d.Kb.p, 0, sizeof(double) * NN, s));
      }
      if (verbose) HFG_HIP_CHECK(hipStreamSynchronize(s));
    }
    res.tK = wall() - tK0;
    double tX0 = wall();
    if (dft) {
      if (restr) xc_fock_dev(ctx, hb, opt.x_func, opt.c_func, d.P.p, d.XCa.p, d.scal.p, opt.dftthr);
      else xc_fock_pol_dev(ctx, hb, opt.x_func, opt.c_func, d.Pa.p, d.Pb.p, d.XCa.p, d.XCb.p, d.scal.p, opt.dftthr);
      if (verbose) HFG_HIP_CHECK(hipStreamSynchronize(s));
    }
    res.tXC = wall() - tX0;

    // energies: res slots 0 Ekin, 1 Epot, 2 2*Ecoul, 3 Tr PaKa, 4 Tr PbKb
    d.dot(d.P.p, d.T.p, NN, 0);
    d.dot(d.P.p, d.V.p, NN, 1);
    d.dot(d.P.p, d.J.p, NN, 2);
    if (anyK) {
      d.dot(d.Pa.p, d.Ka.p, NN, 3);
      if (!restr) d.dot(d.Pb.p, d.Kb.p, NN, 4);
    }

    // Fock matrices
    for (int sp = 0; sp < nspin; sp++) {
      double *F = sp ? d.Fb.p : d.Fa.p;
      HFG_HIP_CHECK(hipMemcpyAsync(F, d.H0.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
      d.axpby(1.0, d.J.p, 1.0, F, NN);
      if (anyK) d.axpby(1.0, sp ? d.Kb.p : d.Ka.p, 1.0, F, NN);
      if (dft) d.axpby(1.0, sp ? d.XCb.p : d.XCa.p, 1.0, F, NN);
      for (size_t gi = 0; gi < avg_idx.size(); gi++) {
        if (avg_idx[gi].empty()) continue;
        const int ng = (int)avg_idx[gi].size(), nn = (int)avg_idx[gi][0].size();
        hipLaunchKernelGGL(k_fock_average, dim3((nn + 255) / 256, nn), dim3(256), 0, s, F, n, avgdev[gi].p, ng, nn);
      }
      if (symm) hipLaunchKernelGGL(k_mask_blocks, dim3((n + 255) / 256, n), dim3(256), 0, s, F, n, d.blockid.p);
    }

    if (rohf) {
      // scf::ROHF_update (scf_helpers.cpp:470-523) with every product on the matrix cores
      gemm_dev(ctx, false, false, n, n, n, 1.0, d.P.p, n, Sh.p, n, 0.0, d.T1.p, n);
      gemm_dev(ctx, true, false, n, n, n, 1.0, Sh.p, n, d.T1.p, n, 0.0, d.T2.p, n);  // P in the orthonormal basis
      eig_sym_dev(ctx, n, d.T2.p, occ.p, Pvec.p);                                    // natural orbitals, ascending
      gemm_dev(ctx, false, false, n, n, n, 1.0, d.Sinvh.p, n, Pvec.p, n, 0.0, A2N.p, n);
      gemm_dev(ctx, false, false, n, n, n, 1.0, Sh.p, n, Pvec.p, n, 0.0, ShPv.p, n);
      d.axpby(0.5, d.Fa.p, 0.0, d.T1.p, NN);
      d.axpby(-0.5, d.Fb.p, 1.0, d.T1.p, NN);  // Delta
      gemm_dev(ctx, false, false, n, n, n, 1.0, d.T1.p, n, A2N.p, n, 0.0, d.T2.p, n);
      gemm_dev(ctx, true, false, n, n, n, 1.0, A2N.p, n, d.T2.p, n, 0.0, d.T1.p, n);  // Delta in the NO basis
      const int Nc = std::min(nela, nelb), Nv = n - std::max(nela, nelb);
      hipLaunchKernelGGL(k_rohf_lambda, dim3((n + 255) / 256, n), dim3(256), 0, s, d.T1.p, n, Nc, Nv);
      gemm_dev(ctx, false, false, n, n, n, 1.0, ShPv.p, n, d.T1.p, n, 0.0, d.T2.p, n);
      gemm_dev(ctx, false, true, n, n, n, 1.0, d.T2.p, n, ShPv.p, n, 0.0, d.T1.p, n);  // lambda in the AO basis
      d.axpby(1.0, d.T1.p, 1.0, d.Fa.p, NN);
      d.axpby(-1.0, d.T1.p, 1.0, d.Fb.p, NN);
    }

    // DIIS: store (F, P, err) in a ring slot; new row of B = err.err, new row and column of T = Tr P F; slot 5 = max |err|
    t0 = wall();
    int slot;
    if (mixer.full()) {
      mixer.pop_oldest();
      slot = slots.front();
      slots.pop_front();
    } else {
      slot = -1;  // lowest free ring slot (the extrapolation may have dropped old entries: DiisMixer::solve)
      for (int q = 0; q < order && slot < 0; q++) {
        bool used = false;
        for (int v : slots) used = used || (v == q);
        if (!used) slot = q;
      }
    }
    slots.push_back(slot);
    if (blocked_err && it == 1 && opt.have_guess) {
      // Orbitals that did not come from this loop's block eigensolve (a checkpoint made with another symmetry setting, a
      // projection from another basis) may mix the symmetry blocks: P then has off-block parts, and so has the DIIS error,
      // which the per-block form would drop (the reference's uDIIS works on the full matrices).  Compare the squared norm
      // of the dense error of this first density with that of its diagonal blocks; unless they agree, the run keeps the
      // dense products.
      const double *Fs0[2] = {d.Fa.p, nspin == 2 ? d.Fb.p : nullptr}, *Ps0[2] = {d.Pa.p, nspin == 2 ? d.Pb.p : nullptr};
      double *Es0[2] = {d.histE[slot].p, d.histE[slot].p + d.be.etot};
      d.blocked_error(Fs0, Ps0, Es0);
      d.multidot(std::vector<const double *>(1, d.histE[slot].p), d.histE[slot].p, nspin * d.be.etot, 6);
      double dense2 = 0.0;
      for (int sp = 0; sp < nspin; sp++) {
        d.diis_error(sp ? d.Fb.p : d.Fa.p, sp ? d.Pb.p : d.Pa.p, d.Err.p);
        d.multidot(std::vector<const double *>(1, d.Err.p), d.Err.p, NN, 7);
        d.fetch(8);
        dense2 += d.hres[7];
      }
      const double blocked2 = d.hres[6];
      if (std::fabs(dense2 - blocked2) > 1e-10 * std::max(dense2, 1e-300)) {
        blocked_err = false;
        if (verbose) printf("The guess orbitals mix the symmetry blocks: DIIS error from the full matrices in this run\n");
      }
    }
    if (blocked_err) {
      const double *Fs[2] = {d.Fa.p, nspin == 2 ? d.Fb.p : nullptr}, *Ps[2] = {d.Pa.p, nspin == 2 ? d.Pb.p : nullptr};
      double *Es[2] = {d.histE[slot].p, d.histE[slot].p + d.be.etot};  // the spins' blocks one after the other
      // P = C_occ C_occ^T of this iteration's orbitals: the error from the occupied columns (HELFEM_DIIS_LOWRANK=0: the
      // four n^3 products per block from F and P, the checker)
      static const bool lowrank_off = getenv("HELFEM_DIIS_LOWRANK") && atoi(getenv("HELFEM_DIIS_LOWRANK")) == 0;
      const double *Cs[2] = {d.Ca.p, nspin == 2 ? d.Cb.p : nullptr};
      const int noccs[2] = {nela, nelb};
      if (lowrank_off) d.blocked_error(Fs, Ps, Es);
      else d.blocked_error_lowrank(Fs, Cs, noccs, Es);
    }
    for (int sp = 0; sp < nspin; sp++) {
      double *F = sp ? d.Fb.p : d.Fa.p;
      if (!blocked_err) d.diis_error(F, sp ? d.Pb.p : d.Pa.p, d.histE[slot].p + sp * NN);
      HFG_HIP_CHECK(hipMemcpyAsync(d.histF[slot].p + sp * NN, F, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
      HFG_HIP_CHECK(hipMemcpyAsync(d.histP[slot].p + sp * NN, sp ? d.Pb.p : d.Pa.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
    }
    const size_t elen = blocked_err ? nspin * d.be.etot : nspin * NN;  // the stored part of an error
    d.maxabs(d.histE[slot].p, elen, 5);
    const int nh0 = (int)slots.size();
    {
      // the new row of B and the new row and column of T: three passes (one per fixed vector) instead of 3 nh0 dot products
      std::vector<const double *> xe, xp, xf;
      for (int k = 0; k < nh0; k++) {
        xe.push_back(d.histE[slots[k]].p);
        xp.push_back(d.histP[slots[k]].p);
        xf.push_back(d.histF[slots[k]].p);
      }
      d.multidot(xe, d.histE[slot].p, elen, 8);
      d.multidot(xp, d.histF[slot].p, nspin * NN, 8 + nh0);      // T(k, n) = P_k . F_n
      d.multidot(xf, d.histP[slot].p, nspin * NN, 8 + 2 * nh0);  // T(n, k) = P_n . F_k
    }
    d.fetch(8 + 3 * nh0);
    if (dft) {
      double sc[3];
      HFG_HIP_CHECK(hipMemcpyAsync(sc, d.scal.p, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
      HFG_HIP_CHECK(hipStreamSynchronize(s));
      res.Exc = sc[0];
      if (verbose) {
        printf("DFT energy %.10e % .6f\n", res.Exc, res.tXC);
        printf("Error in integrated number of electrons % e\n", sc[1] - nel);
      }
    }
    res.Ekin = d.hres[0];
    res.Epot = d.hres[1];
    res.Ecoul = 0.5 * d.hres[2];
    res.Exx = 0.0;
    if (anyK) res.Exx = restr ? d.hres[3] : 0.5 * d.hres[3] + 0.5 * d.hres[4];
    const double diiserr = d.hres[5];
    if (verbose) {
      printf("Coulomb energy %.10e % .6f\n", res.Ecoul, res.tJ);
      if (anyK) printf("Exchange energy %.10e % .6f\n", res.Exx, res.tK);
    }
    res.Etot = res.Ekin + res.Epot + res.Ecoul + res.Exx + res.Exc + res.Enucr;
    const double dE = res.Etot - Eold;
    mixer.push(res.Etot, diiserr);
    for (int k = 0; k < nh0; k++) {
      mixer.set_B((size_t)k, (size_t)nh0 - 1, spinfac * d.hres[8 + k]);
      mixer.set_T((size_t)k, (size_t)nh0 - 1, spinfac * d.hres[8 + nh0 + k]);
      mixer.set_T((size_t)nh0 - 1, (size_t)k, spinfac * d.hres[8 + 2 * nh0 + k]);
    }
    if (verbose) {
      printf("Total energy is % .10f\n", res.Etot);
      if (it > 1) printf("Energy changed by %e\n", dE);
      printf("DIIS error is %e, update done in %.6f\n", diiserr, wall() - t0);
    }
    Eold = res.Etot;

    // weights on the host (tiny), extrapolated Fock matrices on the device
    t0 = wall();
    size_t dropped = 0;
    const std::vector<double> coef = mixer.solve(dropped);
    for (size_t k = 0; k < dropped; k++) slots.pop_front();
    const size_t nh = slots.size();
    if (verbose) printf("DIIS solution done in %.6f\n", wall() - t0);
    const bool convd = (diiserr < opt.convthr) && (fabs(dE) < opt.convthr);

    t0 = wall();
    const bool damping = (opt.dampfock != 1.0 && diiserr >= opt.dampthr);  // atomic/main.cpp:917-936
    if (damping && verbose) printf("Damping off-diagonal elements of Fock matrix by % .3f\n", opt.dampfock);
    // both spins' extrapolated (and damped) Fock matrices first, then ONE batched eigensolve for them: the
    // tridiagonalisation's chain of dependent launches is as long for six blocks as for three
    static const bool pair_eig = !(getenv("HELFEM_EIG_PAIR") && atoi(getenv("HELFEM_EIG_PAIR")) == 0);
    if (nspin == 2) FdB.resize(NN);
    double *Fds[2] = {d.T1.p, nspin == 2 ? FdB.p : nullptr};
    for (int sp = 0; sp < nspin; sp++) {
      double *Fd = Fds[sp];
      if (nh >= 1 && nh <= 16) {
        LinComb lc;
        lc.n = (int)nh;
        for (size_t a = 0; a < nh; a++) {
          lc.x[a] = d.histF[slots[a]].p + sp * NN;
          lc.c[a] = coef[a];
        }
        hipLaunchKernelGGL(k_lincomb, dim3(2048), dim3(256), 0, s, NN, lc, Fd);
      } else
        for (size_t a = 0; a < nh; a++) d.axpby(coef[a], d.histF[slots[a]].p + sp * NN, a ? 1.0 : 0.0, Fd, NN);
      const int nocc = sp ? nelb : nela;
      if (damping && nocc > 0 && n > nocc) {
        // F <- S C f C^T S with f = C^T F C, its occupied-virtual blocks scaled (C: the orbitals that built this density)
        const double *Cs = sp ? d.Cb.p : d.Ca.p;
        gemm_dev(ctx, true, false, n, n, n, 1.0, Cs, n, Fd, n, 0.0, d.T2.p, n);         // C^T F
        gemm_dev(ctx, false, false, n, n, n, 1.0, d.T2.p, n, Cs, n, 0.0, d.Err.p, n);    // f
        hipLaunchKernelGGL(k_scale_offdiag_blocks, dim3((n + 255) / 256, n), dim3(256), 0, s, d.Err.p, n, nocc, opt.dampfock);
        gemm_dev(ctx, false, false, n, n, n, 1.0, d.S.p, n, Cs, n, 0.0, d.T2.p, n);      // S C
        gemm_dev(ctx, false, false, n, n, n, 1.0, d.T2.p, n, d.Err.p, n, 0.0, Fd, n);    // S C f
        gemm_dev(ctx, false, true, n, n, n, 1.0, Fd, n, d.T2.p, n, 0.0, d.Err.p, n);     // S C f (S C)^T
        HFG_HIP_CHECK(hipMemcpyAsync(Fd, d.Err.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, s));
      }
    }
    if (nspin == 2 && pair_eig)
      eig_gsym_sub_pair_dev(ctx, n, Fds[0], Fds[1], d.Sinvh.p, (int)dsym.size(), ptr.data(), idx.data(), d.Ea.p, d.Ca.p, d.Eb.p, d.Cb.p);
    else
      for (int sp = 0; sp < nspin; sp++)
        eig_gsym_sub_dev(ctx, n, Fds[sp], d.Sinvh.p, (int)dsym.size(), ptr.data(), idx.data(), sp ? d.Eb.p : d.Ea.p, sp ? d.Cb.p : d.Ca.p);
    for (int sp = 0; sp < nspin; sp++)
      if (focc.active(it)) enforce_occupations(sp ? d.Cb.p : d.Ca.p, sp ? d.Eb.p : d.Ea.p, sp ? focc.nb : focc.na);  // main.cpp:942-958
    if (verbose) HFG_HIP_CHECK(hipStreamSynchronize(s));
    res.tdiag = wall() - t0;
    if (verbose) {
      printf("%s diagonalization done in %.6f\n", symm ? "Subspace" : "Full", res.tdiag);
      fflush(stdout);
    }
    res.iterations = it;
    if (convd) {
      res.converged = true;
      break;
    }
  }
  // results the callers read back
  res.E.resize(N);
  HFG_HIP_CHECK(hipMemcpyAsync(res.E.data(), d.Ea.p, sizeof(double) * N, hipMemcpyDeviceToHost, s));
  res.C.zeros(N, N);
  HFG_HIP_CHECK(hipMemcpyAsync(res.C.memptr(), d.Ca.p, sizeof(double) * NN, hipMemcpyDeviceToHost, s));
  HFG_HIP_CHECK(hipStreamSynchronize(s));
  if (opt.keep_matrices) {  // what the reference's drivers write to their checkpoint, from HBM
    auto down = [&](const char *name, const double *p, size_t rows, size_t cols) {
      helfem::Mat m(rows, cols);
      if (p) HFG_HIP_CHECK(hipMemcpy(m.memptr(), p, sizeof(double) * rows * cols, hipMemcpyDeviceToHost));
      res.mats[name] = m;
    };
    down("S", d.S.p, N, N);
    down("T", d.T.p, N, N);
    down("Vnuc", d.V.p, N, N);
    down("H0", d.H0.p, N, N);
    down("Sinvh", d.Sinvh.p, N, N);
    down("P", d.P.p, N, N);
    down("Pa", d.Pa.p, N, N);
    down("Pb", restr ? d.Pa.p : d.Pb.p, N, N);
    down("J", d.J.p, N, N);
    down("Ka", anyK ? d.Ka.p : nullptr, anyK ? N : 0, anyK ? N : 0);
    down("Kb", anyK ? (restr ? d.Ka.p : d.Kb.p) : nullptr, anyK ? N : 0, anyK ? N : 0);
    down("XCa", dft ? d.XCa.p : nullptr, dft ? N : 0, dft ? N : 0);
    down("XCb", dft ? (restr ? d.XCa.p : d.XCb.p) : nullptr, dft ? N : 0, dft ? N : 0);
    down("Fa", d.Fa.p, N, N);
    down("Fb", restr ? d.Fa.p : d.Fb.p, N, N);
    down("Cb", restr ? d.Ca.p : d.Cb.p, N, N);
    res.mats["Ca"] = res.C;
    res.Eb.resize(N);
    HFG_HIP_CHECK(hipMemcpy(res.Eb.data(), restr ? d.Ea.p : d.Eb.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  }
  if (verbose) {
    printf("%-21s energy: % .16f\n", "Kinetic", res.Ekin);
    printf("%-21s energy: % .16f\n", "Nuclear attraction", res.Epot);
    printf("%-21s energy: % .16f\n", "Nuclear repulsion", res.Enucr);
    printf("%-21s energy: % .16f\n", "Coulomb", res.Ecoul);
    printf("%-21s energy: % .16f\n", "Exact exchange", res.Exx);
    printf("%-21s energy: % .16f\n", "Exchange-correlation", res.Exc);
    printf("%-21s energy: % .16f\n", "Total", res.Etot);
    printf("%-21s energy: % .16f\n", "Virial ratio", -res.Etot / res.Ekin);
  }
  return res;
}

}  // namespace hfg
