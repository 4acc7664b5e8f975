// Dense symmetric / symmetric-definite generalized eigensolver on gfx950.
//
// Replaces (reference, /root/reference):
//   scf::eig_gsym      src/general/scf_helpers.cpp:131-140   Forth = Sinvh^T F Sinvh; eig_sym; C = Sinvh C
//   scf::eig_gsym_sub  src/general/scf_helpers.cpp:142-186   the same per symmetry block + global sort
//   arma::eig_sym      (LAPACK dsyevd) as called there and in utils::invh, libhelfem/src/utils.cpp:172
//
// Pipeline per (symmetry) block, all blocks batched in the same launches (grid.y = block):
//   1. gather F(idx,idx), Sinvh(idx,cols)                       k_gather_block
//   2. Forth = X^T (F X)                                        FP64 MFMA GEMM (gemm.hip)
//   3. Householder tridiagonalisation, two launches per column   k_trd_gemv / k_trd_update
//        (symv-like sweep over the trailing matrix: bandwidth bound, matrix resident in L2/MALL)
//   4. tridiagonal eigenproblem                                  implicit QL, rotations logged then
//                                                                applied row-parallel (k_tql_*)
//   5. back-transformation Z <- H_0 ... H_{n-3} Z                k_backtransform (column-parallel, no
//                                                                inter-workgroup dependency at all)
//   6. C = X Z                                                   GEMM, then rank sort + scatter
#include "common.h"
#include "wave.h"
#include <cstdlib>
#include <cstring>

namespace hfg {

void gemm_tasklist_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_rect_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist_split2_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);
bool gemm_prefers_128(hfg_ctx *ctx, long tiles128);
bool tridiagonalize_takes_chain(int nblk, const int *ns);  // trdp.hip
void gemm_tasklist_acc_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN, bool tile64);
void gemm_mirror_lower_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxN);
void gemm_dev(hfg_ctx *ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double *A, int lda,
              const double *B, int ldb, double beta, double *C, int ldc);

void tridiag_dc_batch(hfg_ctx *ctx, int nblk, const int *ns, double *const *d, double *const *e, double *const *Z);
int dc_status(hfg_ctx *ctx);
void tridiagonalize_batch(hfg_ctx *ctx, int nblk, const int *ns, double *const *A, double *const *d, double *const *e,
                          double *const *tau);

constexpr int MAXB = 8;
struct EigBatch {
  int nblk;
  int n[MAXB];
  double *A[MAXB];    // n x n, lda = n; on exit Householder vectors below the subdiagonal
  double *d[MAXB];    // n
  double *e[MAXB];    // n
  double *tau[MAXB];  // n
  double *v[MAXB];    // n   current Householder vector (v[0]=1 stored explicitly)
  double *pp[MAXB];   // NCS x n partial gemv results
  double *dots[MAXB];  // partial v^T A v per gemv workgroup
  double *Z[MAXB];    // n x n eigenvectors
  double *rot[MAXB];  // rotation log (c,s pairs)
  int *sweeps[MAXB];  // (l, m, offset, count) per QL sweep; sweeps[0] = number of sweeps, [1]=status
  long rotcap[MAXB];
};

constexpr int TRD_NCS = 8;  // column slabs of the gemv

// ---- 1. gather -----------------------------------------------------------------------------------
__global__ void k_gather_block(const double *__restrict__ F, const double *__restrict__ S, int N,
                               const int64_t *__restrict__ rows, const int64_t *__restrict__ cols, int n,
                               double *__restrict__ Fb, double *__restrict__ Xb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (i >= n) return;
  Fb[(size_t)j * n + i] = F[(size_t)rows[j] * N + rows[i]];
  Xb[(size_t)j * n + i] = S[(size_t)cols[j] * N + rows[i]];
}

// column support of Sinvh on the block rows (scf_helpers.cpp:150-157): flag[c] = any(Sinvh(rows,c) != 0)
__global__ void k_col_support(const double *__restrict__ S, int N, const int64_t *__restrict__ rows, int n,
                              int *__restrict__ flag) {
  int c = blockIdx.x;
  int any = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (S[(size_t)c * N + rows[i]] != 0.0) any = 1;
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) flag[c] = any;
}

// ---- 3. tridiagonalisation -----------------------------------------------------------------------
// Householder vector of column k (LAPACK dlarfg), computed identically by every workgroup.
__device__ inline void householder_of_column(const double *__restrict__ A, int n, int k, double *vsh, double *red,
                                             double &tau, double &beta) {
  const int m = n - k - 1;  // length of x = A[k+1:n, k]
  const double *x = A + (size_t)k * n + k + 1;
  double s = 0.0;
  for (int i = 1 + threadIdx.x; i < m; i += blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  int nwave = blockDim.x / 64;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  double xn2 = 0.0;
  for (int w = 0; w < nwave; w++) xn2 += red[w];
  double alpha = x[0];
  double scale;
  if (xn2 == 0.0) {
    tau = 0.0;
    beta = alpha;
    scale = 0.0;
  } else {
    double nrm = sqrt(alpha * alpha + xn2);
    beta = (alpha >= 0.0) ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  for (int i = threadIdx.x; i < m; i += blockDim.x) vsh[i] = (i == 0) ? 1.0 : x[i] * scale;
  __syncthreads();
}

// S1: partial p = A22 v over (64-row slab) x (column slab); partial v^T A22 v
__global__ __launch_bounds__(256) void k_trd_gemv(EigBatch b, int k) {
  extern __shared__ double sh[];  // v[nmax], red[4*64 + 8]
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (k > n - 3) return;
  const int m = n - k - 1;
  const int nrs = (m + 63) / 64;
  const int rs = blockIdx.x / TRD_NCS, cs = blockIdx.x % TRD_NCS;
  if (rs >= nrs) return;
  double *A = b.A[blk];
  double *vsh = sh;
  double *red = sh + n;
  double tau, beta;
  householder_of_column(A, n, k, vsh, red, tau, beta);
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < m; i += blockDim.x) b.v[blk][i] = vsh[i];
    if (threadIdx.x == 0) {
      b.tau[blk][k] = tau;
      b.e[blk][k] = beta;
      b.d[blk][k] = A[(size_t)k * n + k];
    }
  }
  const int cchunk = (m + TRD_NCS - 1) / TRD_NCS;
  const int c0 = cs * cchunk, c1 = min(m, c0 + cchunk);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = rs * 64 + lane;  // local index in trailing block
  double acc = 0.0;
  if (row < m) {
    const double *a = A + (size_t)(k + 1) * n + (k + 1) + row;  // A22(row, 0)
    for (int c = c0 + wave; c < c1; c += 4) acc += a[(size_t)c * n] * vsh[c];
  }
  __syncthreads();
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0) {
    double p = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    if (row < m) b.pp[blk][(size_t)cs * n + row] = p;
    double dv = (row < m) ? p * vsh[row] : 0.0;
    dv = wave_sum(dv);
    if (lane == 0) b.dots[blk][blockIdx.x] = dv;
  }
}

// S2: w = tau*p - (tau^2/2)(v^T A v) v ; A22 -= v w^T + w v^T on a 64x64 tile; first tile stores v into column k
__global__ __launch_bounds__(256) void k_trd_update(EigBatch b, int k) {
  __shared__ double vr[64], vc[64], wr[64], wc[64];
  __shared__ double Ksh;
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  if (k > n - 3) return;
  const int m = n - k - 1;
  const int nt = (m + 63) / 64;
  if ((int)blockIdx.x >= nt * nt) return;
  const int tr = blockIdx.x % nt, tc = blockIdx.x / nt;
  double *A = b.A[blk];
  const double tau = b.tau[blk][k];
  if (threadIdx.x == 0) {
    const int nrs = (m + 63) / 64;
    double s = 0.0;
    for (int i = 0; i < nrs * TRD_NCS; i++) s += b.dots[blk][i];
    Ksh = 0.5 * tau * tau * s;  // (tau/2) * p^T v with p = tau A v
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    int which = threadIdx.x >> 6, l = threadIdx.x & 63;
    int idx = (which ? tc : tr) * 64 + l;
    double v = 0.0, w = 0.0;
    if (idx < m) {
      v = b.v[blk][idx];
      double p = 0.0;
      for (int cs = 0; cs < TRD_NCS; cs++) p += b.pp[blk][(size_t)cs * n + idx];
      w = tau * p - Ksh * v;
    }
    if (which) {
      vc[l] = v;
      wc[l] = w;
    } else {
      vr[l] = v;
      wr[l] = w;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = tr * 64 + lane;
  if (row < m) {
    double *a = A + (size_t)(k + 1) * n + (k + 1) + row;
    for (int c = wave; c < 64; c += 4) {
      int col = tc * 64 + c;
      if (col < m) a[(size_t)col * n] -= vr[lane] * wc[c] + wr[lane] * vc[c];
    }
  }
  if (blockIdx.x == 0) {
    // store v (without the implicit leading 1) below the subdiagonal of column k, beta on the subdiagonal
    for (int i = 1 + threadIdx.x; i < m; i += blockDim.x) A[(size_t)k * n + k + 1 + i] = b.v[blk][i];
  }
}

__global__ void k_trd_finish(EigBatch b) {
  int blk = blockIdx.x;
  if (threadIdx.x != 0) return;
  int n = b.n[blk];
  double *A = b.A[blk];
  if (n >= 2) {
    b.d[blk][n - 2] = A[(size_t)(n - 2) * n + (n - 2)];
    b.e[blk][n - 2] = A[(size_t)(n - 2) * n + (n - 1)];
    b.tau[blk][n - 2] = 0.0;
  }
  b.d[blk][n - 1] = A[(size_t)(n - 1) * n + (n - 1)];
  b.e[blk][n - 1] = 0.0;
}

// ---- 4. tridiagonal eigenproblem: implicit-shift QL, rotations logged --------------------------------
__global__ void k_tql_values(EigBatch b) {
  int blk = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int n = b.n[blk];
  double *d = b.d[blk], *e = b.e[blk];
  double *rot = b.rot[blk];
  int *sw = b.sweeps[blk];
  long nrot = 0;
  int nsw = 0;
  int status = 0;
  const long cap = b.rotcap[blk];
  for (int l = 0; l < n && !status; l++) {
    int iter = 0;
    int m;
    do {
      for (m = l; m + 1 < n; m++) {
        double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 300) {
          status = 1;
          break;
        }
        if (nrot + (m - l) > cap || 4 * (nsw + 1) + 2 >= 4 * 64 * n) {
          status = 2;
          break;
        }
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int cnt = 0;
        bool under = false;
        int i;
        for (i = m - 1; i >= l; i--) {
          double f = s * e[i], bb = c * e[i];
          r = hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            under = true;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * bb;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - bb;
          rot[2 * (nrot + cnt)] = c;
          rot[2 * (nrot + cnt) + 1] = s;
          cnt++;
        }
        sw[2 + 4 * nsw + 0] = l;
        sw[2 + 4 * nsw + 1] = m;
        sw[2 + 4 * nsw + 2] = (int)(nrot & 0x7fffffff);
        sw[2 + 4 * nsw + 3] = cnt;
        // offsets beyond 2^31 are reconstructed by the consumer as a running sum of cnt
        nsw++;
        nrot += cnt;
        if (under) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  sw[0] = nsw;
  sw[1] = status;
}

__global__ void k_set_identity(EigBatch b) {
  int blk = blockIdx.z;
  int n = b.n[blk];
  int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i < n && j < n) b.Z[blk][(size_t)j * n + i] = (i == j) ? 1.0 : 0.0;
}

// apply the logged rotations to the rows of Z; one thread per row, columns walked as tql2 does
__global__ void k_tql_apply(EigBatch b) {
  int blk = blockIdx.y;
  int n = b.n[blk];
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double *Z = b.Z[blk] + k;
  const double *rot = b.rot[blk];
  const int *sw = b.sweeps[blk];
  int nsw = sw[0];
  long off = 0;
  for (int q = 0; q < nsw; q++) {
    int m = sw[2 + 4 * q + 1], cnt = sw[2 + 4 * q + 3];
    double hi = Z[(size_t)m * n];
    int i = m - 1;
    for (int t = 0; t < cnt; t++, i--) {
      double c = rot[2 * (off + t)], s = rot[2 * (off + t) + 1];
      double lo = Z[(size_t)i * n];
      Z[(size_t)(i + 1) * n] = s * lo + c * hi;
      hi = c * lo - s * hi;
    }
    Z[(size_t)(i + 1) * n] = hi;
    off += cnt;
  }
}

// ---- 5. back-transformation ----------------------------------------------------------------------------
// Each wave owns one column z of Z (kept in registers, strided over the lanes) and applies
// H_{n-3} ... H_0 in turn:  z[k+1:] -= tau_k v_k (v_k^T z[k+1:]).
template <int NR>
__global__ __launch_bounds__(256) void k_backtransform(EigBatch b) {
  int blk = blockIdx.y;
  int n = b.n[blk];
  int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (col >= n) return;
  const double *A = b.A[blk];
  const double *tau = b.tau[blk];
  double *z = b.Z[blk] + (size_t)col * n;
  double zr[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    int i = lane + 64 * r;
    zr[r] = (i < n) ? z[i] : 0.0;
  }
  for (int k = n - 3; k >= 0; k--) {
    double t = tau[k];
    if (t == 0.0) continue;
    const double *v = A + (size_t)k * n;  // v_i for i>k+1 at v[i], v_{k+1}=1
    double s = 0.0;
    double vr[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
      int i = lane + 64 * r;
      double vi = 0.0;
      if (i == k + 1) vi = 1.0;
      else if (i > k + 1 && i < n) vi = v[i];
      vr[r] = vi;
      s += vi * zr[r];
    }
    s = wave_sum(s);
    s *= t;
#pragma unroll
    for (int r = 0; r < NR; r++) zr[r] -= s * vr[r];
  }
#pragma unroll
  for (int r = 0; r < NR; r++) {
    int i = lane + 64 * r;
    if (i < n) z[i] = zr[r];
  }
}

// generic (any n) version: column kept in LDS
__global__ __launch_bounds__(256) void k_backtransform_lds(EigBatch b) {
  extern __shared__ double sh[];  // 4 * n
  int blk = blockIdx.y;
  int n = b.n[blk];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int col = blockIdx.x * 4 + wave;
  if (col >= n) return;
  const double *A = b.A[blk];
  const double *tau = b.tau[blk];
  double *z = b.Z[blk] + (size_t)col * n;
  double *zs = sh + (size_t)wave * n;
  for (int i = lane; i < n; i += 64) zs[i] = z[i];
  for (int k = n - 3; k >= 0; k--) {
    double t = tau[k];
    if (t == 0.0) continue;
    const double *v = A + (size_t)k * n;
    double s = 0.0;
    for (int i = k + 1 + lane; i < n; i += 64) s += ((i == k + 1) ? 1.0 : v[i]) * zs[i];
    s = wave_sum(s);
    s *= t;
    for (int i = k + 1 + lane; i < n; i += 64) zs[i] -= s * ((i == k + 1) ? 1.0 : v[i]);
  }
  for (int i = lane; i < n; i += 64) z[i] = zs[i];
}

// ---- 6. sort + scatter ---------------------------------------------------------------------------------
// rank[i] = number of eigenvalues ordered before E[i] (ties by index).  Grid (ceil(n/256), segments): a workgroup
// compares its 256 values with one segment of E staged through LDS in chunks of 256; the segment counts are integer
// atomic adds into the zero-initialised rank array (order independent).
constexpr int RANK_SEG = 8;
__global__ __launch_bounds__(256) void k_rank(const double *__restrict__ E, int n, int *__restrict__ rank) {
  __shared__ double sE[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const double ei = (i < n) ? E[i] : 0.0;
  const int seg = (n + RANK_SEG - 1) / RANK_SEG;
  const int j0 = blockIdx.y * seg, j1 = min(n, j0 + seg);
  int r = 0;
  for (int c = j0; c < j1; c += 256) {
    __syncthreads();
    sE[threadIdx.x] = (c + (int)threadIdx.x < j1) ? E[c + threadIdx.x] : 0.0;
    __syncthreads();
    const int m = min(256, j1 - c);
    if (m == 256) {
#pragma unroll 16
      for (int jj = 0; jj < 256; jj++) {
        const double ej = sE[jj];
        r += (ej < ei) || (ej == ei && c + jj < i);
      }
    } else {
      for (int jj = 0; jj < m; jj++) {
        const double ej = sE[jj];
        r += (ej < ei) || (ej == ei && c + jj < i);
      }
    }
  }
  if (i < n && r) atomicAdd(&rank[i], r);
}
static void launch_rank(hfg_ctx *ctx, const double *E, int n, int *rank) {
  HFG_HIP_CHECK(hipMemsetAsync(rank, 0, sizeof(int) * n, ctx->stream));
  hipLaunchKernelGGL(k_rank, dim3((n + 255) / 256, RANK_SEG), dim3(256), 0, ctx->stream, E, n, rank);
}

// Cout(rows[i], rank[coff+j]) = Cb(i,j) ; Eout[rank[coff+j]] = Eb[j]   (rows==nullptr: identity)
__global__ void k_scatter_cols(const double *__restrict__ Cb, int nrow, int ncol, const int64_t *__restrict__ rows,
                               const int *__restrict__ rank, int coff, const double *__restrict__ Eb, int ldc,
                               double *__restrict__ Cout, double *__restrict__ Eout) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (i >= nrow) return;
  int c = rank[coff + j];
  size_t r = rows ? (size_t)rows[i] : (size_t)i;
  Cout[(size_t)c * ldc + r] = Cb[(size_t)j * nrow + i];
  if (i == 0) Eout[c] = Eb[j];
}

// -------------------------------------------------------------------------------------------------
// host-side drivers
// -------------------------------------------------------------------------------------------------
// ---- 5b. back-transformation through compact-WY blocks on the matrix cores --------------------------------------
// Q = H_0 ... H_{n-3} = Q_0 ... Q_{P-1},  Q_p = I - V_p T_p V_p^T over BT_KB consecutive reflectors (dlarft,
// forward / columnwise), applied from the last block to the first:  Z <- Z - (V_p T_p)(V_p^T Z).
constexpr int BT_KB = 64;

// explicit reflector matrix: Vx[g, j] = 1 at g = j+1, A[g, j] below, 0 above; columns j > n-3 are zero
__global__ void k_bt_extract(EigBatch b, double *const *__restrict__ Vx) {
  const int blk = blockIdx.z;
  const int n = b.n[blk];
  const int j = blockIdx.y;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || g >= n) return;
  double v = 0.0;
  if (j <= n - 3) {
    if (g == j + 1) v = 1.0;
    else if (g > j + 1) v = b.A[blk][(size_t)j * n + g];
  }
  Vx[blk][(size_t)j * n + g] = v;
}

// Two consecutive reflector blocks are applied in one pass over Z:  Q_p Q_{p+1} Z = Z - [VT_p | VT_{p+1}] [W0; W1] with
//   W1 = V_{p+1}^T Z,   W0 = V_p^T Z - (V_p^T VT_{p+1}) W1.
// The split-K product delivers S = [V_p | V_{p+1}]^T Z in BT_S slabs (2 BT_KB x n each, ld 2 BT_KB); this kernel adds the
// slabs and applies the coupling C = V_p^T VT_{p+1} (itself the sum of BT_GS partial products).  Workgroup = 4 columns.
constexpr int BT_PW = 2 * BT_KB;  // rows of the pair's W
constexpr int BT_WREP = 4;         // passes of 4 columns per workgroup of k_bt_wpair
__global__ __launch_bounds__(256) void k_bt_wpair(EigBatch b, double *const *__restrict__ Wpart, double *const *__restrict__ W,
                                                  double *const *__restrict__ Cc, int S, int GS, int npair, int pair,
                                                  int mirror) {
  __shared__ double sC[BT_KB][BT_KB + 1];
  __shared__ double s1[4][BT_KB];
  const int blk = blockIdx.y;
  const int n = b.n[blk];
  const int j0 = pair * BT_PW;
  if (j0 > n - 3) return;
  const int kb1 = min(BT_KB, n - 2 - (j0 + BT_KB));  // reflectors of the pair's second block (<= 0: there is none)
  const int tid = threadIdx.x, cl = tid >> 6, i = tid & 63;
  if (kb1 > 0) {
    // all loads of a slab in flight together (a rolled "load GS values, add, store" loop over the 16 elements of a
    // thread paid 16 memory round trips: 30 of this kernel's 37 us)
    constexpr int NE = BT_KB * BT_KB / 256;
    double c[NE];
#pragma unroll
    for (int u = 0; u < NE; u++) c[u] = 0.0;
    for (int sl = 0; sl < GS; sl++) {
      const double *src = Cc[blk] + ((size_t)sl * npair + pair) * BT_KB * BT_KB + tid;
#pragma unroll
      for (int u = 0; u < NE; u++) c[u] += src[256 * u];
    }
#pragma unroll
    for (int u = 0; u < NE; u++) {
      const int t = tid + 256 * u;
      sC[t % BT_KB][t / BT_KB] = c[u];  // C is column-major, ld BT_KB: element (row t % KB, col t / KB)
    }
  }
  // 16 columns per workgroup, four at a time: the 64 x 64 coupling (GS slabs to add) is staged once for all of them
  // (with 4 columns per workgroup its 128 KB of loads per workgroup made this kernel 43 us per pair)
  const size_t tot = (size_t)BT_PW * n;
  // the slab sums of all 16 columns first (independent loads, one memory round trip), then the coupling pass by pass
  double a0[BT_WREP], a1[BT_WREP];
#pragma unroll
  for (int rep = 0; rep < BT_WREP; rep++) {
    const int col = (blockIdx.x * BT_WREP + rep) * 4 + cl;
    a0[rep] = 0.0;
    a1[rep] = 0.0;
    if (col < n) {
      for (int sl = 0; sl < S; sl++) {
        a0[rep] += Wpart[blk][(size_t)sl * tot + (size_t)col * BT_PW + i];
        a1[rep] += Wpart[blk][(size_t)sl * tot + (size_t)col * BT_PW + BT_KB + i];
      }
    }
  }
  if (mirror) {
    // right application (X Q_p Q_p+1, worked on Y = X^T): rows 0..63 pass through, rows 64.. get the coupling,
    //   W1 = A1 - C^T A0   (C = V_p^T VT_p+1 as above)
    const int kb0 = min(BT_KB, n - 2 - j0);
#pragma unroll
    for (int rep = 0; rep < BT_WREP; rep++) {
      const int col = (blockIdx.x * BT_WREP + rep) * 4 + cl;
      __syncthreads();
      s1[cl][i] = (i < kb0) ? a0[rep] : 0.0;
      __syncthreads();
      if (col < n) {
        double r1 = 0.0;
        if (kb1 > 0 && i < kb1) {
          double corr = 0.0;
          for (int j = 0; j < kb0; j++) corr += sC[j][i] * s1[cl][j];
          r1 = a1[rep] - corr;
        }
        W[blk][(size_t)col * BT_PW + i] = a0[rep];
        W[blk][(size_t)col * BT_PW + BT_KB + i] = r1;
      }
    }
    return;
  }
#pragma unroll
  for (int rep = 0; rep < BT_WREP; rep++) {
    const int col = (blockIdx.x * BT_WREP + rep) * 4 + cl;
    __syncthreads();  // s1 of the previous four columns has been read (first pass: sC is complete)
    s1[cl][i] = (kb1 > 0 && i < kb1) ? a1[rep] : 0.0;
    __syncthreads();
    if (col < n) {
      double r0 = a0[rep];
      if (kb1 > 0) {
        double corr = 0.0;
        for (int j = 0; j < kb1; j++) corr += sC[i][j] * s1[cl][j];
        r0 -= corr;
      }
      W[blk][(size_t)col * BT_PW + i] = r0;
      W[blk][(size_t)col * BT_PW + BT_KB + i] = (kb1 > 0 && i < kb1) ? a1[rep] : 0.0;
    }
  }
}

// T_p from the Gram matrix G_p = V_p^T V_p and tau (one workgroup per block p of one matrix):
//   T(i,i) = tau_i,   T(0:i, i) = -tau_i T(0:i,0:i) G(0:i, i)
constexpr int BT_GS = 4;  // split-K slabs of the Gram products (K = rows of the reflector block, one 64 x 64 tile each)
__global__ __launch_bounds__(64) void k_bt_T(EigBatch b, double *const *__restrict__ G, double *const *__restrict__ T,
                                             int P) {
  __shared__ double sT[BT_KB][BT_KB + 1];
  __shared__ double sG[BT_KB][BT_KB + 1];
  __shared__ double sg[BT_KB];
  const int blk = blockIdx.y, p = blockIdx.x;
  const int n = b.n[blk];
  const int j0 = p * BT_KB;
  if (j0 > n - 3) return;
  const int kb = min(BT_KB, n - 2 - j0);
  double *Tp = T[blk] + (size_t)p * BT_KB * BT_KB;
  const int t = threadIdx.x;
  // Gram matrix = sum of the BT_GS partial products (fixed order), staged in LDS
  // (16 columns x BT_GS slabs of loads in flight at a time: the rolled "load, add, store" loop over the 64 columns paid
  // 64 memory round trips, 100 of this kernel's 115 us)
  for (int c0 = 0; c0 < BT_KB; c0 += 16) {
    double g[16];
#pragma unroll
    for (int u = 0; u < 16; u++) g[u] = 0.0;
#pragma unroll
    for (int sl = 0; sl < BT_GS; sl++) {
      const double *src = G[blk] + ((size_t)sl * P + p) * BT_KB * BT_KB + (size_t)c0 * BT_KB + t;
#pragma unroll
      for (int u = 0; u < 16; u++) g[u] += src[(size_t)u * BT_KB];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      sG[c0 + u][t] = g[u];  // column c of G
      sT[t][c0 + u] = 0.0;
    }
  }
  __shared__ double stau[BT_KB];
  stau[t] = (t < kb) ? b.tau[blk][j0 + t] : 0.0;  // one load per lane instead of one memory round trip per column
  __syncthreads();
  for (int i = 0; i < kb; i++) {
    const double ti = stau[i];
    sg[t] = (t < i) ? sG[i][t] : 0.0;
    __syncthreads();
    if (t < i) {
      // row t of the triangle times G(0:i, i).  Full-length and unrolled: T(t, k) is still zero for k < t and for the
      // columns >= i not built yet, sg is zero from i on; the rolled loop over k = t .. i-1 was bound by the latency of
      // its two LDS reads per term (2016 dependent terms per block: 85 of the kernel's 100 us)
      double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
      for (int k = 0; k < BT_KB; k += 4) {
        acc0 += sT[t][k] * sg[k];
        acc1 += sT[t][k + 1] * sg[k + 1];
        acc2 += sT[t][k + 2] * sg[k + 2];
        acc3 += sT[t][k + 3] * sg[k + 3];
      }
      sT[t][i] = -ti * ((acc0 + acc1) + (acc2 + acc3));
    }
    if (t == i) sT[i][i] = ti;
    __syncthreads();
  }
  for (int c = 0; c < BT_KB; c++) Tp[(size_t)c * BT_KB + t] = sT[t][c];
}

struct EigWork {
  bool split_full = false;  // this batch's full products run as two half-K workgroups per tile
  bool tile64 = false;
  std::vector<int64_t> asm_rows;  // eig_assemble_dev: the index list whose device copy is asm_rows_dev
  DevBuf<long long> asm_rows_dev;
  bool folded = false;      // the last batch formed Y = (X Q)^T beside the divide-and-conquer stage (bt_wy_fold_x)
  DevBuf<double> Y[MAXB];
  DevBuf<GemmTask> btslabR, btupdR;
  std::vector<GemmTask> h_btslabR, h_btupdR;
  DevBuf<double> A[MAXB], d[MAXB], e[MAXB], tau[MAXB], v[MAXB], pp[MAXB], dots[MAXB], Z[MAXB], rot[MAXB];
  DevBuf<int> sweeps[MAXB];
  DevBuf<int> ibuf1, ibuf2;
  DevBuf<GemmTask> gtasks, bttasks, btslab, btgram, btcpl, btupd;
  std::vector<GemmTask> h_gtasks, h_bttasks, h_btslab, h_btgram, h_btcpl, h_btupd;  // host copies (upload_cached)
  std::vector<double *> h_btptr;
  DevBuf<double> Vx[MAXB], G[MAXB], T[MAXB], VT[MAXB], Wb[MAXB], Wp[MAXB], Cc[MAXB];
  DevBuf<double *> btptr;
  bool used_dc = true;
  // block row / column index lists of the last eig_blocks_dev call, and what they were derived from (hfg_ctx_fix_sinvh)
  DevBuf<double> idx;  // raw storage for int64 rows + cols
  const double *sup_S = nullptr;
  unsigned long sup_gen = 0;
  int sup_N = 0;
  std::vector<int64_t> sup_ptr, sup_idx;
};
static std::map<hfg_ctx *, EigWork *> g_work;
static EigWork &work_for(hfg_ctx *ctx) {
  auto it = g_work.find(ctx);
  if (it != g_work.end()) return *it->second;
  EigWork *w = new EigWork();
  g_work[ctx] = w;
  return *w;
}
void eig_release(hfg_ctx *ctx) {
  auto it = g_work.find(ctx);
  if (it != g_work.end()) {
    delete it->second;
    g_work.erase(it);
  }
}

/// Eigen-decomposition of nblk symmetric matrices already in w.A[blk] (n x n); eigenvalues end up in
/// w.d[blk] (unsorted), eigenvectors in w.Z[blk].
// (1) buffers and task lists (host work + one stream synchronisation: done before the tridiagonalisation is queued),
// (2) the part that needs only the reflectors -- explicit V, Gram matrices, T, V T -- queued on the context's side
//     stream so that it runs beside the divide-and-conquer stage (which keeps few CUs busy),
// (3) the sweep over the reflector blocks on the main stream.
// up to 16 short vectors copied by ONE launch (eigenvalues into / out of the block slots: a device-to-device
// hipMemcpyAsync costs 15-20 us of stream time each, seven of them 0.13 ms per step)
struct CopySlices {
  const double *src[16];
  double *dst[16];
  int n[16];
};
__global__ void k_copy_slices(CopySlices c) {
  const int q = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n[q]; i += gridDim.x * blockDim.x) c.dst[q][i] = c.src[q][i];
}
static void copy_slices(hipStream_t s, const std::vector<const double *> &src, const std::vector<double *> &dst, const std::vector<int> &n) {
  for (size_t q0 = 0; q0 < src.size(); q0 += 16) {
    CopySlices c{};
    const int m = (int)std::min<size_t>(16, src.size() - q0);
    int nmax = 0;
    for (int q = 0; q < m; q++) {
      c.src[q] = src[q0 + q];
      c.dst[q] = dst[q0 + q];
      c.n[q] = n[q0 + q];
      nmax = std::max(nmax, n[q0 + q]);
    }
    if (nmax > 0) hipLaunchKernelGGL(k_copy_slices, dim3((nmax + 255) / 256, m), dim3(256), 0, s, c);
  }
}

// Y = X^T for every matrix of the batch (64 x 64 blocks through LDS)
struct TrPtrs {
  const double *src[MAXB];
  double *dst[MAXB];
  int n[MAXB];
};
__global__ __launch_bounds__(256) void k_transpose_batch(TrPtrs t) {
  __shared__ double tile[64][65];
  const int blk = blockIdx.z, n = t.n[blk];
  const int bi = blockIdx.x * 64, bj = blockIdx.y * 64;
  if (bi >= n || bj >= n) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) {
    const int gm = bi + tx, gn = bj + c;
    tile[c][tx] = (gm < n && gn < n) ? t.src[blk][(size_t)gn * n + gm] : 0.0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int gm = bj + tx, gn = bi + c;  // dst(gm, gn) = src(gn, gm)
    if (gm < n && gn < n) t.dst[blk][(size_t)gn * n + gm] = tile[tx][c];
  }
}

static void bt_wy_setup(hfg_ctx *ctx, EigWork &w, const EigBatch &b, int nblk, const int *ns, int nmax, bool fold = false) {
  hipStream_t s = ctx->stream;
  const int P = (nmax - 3) / BT_KB + 1;  // reflector blocks of the largest matrix
  const int NP = (P + 1) / 2;            // pairs of blocks, applied together (k_bt_wpair)
  constexpr int BT_S = 6;                // split-K slabs of the skinny product S = [V_p | V_p+1]^T Z
  std::vector<double *> ptrs(6 * (size_t)nblk);
  for (int k = 0; k < nblk; k++) {
    const int n = ns[k];
    w.Vx[k].resize((size_t)n * n);
    w.G[k].resize((size_t)BT_GS * P * BT_KB * BT_KB);
    w.T[k].resize((size_t)P * BT_KB * BT_KB);
    w.VT[k].resize((size_t)n * P * BT_KB);
    w.Wb[k].resize((size_t)BT_PW * n);
    w.Wp[k].resize((size_t)BT_S * BT_PW * n);
    w.Cc[k].resize((size_t)BT_GS * NP * BT_KB * BT_KB);
    ptrs[k] = w.Vx[k].p;
    ptrs[nblk + k] = w.G[k].p;
    ptrs[2 * nblk + k] = w.T[k].p;
    ptrs[3 * nblk + k] = w.Wp[k].p;
    ptrs[4 * nblk + k] = w.Wb[k].p;
    ptrs[5 * nblk + k] = w.Cc[k].p;
  }
  upload_cached(w.btptr, w.h_btptr, ptrs, s);
  GemmTask none;  // inactive
  none.A = none.B = nullptr;
  none.C = nullptr;
  none.M = none.N = none.K = 0;
  none.lda = none.ldb = none.ldc = 1;
  auto kb_of = [&](int k, int p) { return std::max(0, std::min(BT_KB, ns[k] - 2 - p * BT_KB)); };
  auto chunk_of = [](int len, int parts) { return ((len + parts - 1) / parts + 15) / 16 * 16; };

  // ---- per block p: Gram products (split over BT_GS row slabs) and VT = V T.  The VT columns of an odd block are
  //      formed from the first row of its PAIR on (V is zero there), so that [VT_p | VT_p+1] is one operand ----
  std::vector<GemmTask> gram((size_t)BT_GS * P * nblk, none), vt((size_t)P * nblk, none);
  for (int p = 0; p < P; p++)
    for (int k = 0; k < nblk; k++) {
      const int n = ns[k], j0 = p * BT_KB, kb = kb_of(k, p);
      if (kb <= 0) continue;
      const int r0 = j0 + 1, mr = n - r0;  // first row where the block's reflectors are non-zero
      const double *Vp = w.Vx[k].p + (size_t)j0 * n + r0;
      const int chunk = chunk_of(mr, BT_GS);
      for (int sl = 0; sl < BT_GS; sl++) {
        GemmTask g = none;
        const int k0 = sl * chunk, kk = std::max(0, std::min(chunk, mr - k0));
        g.A = g.B = Vp + (kk > 0 ? k0 : 0);
        g.C = w.G[k].p + ((size_t)sl * P + p) * BT_KB * BT_KB;
        g.tA = 1;
        g.M = g.N = kb;
        g.K = kk;  // K = 0 writes zeros
        g.lda = g.ldb = n;
        g.ldc = BT_KB;
        gram[((size_t)sl * P + p) * nblk + k] = g;
      }
      const int rp = (p & ~1) * BT_KB + 1;  // first row of the pair
      GemmTask g = none;
      g.A = w.Vx[k].p + (size_t)j0 * n + rp;
      g.B = w.T[k].p + (size_t)p * BT_KB * BT_KB;
      g.C = w.VT[k].p + (size_t)j0 * n + rp;
      g.M = n - rp;
      g.N = g.K = kb;
      g.lda = n;
      g.ldb = BT_KB;
      g.ldc = n;
      vt[(size_t)p * nblk + k] = g;
    }
  upload_cached(w.btgram, w.h_btgram, gram, s);
  upload_cached(w.bttasks, w.h_bttasks, vt, s);

  // ---- per pair: coupling C = V_p^T VT_p+1 (split over BT_GS slabs), S = [V_p | V_p+1]^T Z (BT_S slabs), the update ----
  std::vector<GemmTask> cpl((size_t)BT_GS * NP * nblk, none), slab((size_t)NP * BT_S * nblk, none), upd((size_t)NP * nblk, none);
  for (int g2 = 0; g2 < NP; g2++)
    for (int k = 0; k < nblk; k++) {
      const int n = ns[k], p0 = 2 * g2, j0 = p0 * BT_KB;
      const int kb0 = kb_of(k, p0), kb1 = (p0 + 1 < P) ? kb_of(k, p0 + 1) : 0;
      if (kb0 <= 0) continue;
      const int r0 = j0 + 1, mr = n - r0, kbp = kb0 + kb1;
      if (kb1 > 0) {
        const int r1 = j0 + BT_KB + 1, m1 = n - r1;  // rows where V_p+1 (and VT_p+1) are non-zero
        const int chunk = chunk_of(m1, BT_GS);
        for (int sl = 0; sl < BT_GS; sl++) {
          GemmTask c = none;
          const int k0 = sl * chunk, kk = std::max(0, std::min(chunk, m1 - k0));
          c.A = w.Vx[k].p + (size_t)j0 * n + r1 + (kk > 0 ? k0 : 0);
          c.B = w.VT[k].p + (size_t)(j0 + BT_KB) * n + r1 + (kk > 0 ? k0 : 0);
          c.C = w.Cc[k].p + ((size_t)sl * NP + g2) * BT_KB * BT_KB;
          c.tA = 1;
          c.M = kb0;
          c.N = kb1;
          c.K = kk;
          c.lda = c.ldb = n;
          c.ldc = BT_KB;
          cpl[((size_t)sl * NP + g2) * nblk + k] = c;
        }
      }
      const int chunk = chunk_of(mr, BT_S);
      for (int sl = 0; sl < BT_S; sl++) {
        GemmTask q = none;
        const int k0 = sl * chunk, kk = std::max(0, std::min(chunk, mr - k0));
        q.A = w.Vx[k].p + (size_t)j0 * n + r0 + (kk > 0 ? k0 : 0);
        q.B = b.Z[k] + r0 + (kk > 0 ? k0 : 0);
        q.C = w.Wp[k].p + (size_t)sl * BT_PW * n;
        q.tA = 1;
        q.M = kbp;
        q.N = n;
        q.K = kk;  // empty slab: K = 0 writes zeros
        q.lda = q.ldb = n;
        q.ldc = BT_PW;
        slab[((size_t)g2 * BT_S + sl) * nblk + k] = q;
      }
      GemmTask u = none;
      u.A = w.VT[k].p + (size_t)j0 * n + r0;
      u.B = w.Wb[k].p;
      u.C = b.Z[k] + r0;
      u.M = mr;
      u.N = n;
      u.K = kbp;
      u.lda = n;
      u.ldb = BT_PW;
      u.ldc = n;
      u.alpha = -1.0;
      u.beta = 1.0;
      upd[(size_t)g2 * nblk + k] = u;
    }
  upload_cached(w.btcpl, w.h_btcpl, cpl, s);
  upload_cached(w.btslab, w.h_btslab, slab, s);
  upload_cached(w.btupd, w.h_btupd, upd, s);
  if (fold) {
    // ---- the same pairs applied from the RIGHT to X, worked on Y = X^T (eig_blocks_dev: C = (X Q) Z):
    //   X Q_p Q_p+1 = X - B0 V_p^T - (B1 - B0 C) V_p+1^T,  B = X [VT_p | VT_p+1]   <=>
    //   Y <- Y - [V_p | V_p+1] [W0; W1],  W0 = VT_p^T Y,  W1 = VT_p+1^T Y - C^T W0
    // i.e. the tasks above with V and VT exchanged, on Y, pairs first to last; needs only the reflectors, not Z ----
    std::vector<GemmTask> slabR((size_t)NP * BT_S * nblk, none), updR((size_t)NP * nblk, none);
    for (int g2 = 0; g2 < NP; g2++)
      for (int k = 0; k < nblk; k++) {
        const int n = ns[k], p0 = 2 * g2, j0 = p0 * BT_KB;
        const int kb0 = kb_of(k, p0), kb1 = (p0 + 1 < P) ? kb_of(k, p0 + 1) : 0;
        if (kb0 <= 0) continue;
        const int r0 = j0 + 1, mr = n - r0, kbp = kb0 + kb1;
        const int chunk = chunk_of(mr, BT_S);
        for (int sl = 0; sl < BT_S; sl++) {
          GemmTask q = none;
          const int k0 = sl * chunk, kk = std::max(0, std::min(chunk, mr - k0));
          q.A = w.VT[k].p + (size_t)j0 * n + r0 + (kk > 0 ? k0 : 0);
          q.B = w.Y[k].p + r0 + (kk > 0 ? k0 : 0);
          q.C = w.Wp[k].p + (size_t)sl * BT_PW * n;
          q.tA = 1;
          q.M = kbp;
          q.N = n;
          q.K = kk;
          q.lda = q.ldb = n;
          q.ldc = BT_PW;
          slabR[((size_t)g2 * BT_S + sl) * nblk + k] = q;
        }
        GemmTask u = none;
        u.A = w.Vx[k].p + (size_t)j0 * n + r0;
        u.B = w.Wb[k].p;
        u.C = w.Y[k].p + r0;
        u.M = mr;
        u.N = n;
        u.K = kbp;
        u.lda = n;
        u.ldb = BT_PW;
        u.ldc = n;
        u.alpha = -1.0;
        u.beta = 1.0;
        updR[(size_t)g2 * nblk + k] = u;
      }
    upload_cached(w.btslabR, w.h_btslabR, slabR, s);
    upload_cached(w.btupdR, w.h_btupdR, updR, s);
  }
}

static bool bt_use_side() {
  static const bool v = (getenv("HELFEM_BT_SIDE") && atoi(getenv("HELFEM_BT_SIDE")) == 1);
  return v;
}

static void bt_wy_prepare(hfg_ctx *ctx, EigWork &w, const EigBatch &b, int nblk, int nmax, bool on_side = false) {
  const int P = (nmax - 3) / BT_KB + 1;
  // HELFEM_BT_SIDE=1 queues this part on a second stream beside the divide-and-conquer stage.  Measured: the 0.25 ms it
  // hides are outweighed by what a second live stream costs every launch on the framework's null stream (the
  // tridiagonalisation alone went from 17.4 to 19.1 ms), so the default keeps everything on the context's stream.
  // (Round 3: with X Q folded beside the divide-and-conquer stage -- bt_wy_fold_x, which needs these operands -- the set-up
  // goes to the side stream with it; the launch chain whose launches a second stream slowed is no longer the default.)
  const bool use_side = (bt_use_side() && !ctx->avoid_side) || on_side;
  hipStream_t main = ctx->stream, q = main;
  const bool prof = ctx->profiling;
  if (use_side) {
    q = ctx->side();
    HFG_HIP_CHECK(hipEventRecord(ctx->side_ev[0], main));  // reflectors and tau are complete
    HFG_HIP_CHECK(hipStreamWaitEvent(q, ctx->side_ev[0], 0));
    ctx->profiling = false;  // the profiling brackets are events of the main stream
    ctx->stream = q;
  }
  try {
    double *const *dptr = w.btptr.p;
    hipLaunchKernelGGL(k_bt_extract, dim3((nmax + 255) / 256, nmax, nblk), dim3(256), 0, q, b, dptr);
    gemm_tasklist64_dev(ctx, w.btgram.p, BT_GS * P * nblk, BT_KB, BT_KB);
    hipLaunchKernelGGL(k_bt_T, dim3(P, nblk), dim3(BT_KB), 0, q, b, dptr + nblk, dptr + 2 * nblk, P);
    gemm_tasklist64_dev(ctx, w.bttasks.p, P * nblk, nmax, BT_KB);
    gemm_tasklist64_dev(ctx, w.btcpl.p, BT_GS * ((P + 1) / 2) * nblk, BT_KB, BT_KB);
  } catch (...) {
    ctx->stream = main;
    ctx->profiling = prof;
    throw;
  }
  ctx->stream = main;
  ctx->profiling = prof;
  if (use_side) HFG_HIP_CHECK(hipEventRecord(ctx->side_ev[1], q));
}

static void bt_wy_apply(hfg_ctx *ctx, EigWork &w, const EigBatch &b, int nblk, int nmax) {
  hipStream_t s = ctx->stream;
  const int P = (nmax - 3) / BT_KB + 1, NP = (P + 1) / 2;
  constexpr int BT_S = 6;
  double *const *dptr = w.btptr.p;
  if (bt_use_side() && !ctx->avoid_side) HFG_HIP_CHECK(hipStreamWaitEvent(s, ctx->side_ev[1], 0));  // (not reached when X Q was folded)
  static const int acc_tile = getenv("HELFEM_ACC_TILE") ? atoi(getenv("HELFEM_ACC_TILE")) : 0;  // A/B runs: 64 or 128
  // pairs of reflector blocks, last to first: three launches and one read-modify-write of Z per 128 reflectors
  for (int g = NP - 1; g >= 0; g--) {
    gemm_tasklist64_dev(ctx, w.btslab.p + (size_t)g * BT_S * nblk, BT_S * nblk, BT_PW, nmax);
    // (summing the slabs inside the update's operand loads instead was measured: 2.73 -> 3.11 ms, six times the operand
    // traffic on every tile's critical path)
    hipLaunchKernelGGL(k_bt_wpair, dim3((nmax + 4 * BT_WREP - 1) / (4 * BT_WREP), nblk), dim3(256), 0, s, b, dptr + 3 * nblk, dptr + 4 * nblk,
                       dptr + 5 * nblk, BT_S, BT_GS, NP, g, 0);
    gemm_tasklist_acc_dev(ctx, w.btupd.p + (size_t)g * nblk, nblk, nmax, nmax, acc_tile != 128);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

// X Q on the side stream, beside the divide-and-conquer stage (which needs d and e only and keeps few CUs busy):
// the compact-WY set-up (bt_wy_prepare, same stream), Y = X^T, then the pairs of reflector blocks first to last.
// side_ev[0]: reflectors and tau are complete (main stream); side_ev[1]: Y = (X Q)^T is complete.
static void bt_wy_fold_x(hfg_ctx *ctx, EigWork &w, const EigBatch &b, int nblk, const int *ns, int nmax, const double *const *X) {
  const int P = (nmax - 3) / BT_KB + 1, NP = (P + 1) / 2;
  constexpr int BT_S = 6;
  hipStream_t main = ctx->stream, q = ctx->side();  // (bt_wy_prepare ran on q behind the tridiagonalisation: stream order)
  const bool prof = ctx->profiling;
  ctx->profiling = false;  // the profiling brackets are events of the main stream
  ctx->stream = q;
  try {
    TrPtrs t;
    for (int k = 0; k < MAXB; k++) {
      t.src[k] = k < nblk ? X[k] : nullptr;
      t.dst[k] = k < nblk ? w.Y[k].p : nullptr;
      t.n[k] = k < nblk ? ns[k] : 0;
    }
    hipLaunchKernelGGL(k_transpose_batch, dim3((nmax + 63) / 64, (nmax + 63) / 64, nblk), dim3(256), 0, q, t);
    double *const *dptr = w.btptr.p;
    for (int g = 0; g < NP; g++) {
      gemm_tasklist64_dev(ctx, w.btslabR.p + (size_t)g * BT_S * nblk, BT_S * nblk, BT_PW, nmax);
      hipLaunchKernelGGL(k_bt_wpair, dim3((nmax + 4 * BT_WREP - 1) / (4 * BT_WREP), nblk), dim3(256), 0, q, b, dptr + 3 * nblk, dptr + 4 * nblk,
                         dptr + 5 * nblk, BT_S, BT_GS, NP, g, 1);
      gemm_tasklist_acc_dev(ctx, w.btupdR.p + (size_t)g * nblk, nblk, nmax, nmax, true);
    }
  } catch (...) {
    ctx->stream = main;
    ctx->profiling = prof;
    throw;
  }
  ctx->stream = main;
  ctx->profiling = prof;
  HFG_HIP_CHECK(hipGetLastError());
  HFG_HIP_CHECK(hipEventRecord(ctx->side_ev[1], q));
}

/// foldX != nullptr (device pointers of the blocks' X, n x n, ld n): the caller wants C = X Q Z; then X Q is formed beside
/// the divide-and-conquer stage (w.Y = (X Q)^T, w.folded = true), w.Z keeps the tridiagonal matrix's eigenvectors and
/// the caller multiplies Y^T Z.  Otherwise (and when the compact-WY path does not apply) Z <- Q Z as before.
static void eig_sym_batch(hfg_ctx *ctx, EigWork &w, int nblk, const int *ns, const double *const *foldX = nullptr) {
  EigBatch b;
  b.nblk = nblk;
  int nmax = 0;
  for (int i = 0; i < nblk; i++) {
    int n = ns[i];
    nmax = std::max(nmax, n);
    b.n[i] = n;
    w.d[i].resize(n);
    w.e[i].resize(n);
    w.tau[i].resize(n);
    w.v[i].resize(n);
    w.pp[i].resize((size_t)TRD_NCS * n);
    w.dots[i].resize((size_t)TRD_NCS * ((n + 63) / 64) + 8);
    w.Z[i].resize((size_t)n * n);
    long cap = 3L * n * n + 1024;
    w.rot[i].resize(2 * (size_t)cap);
    w.sweeps[i].resize((size_t)4 * 64 * n + 8);
    b.rotcap[i] = cap;
    b.A[i] = w.A[i].p;
    b.d[i] = w.d[i].p;
    b.e[i] = w.e[i].p;
    b.tau[i] = w.tau[i].p;
    b.v[i] = w.v[i].p;
    b.pp[i] = w.pp[i].p;
    b.dots[i] = w.dots[i].p;
    b.Z[i] = w.Z[i].p;
    b.rot[i] = w.rot[i].p;
    b.sweeps[i] = w.sweeps[i].p;
  }
  hipStream_t s = ctx->stream;
  static const bool bt_column = (getenv("HELFEM_BT") && !strcmp(getenv("HELFEM_BT"), "column"));
  const bool bt_wy = !bt_column && nmax >= 4 * BT_KB;
  static const bool fold_on = !(getenv("HELFEM_BT_FOLD") && atoi(getenv("HELFEM_BT_FOLD")) == 0);
  if (tridiagonalize_takes_chain(nblk, ns)) {  // thousands of dependent launches ahead: no second stream beside them
    ctx->avoid_side = true;
    ctx->drop_side();
  }
  const bool fold = bt_wy && foldX != nullptr && fold_on && !ctx->avoid_side;
  w.folded = fold;
  if (fold)
    for (int i = 0; i < nblk; i++) w.Y[i].resize((size_t)ns[i] * ns[i]);
  if (bt_wy) bt_wy_setup(ctx, w, b, nblk, ns, nmax, fold);
  {
    ProfScope ps(ctx, "eig_tridiag");
    static const bool unblocked = (getenv("HELFEM_TRD") && !strcmp(getenv("HELFEM_TRD"), "unblocked"));
    if (unblocked) {
      // first-generation path (rank-2 update of the whole trailing matrix at every column), kept for A/B runs
      size_t shb = (size_t)(nmax + 4 * 64 + 8) * sizeof(double);
      if (shb > 64 * 1024)
        HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_trd_gemv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
      for (int k = 0; k <= nmax - 3; k++) {
        int m = nmax - k - 1;
        int nrs = (m + 63) / 64;
        hipLaunchKernelGGL(k_trd_gemv, dim3(nrs * TRD_NCS, nblk), dim3(256), shb, s, b, k);
        hipLaunchKernelGGL(k_trd_update, dim3(nrs * nrs, nblk), dim3(256), 0, s, b, k);
      }
      hipLaunchKernelGGL(k_trd_finish, dim3(nblk), dim3(64), 0, s, b);
    } else {
      double *Ap[MAXB], *dp[MAXB], *ep[MAXB], *tp[MAXB];
      for (int i = 0; i < nblk; i++) {
        Ap[i] = w.A[i].p;
        dp[i] = w.d[i].p;
        ep[i] = w.e[i].p;
        tp[i] = w.tau[i].p;
      }
      tridiagonalize_batch(ctx, nblk, ns, Ap, dp, ep, tp);
    }
  }
  if (bt_wy) bt_wy_prepare(ctx, w, b, nblk, nmax, fold);
  if (fold) bt_wy_fold_x(ctx, w, b, nblk, ns, nmax, foldX);
  {
    ProfScope ps(ctx, "eig_tridiag_solve");
    static const bool use_ql = (getenv("HELFEM_TRIDIAG") && !strcmp(getenv("HELFEM_TRIDIAG"), "ql"));
    w.used_dc = !use_ql;
    if (use_ql) {
      // reference implementation kept for cross-checks: implicit QL by one lane, rotations logged then applied
      hipLaunchKernelGGL(k_set_identity, dim3((nmax + 255) / 256, nmax, nblk), dim3(256), 0, s, b);
      hipLaunchKernelGGL(k_tql_values, dim3(nblk), dim3(64), 0, s, b);
      hipLaunchKernelGGL(k_tql_apply, dim3((nmax + 63) / 64, nblk), dim3(64), 0, s, b);
    } else {
      double *dp[MAXB], *ep[MAXB], *zp[MAXB];
      for (int i = 0; i < nblk; i++) {
        dp[i] = w.d[i].p;
        ep[i] = w.e[i].p;
        zp[i] = w.Z[i].p;
      }
      tridiag_dc_batch(ctx, nblk, ns, dp, ep, zp);
    }
  }
  {
    ProfScope ps(ctx, "eig_backtransform");
    if (fold) {
      HFG_HIP_CHECK(hipStreamWaitEvent(s, ctx->side_ev[1], 0));  // Y = (X Q)^T is complete
    } else if (bt_wy) {
      bt_wy_apply(ctx, w, b, nblk, nmax);
    } else {
    dim3 grid((nmax + 3) / 4, nblk);
    if (nmax <= 64 * 8)
      hipLaunchKernelGGL(k_backtransform<8>, grid, dim3(256), 0, s, b);
    else if (nmax <= 64 * 24)
      hipLaunchKernelGGL(k_backtransform<24>, grid, dim3(256), 0, s, b);
    else {
      size_t shb = (size_t)4 * nmax * sizeof(double);
      if (shb > 64 * 1024)
        HFG_HIP_CHECK(
            hipFuncSetAttribute((const void *)k_backtransform_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
      hipLaunchKernelGGL(k_backtransform_lds, grid, dim3(256), shb, s, b);
    }
    }
  }
  HFG_HIP_CHECK(hipGetLastError());
}

bool tridiagonalize_takes_chain(int nblk, const int *ns);  // trdp.hip
void trdp_check_status(hfg_ctx *ctx);        // trdp.hip
static void check_status(hfg_ctx *ctx, EigWork &w, int nblk) {
  if (w.used_dc) {
    if (dc_status(ctx) != 0) throw std::logic_error("Eigendecomposition failed!\n");
    trdp_check_status(ctx);  // the stream has just been synchronised: the status word of a persistent launch is home
    return;
  }
  for (int i = 0; i < nblk; i++) {
    int st[2];
    HFG_HIP_CHECK(hipMemcpyAsync(st, w.sweeps[i].p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (st[1] != 0) throw std::logic_error("Eigendecomposition failed!\n");
  }
}

// E (n), C (n x n) <- eig_sym(A) ; all device pointers
void eig_sym_dev(hfg_ctx *ctx, int n, const double *dA, double *dE, double *dC) {
  EigWork &w = work_for(ctx);
  w.A[0].resize((size_t)n * n + 2);
  HFG_HIP_CHECK(hipMemcpyAsync(w.A[0].p, dA, sizeof(double) * n * n, hipMemcpyDeviceToDevice, ctx->stream));
  eig_sym_batch(ctx, w, 1, &n);
  DevBuf<int> &rank = w.ibuf1;
  rank.resize(n + 8);
  launch_rank(ctx, w.d[0].p, n, rank.p);
  hipLaunchKernelGGL(k_scatter_cols, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, w.Z[0].p, n, n,
                     (const int64_t *)nullptr, rank.p, 0, w.d[0].p, n, dC, dE);
  check_status(ctx, w, 1);
}

// scf::eig_gsym: F N x N, Sinvh N x n  ->  E (n), C (N x n)
void eig_gsym_dev(hfg_ctx *ctx, int N, int n, const double *dF, const double *dS, double *dE, double *dC) {
  EigWork &w = work_for(ctx);
  w.A[0].resize((size_t)n * n + 2);
  DevBuf<double> &T1 = ctx->ws[0];
  DevBuf<double> &Ctmp = ctx->ws[1];
  T1.resize((size_t)N * n);
  Ctmp.resize((size_t)N * n);
  {
    ProfScope ps(ctx, "eig_reduce");
    gemm_dev(ctx, false, false, N, n, N, 1.0, dF, N, dS, N, 0.0, T1.p, N);
    gemm_dev(ctx, true, false, n, n, N, 1.0, dS, N, T1.p, N, 0.0, w.A[0].p, n);
  }
  eig_sym_batch(ctx, w, 1, &n);
  {
    ProfScope ps(ctx, "eig_backtransform");
    gemm_dev(ctx, false, false, N, n, n, 1.0, dS, N, w.Z[0].p, n, 0.0, Ctmp.p, N);
  }
  DevBuf<int> &rank = w.ibuf1;
  rank.resize(n + 8);
  launch_rank(ctx, w.d[0].p, n, rank.p);
  hipLaunchKernelGGL(k_scatter_cols, dim3((N + 255) / 256, n), dim3(256), 0, ctx->stream, Ctmp.p, N, n,
                     (const int64_t *)nullptr, rank.p, 0, w.d[0].p, N, dC, dE);
  check_status(ctx, w, 1);
}

// scf::eig_gsym_sub, phase 1: the symmetry blocks owned by this shard (block ib belongs to rank ib % nranks).
// dBlockBuf: nblk slots of (nmax*nmax + nmax) doubles, slot ib = [C block (n x n, ld n) | pad | eigenvalues (n) at
// offset nmax*nmax]; slots of blocks owned by other ranks are zeroed so that a sum all-reduce over ranks
// completes the buffer.
size_t eig_block_buf_size(int nblk, const int64_t *blk_ptr) {
  size_t nmax = 0;
  for (int ib = 0; ib < nblk; ib++) nmax = std::max<size_t>(nmax, blk_ptr[ib + 1] - blk_ptr[ib]);
  return (size_t)nblk * (nmax * nmax + nmax);
}

// nF Fock matrices at once (the two spins of an unrestricted iteration): their blocks join ONE batch -- the
// tridiagonalisation is a chain of dependent launches whose length does not depend on the number of blocks in it, so two
// spins cost about what one costs
void eig_blocks_multi_dev(hfg_ctx *ctx, int N, int nF, const double *const *dFs, const double *dS, int nblk, const int64_t *blk_ptr,
                          const int64_t *blk_idx, double *const *dBlockBufs);
static int eig_num_cus(hfg_ctx *ctx) {
  static int ncu = 0;
  if (!ncu) {
    hipDeviceProp_t prop;
    ncu = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return ncu;
}
void eig_blocks_dev(hfg_ctx *ctx, int N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                    const int64_t *blk_idx, double *dBlockBuf) {
  eig_blocks_multi_dev(ctx, N, 1, &dF, dS, nblk, blk_ptr, blk_idx, &dBlockBuf);
}
void eig_blocks_multi_dev(hfg_ctx *ctx, int N, int nF, const double *const *dFs, const double *dS, int nblk, const int64_t *blk_ptr,
                          const int64_t *blk_idx, double *const *dBlockBufs) {
  EigWork &w = work_for(ctx);
  hipStream_t s = ctx->stream;
  if (blk_ptr[nblk] != N) throw std::logic_error("Symmetry mismatch in eig_gsym_sub\n");
  // block row indices on the device
  DevBuf<double> &idxbuf = w.idx;  // raw storage for int64 rows + cols
  idxbuf.resize(2 * (size_t)N + 16);
  int64_t *drows = (int64_t *)idxbuf.p;
  int64_t *dcols = drows + N;
  for (int ib = 0; ib < nblk; ib++)
    if (blk_ptr[ib + 1] == blk_ptr[ib]) throw std::logic_error("eig_gsym_sub: empty symmetry block\n");
  // column support of every block (scf_helpers.cpp:150-157).  It depends on Sinvh and the blocks only: when the caller
  // has declared Sinvh fixed (hfg_ctx_fix_sinvh) the lists of the previous call are still on the device
  const bool cached = ctx->fixed_sinvh && ctx->fixed_sinvh == dS && w.sup_S == dS && w.sup_gen == ctx->fixed_gen &&
                      w.sup_N == N &&
                      w.sup_ptr.size() == (size_t)nblk + 1 && std::equal(w.sup_ptr.begin(), w.sup_ptr.end(), blk_ptr) &&
                      w.sup_idx.size() == (size_t)N && std::equal(w.sup_idx.begin(), w.sup_idx.end(), blk_idx);
  if (!cached) {
    w.sup_S = nullptr;
    HFG_HIP_CHECK(hipMemcpyAsync(drows, blk_idx, sizeof(int64_t) * N, hipMemcpyHostToDevice, s));
    DevBuf<int> &flag = w.ibuf2;
    flag.resize((size_t)N * nblk + 8);
    for (int ib = 0; ib < nblk; ib++) {
      int n = (int)(blk_ptr[ib + 1] - blk_ptr[ib]);
      hipLaunchKernelGGL(k_col_support, dim3(N), dim3(256), 0, s, dS, N, drows + blk_ptr[ib], n, flag.p + (size_t)ib * N);
    }
    std::vector<int> hflag((size_t)N * nblk);
    HFG_HIP_CHECK(hipMemcpyAsync(hflag.data(), flag.p, sizeof(int) * hflag.size(), hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int64_t> allcols;
    for (int ib = 0; ib < nblk; ib++) {
      int cnt = 0;
      for (int c = 0; c < N; c++)
        if (hflag[(size_t)ib * N + c]) {
          allcols.push_back(c);
          cnt++;
        }
      if (cnt != (int)(blk_ptr[ib + 1] - blk_ptr[ib]))
        throw std::logic_error("eig_gsym_sub: Sinvh is not block structured (columns with support != block size)\n");
    }
    HFG_HIP_CHECK(hipMemcpyAsync(dcols, allcols.data(), sizeof(int64_t) * N, hipMemcpyHostToDevice, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));  // allcols lives on this stack frame
    if (ctx->fixed_sinvh == dS) {
      w.sup_S = dS;
      w.sup_gen = ctx->fixed_gen;
      w.sup_N = N;
      w.sup_ptr.assign(blk_ptr, blk_ptr + nblk + 1);
      w.sup_idx.assign(blk_idx, blk_idx + N);
    }
  }

  size_t nmax = 0;
  for (int ib = 0; ib < nblk; ib++) nmax = std::max<size_t>(nmax, blk_ptr[ib + 1] - blk_ptr[ib]);
  const size_t slot = nmax * nmax + nmax;
  for (int f = 0; f < nF; f++) HFG_HIP_CHECK(hipMemsetAsync(dBlockBufs[f], 0, sizeof(double) * slot * nblk, s));

  // work items: (matrix, block) pairs; `mine` holds matrix * nblk + block
  std::vector<int> mine;
  for (int f = 0; f < nF; f++)
    for (int ib = 0; ib < nblk; ib++)
      if (ib % ctx->shard_n == ctx->shard_rank) mine.push_back(f * nblk + ib);
  DevBuf<double> &Fb = ctx->ws[5];
  DevBuf<double> &T1 = ctx->ws[0];
  DevBuf<double> &Xall = ctx->ws[4];
  Fb.resize(nmax * nmax * MAXB);
  T1.resize(nmax * nmax * MAXB);
  Xall.resize(nmax * nmax * MAXB);
  for (size_t c0 = 0; c0 < mine.size(); c0 += MAXB) {
    int nb = (int)std::min<size_t>(MAXB, mine.size() - c0);
    std::vector<int> ns(nb);
    // the three products of every block (F X, X^T (F X), X Z) go through one task-list launch each, so that the
    // blocks fill the chip together with 128 x 128 tiles
    std::vector<GemmTask> gt(4 * (size_t)nb);  // [3 nb + k]: the last product when X Q was folded (eig_sym_batch): (X Q) Z = Y^T Z
    const double *Xptr[MAXB];
    int nm = 0;
    for (int k = 0; k < nb; k++) {
      const int ib = mine[c0 + k] % nblk;
      double *dBlockBuf = dBlockBufs[mine[c0 + k] / nblk];
      int n = (int)(blk_ptr[ib + 1] - blk_ptr[ib]);
      ns[k] = n;
      nm = std::max(nm, n);
      w.A[k].resize((size_t)n * n + 2);  // + 2: the sweep's 16-byte row pairs may straddle the last element
      w.Z[k].resize((size_t)n * n);
      w.Y[k].resize((size_t)n * n);
      double *Xb = Xall.p + (size_t)k * nmax * nmax, *Fk = Fb.p + (size_t)k * nmax * nmax, *Tk = T1.p + (size_t)k * nmax * nmax;
      Xptr[k] = Xb;
      GemmTask g;
      g.M = g.N = g.K = n;
      g.lda = g.ldb = g.ldc = n;
      g.A = Fk;
      g.B = Xb;
      g.C = Tk;
      gt[k] = g;
      g.A = Xb;
      g.tA = 1;
      g.B = Tk;
      g.C = w.A[k].p;
      g.sym = 1;  // X^T (F X): the lower tiles are computed, the upper ones mirrored
      gt[nb + k] = g;
      g.sym = 0;
      g.tA = 0;
      g.A = Xb;
      g.B = w.Z[k].p;
      g.C = dBlockBuf + (size_t)ib * slot;
      gt[2 * nb + k] = g;
      g.A = w.Y[k].p;
      g.tA = 1;
      gt[3 * nb + k] = g;
    }
    upload_cached(w.gtasks, w.h_gtasks, gt, s);
    {
      ProfScope ps(ctx, "eig_reduce");
      for (int k = 0; k < nb; k++) {
        const int ib = mine[c0 + k] % nblk;
        const double *dF = dFs[mine[c0 + k] / nblk];
        int n = ns[k];
        hipLaunchKernelGGL(k_gather_block, dim3((n + 255) / 256, n), dim3(256), 0, s, dF, dS, N, drows + blk_ptr[ib],
                           dcols + blk_ptr[ib], n, Fb.p + (size_t)k * nmax * nmax, Xall.p + (size_t)k * nmax * nmax);
      }
      static const bool rect = getenv("HELFEM_GEMM_RECT") && atoi(getenv("HELFEM_GEMM_RECT"));
      // two workgroups per tile (split K) when the batch's tiles would not fill the chip evenly: more than half, fewer
      // than all of the 2 x CU slots; HELFEM_GEMM_SPLITK = 0 / 1 forces it off / on
      static const int force_split = getenv("HELFEM_GEMM_SPLITK") ? atoi(getenv("HELFEM_GEMM_SPLITK")) : -1;
      long full_tiles = 0, low_tiles = 0;
      for (int k = 0; k < nb; k++) {
        const long t1 = (ns[k] + 127) / 128;
        full_tiles += t1 * t1;
        low_tiles += t1 * (t1 + 1) / 2;
      }
      const int slots = 2 * eig_num_cus(ctx);
      const bool split_full = force_split >= 0 ? force_split != 0 : (full_tiles < slots && nm >= 256);
      const bool split_low = force_split >= 0 ? force_split != 0 : (low_tiles < slots && nm >= 256);
      w.split_full = split_full;
      ProfScope pp3(ctx, "eig_products");  // the N^3 products alone (bench.py: MFMA fraction of the tile engine)
      // 64 x 64 tiles unless the batch's 128 x 128 tiles would fill whole rounds of the chip (gemm_prefers_128): 386 large
      // tiles on 512 slots ran at 34 TFLOP/s as two half-K workgroups each, 1452 small ones at 37 and without zeroing C.
      // HELFEM_GEMM_SPLITK=1 / HELFEM_GEMM_TILE=128 bring the large tiles back (checkers).
      const bool tile64 = force_split < 0 && !gemm_prefers_128(ctx, full_tiles);
      w.tile64 = tile64;
      if (tile64) {
        gemm_tasklist64_dev(ctx, w.gtasks.p, nb, nm, nm);
        gemm_tasklist64_dev(ctx, w.gtasks.p + nb, nb, nm, nm);
        gemm_mirror_lower_dev(ctx, w.gtasks.p + nb, nb, nm);
      } else {
      if (split_full) {
        for (int k = 0; k < nb; k++) HFG_HIP_CHECK(hipMemsetAsync(T1.p + (size_t)k * nmax * nmax, 0, sizeof(double) * (size_t)ns[k] * ns[k], s));
        gemm_tasklist_split2_dev(ctx, w.gtasks.p, nb, nm, nm);
      } else if (rect) gemm_tasklist_rect_dev(ctx, w.gtasks.p, nb, nm, nm);
      else gemm_tasklist_dev(ctx, w.gtasks.p, nb, nm, nm);
      if (split_low) {
        for (int k = 0; k < nb; k++) HFG_HIP_CHECK(hipMemsetAsync(w.A[k].p, 0, sizeof(double) * (size_t)ns[k] * ns[k], s));
        gemm_tasklist_split2_dev(ctx, w.gtasks.p + nb, nb, nm, nm);
      } else
        gemm_tasklist_dev(ctx, w.gtasks.p + nb, nb, nm, nm);      // lower tiles of X^T (F X) only (GemmTask::sym)
      gemm_mirror_lower_dev(ctx, w.gtasks.p + nb, nb, nm);        // the tridiagonalisation sweeps the full square
      }
    }
    eig_sym_batch(ctx, w, nb, ns.data(), Xptr);
    {
      ProfScope ps(ctx, "eig_backtransform");
      static const bool rect = getenv("HELFEM_GEMM_RECT") && atoi(getenv("HELFEM_GEMM_RECT"));
      {
        ProfScope pp3(ctx, "eig_products");
        const GemmTask *last = w.gtasks.p + (w.folded ? 3 : 2) * (size_t)nb;
        if (w.tile64) gemm_tasklist64_dev(ctx, last, nb, nm, nm);
        else if (w.split_full) gemm_tasklist_split2_dev(ctx, last, nb, nm, nm);  // the block slots were zeroed above
        else if (rect) gemm_tasklist_rect_dev(ctx, last, nb, nm, nm);
        else gemm_tasklist_dev(ctx, last, nb, nm, nm);
      }
      std::vector<const double *> csrc;
      std::vector<double *> cdst;
      std::vector<int> cn;
      for (int k = 0; k < nb; k++) {
        const int ib = mine[c0 + k] % nblk;
        double *slotp = dBlockBufs[mine[c0 + k] / nblk] + (size_t)ib * slot;
        csrc.push_back(w.d[k].p);
        cdst.push_back(slotp + nmax * nmax);
        cn.push_back(ns[k]);
      }
      copy_slices(s, csrc, cdst, cn);
    }
    check_status(ctx, w, nb);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

// phase 2: global sort of all eigenvalues and scatter of the block eigenvectors into C (N x N)
void eig_assemble_dev(hfg_ctx *ctx, int N, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                      const double *dBlockBuf, double *dE, double *dC) {
  EigWork &w = work_for(ctx);
  hipStream_t s = ctx->stream;
  ProfScope ps(ctx, "scatter");
  // the row indices of the blocks: uploaded once per index list (an SCF run passes the same list every iteration; the copy
  // from pageable host memory cost 25 us of stream time per call)
  if (w.asm_rows.size() != (size_t)N || !std::equal(w.asm_rows.begin(), w.asm_rows.end(), blk_idx)) {
    w.asm_rows.assign(blk_idx, blk_idx + N);
    w.asm_rows_dev.resize((size_t)N + 2);
    HFG_HIP_CHECK(hipMemcpyAsync(w.asm_rows_dev.p, w.asm_rows.data(), sizeof(int64_t) * N, hipMemcpyHostToDevice, s));
  }
  int64_t *drows = (int64_t *)w.asm_rows_dev.p;
  size_t nmax = 0;
  for (int ib = 0; ib < nblk; ib++) nmax = std::max<size_t>(nmax, blk_ptr[ib + 1] - blk_ptr[ib]);
  const size_t slot = nmax * nmax + nmax;
  DevBuf<double> &Etmp = ctx->ws[3];
  Etmp.resize(N);
  {
    std::vector<const double *> csrc;
    std::vector<double *> cdst;
    std::vector<int> cn;
    for (int ib = 0; ib < nblk; ib++) {
      csrc.push_back(dBlockBuf + (size_t)ib * slot + nmax * nmax);
      cdst.push_back(Etmp.p + blk_ptr[ib]);
      cn.push_back((int)(blk_ptr[ib + 1] - blk_ptr[ib]));
    }
    copy_slices(s, csrc, cdst, cn);
  }
  DevBuf<int> &rank = w.ibuf1;
  rank.resize(N + 8);
  HFG_HIP_CHECK(hipMemsetAsync(dC, 0, sizeof(double) * (size_t)N * N, s));
  launch_rank(ctx, Etmp.p, N, rank.p);
  for (int ib = 0; ib < nblk; ib++) {
    int n = (int)(blk_ptr[ib + 1] - blk_ptr[ib]);
    hipLaunchKernelGGL(k_scatter_cols, dim3((n + 255) / 256, n), dim3(256), 0, s, dBlockBuf + (size_t)ib * slot, n, n,
                       drows + blk_ptr[ib], rank.p, (int)blk_ptr[ib], Etmp.p + blk_ptr[ib], N, dC, dE);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

// scf::eig_gsym_sub on one device
void eig_gsym_sub_dev(hfg_ctx *ctx, int N, const double *dF, const double *dS, int nblk, const int64_t *blk_ptr,
                      const int64_t *blk_idx, double *dE, double *dC) {
  int save_rank = ctx->shard_rank, save_n = ctx->shard_n;
  ctx->shard_rank = 0;
  ctx->shard_n = 1;
  try {
    DevBuf<double> &buf = ctx->ws[6];
    buf.resize(eig_block_buf_size(nblk, blk_ptr));
    eig_blocks_dev(ctx, N, dF, dS, nblk, blk_ptr, blk_idx, buf.p);
    eig_assemble_dev(ctx, N, nblk, blk_ptr, blk_idx, buf.p, dE, dC);
  } catch (...) {
    ctx->shard_rank = save_rank;
    ctx->shard_n = save_n;
    throw;
  }
  ctx->shard_rank = save_rank;
  ctx->shard_n = save_n;
}

/// column supports of the symmetry blocks of a block-structured Sinvh (scf_helpers.cpp:150-157): cols[blk_ptr[ib] ...] =
/// the columns with support on block ib's rows, ascending; throws like eig_blocks_dev when Sinvh is not block structured
void eig_block_supports(hfg_ctx *ctx, int N, const double *dS, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                        std::vector<int64_t> &cols) {
  hipStream_t s = ctx->stream;
  DevBuf<int64_t> drows;
  DevBuf<int> flag;
  drows.resize(N);
  flag.resize((size_t)N * nblk + 8);
  HFG_HIP_CHECK(hipMemcpyAsync(drows.p, blk_idx, sizeof(int64_t) * N, hipMemcpyHostToDevice, s));
  for (int ib = 0; ib < nblk; ib++) {
    const int n = (int)(blk_ptr[ib + 1] - blk_ptr[ib]);
    hipLaunchKernelGGL(k_col_support, dim3(N), dim3(256), 0, s, dS, N, drows.p + blk_ptr[ib], n, flag.p + (size_t)ib * N);
  }
  std::vector<int> hflag((size_t)N * nblk);
  HFG_HIP_CHECK(hipMemcpyAsync(hflag.data(), flag.p, sizeof(int) * hflag.size(), hipMemcpyDeviceToHost, s));
  HFG_HIP_CHECK(hipStreamSynchronize(s));
  cols.clear();
  for (int ib = 0; ib < nblk; ib++) {
    int cnt = 0;
    for (int c = 0; c < N; c++)
      if (hflag[(size_t)ib * N + c]) {
        cols.push_back(c);
        cnt++;
      }
    if (cnt != (int)(blk_ptr[ib + 1] - blk_ptr[ib]))
      throw std::logic_error("eig_gsym_sub: Sinvh is not block structured (columns with support != block size)\n");
  }
}

// scf::eig_gsym_sub for the two spin matrices of an unrestricted iteration in one batch (diatomic/main.cpp:936-958 calls
// it twice in a row with the same Sinvh and symmetry blocks)
void eig_gsym_sub_pair_dev(hfg_ctx *ctx, int N, const double *dFa, const double *dFb, const double *dS, int nblk, const int64_t *blk_ptr,
                           const int64_t *blk_idx, double *dEa, double *dCa, double *dEb, double *dCb) {
  int save_rank = ctx->shard_rank, save_n = ctx->shard_n;
  ctx->shard_rank = 0;
  ctx->shard_n = 1;
  try {
    const size_t sz = eig_block_buf_size(nblk, blk_ptr);
    DevBuf<double> &buf = ctx->ws[6];
    buf.resize(2 * sz);
    const double *Fs[2] = {dFa, dFb};
    double *bufs[2] = {buf.p, buf.p + sz};
    eig_blocks_multi_dev(ctx, N, 2, Fs, dS, nblk, blk_ptr, blk_idx, bufs);
    eig_assemble_dev(ctx, N, nblk, blk_ptr, blk_idx, bufs[0], dEa, dCa);
    eig_assemble_dev(ctx, N, nblk, blk_ptr, blk_idx, bufs[1], dEb, dCb);
  } catch (...) {
    ctx->shard_rank = save_rank;
    ctx->shard_n = save_n;
    throw;
  }
  ctx->shard_rank = save_rank;
  ctx->shard_n = save_n;
}

}  // namespace hfg
