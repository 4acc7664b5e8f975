// Divide-and-conquer eigensolver for symmetric tridiagonal matrices on gfx950 (Cuppen's method with
// Gu-Eisenstat eigenvectors), the tridiagonal stage of the dense eigensolve that the reference obtains
// from LAPACK dsyevd through arma::eig_sym (src/general/scf_helpers.cpp:135, libhelfem/src/utils.cpp:172).
//
// Why this shape on MI355X: the implicit QL/QR iteration is an O(n^2) chain of dependent scalar
// rotations (one lane busy, ~1.5 s at n=1400 measured), whereas every step of the rank-one-update
// merge is data parallel: sorting by rank counting, one wavefront per secular-equation root with
// wave-wide reductions for the pole sums, one wavefront per column of the eigenvector update, and
// an FP64 MFMA GEMM for Q <- Q U.  The only serial part is the deflation scan (O(n) simple steps
// in LDS by one lane per merge).  All problems of a batch (the symmetry blocks) and all nodes of a
// tree level run in the same launches; nothing is read back to the host.
//
// Algorithm (checked against a numpy prototype and LAPACK on random, graded, clustered, Wilkinson
// and decoupled matrices):
//   tear:   T = diag(T1,T2) + |rho| v v^T,  d[mid-1] -= |rho|, d[mid] -= |rho|
//   leaves: <= 32 rows, implicit QL by one wavefront (rows of Z on the lanes)
//   merge:  z = Q^T v / sqrt2, rho <- 2|rho|; sort; deflate (|rho z_j| <= tol, and close pairs by a
//           Givens rotation, tol = 8 eps max(|d|,|z|) as LAPACK dlaed2); secular roots
//           lam_i = d[org_i] + mu_i by the two-pole "middle way" iteration with bisection safeguard;
//           zhat_j^2 = prod_i (lam_i - d_j) / (rho prod_{i != j} (d_i - d_j));  U_ji = zhat_j / (d_j - lam_i)
//           (columns normalised); Q <- [Q_nd U | Q_deflated] sorted by eigenvalue.
#include "common.h"
#include <cstring>
#include "wave.h"

namespace hfg {

constexpr int DC_LEAF = 16;  // measured at 3 x n ~ 1400: 32 -> 1.96 ms, 16 -> 1.81 ms, 8 -> 1.85 ms for the whole stage
constexpr int DC_MAXB = 8;
#define DC_EPS 2.220446049250313e-16

struct DCNode {
  int blk, lo, mid, hi;
};

struct DCBatch {
  int n[DC_MAXB];
  double *d[DC_MAXB];   // in: torn diagonal / leaf+merge eigenvalues (ping)
  double *d2[DC_MAXB];  // pong
  double *e[DC_MAXB];
  double *Qa[DC_MAXB], *Qb[DC_MAXB];  // eigenvector ping-pong, ld = n
  double *U[DC_MAXB], *Qg[DC_MAXB], *Qn[DC_MAXB];
  double *Ds[DC_MAXB], *zs[DC_MAXB], *dnd[DC_MAXB], *znd[DC_MAXB], *mu[DC_MAXB], *lam[DC_MAXB], *zhat[DC_MAXB];
  double *rotc[DC_MAXB], *rots[DC_MAXB];
  int *src[DC_MAXB], *flag[DC_MAXB], *nd[DC_MAXB], *org[DC_MAXB], *roti[DC_MAXB], *rotj[DC_MAXB], *rank[DC_MAXB];
  int *ndpos[DC_MAXB];  // position of a non-deflated sorted slot in the nd list (written by k_dc_prepare for k_dc_rank)
};

// ---- tear ------------------------------------------------------------------------------------------
__global__ void k_dc_tear(DCBatch b, const DCNode *__restrict__ nodes, int nnodes, double *__restrict__ rho) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnodes) return;
  DCNode nd = nodes[i];
  double r = b.e[nd.blk][nd.mid - 1];
  rho[i] = r;
  b.d[nd.blk][nd.mid - 1] -= fabs(r);
  b.d[nd.blk][nd.mid] -= fabs(r);
}

// ---- leaves: implicit QL with eigenvectors, one wavefront per leaf -------------------------------------
__global__ __launch_bounds__(64) void k_dc_leaf(DCBatch b, const DCNode *__restrict__ leaves, int nleaves,
                                                int *__restrict__ status) {
  __shared__ double ds[DC_LEAF], es[DC_LEAF], zt[DC_LEAF][DC_LEAF + 1];
  int li = blockIdx.x;
  if (li >= nleaves) return;
  DCNode lf = leaves[li];
  const int lo = lf.lo, s = lf.hi - lf.lo;
  const int lane = threadIdx.x;
  const int ld = b.n[lf.blk];
  double *d = b.d[lf.blk], *e = b.e[lf.blk];
  if (lane < s) {
    ds[lane] = d[lo + lane];
    es[lane] = (lane < s - 1) ? e[lo + lane] : 0.0;
    for (int c = 0; c < s; c++) zt[c][lane] = (c == lane) ? 1.0 : 0.0;
  }
  __syncthreads();
  // all lanes execute the same scalar recurrence (wave-uniform); lane k owns row k of Z
  for (int l = 0; l < s; l++) {
    int iter = 0;
    int m;
    do {
      for (m = l; m + 1 < s; m++) {
        double dd = fabs(ds[m]) + fabs(ds[m + 1]);
        if (fabs(es[m]) <= DC_EPS * dd) break;
      }
      if (m != l) {
        if (iter++ == 300) {
          if (lane == 0) status[0] = 1;
          break;
        }
        double g = (ds[l + 1] - ds[l]) / (2.0 * es[l]);
        double r = hypot(g, 1.0);
        g = ds[m] - ds[l] + es[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
        double sn = 1.0, c = 1.0, p = 0.0;
        bool under = false;
        for (int i = m - 1; i >= l; i--) {
          double f = sn * es[i], bb = c * es[i];
          r = hypot(f, g);
          __syncthreads();
          es[i + 1] = r;
          if (r == 0.0) {
            ds[i + 1] -= p;
            es[m] = 0.0;
            under = true;
            __syncthreads();
            break;
          }
          sn = f / r;
          c = g / r;
          g = ds[i + 1] - p;
          r = (ds[i] - g) * sn + 2.0 * c * bb;
          p = sn * r;
          __syncthreads();
          ds[i + 1] = g + p;
          g = c * r - bb;
          if (lane < s) {
            double fz = zt[i + 1][lane], zi = zt[i][lane];
            zt[i + 1][lane] = sn * zi + c * fz;
            zt[i][lane] = c * zi - sn * fz;
          }
          __syncthreads();
        }
        if (under) continue;
        __syncthreads();
        ds[l] -= p;
        es[l] = g;
        es[m] = 0.0;
        __syncthreads();
      }
    } while (m != l);
  }
  __syncthreads();
  // sort ascending (rank counting) and write out
  if (lane < s) {
    double v = ds[lane];
    int rk = 0;
    for (int j = 0; j < s; j++) rk += (ds[j] < v) || (ds[j] == v && j < lane);
    d[lo + rk] = v;
    double *Q = b.Qa[lf.blk];
    for (int k = 0; k < s; k++) Q[(size_t)(lo + rk) * ld + lo + k] = zt[lane][k];
  }
}

// ---- merge, step 1: z vector, merged sort, deflation ---------------------------------------------------
constexpr int DC_SQZ = 20;  // candidate-list entries per thread in the parallel squeeze of k_dc_prepare: merges up to 5120
__global__ __launch_bounds__(256) void k_dc_prepare(DCBatch b, const DCNode *__restrict__ nodes,
                                                    const double *__restrict__ rho_all, int node0,
                                                    int *__restrict__ kcount, int *__restrict__ nrot,
                                                    double *__restrict__ rho_eff, GemmTask *__restrict__ tasks,
                                                    double *__restrict__ gscratch, size_t gstride) {
  // work arrays of the node: LDS, or -- for merges whose arrays exceed it (n > ~5000) -- a slice of a global buffer
  // (same code: a workgroup's global stores are visible to its own threads behind __syncthreads)
  extern __shared__ double lds_[];
  double *sh = gscratch ? gscratch + (size_t)blockIdx.x * gstride : lds_;
  const int ni = node0 + blockIdx.x;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo, mid = nd.mid, hi = nd.hi;
  const int n1 = mid - lo, n = hi - lo;
  const int ld = b.n[blk];
  double *sD = sh;
  double *sz = sh + n;
  int *ssrc = (int *)(sh + 2 * n);
  int *sflag = ssrc + n;
  int *snd = sflag + n;
  int *rm = snd + n;                                                  // positions (in the candidate list) deflated by a rotation
  unsigned long long *rotw = (unsigned long long *)(rm + n);          // one bit per neighbouring candidate pair
  __shared__ int nrm_sh, candk_sh;
  __shared__ double red[8];
  __shared__ double tol_sh;
  __shared__ int k_sh;
  const double *d = b.d[blk];
  const double *Q = b.Qa[blk];
  const double rs = rho_all[ni];
  const double sgn = (rs >= 0.0) ? 1.0 : -1.0;
  const double rho = 2.0 * fabs(rs);
  const double isq2 = 0.7071067811865475244;

  double dmax = 0.0, zmax = 0.0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    double Dj = d[lo + j];
    double zj = (j < n1 ? Q[(size_t)(lo + j) * ld + (mid - 1)] : sgn * Q[(size_t)(lo + j) * ld + mid]) * isq2;
    // position in the merge of the two sorted lists (ties: first list first)
    int pos;
    if (j < n1) {
      int a = 0, c = n - n1;  // count of second-list elements < Dj
      while (a < c) {
        int h = (a + c) >> 1;
        if (d[mid + h] < Dj) a = h + 1;
        else c = h;
      }
      pos = j + a;
    } else {
      int a = 0, c = n1;  // count of first-list elements <= Dj
      while (a < c) {
        int h = (a + c) >> 1;
        if (d[lo + h] <= Dj) a = h + 1;
        else c = h;
      }
      pos = (j - n1) + a;
    }
    sD[pos] = Dj;
    sz[pos] = zj;
    ssrc[pos] = j;
    dmax = fmax(dmax, fabs(Dj));
    zmax = fmax(zmax, fabs(zj));
  }
  double m2 = fmax(dmax, zmax);
  m2 = wave_max(m2);
  zmax = wave_max(zmax);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = m2;
    red[4 + (threadIdx.x >> 6)] = zmax;
  }
  __syncthreads();
  const double mm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const double zz = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  const double tol = 8.0 * DC_EPS * mm;
  // ---- parallel pre-pass of the deflation scan.  The scan below is sequential only through its rotations (two
  // candidates with close poles are combined and the combination is the next candidate); they are rare.  So: flag the
  // small z in parallel, compact the remaining candidates, test every neighbouring pair of candidates with the
  // rotation criterion.  If no pair meets it the sequential scan would not have rotated either and its result is the
  // compacted list (exactly: the criterion of a pair only involves unmodified values then).  Otherwise the
  // sequential scan runs as before.  (0.45 ms of one-lane scans per eigensolve at 3 x 1400 before.) ----
  __shared__ int scan[256];
  int fast = 0, fast_k = 0;
  if (!(rho * zz <= tol)) {
    const int tid = threadIdx.x;
    const int chunk = (n + 255) / 256, c0 = tid * chunk, c1 = min(n, c0 + chunk);
    int cnt = 0;
    for (int j = c0; j < c1; j++) {
      const int small = (rho * fabs(sz[j]) <= tol) ? 1 : 0;
      sflag[j] = small;
      cnt += 1 - small;
    }
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int v = (tid >= off) ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    fast_k = scan[255];
    int r = scan[tid] - cnt;
    for (int j = c0; j < c1; j++)
      if (!sflag[j]) snd[r++] = j;
    __syncthreads();
    int rot = 0;
    for (int base = 0; base < fast_k; base += 256) {
      const int q = base + tid;
      int f = 0;
      if (q >= 1 && q < fast_k) {
        const int pj = snd[q - 1], j = snd[q];
        const double zj = sz[j], zp = sz[pj], t = sD[j] - sD[pj];
        f = (fabs(t * zj * zp) <= tol * (zj * zj + zp * zp)) ? 1 : 0;
      }
      const unsigned long long m = __ballot(f);
      if ((tid & 63) == 0) rotw[(base >> 6) + (tid >> 6)] = m;
      rot |= f;
    }
    fast = __syncthreads_or(rot) ? 0 : 1;
  }
  if (threadIdx.x == 0) {
    nrm_sh = 0;
    candk_sh = fast_k;
  }
  if (threadIdx.x == 0) {
    tol_sh = tol;
    // ---- serial deflation scan (LAPACK dlaed2 logic) ----
    int k = 0, nr = 0;
    int *ndl = snd;
    int *ri = b.roti[blk] + lo, *rj = b.rotj[blk] + lo;
    double *rc = b.rotc[blk] + lo, *rsn = b.rots[blk] + lo;
    if (rho * zz <= tol) {
      for (int j = 0; j < n; j++) sflag[j] = 1;
    } else if (fast) {
      k = fast_k;  // sflag and the candidate list are already in place
    } else if (n <= 256 * DC_SQZ) {
      // Rotations exist, but they are sparse (64-73 among ~620 candidates in the top merges of the 3 x 1400 bench
      // problem).  Only the chains that start at a flagged pair are sequential: the lane jumps from flagged pair to
      // flagged pair through the bit words, follows a chain while its running (modified) candidate keeps rotating with
      // the next one, and is back in step with the parallel flags after the first pair that does not rotate (both of
      // the next pair's members are unmodified then).  Deflated list positions are collected in rm and squeezed out
      // of the candidate list by all threads afterwards.
      const int K = fast_k, nw = (K + 63) >> 6;
      int q = 1;
      while (q < K) {
        int wq = q >> 6;
        unsigned long long m = rotw[wq] & (~0ull << (q & 63));
        while (m == 0 && ++wq < nw) m = rotw[wq];
        if (m == 0) break;
        q = (wq << 6) + __ffsll((long long)m) - 1;
        if (q >= K) break;
        int pq = q - 1, pj = snd[pq];
        double zp = sz[pj], dp = sD[pj];
        while (q < K) {
          const int j = snd[q];
          const double zj = sz[j], dj = sD[j], t = dj - dp;
          if (!(fabs(t * zj * zp) <= tol * (zj * zj + zp * zp))) {
            q++;
            break;
          }
          const double tau = hypot(zj, zp);
          const double c = zj / tau, sn = -zp / tau;
          sz[j] = tau;
          sz[pj] = 0.0;
          ri[nr] = pj;
          rj[nr] = j;
          rc[nr] = c;
          rsn[nr] = sn;
          rm[nr] = pq;
          nr++;
          const double tt = dp * c * c + dj * sn * sn;
          const double dn = dp * sn * sn + dj * c * c;
          sD[j] = dn;
          sD[pj] = tt;
          sflag[pj] = 1;
          pj = j;
          pq = q;
          zp = tau;
          dp = dn;
          q++;
        }
      }
      k = K - nr;
      nrm_sh = nr;
    } else {
      // One lane walks the sorted list (the chain through pj is sequential).  z and D of the running candidate pj stay
      // in registers, and the rotation test |t c s| <= tol is made on the unnormalised pair, |t c0 s0| <= tol (c0^2 +
      // s0^2), so that hypot and the two divisions are only paid when a rotation really happens (rare): this loop is
      // the critical path of the upper merge levels.
      int pj = -1;
      double zp = 0.0, dp = 0.0;
      for (int j = 0; j < n; j++) {
        const double zj = sz[j], dj = sD[j];
        if (rho * fabs(zj) <= tol) {
          sflag[j] = 1;
          continue;
        }
        sflag[j] = 0;
        if (pj < 0) {
          pj = j;
          zp = zj;
          dp = dj;
          continue;
        }
        const double t = dj - dp;
        if (fabs(t * zj * zp) <= tol * (zj * zj + zp * zp)) {
          const double tau = hypot(zj, zp);
          const double c = zj / tau, s = -zp / tau;
          sz[j] = tau;
          sz[pj] = 0.0;
          ri[nr] = pj;
          rj[nr] = j;
          rc[nr] = c;
          rsn[nr] = s;
          nr++;
          const double tt = dp * c * c + dj * s * s;
          const double dn = dp * s * s + dj * c * c;
          sD[j] = dn;
          sD[pj] = tt;
          sflag[pj] = 1;
          pj = j;
          zp = tau;
          dp = dn;
        } else {
          ndl[k++] = pj;
          pj = j;
          zp = zj;
          dp = dj;
        }
      }
      if (pj >= 0) ndl[k++] = pj;
    }
    k_sh = k;
    kcount[ni] = k;
    nrot[ni] = nr;
    rho_eff[ni] = rho;
    GemmTask t;
    t.A = b.Qg[blk] + (size_t)lo * ld + lo;
    t.B = b.U[blk] + (size_t)lo * ld + lo;
    t.C = b.Qn[blk] + (size_t)lo * ld + lo;
    t.M = n;
    t.N = k;
    t.K = k;
    t.lda = t.ldb = t.ldc = ld;
    tasks[blockIdx.x] = t;
  }
  __syncthreads();
  if (nrm_sh > 0) {  // squeeze the deflated positions out of the candidate list (uniform branch)
    const int K = candk_sh, nrm = nrm_sh;
    const int chunk = (K + 255) / 256, c0 = threadIdx.x * chunk;  // <= DC_SQZ entries per thread
    int vals[DC_SQZ], newi[DC_SQZ];
#pragma unroll
    for (int e = 0; e < DC_SQZ; e++) {
      const int q = c0 + e;
      newi[e] = -1;
      vals[e] = 0;
      if (e < chunk && q < K) {
        int a = 0, c = nrm;  // number of removed positions < q
        while (a < c) {
          const int h = (a + c) >> 1;
          if (rm[h] < q) a = h + 1;
          else c = h;
        }
        if (!(a < nrm && rm[a] == q)) {
          newi[e] = q - a;
          vals[e] = snd[q];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < DC_SQZ; e++)
      if (newi[e] >= 0) snd[newi[e]] = vals[e];
    __syncthreads();
  }
  const int k = k_sh;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    b.Ds[blk][lo + j] = sD[j];
    b.zs[blk][lo + j] = sz[j];
    b.src[blk][lo + j] = ssrc[j];
    b.flag[blk][lo + j] = sflag[j];
  }
  for (int c = threadIdx.x; c < k; c += blockDim.x) {
    int sidx = snd[c];
    b.ndpos[blk][lo + sidx] = c;
    b.nd[blk][lo + c] = sidx;
    b.dnd[blk][lo + c] = sD[sidx];
    b.znd[blk][lo + c] = sz[sidx];
  }
}

// ---- step 2: Givens rotations of the deflation on the columns of Q -----------------------------------
__global__ __launch_bounds__(256) void k_dc_rotate(DCBatch b, const DCNode *__restrict__ nodes, int node0,
                                                   const int *__restrict__ nrot) {
  const int ni = node0 + blockIdx.y;
  const DCNode nd = nodes[ni];
  const int nr = nrot[ni];
  if (nr == 0) return;
  const int lo = nd.lo, n = nd.hi - nd.lo, blk = nd.blk;
  const int ld = b.n[blk];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  // the rotations are staged 256 at a time (column offsets already resolved through src): read from global memory inside
  // the loop, every rotation paid the chain "index -> src[index] -> Q" of dependent round trips
  __shared__ int scp[256], scj[256];
  __shared__ double sc[256], ss[256];
  const int *src = b.src[blk] + lo;
  double *Q = b.Qa[blk] + lo + r;
  for (int t0 = 0; t0 < nr; t0 += 256) {
    const int t = t0 + threadIdx.x;
    if (t < nr) {
      scp[threadIdx.x] = lo + src[b.roti[blk][lo + t]];
      scj[threadIdx.x] = lo + src[b.rotj[blk][lo + t]];
      sc[threadIdx.x] = b.rotc[blk][lo + t];
      ss[threadIdx.x] = b.rots[blk][lo + t];
    }
    __syncthreads();
    if (r < n) {
      const int cnt = min(256, nr - t0);
      for (int u = 0; u < cnt; u++) {
        const int cp = scp[u], cj = scj[u];
        const double c = sc[u], s = ss[u];
        const double qp = Q[(size_t)cp * ld], qj = Q[(size_t)cj * ld];
        Q[(size_t)cp * ld] = c * qp + s * qj;
        Q[(size_t)cj * ld] = -s * qp + c * qj;
      }
    }
    __syncthreads();
  }
}

// ---- step 3: secular equation, one wavefront per root -------------------------------------------------

__global__ __launch_bounds__(256) void k_dc_secular(DCBatch b, const DCNode *__restrict__ nodes, int node0,
                                                    const int *__restrict__ kcount,
                                                    const double *__restrict__ rho_eff) {
  const int ni = node0 + blockIdx.y;
  const int k = kcount[ni];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= k) return;
  const int lane = threadIdx.x & 63;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo;
  const double *d = b.dnd[blk] + lo, *z = b.znd[blk] + lo;
  const double rho = rho_eff[ni];
  int org;
  double lob, hib, y;
  if (i < k - 1) {
    double gap = d[i + 1] - d[i];
    double midp = 0.5 * gap;
    double di = d[i];
    double s = 0.0;
    for (int j = lane; j < k; j += 64) s += z[j] * z[j] / ((d[j] - di) - midp);
    double fm = 1.0 + rho * wave_sum(s);
    if (fm > 0.0) {
      org = i;
      lob = 0.0;
      hib = midp;
    } else {
      org = i + 1;
      lob = -midp;
      hib = 0.0;
    }
  } else {
    org = k - 1;
    double s = 0.0;
    for (int j = lane; j < k; j += 64) s += z[j] * z[j];
    lob = 0.0;
    hib = rho * wave_sum(s);
  }
  const double dorg = d[org];
  y = 0.5 * (lob + hib);
  for (int it = 0; it < 100; it++) {
    double psi = 0.0, phi = 0.0, dpsi = 0.0, dphi = 0.0;
    for (int j = lane; j < k; j += 64) {
      double D = (d[j] - dorg) - y;
      double t = z[j] * z[j] / D;
      if (j <= i) {
        psi += t;
        dpsi += t / D;
      } else {
        phi += t;
        dphi += t / D;
      }
    }
    psi = rho * wave_sum(psi);
    phi = rho * wave_sum(phi);
    dpsi = rho * wave_sum(dpsi);
    dphi = rho * wave_sum(dphi);
    double w = 1.0 + psi + phi;
    double err = 16.0 * DC_EPS * (1.0 + fabs(psi) + fabs(phi)) + DC_EPS * fabs(y) * (dpsi + dphi);
    if (fabs(w) <= err) break;
    if (w < 0.0) lob = y;
    else hib = y;
    double eta;
    if (i < k - 1) {
      double Di = (d[i] - dorg) - y, Dq = (d[i + 1] - dorg) - y;
      double c = w - Di * dpsi - Dq * dphi;
      double a = (Di + Dq) * w - Di * Dq * (dpsi + dphi);
      double bb = Di * Dq * w;
      if (c == 0.0) eta = (a != 0.0) ? bb / a : 0.0;
      else {
        double disc = a * a - 4.0 * bb * c;
        disc = sqrt(fmax(disc, 0.0));
        eta = (a <= 0.0) ? (a - disc) / (2.0 * c) : 2.0 * bb / (a + disc);
      }
    } else {
      double Di = -y;  // (d[org]-dorg) - y
      double c = w - Di * dpsi;
      double S = Di * Di * dpsi;
      eta = (c != 0.0) ? Di + S / c : 0.0;
    }
    double yn = y + eta;
    if (!(yn > lob && yn < hib)) yn = 0.5 * (lob + hib);
    bool done = (yn == y) || (hib - lob <= 2.0 * DC_EPS * fmax(fabs(lob), fabs(hib)));
    y = yn;
    if (done) break;
  }
  if (lane == 0) {
    b.org[blk][lo + i] = org;
    b.mu[blk][lo + i] = y;
    b.lam[blk][lo + i] = dorg + y;
  }
}

// ---- step 4: Gu-Eisenstat z-hat, one wavefront per component ---------------------------------------------
__global__ __launch_bounds__(256) void k_dc_zhat(DCBatch b, const DCNode *__restrict__ nodes, int node0,
                                                 const int *__restrict__ kcount, const double *__restrict__ rho_eff) {
  const int ni = node0 + blockIdx.y;
  const int k = kcount[ni];
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= k) return;
  const int lane = threadIdx.x & 63;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo;
  const double *d = b.dnd[blk] + lo, *mu = b.mu[blk] + lo;
  const int *org = b.org[blk] + lo;
  const double dj = d[j];
  double prod = 1.0;
  for (int i = lane; i < k; i += 64) {
    double num = (d[org[i]] - dj) + mu[i];  // lam_i - d_j
    prod *= (i == j) ? num : num / (d[i] - dj);
  }
  prod = wave_prod(prod);
  if (lane == 0) {
    double zh = sqrt(fabs(prod) / rho_eff[ni]);
    b.zhat[blk][lo + j] = (b.znd[blk][lo + j] >= 0.0) ? zh : -zh;
  }
}

// ---- step 5: U(:,i) = zhat / (d - lam_i), normalised; one wavefront per column -----------------------------
__global__ __launch_bounds__(256) void k_dc_U(DCBatch b, const DCNode *__restrict__ nodes, int node0,
                                              const int *__restrict__ kcount) {
  const int ni = node0 + blockIdx.y;
  const int k = kcount[ni];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= k) return;
  const int lane = threadIdx.x & 63;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo;
  const int ld = b.n[blk];
  const double *d = b.dnd[blk] + lo, *zh = b.zhat[blk] + lo;
  const double dorg = d[b.org[blk][lo + i]], mui = b.mu[blk][lo + i];
  double s = 0.0;
  for (int j = lane; j < k; j += 64) {
    double v = zh[j] / ((d[j] - dorg) - mui);
    s += v * v;
  }
  s = 1.0 / sqrt(wave_sum(s));
  double *Ucol = b.U[blk] + (size_t)(lo + i) * ld + lo;
  for (int j = lane; j < k; j += 64) Ucol[j] = zh[j] / ((d[j] - dorg) - mui) * s;
}

// ---- step 6: gather the non-deflated columns of Q ---------------------------------------------------------
__global__ void k_dc_gather(DCBatch b, const DCNode *__restrict__ nodes, int node0,
                            const int *__restrict__ kcount) {
  const int ni = node0 + blockIdx.z;
  const int k = kcount[ni];
  const int c = blockIdx.y;
  if (c >= k) return;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo, n = nd.hi - nd.lo;
  const int ld = b.n[blk];
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double *Q = b.Qa[blk];
  int sidx = b.nd[blk][lo + c];
  int col = lo + b.src[blk][lo + sidx];
  b.Qg[blk][(size_t)(lo + c) * ld + lo + r] = Q[(size_t)col * ld + lo + r];
}

// ---- step 8: final order of the node's eigenvalues --------------------------------------------------------
__global__ __launch_bounds__(256) void k_dc_rank(DCBatch b, const DCNode *__restrict__ nodes, int node0,
                                                 const int *__restrict__ kcount) {
  // grid (node, slice): every workgroup stages the node's n values in LDS, then ranks its own slice of 32 of them, eight
  // lanes per value (each counts over an eighth of the list, the counts are summed by DPP)
  // (one workgroup per node left the top merges, n ~ 1400, with three busy CUs for 250 us; one thread per value and 256
  // values per workgroup 104 us: every thread walked the whole list)
  extern __shared__ double sh[];  // values[n]
  const int ni = node0 + blockIdx.x;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo, n = nd.hi - nd.lo;
  if ((int)blockIdx.y * 32 >= n) return;
  // value of sorted slot s: the new root if s is non-deflated (position ndpos[s] in the nd list), else the deflated Ds[s]
  const int *flag = b.flag[blk] + lo;
  for (int s = threadIdx.x; s < n; s += blockDim.x) {
    sh[s] = flag[s] ? b.Ds[blk][lo + s] : b.lam[blk][lo + b.ndpos[blk][lo + s]];
  }
  __syncthreads();
  const int s = blockIdx.y * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
  const int per = (n + 7) >> 3, j0 = part * per, j1 = min(n, j0 + per);
  const bool ok = s < n;
  const double v = ok ? sh[s] : 0.0;
  int rk = 0;
  if (ok) {
    int j = j0;
    for (; j + 8 <= j1; j += 8) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const double x = sh[j + u];
        rk += (x < v) || (x == v && j + u < s);
      }
    }
    for (; j < j1; j++) {
      const double x = sh[j];
      rk += (x < v) || (x == v && j < s);
    }
  }
  rk += __builtin_amdgcn_update_dpp(0, rk, 0xb1, 0xf, 0xf, true);   // lane ^ 1
  rk += __builtin_amdgcn_update_dpp(0, rk, 0x4e, 0xf, 0xf, true);   // lane ^ 2
  rk += __builtin_amdgcn_update_dpp(0, rk, 0x141, 0xf, 0xf, true);  // the other quad of the eight (row_half_mirror)
  if (ok && part == 0) {
    b.rank[blk][lo + s] = rk;
    b.d2[blk][lo + rk] = v;
  }
}

__global__ void k_dc_scatter(DCBatch b, const DCNode *__restrict__ nodes, int node0, const int *__restrict__ kcount) {
  const int ni = node0 + blockIdx.z;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo, n = nd.hi - nd.lo;
  const int s = blockIdx.y;
  if (s >= n) return;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int ld = b.n[blk];
  const int k = kcount[ni];
  const double *Q = b.Qa[blk];
  int rk = b.rank[blk][lo + s];
  double v;
  if (b.flag[blk][lo + s]) {
    v = Q[(size_t)(lo + b.src[blk][lo + s]) * ld + lo + r];
  } else {
    const int *ndl = b.nd[blk] + lo;
    int a = 0, c = k;
    while (a < c) {
      int h = (a + c) >> 1;
      if (ndl[h] < s) a = h + 1;
      else c = h;
    }
    v = b.Qn[blk][(size_t)(lo + a) * ld + lo + r];
  }
  b.Qb[blk][(size_t)(lo + rk) * ld + lo + r] = v;
}

// ---- step 10: copy the node's results back into (d, Qa) -------------------------------------------------------
__global__ void k_dc_copyback(DCBatch b, const DCNode *__restrict__ nodes, int node0) {
  const int ni = node0 + blockIdx.z;
  const DCNode nd = nodes[ni];
  const int blk = nd.blk, lo = nd.lo, n = nd.hi - nd.lo;
  const int c = blockIdx.y;
  if (c >= n) return;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int ld = b.n[blk];
  b.Qa[blk][(size_t)(lo + c) * ld + lo + r] = b.Qb[blk][(size_t)(lo + c) * ld + lo + r];
  if (c == 0) b.d[blk][lo + r] = b.d2[blk][lo + r];
}

// ---- batched FP64 MFMA GEMM over device-side task descriptors (C = A B, column-major) ----------------------------
void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);  // gemm.hip
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_dgemm_tasks(const GemmTask *__restrict__ tasks) {
  constexpr int BM = 64, BN = 64, BK = 16, PAD = 16;
  __shared__ double As[BK][BM + PAD];
  __shared__ double Bs[BK][BN + PAD];
  const GemmTask t = tasks[blockIdx.y];
  const int M = t.M, N = t.N, K = t.K;
  if (M <= 0 || N <= 0) return;
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  if ((int)blockIdx.x >= nbm * nbn) return;
  const int bm = (blockIdx.x % nbm) * BM, bn = (blockIdx.x / nbm) * BN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;
  double4_t acc[2][2];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int e = tid + 256 * r;
      int m = e % BM, kk = e / BM;
      int gm = bm + m, gk = k0 + kk;
      As[kk][m] = (gm < M && gk < K) ? t.A[(size_t)gk * t.lda + gm] : 0.0;
      int k2 = e % BK, n2 = e / BK;
      int gn = bn + n2, gk2 = k0 + k2;
      Bs[k2][n2] = (gn < N && gk2 < K) ? t.B[(size_t)gn * t.ldb + gk2] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double fa[2], fb[2];
      for (int i = 0; i < 2; i++) fa[i] = As[kk + l4][wm + i * 16 + l15];
      for (int j = 0; j < 2; j++) fb[j] = Bs[kk + l4][wn + j * 16 + l15];
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int r = 0; r < 4; r++) {
        int gm = bm + wm + i * 16 + l15, gn = bn + wn + j * 16 + l4 + 4 * r;
        if (gm < M && gn < N) t.C[(size_t)gn * t.ldc + gm] = acc[i][j][r];
      }
}

void gemm_tasks_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int max_tiles) {
  if (ntasks <= 0 || max_tiles <= 0) return;
  for (int t0 = 0; t0 < ntasks; t0 += 65535) {
    int nt = std::min(65535, ntasks - t0);
    hipLaunchKernelGGL(k_dgemm_tasks, dim3(max_tiles, nt), dim3(256), 0, ctx->stream, dtasks + t0);
  }
  HFG_HIP_CHECK(hipGetLastError());
}

// -------------------------------------------------------------------------------------------------
// host driver
// -------------------------------------------------------------------------------------------------
struct DCWork {
  DevBuf<double> prep_scratch;  // work arrays of k_dc_prepare for merges too large for LDS
  DevBuf<double> d2[DC_MAXB], Qb[DC_MAXB], U[DC_MAXB], Qg[DC_MAXB], Qn[DC_MAXB];
  DevBuf<double> vec[DC_MAXB];  // 9 double vectors of length n
  DevBuf<int> ivec[DC_MAXB];    // 7 int vectors of length n
  DevBuf<DCNode> nodes;
  DevBuf<double> rho, rho_eff;
  DevBuf<int> kcount, nrot, status;
  DevBuf<GemmTask> tasks;
  std::vector<int> key;  // block sizes the tree was built for
  std::vector<DCNode> hnodes;
  std::vector<int> level_off;  // nodes of height h>=1 are hnodes[level_off[h-1] .. level_off[h])
  std::vector<int> level_maxn;
  int nleaves = 0;
};
static std::map<hfg_ctx *, DCWork *> g_dc;
void dc_release(hfg_ctx *ctx) {
  auto it = g_dc.find(ctx);
  if (it != g_dc.end()) {
    delete it->second;
    g_dc.erase(it);
  }
}

static int build_tree(int blk, int lo, int hi, std::vector<std::vector<DCNode> > &byheight) {
  DCNode nd;
  nd.blk = blk;
  nd.lo = lo;
  nd.hi = hi;
  if (hi - lo <= DC_LEAF) {
    nd.mid = lo;
    if (byheight.empty()) byheight.resize(1);
    byheight[0].push_back(nd);
    return 0;
  }
  int mid = lo + (hi - lo) / 2;
  nd.mid = mid;
  int h = 1 + std::max(build_tree(blk, lo, mid, byheight), build_tree(blk, mid, hi, byheight));
  if ((int)byheight.size() <= h) byheight.resize(h + 1);
  byheight[h].push_back(nd);
  return h;
}

/// Eigen-decomposition of nblk symmetric tridiagonal matrices (d[blk], e[blk]); on return the eigenvalues are in
/// d[blk] (ascending) and the eigenvectors in Z[blk] (n x n, ld n).  d, e are overwritten.
void tridiag_dc_batch(hfg_ctx *ctx, int nblk, const int *ns, double *const *d, double *const *e, double *const *Z) {
  if (nblk > DC_MAXB) throw std::logic_error("tridiag_dc_batch: too many blocks");
  DCWork *wp;
  auto it = g_dc.find(ctx);
  if (it == g_dc.end()) {
    wp = new DCWork();
    g_dc[ctx] = wp;
  } else
    wp = it->second;
  DCWork &w = *wp;
  hipStream_t s = ctx->stream;
  std::vector<int> key(ns, ns + nblk);
  if (key != w.key) {
    std::vector<std::vector<DCNode> > byh;
    for (int b = 0; b < nblk; b++) build_tree(b, 0, ns[b], byh);
    w.hnodes.clear();
    w.level_off.clear();
    w.level_maxn.clear();
    w.hnodes.insert(w.hnodes.end(), byh[0].begin(), byh[0].end());
    w.nleaves = (int)byh[0].size();
    for (size_t h = 1; h < byh.size(); h++) {
      w.level_off.push_back((int)w.hnodes.size());
      int mx = 0;
      for (auto &nd : byh[h]) mx = std::max(mx, nd.hi - nd.lo);
      w.level_maxn.push_back(mx);
      w.hnodes.insert(w.hnodes.end(), byh[h].begin(), byh[h].end());
    }
    w.level_off.push_back((int)w.hnodes.size());
    w.nodes.resize(w.hnodes.size());
    HFG_HIP_CHECK(hipMemcpyAsync(w.nodes.p, w.hnodes.data(), sizeof(DCNode) * w.hnodes.size(), hipMemcpyHostToDevice, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    w.key = key;
    w.rho.resize(w.hnodes.size());
    w.rho_eff.resize(w.hnodes.size());
    w.kcount.resize(w.hnodes.size());
    w.nrot.resize(w.hnodes.size());
    w.tasks.resize(w.hnodes.size());
    w.status.resize(4);
  }
  DCBatch b;
  int nmax = 0;
  for (int i = 0; i < nblk; i++) {
    int n = ns[i];
    nmax = std::max(nmax, n);
    size_t nn = (size_t)n * n;
    w.d2[i].resize(n);
    w.Qb[i].resize(nn);
    w.U[i].resize(nn);
    w.Qg[i].resize(nn);
    w.Qn[i].resize(nn);
    w.vec[i].resize((size_t)9 * n);
    w.ivec[i].resize((size_t)8 * n);
    b.n[i] = n;
    b.d[i] = d[i];
    b.d2[i] = w.d2[i].p;
    b.e[i] = e[i];
    b.Qa[i] = Z[i];
    b.Qb[i] = w.Qb[i].p;
    b.U[i] = w.U[i].p;
    b.Qg[i] = w.Qg[i].p;
    b.Qn[i] = w.Qn[i].p;
    double *v = w.vec[i].p;
    b.Ds[i] = v;
    b.zs[i] = v + n;
    b.dnd[i] = v + 2 * n;
    b.znd[i] = v + 3 * n;
    b.mu[i] = v + 4 * n;
    b.lam[i] = v + 5 * n;
    b.zhat[i] = v + 6 * n;
    b.rotc[i] = v + 7 * n;
    b.rots[i] = v + 8 * n;
    int *iv = w.ivec[i].p;
    b.src[i] = iv;
    b.flag[i] = iv + n;
    b.nd[i] = iv + 2 * n;
    b.org[i] = iv + 3 * n;
    b.roti[i] = iv + 4 * n;
    b.rotj[i] = iv + 5 * n;
    b.rank[i] = iv + 6 * n;
    b.ndpos[i] = iv + 7 * n;
    HFG_HIP_CHECK(hipMemsetAsync(Z[i], 0, sizeof(double) * nn, s));
  }
  HFG_HIP_CHECK(hipMemsetAsync(w.status.p, 0, sizeof(int) * 4, s));
  const int ninternal = (int)w.hnodes.size() - w.nleaves;
  if (ninternal > 0)
    hipLaunchKernelGGL(k_dc_tear, dim3((ninternal + 255) / 256), dim3(256), 0, s, b, w.nodes.p + w.nleaves, ninternal,
                       w.rho.p + w.nleaves);
  hipLaunchKernelGGL(k_dc_leaf, dim3(w.nleaves), dim3(64), 0, s, b, w.nodes.p, w.nleaves, w.status.p);
  const int nlevels = (int)w.level_off.size() - 1;
  for (int h = 0; h < nlevels; h++) {
    const int node0 = w.level_off[h], nn = w.level_off[h + 1] - node0;
    const int mx = w.level_maxn[h];
    size_t shb = (size_t)mx * (2 * sizeof(double) + 4 * sizeof(int)) + ((size_t)mx / 64 + 2) * sizeof(unsigned long long);
    double *gscr = nullptr;
    size_t gstride = 0;
    if (shb > 150 * 1024) {  // beyond the LDS of a CU (merges of more than ~4800 values): global work arrays
      gstride = (shb + 7) / 8 + 8;
      w.prep_scratch.resize(gstride * nn);
      gscr = w.prep_scratch.p;
      shb = 0;
    } else if (shb > 64 * 1024)
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_dc_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
    hipLaunchKernelGGL(k_dc_prepare, dim3(nn), dim3(256), shb, s, b, w.nodes.p, w.rho.p, node0, w.kcount.p, w.nrot.p,
                       w.rho_eff.p, w.tasks.p, gscr, gstride);
    hipLaunchKernelGGL(k_dc_rotate, dim3((mx + 255) / 256, nn), dim3(256), 0, s, b, w.nodes.p, node0, w.nrot.p);
    hipLaunchKernelGGL(k_dc_secular, dim3((mx + 3) / 4, nn), dim3(256), 0, s, b, w.nodes.p, node0, w.kcount.p,
                       w.rho_eff.p);
    hipLaunchKernelGGL(k_dc_zhat, dim3((mx + 3) / 4, nn), dim3(256), 0, s, b, w.nodes.p, node0, w.kcount.p, w.rho_eff.p);
    hipLaunchKernelGGL(k_dc_U, dim3((mx + 3) / 4, nn), dim3(256), 0, s, b, w.nodes.p, node0, w.kcount.p);
    hipLaunchKernelGGL(k_dc_gather, dim3((mx + 255) / 256, mx, nn), dim3(256), 0, s, b, w.nodes.p, node0, w.kcount.p);
    // Q <- Q U of every node of the level: the tile engine of gemm.hip (16-byte staging loads, conflict-free LDS rows, two
    // workgroups per CU).  The small kernel below, which this call replaced, spent 68 % of its LDS cycles in bank
    // conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r03_gemm_sq_counters_bench_step.txt): its B tile is
    // read along k and stored transposed.  HELFEM_DC_GEMM=small keeps it as the checker.
    static const bool small_gemm = getenv("HELFEM_DC_GEMM") && !strcmp(getenv("HELFEM_DC_GEMM"), "small");
    if (small_gemm) {
      int tiles = ((mx + 63) / 64) * ((mx + 63) / 64);
      hipLaunchKernelGGL(k_dgemm_tasks, dim3(tiles, nn), dim3(256), 0, s, w.tasks.p);
    } else
      gemm_tasklist64_dev(ctx, w.tasks.p, nn, mx, mx);
    size_t shr = (size_t)mx * sizeof(double);
    if (shr > 64 * 1024)
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_dc_rank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shr));
    hipLaunchKernelGGL(k_dc_rank, dim3(nn, (mx + 31) / 32), dim3(256), shr, s, b, w.nodes.p, node0, w.kcount.p);
    hipLaunchKernelGGL(k_dc_scatter, dim3((mx + 255) / 256, mx, nn), dim3(256), 0, s, b, w.nodes.p, node0, w.kcount.p);
    hipLaunchKernelGGL(k_dc_copyback, dim3((mx + 255) / 256, mx, nn), dim3(256), 0, s, b, w.nodes.p, node0);
  }
  HFG_HIP_CHECK(hipGetLastError());
  static const bool dbg = getenv("HELFEM_DC_DBG") != nullptr;  // merge statistics: order, secular roots, rotations
  if (dbg) {
    std::vector<int> hk(w.hnodes.size()), hr(w.hnodes.size());
    HFG_HIP_CHECK(hipMemcpyAsync(hk.data(), w.kcount.p, sizeof(int) * hk.size(), hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipMemcpyAsync(hr.data(), w.nrot.p, sizeof(int) * hr.size(), hipMemcpyDeviceToHost, s));
    HFG_HIP_CHECK(hipStreamSynchronize(s));
    for (int h = std::max(0, nlevels - 3); h < nlevels; h++)
      for (int q = w.level_off[h]; q < w.level_off[h + 1]; q++)
        fprintf(stderr, "dc level %d node %d: n %d, roots %d, rotations %d\n", h, q, w.hnodes[q].hi - w.hnodes[q].lo, hk[q], hr[q]);
  }
}

int dc_status(hfg_ctx *ctx) {
  auto it = g_dc.find(ctx);
  if (it == g_dc.end()) return 0;
  int st = 0;
  HFG_HIP_CHECK(hipMemcpyAsync(&st, it->second->status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HFG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return st;
}

}  // namespace hfg
