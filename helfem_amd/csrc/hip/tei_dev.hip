// In-element two-electron integral tables built on the GPU:
// diatomic TwoDBasis::compute_tei, /root/reference/src/diatomic/basis.cpp:1166-1302 with quadrature::twoe_integral,
// src/diatomic/quadrature.cpp:22-123; atomic TwoDBasis::compute_tei, src/atomic/TwoDBasis.cpp:666-739 with
// quadrature::twoe_integral, libhelfem/src/quadrature.cpp:22-130 (one operand type, kernel r_<^L / r_>^{L+1}).
//
// The host keeps what is cheap and needs its special-function code (quadrature points, LIP products, P_L^M/Q_L^M
// values: TwoDBasis::tei_element_tables); the O(Nlm nq p^4) part runs here, per radial element:
//   inner_l[ilm][(ij), isub] = sum_{s <= (isub, .)} wP_l[ilm][s] bbs[(ij), s]        (prefix over the sub-intervals)
//   W_kl[ilm] = (bb0 diag(wQ_k[ilm])) inner_l[ilm]^T                               (p^2 x p^2, FP64 MFMA task list)
//   tei_kl = W_kl + W_lk^T   written straight into the padded device layout of hip/tables.h
// so the 1 GB of tables never exists on the host and never crosses PCIe.
#include "tables.h"
#include <vector>

namespace hfg {

void gemm_tasklist64_dev(hfg_ctx *ctx, const GemmTask *dtasks, int ntasks, int maxM, int maxN);

// inner[(l*Nlm + ilm)][isub][(ij)]: one workgroup per (ilm, l); thread = (ij)
__global__ __launch_bounds__(256) void k_tei_inner(const double *__restrict__ bbs, const double *__restrict__ wP, int Np,
                                                   int nq, int Nlm, double *__restrict__ inner) {
  extern __shared__ double sw[];  // wP of the channel, nq*nq
  const int ilm = blockIdx.x, l = blockIdx.y;
  const double *w = wP + ((size_t)l * Nlm + ilm) * nq * nq;
  for (int t = threadIdx.x; t < nq * nq; t += blockDim.x) sw[t] = w[t];
  __syncthreads();
  double *out = inner + ((size_t)l * Nlm + ilm) * (size_t)nq * Np;
  for (int k = threadIdx.x; k < Np; k += blockDim.x) {
    double acc = 0.0;
    for (int isub = 0; isub < nq; isub++) {
      const double *b = bbs + (size_t)isub * nq * Np + k;
      double s = 0.0;
      for (int q = 0; q < nq; q++) s += sw[isub * nq + q] * b[(size_t)q * Np];
      acc += s;
      out[(size_t)isub * Np + k] = acc;
    }
  }
}

// bq[(k*Nlm + ilm)][q][(ij)] = bb0[(ij), q] * wQ_k[ilm][q]
__global__ void k_tei_scale(const double *__restrict__ bb0, const double *__restrict__ wQ, int Np, int nq, int Nlm,
                            double *__restrict__ bq) {
  const int ilm = blockIdx.x, k = blockIdx.y;
  const double *w = wQ + ((size_t)k * Nlm + ilm) * nq;
  double *out = bq + ((size_t)k * Nlm + ilm) * (size_t)nq * Np;
  for (int t = threadIdx.x; t < nq * Np; t += blockDim.x) out[t] = bb0[t] * w[t / Np];
}

// tei[tt][ilm][e][(c)][(r)] (p^2 x p^2 padded, primitives shifted by lo) = W_kl + W_lk^T, tt = nty k + l
__global__ void k_tei_combine(const double *__restrict__ W, int Ni, int p, int lo, int Nlm, int E, int e, int nty,
                              double *__restrict__ tei) {
  const int ilm = blockIdx.x, tt = blockIdx.y;
  const int k = tt / nty, l = tt % nty;
  const int Np = Ni * Ni, pp = p * p;
  const double *Wkl = W + ((size_t)(k * nty + l) * Nlm + ilm) * (size_t)Np * Np;
  const double *Wlk = W + ((size_t)(l * nty + k) * Nlm + ilm) * (size_t)Np * Np;
  double *T = tei + (((size_t)tt * Nlm + ilm) * E + e) * (size_t)pp * pp;
  for (int t = threadIdx.x; t < pp * pp; t += blockDim.x) {
    int r = t % pp, c = t / pp;
    int ri = r % p - lo, rj = r / p - lo, ci = c % p - lo, cj = c / p - lo;
    double v = 0.0;
    if (ri >= 0 && ri < Ni && rj >= 0 && rj < Ni && ci >= 0 && ci < Ni && cj >= 0 && cj < Ni) {
      int rr = rj * Ni + ri, cc = cj * Ni + ci;
      v = Wkl[(size_t)cc * Np + rr] + Wlk[(size_t)rr * Np + cc];
    }
    T[t] = v;
  }
}

/// builds basis->dev_tei (device); the host keeps only the disjoint tables.  Diatomic basis: two operand types
/// (weights with and without cosh^2 mu), one channel per (L,|M|); atomic basis: one type, one channel per L, the first
/// element's primitives shifted by one in the padded layout (its first primitive is dropped at the nucleus).
void compute_tei_dev(hfg_ctx *ctx, hfg_basis *basis) {
  const bool atomic = basis->kind != 0;
  HFG_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  int E, p, Nlm, nq;
  if (atomic) {
    helfem::atomic::TwoDBasis &b = basis->ab;
    b.compute_disjoint();
    E = (int)b.Nel();
    p = (int)b.max_Nprim();
    Nlm = b.N_L();
    nq = b.nquad();
  } else {
    helfem::diatomic::TwoDBasis &b = basis->b;
    b.compute_disjoint();
    E = (int)b.Nel();
    p = (int)b.max_Nprim();
    Nlm = (int)b.lm_map.size();
    nq = b.nquad();
  }
  const int nty = atomic ? 1 : 2;
  const size_t pp = (size_t)p * p;
  basis->dev_tei.resize((size_t)nty * nty * Nlm * E * pp * pp);
  DevBuf<double> d_bb0, d_bbs, d_wQ, d_wP, d_inner, d_bq, d_W;
  DevBuf<GemmTask> d_tasks;
  for (int e = 0; e < E; e++) {
    helfem::diatomic::TwoDBasis::TeiElementTables t;
    if (atomic) basis->ab.tei_element_tables(e, t);
    else basis->b.tei_element_tables(e, t);
    const int Ni = (int)t.Ni, Np = (int)t.Np;
    d_bb0.upload(t.bb0.d, s);
    d_bbs.upload(t.bbs.d, s);
    d_wQ.upload(t.wQ, s);
    d_wP.upload(t.wP, s);
    d_inner.resize((size_t)nty * Nlm * nq * Np);
    d_bq.resize((size_t)nty * Nlm * nq * Np);
    d_W.resize((size_t)nty * nty * Nlm * Np * Np);
    size_t shb = (size_t)nq * nq * sizeof(double);
    if (shb > 64 * 1024)
      HFG_HIP_CHECK(hipFuncSetAttribute((const void *)k_tei_inner, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
    hipLaunchKernelGGL(k_tei_inner, dim3(Nlm, nty), dim3(256), shb, s, d_bbs.p, d_wP.p, Np, nq, Nlm, d_inner.p);
    hipLaunchKernelGGL(k_tei_scale, dim3(Nlm, nty), dim3(256), 0, s, d_bb0.p, d_wQ.p, Np, nq, Nlm, d_bq.p);
    // W_kl[ilm] (Np x Np) = bq_k[ilm] (Np x nq) * inner_l[ilm]^T (nq x Np)
    std::vector<GemmTask> tasks((size_t)nty * nty * Nlm);
    for (int k = 0; k < nty; k++)
      for (int l = 0; l < nty; l++)
        for (int ilm = 0; ilm < Nlm; ilm++) {
          GemmTask g;
          g.A = d_bq.p + ((size_t)k * Nlm + ilm) * (size_t)nq * Np;
          g.B = d_inner.p + ((size_t)l * Nlm + ilm) * (size_t)nq * Np;
          g.C = d_W.p + ((size_t)(k * nty + l) * Nlm + ilm) * (size_t)Np * Np;
          g.M = g.N = Np;
          g.K = nq;
          g.lda = g.ldb = g.ldc = Np;
          g.tB = 1;
          tasks[((size_t)(k * nty + l)) * Nlm + ilm] = g;
        }
    d_tasks.upload(tasks, s);
    HFG_HIP_CHECK(hipStreamSynchronize(s));  // host vectors of this element live on this stack frame
    gemm_tasklist64_dev(ctx, d_tasks.p, nty * nty * Nlm, Np, Np);
    // diatomic elements hold their primitives from index 0 (the last one has p-1 of them); the first atomic element has
    // lost its first primitive (hip/tables.cpp: lo)
    const int lo = (atomic && e == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_tei_combine, dim3(Nlm, nty * nty), dim3(256), 0, s, d_W.p, Ni, p, lo, Nlm, E, e, nty, basis->dev_tei.p);
    HFG_HIP_CHECK(hipGetLastError());
    HFG_HIP_CHECK(hipStreamSynchronize(s));
  }
  basis->tei_on_device = true;
}

}  // namespace hfg
