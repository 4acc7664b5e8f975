// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Geometry-independent part of the reference's DFT grid worker, dense complex formulation:
//   update_density  src/diatomic/dftgrid.cpp:51-117 (restricted), :119-200 (polarized)
//   compute_xc      src/diatomic/dftgrid.cpp:343-458
//   eval_Fxc        src/diatomic/dftgrid.cpp:499-545 (restricted), :547-640 (polarized)
//   increment_lda / increment_gga  src/diatomic/dftgrid.h:190-253
// (the atomic worker, src/atomic/dftgrid.cpp, has the same members and algebra).  The program-specific part
// (compute_bf: basis values, weights, scale factors) fills the arrays below.
#pragma once
#include "oracle.h"
#include <complex>

namespace oracle {
typedef std::complex<double> cplx;

struct DenseGrid {
  bool do_grad = false, do_gga = false, polarized = false;
  bool do_tau = false, do_mgga_t = false;  // kinetic energy density
  Vec tau, vtau;                           // polarised: 2 x Ng, point-major
  std::vector<size_t> bf_ind;
  size_t ne = 0, Ng = 0;
  std::vector<cplx> bf, bf_rho, bf_theta, bf_phi;  // ne x Ng column-major
  Vec wtot, scale_r, scale_theta, scale_phi;
  Vec rho, sigma, exc, vxc, vsigma;  // polarized: rho 2 x Ng, sigma 3 x Ng, vxc 2 x Ng, vsigma 3 x Ng (point-major)
  std::vector<double> grho;          // restricted 3 x Ng; polarized 6 x Ng (component-major: [c*Ng+ip])

  Mat gather(const Mat &Pdummy) const {
    Mat P(ne, ne);
    for (size_t j = 0; j < ne; j++)
      for (size_t i = 0; i < ne; i++) P(i, j) = Pdummy(bf_ind[i], bf_ind[j]);
    return P;
  }

  // rho and gradient of one spin density at every grid point
  void density_of(const Mat &P, double *rho_out, size_t rho_stride, double *g_out /* 3 x Ng or null */) const {
    std::vector<cplx> Pv(ne);
    for (size_t ip = 0; ip < Ng; ip++) {
      for (size_t i = 0; i < ne; i++) Pv[i] = 0;
      for (size_t j = 0; j < ne; j++) {
        cplx cb = std::conj(bf[ip * ne + j]);
        for (size_t i = 0; i < ne; i++) Pv[i] += P(i, j) * cb;
      }
      cplx d(0);
      for (size_t i = 0; i < ne; i++) d += Pv[i] * bf[ip * ne + i];
      rho_out[ip * rho_stride] = d.real();
      if (g_out) {
        cplx g0(0), g1(0), g2(0);
        for (size_t i = 0; i < ne; i++) {
          g0 += Pv[i] * bf_rho[ip * ne + i];
          g1 += Pv[i] * bf_theta[ip * ne + i];
          g2 += Pv[i] * bf_phi[ip * ne + i];
        }
        g_out[0 * Ng + ip] = 2.0 * g0.real() / scale_r[ip];
        g_out[1 * Ng + ip] = 2.0 * g1.real() / scale_theta[ip];
        g_out[2 * Ng + ip] = 2.0 * g2.real() / scale_phi[ip];
      }
    }
  }

  void update_density(const Mat &Pdummy) {
    polarized = false;
    Mat P = gather(Pdummy);
    rho.assign(Ng, 0.0);
    if (do_grad) {
      grho.assign(3 * Ng, 0.0);
      sigma.assign(Ng, 0.0);
    }
    density_of(P, rho.data(), 1, do_grad ? grho.data() : nullptr);
    if (do_grad)
      for (size_t ip = 0; ip < Ng; ip++) {
        double gr = grho[ip], gt = grho[Ng + ip], gp = grho[2 * Ng + ip];
        sigma[ip] = gr * gr + gt * gt + gp * gp;
      }
    if (do_tau) {
      tau.assign(Ng, 0.0);
      tau_of(P, tau.data(), 1);
    }
  }
  // tau = 1/2 sum_c Re[(P conj(d_c bf)) . d_c bf] / h_c^2   (dftgrid.cpp:90-112, 159-200)
  void tau_of(const Mat &P, double *tau_out, size_t stride) const {
    const std::vector<cplx> *dbf[3] = {&bf_rho, &bf_theta, &bf_phi};
    const Vec *sc[3] = {&scale_r, &scale_theta, &scale_phi};
    std::vector<cplx> Pv(ne);
    for (int c = 0; c < 3; c++)
      for (size_t ip = 0; ip < Ng; ip++) {
        const cplx *f = &(*dbf[c])[ip * ne];
        for (size_t i = 0; i < ne; i++) Pv[i] = 0;
        for (size_t j = 0; j < ne; j++) {
          cplx cb = std::conj(f[j]);
          for (size_t i = 0; i < ne; i++) Pv[i] += P(i, j) * cb;
        }
        cplx k(0);
        for (size_t i = 0; i < ne; i++) k += Pv[i] * f[i];
        tau_out[ip * stride] += 0.5 * k.real() / ((*sc[c])[ip] * (*sc[c])[ip]);
      }
  }
  double compute_Ekin() const {
    double e = 0.0;
    if (do_tau)
      for (size_t ip = 0; ip < Ng; ip++) e += wtot[ip] * (polarized ? tau[2 * ip] + tau[2 * ip + 1] : tau[ip]);
    return e;
  }

  void update_density(const Mat &Padummy, const Mat &Pbdummy) {
    polarized = true;
    Mat Pa = gather(Padummy), Pb = gather(Pbdummy);
    rho.assign(2 * Ng, 0.0);
    if (do_grad) {
      grho.assign(6 * Ng, 0.0);
      sigma.assign(3 * Ng, 0.0);
    }
    density_of(Pa, rho.data(), 2, do_grad ? grho.data() : nullptr);
    density_of(Pb, rho.data() + 1, 2, do_grad ? grho.data() + 3 * Ng : nullptr);
    if (do_grad)
      for (size_t ip = 0; ip < Ng; ip++) {
        const double *ga = &grho[ip], *gb = &grho[3 * Ng + ip];
        sigma[3 * ip + 0] = ga[0] * ga[0] + ga[Ng] * ga[Ng] + ga[2 * Ng] * ga[2 * Ng];
        sigma[3 * ip + 1] = ga[0] * gb[0] + ga[Ng] * gb[Ng] + ga[2 * Ng] * gb[2 * Ng];
        sigma[3 * ip + 2] = gb[0] * gb[0] + gb[Ng] * gb[Ng] + gb[2 * Ng] * gb[2 * Ng];
      }
    if (do_tau) {
      tau.assign(2 * Ng, 0.0);
      tau_of(Pa, tau.data(), 2);
      tau_of(Pb, tau.data() + 1, 2);
    }
  }

  double compute_Nel() const {
    double n = 0.0;
    for (size_t ip = 0; ip < Ng; ip++) n += wtot[ip] * (polarized ? rho[2 * ip] + rho[2 * ip + 1] : rho[ip]);
    return n;
  }
  double eval_Exc() const {
    double e = 0.0;
    for (size_t ip = 0; ip < Ng; ip++) e += wtot[ip] * exc[ip] * (polarized ? rho[2 * ip] + rho[2 * ip + 1] : rho[ip]);
    return e;
  }

  void compute_xc(int x_func, int c_func, double thr) {
    // meta-GGAs: floor under the density threshold, see hip/fock.hip xc_compact (tau-dependent terms overflow below 1e-50)
    if ((x_func > 0 && xc_is_mgga(x_func)) || (c_func > 0 && xc_is_mgga(c_func))) thr = std::max(thr, 1e-40);
    const size_t nr = polarized ? 2 : 1, ns = polarized ? 3 : 1;
    exc.assign(Ng, 0.0);
    vxc.assign(nr * Ng, 0.0);
    vsigma.assign(ns * Ng, 0.0);
    do_gga = false;
    do_mgga_t = false;
    vtau.assign(nr * Ng, 0.0);
    Vec e(Ng), v(nr * Ng), vs(ns * Ng);
    for (int id : {x_func, c_func}) {
      if (id <= 0) continue;
      do_gga = do_gga || xc_is_gga(id);
      if (xc_is_mgga(id)) {
        Vec vt(nr * Ng);
        if (polarized)
          xc_polarized_mgga(id, Ng, rho.data(), sigma.data(), tau.data(), e.data(), v.data(), vs.data(), vt.data(), thr);
        else
          xc_unpolarized_mgga(id, Ng, rho.data(), sigma.data(), tau.data(), e.data(), v.data(), vs.data(), vt.data(), thr);
        for (size_t i = 0; i < nr * Ng; i++) vtau[i] += vt[i];
        do_mgga_t = true;
      } else if (polarized)
        xc_polarized(id, Ng, rho.data(), do_grad ? sigma.data() : nullptr, e.data(), v.data(), vs.data(), thr);
      else
        xc_unpolarized(id, Ng, rho.data(), do_grad ? sigma.data() : nullptr, e.data(), v.data(), vs.data(), thr);
      for (size_t i = 0; i < Ng; i++) exc[i] += e[i];
      for (size_t i = 0; i < nr * Ng; i++) vxc[i] += v[i];
      for (size_t i = 0; i < ns * Ng; i++) vsigma[i] += vs[i];
    }
  }

  // H += Re[(f o v) f^H]   (dftgrid.h:190-208)
  void increment_lda(Mat &H, const Vec &v, const std::vector<cplx> &f) const {
    for (size_t ip = 0; ip < Ng; ip++)
      for (size_t j = 0; j < ne; j++) {
        cplx cj = std::conj(f[ip * ne + j]) * v[ip];
        for (size_t i = 0; i < ne; i++) H(i, j) += (f[ip * ne + i] * cj).real();
      }
  }
  // gamma = sum_c gr_c d_c bf ;  H += Re[gamma f^H + f gamma^H]   (dftgrid.h:211-253); gr is 3 x Ng
  void increment_gga(Mat &H, const std::vector<double> &gr) const {
    std::vector<cplx> gamma(ne * Ng);
    for (size_t ip = 0; ip < Ng; ip++)
      for (size_t i = 0; i < ne; i++)
        gamma[ip * ne + i] = gr[ip] * bf_rho[ip * ne + i] + gr[Ng + ip] * bf_theta[ip * ne + i] +
                             gr[2 * Ng + ip] * bf_phi[ip * ne + i];
    for (size_t ip = 0; ip < Ng; ip++)
      for (size_t j = 0; j < ne; j++) {
        cplx cfj = std::conj(bf[ip * ne + j]), cgj = std::conj(gamma[ip * ne + j]);
        for (size_t i = 0; i < ne; i++) H(i, j) += (gamma[ip * ne + i] * cfj + bf[ip * ne + i] * cgj).real();
      }
  }
  void scatter_add(Mat &Hdummy, const Mat &H) const {
    for (size_t j = 0; j < ne; j++)
      for (size_t i = 0; i < ne; i++) Hdummy(bf_ind[i], bf_ind[j]) += H(i, j);
  }

  void eval_Fxc(Mat &Hdummy) const {
    Mat H(ne, ne);
    Vec vr(Ng);
    for (size_t i = 0; i < Ng; i++) vr[i] = vxc[i] * wtot[i];
    increment_lda(H, vr, bf);
    if (do_gga) {
      std::vector<double> gr(3 * Ng);
      for (size_t ip = 0; ip < Ng; ip++) {
        gr[ip] = grho[0 * Ng + ip] * 2.0 * wtot[ip] * vsigma[ip] / scale_r[ip];
        gr[Ng + ip] = grho[1 * Ng + ip] * 2.0 * wtot[ip] * vsigma[ip] / scale_theta[ip];
        gr[2 * Ng + ip] = grho[2 * Ng + ip] * 2.0 * wtot[ip] * vsigma[ip] / scale_phi[ip];
      }
      increment_gga(H, gr);
    }
    if (do_mgga_t) {  // dftgrid.cpp:533-540
      Vec v0(Ng), v1(Ng), v2(Ng);
      for (size_t ip = 0; ip < Ng; ip++) {
        double vt = 0.5 * wtot[ip] * vtau[ip];
        v0[ip] = vt / (scale_r[ip] * scale_r[ip]);
        v1[ip] = vt / (scale_theta[ip] * scale_theta[ip]);
        v2[ip] = vt / (scale_phi[ip] * scale_phi[ip]);
      }
      increment_lda(H, v0, bf_rho);
      increment_lda(H, v1, bf_theta);
      increment_lda(H, v2, bf_phi);
    }
    scatter_add(Hdummy, H);
  }

  void eval_Fxc(Mat &Hadummy, Mat &Hbdummy) const {
    Mat Ha(ne, ne), Hb(ne, ne);
    Vec va(Ng), vb(Ng);
    for (size_t i = 0; i < Ng; i++) {
      va[i] = vxc[2 * i] * wtot[i];
      vb[i] = vxc[2 * i + 1] * wtot[i];
    }
    increment_lda(Ha, va, bf);
    increment_lda(Hb, vb, bf);
    if (do_gga) {
      std::vector<double> gra(3 * Ng), grb(3 * Ng);
      for (size_t ip = 0; ip < Ng; ip++) {
        const double vaa = vsigma[3 * ip], vab = vsigma[3 * ip + 1], vbb = vsigma[3 * ip + 2];
        const double sc[3] = {scale_r[ip], scale_theta[ip], scale_phi[ip]};
        for (int c = 0; c < 3; c++) {
          double ga = grho[c * Ng + ip], gb = grho[(3 + c) * Ng + ip];
          gra[c * Ng + ip] = wtot[ip] * (2.0 * vaa * ga + vab * gb) / sc[c];
          grb[c * Ng + ip] = wtot[ip] * (2.0 * vbb * gb + vab * ga) / sc[c];
        }
      }
      increment_gga(Ha, gra);
      increment_gga(Hb, grb);
    }
    if (do_mgga_t)  // dftgrid.cpp:615-636: the three tau terms of each spin
      for (int sp = 0; sp < 2; sp++) {
        Vec v0(Ng), v1(Ng), v2(Ng);
        for (size_t ip = 0; ip < Ng; ip++) {
          double vt = 0.5 * wtot[ip] * vtau[2 * ip + sp];
          v0[ip] = vt / (scale_r[ip] * scale_r[ip]);
          v1[ip] = vt / (scale_theta[ip] * scale_theta[ip]);
          v2[ip] = vt / (scale_phi[ip] * scale_phi[ip]);
        }
        Mat &H = sp ? Hb : Ha;
        increment_lda(H, v0, bf_rho);
        increment_lda(H, v1, bf_theta);
        increment_lda(H, v2, bf_phi);
      }
    scatter_add(Hadummy, Ha);
    scatter_add(Hbdummy, Hb);
  }
};

}  // namespace oracle
