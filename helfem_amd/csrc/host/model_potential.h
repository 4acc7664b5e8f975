// Screened-nucleus model potentials for the initial guess (reference: src/general/model_potential.{h,cpp},
// src/general/gsz.cpp; `--iguess` of src/diatomic/main.cpp:128 and src/atomic/main.cpp:102).
//   kind 0  point nucleus            V = -Z/r                         (the core guess)
//   kind 1  Green-Sellin-Zachor      V = -[1 + (Z-1)/(1 + H (e^{r/d} - 1))]/r ; the caller supplies d (H = d (Z-1)^0.4
//           when H <= 0): the reference's per-element d table (Green et al. 1969) is data of its source tree and is
//           not reproduced here
//   kind 3  Thomas-Fermi             V = -Z_TF(r)/r, Z_TF = Z (1 + a sqrt(x) + b x e^{-g sqrt(x)})^2 e^{-2 a sqrt(x)},
//           x = r (128 Z/(9 pi^2))^{1/3}, a = 0.7280642371, b = -0.5430794693, g = 0.3612163121 (arXiv:physics/0511017)
//   kind 2  superposition of atomic potentials: needs the reference's 30 000-line tabulation, not available
#pragma once
#include <cmath>
#include <stdexcept>

namespace helfem {

struct ModelPotential {
  int kind = 0;
  int Z = 0;
  double d = 0.0, H = 0.0;  // GSZ parameters

  double effective_charge(double r) const {
    switch (kind) {
      case 0: return (double)Z;
      case 1: {
        if (!(d > 0.0)) throw std::logic_error("GSZ guess: the screening length d_Z must be given\n");
        const double Hz = (H > 0.0) ? H : d * std::pow((double)(Z - 1), 0.4);
        return 1.0 + (Z - 1) / (1.0 + (std::exp(r / d) - 1.0) * Hz);
      }
      case 3: {
        const double alpha = 0.7280642371, beta = -0.5430794693, gamma = 0.3612163121;
        const double x = r * std::cbrt(128.0 * Z / (9.0 * M_PI * M_PI)), sx = std::sqrt(x);
        const double f = 1.0 + alpha * sx + beta * x * std::exp(-gamma * sx);
        return Z * f * f * std::exp(-2.0 * alpha * sx);
      }
      default: throw std::logic_error("Unsupported guess\n");
    }
  }
  /// V(r); non-finite values (r = 0) are skipped by the quadratures, as the reference does with std::isnormal
  double V(double r) const { return -effective_charge(r) / r; }
};

}  // namespace helfem
