#include "scf.h"
#include <cfloat>
#include <climits>
#include "diis.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <deque>
#include <functional>
#include <string>

namespace helfem {
namespace scf {

namespace {
double wall() {
  using namespace std::chrono;
  return duration_cast<duration<double> >(steady_clock::now().time_since_epoch()).count();
}

// History of the Fock-matrix extrapolation (uDIIS of the reference, diis.cpp:129-168: Fock matrices, densities and error
// vectors of both spins; the restricted drivers pass the same matrices for both spins, main.cpp:779-781, 938).  The
// weights come from DiisMixer (ADIIS + CDIIS, diis.h), which only sees inner products.
struct FockHistory {
  DiisMixer mix;
  std::deque<Mat> Fs, Ps, errs;  // [Fa | Fb], [Pa | Pb], [ea | eb]; restricted runs store one spin and count it twice
  double spinfac;
  FockHistory(const Options &opt, bool restricted)
      : mix(true, opt.diiseps, opt.diisthr, true, opt.verbose, (size_t)opt.diisorder), spinfac(restricted ? 2.0 : 1.0) {}
  static double dot(const Mat &a, const Mat &b) {
    double s = 0.0;
    for (size_t k = 0; k < a.d.size(); k++) s += a.d[k] * b.d[k];
    return s;
  }
  void update(const Mat &F, const Mat &P, const Mat &err, double E, double maxerr) {
    if (mix.full()) {
      mix.pop_oldest();
      Fs.pop_front();
      Ps.pop_front();
      errs.pop_front();
    }
    Fs.push_back(F);
    Ps.push_back(P);
    errs.push_back(err);
    mix.push(E, maxerr);
    const size_t n = Fs.size() - 1;
    for (size_t i = 0; i <= n; i++) {
      mix.set_B(i, n, spinfac * dot(errs[i], errs[n]));
      mix.set_T(i, n, spinfac * dot(Ps[i], Fs[n]));  // Tr P F = sum_kl P_kl F_kl for symmetric matrices
      mix.set_T(n, i, spinfac * dot(Ps[n], Fs[i]));
    }
  }
  Mat solve() {
    size_t dropped = 0;
    std::vector<double> w = mix.solve(dropped);
    for (size_t k = 0; k < dropped; k++) {
      Fs.pop_front();
      Ps.pop_front();
      errs.pop_front();
    }
    Mat F(Fs[0].n_rows, Fs[0].n_cols);
    for (size_t i = 0; i < Fs.size(); i++)
      for (size_t k = 0; k < F.d.size(); k++) F.d[k] += w[i] * Fs[i].d[k];
    return F;
  }
};

Mat form_density(const Mat &C, size_t nocc) {
  Mat P(C.n_rows, C.n_rows);
  for (size_t o = 0; o < nocc; o++)
    for (size_t j = 0; j < C.n_rows; j++) {
      double cj = C(j, o);
      for (size_t i = 0; i < C.n_rows; i++) P(i, j) += C(i, o) * cj;
    }
  return P;
}

Mat enforce_sym(const Mat &F, const std::vector<std::vector<size_t> > &sym) {
  Mat out(F.n_rows, F.n_cols);
  for (const auto &idx : sym)
    for (size_t j : idx)
      for (size_t i : idx) out(i, j) = F(i, j);
  return out;
}
}  // namespace

// ---- forced occupations -----------------------------------------------------------------------------------------
static void check_occupation_sums(const OccupationPlan &pl, int nela, int nelb) {
  int sa = 0, sb = 0;
  for (int v : pl.na) sa += v;
  for (int v : pl.nb) sb += v;
  // diatomic/main.cpp:369-380 (the beta message says "alpha" in the reference too)
  if (sa != nela)
    throw std::logic_error("Specified alpha occupations don't match wanted spin state.\nOccupying " + std::to_string(sa) +
                           " orbitals but should have " + std::to_string(nela) + " orbitals.\n");
  if (sb != nelb)
    throw std::logic_error("Specified alpha occupations don't match wanted spin state.\nOccupying " + std::to_string(sb) +
                           " orbitals but should have " + std::to_string(nelb) + " orbitals.\n");
}

OccupationPlan occupation_plan(const Options &opt, const diatomic::TwoDBasis &basis, int nela, int nelb) {
  OccupationPlan pl;
  if (!opt.readocc) return pl;
  pl.until = opt.readocc < 0 ? INT_MAX : opt.readocc;
  if (opt.occs.empty() || opt.occs[0].size() < 3) throw std::logic_error("Must have at least three columns in occupation data.\n");
  const size_t ncol = opt.occs[0].size();
  const bool hetero = basis.Z1 != basis.Z2;
  if (hetero && ncol != 3) throw std::logic_error("Heteronuclear molecule orbital occupations must have three columns.\n");
  if (!hetero && ncol != 3 && ncol != 4) throw std::logic_error("Homonuclear molecule orbital occupations must have three or four columns.\n");
  if (ncol == 4 && opt.symmetry != 2) throw std::logic_error("For use of homonuclear orbital occupations, must turn on use of full symmetry.\n");
  for (size_t i = 0; i < opt.occs.size(); i++) {
    const std::vector<int> &row = opt.occs[i];
    if (row.size() != ncol) throw std::logic_error("Ragged occupation data.\n");
    pl.na.push_back(row[0]);
    pl.nb.push_back(row[1]);
    if (ncol == 3)
      pl.sym.push_back(basis.m_indices(row[2]));
    else {
      if (row[3] != 1 && row[3] != -1)
        throw std::logic_error("Error on line " + std::to_string(i + 1) + " of orbital occupations: parity must be +1 or -1\n");
      pl.sym.push_back(basis.m_indices(row[2], row[3] == -1));
    }
  }
  check_occupation_sums(pl, nela, nelb);
  return pl;
}

OccupationPlan occupation_plan(const Options &opt, const atomic::TwoDBasis &basis, int nela, int nelb) {
  OccupationPlan pl;
  if (!opt.readocc) return pl;
  pl.until = opt.readocc < 0 ? INT_MAX : opt.readocc;
  const size_t ncol = opt.occs.empty() ? 0 : opt.occs[0].size();
  if (opt.symmetry == 2 && ncol != 4) throw std::logic_error("Must have four columns in occupation data to use full atomic symmetry.\n");
  if (opt.symmetry == 1 && ncol != 3) throw std::logic_error("Must have three columns in occupation data to use axial symmetry.\n");
  if (opt.symmetry != 1 && opt.symmetry != 2) throw std::logic_error("Not implemented!\n");
  for (const std::vector<int> &row : opt.occs) {
    if (row.size() != ncol) throw std::logic_error("Ragged occupation data.\n");
    pl.na.push_back(row[0]);
    pl.nb.push_back(row[1]);
    pl.sym.push_back(opt.symmetry == 1 ? basis.m_indices(row[2]) : basis.lm_indices(row[2], row[3]));
  }
  check_occupation_sums(pl, nela, nelb);
  return pl;
}

void guess_from_checkpoint(const Options &opt, const Mat &S, const Mat &Sinvh, const Mat &S12, size_t nela, size_t nelb, Mat &Ca, Mat &Cb,
                           Vec &Ea, Vec &Eb) {
  const size_t N = S.n_rows;
  bool same = opt.guessS.n_rows == N && opt.guessS.n_cols == N && opt.guessCa.n_rows == N;
  if (same) {
    double dmax = 0.0, smax = 0.0;
    for (size_t k = 0; k < S.d.size(); k++) {
      dmax = std::max(dmax, std::fabs(S.d[k] - opt.guessS.d[k]));
      smax = std::max(smax, std::fabs(S.d[k]));
    }
    same = dmax <= 1e-10 * smax;
  }
  Ea = opt.guessEa;
  Eb = opt.guessEb.size() ? opt.guessEb : opt.guessEa;
  if (same) {
    Ca = opt.guessCa;
    Cb = opt.guessCb.n_rows == N ? opt.guessCb : opt.guessCa;
  } else {
    if (S12.n_rows != N || S12.n_cols != opt.guessCa.n_rows)
      throw std::logic_error("The checkpoint to load was made in a different basis set; projection between basis sets "
                             "(interbasis overlap) needs the checkpoint's basis, which could not be read.\n");
    // C = Sinvh Sinvh^T S12 C_old (main.cpp:616-626)
    auto project = [&](const Mat &Cold) {
      Mat t = matmul(S12, false, Cold, false);
      t = matmul(Sinvh, true, t, false);
      return matmul(Sinvh, false, t, false);
    };
    Ca = project(opt.guessCa);
    Cb = opt.guessCb.n_rows == opt.guessCa.n_rows && opt.guessCb.n_cols ? project(opt.guessCb) : Ca;
  }
  if (Ca.n_cols < nela || Cb.n_cols < nelb) throw std::logic_error("The checkpoint holds fewer orbitals than are to be occupied.\n");
  auto gram_schmidt = [&](Mat &C, size_t nocc) {
    std::vector<double> Sc(N);
    for (size_t i = 0; i < nocc; i++) {
      for (size_t j = 0; j <= i; j++) {
        // S c_i with the current c_i
        for (size_t a = 0; a < N; a++) {
          double t = 0.0;
          for (size_t b = 0; b < N; b++) t += S(a, b) * C(b, i);
          Sc[a] = t;
        }
        double dot = 0.0;
        for (size_t a = 0; a < N; a++) dot += C(a, j) * Sc[a];
        if (j < i)
          for (size_t a = 0; a < N; a++) C(a, i) -= C(a, j) * dot;
        else {
          const double inv = 1.0 / std::sqrt(dot);
          for (size_t a = 0; a < N; a++) C(a, i) *= inv;
        }
      }
    }
  };
  gram_schmidt(Ca, nela);
  gram_schmidt(Cb, nelb);
}

std::vector<size_t> occupation_order(const Vec &E, const std::vector<std::vector<double> > &w, const std::vector<int> &nocc) {
  if (nocc.size() != w.size()) throw std::logic_error("nocc vector and symmetry indices don't match!\n");
  const size_t norb = E.size();
  std::vector<char> occupied(norb, 0);
  std::vector<size_t> occ;
  for (size_t isym = 0; isym < w.size(); isym++) {
    if (!nocc[isym]) continue;
    int taken = 0;
    for (size_t o = 0; o < norb && taken < nocc[isym]; o++)
      if (w[isym][o] > 10 * DBL_EPSILON) {
        if (occupied[o]) throw std::logic_error("Duplicates in occupied orbital list!\n");
        occupied[o] = 1;
        occ.push_back(o);
        taken++;
      }
    if (taken < nocc[isym]) throw std::logic_error("Not enough orbitals of the requested symmetry to occupy!\n");
  }
  std::vector<size_t> virt;
  for (size_t o = 0; o < norb; o++)
    if (!occupied[o]) virt.push_back(o);
  auto by_energy = [&E](size_t a, size_t b) { return E[a] < E[b]; };
  std::stable_sort(occ.begin(), occ.end(), by_energy);
  std::stable_sort(virt.begin(), virt.end(), by_energy);
  occ.insert(occ.end(), virt.begin(), virt.end());
  return occ;
}

void enforce_occupations(Mat &C, Vec &E, const Mat &S, const std::vector<int> &nocc, const std::vector<std::vector<size_t> > &sym) {
  // the symmetries must not share basis functions (scf_helpers.cpp:35-50)
  {
    std::vector<char> seen(S.n_rows, 0);
    for (const auto &idx : sym)
      for (size_t i : idx) {
        if (seen[i]) throw std::logic_error("Duplicate basis functions in symmetry list!\n");
        seen[i] = 1;
      }
  }
  const size_t norb = C.n_cols;
  std::vector<std::vector<double> > w(sym.size(), std::vector<double>(norb, 0.0));
  for (size_t isym = 0; isym < sym.size(); isym++) {
    if (!nocc[isym]) continue;
    const std::vector<size_t> &idx = sym[isym];
    std::vector<double> sc(idx.size());
    for (size_t o = 0; o < norb; o++) {
      double nrm = 0.0;
      for (size_t a = 0; a < idx.size(); a++) {
        double t = 0.0;
        for (size_t b = 0; b < idx.size(); b++) t += S(idx[a], idx[b]) * C(idx[b], o);
        nrm += C(idx[a], o) * t;
      }
      w[isym][o] = nrm;
    }
  }
  const std::vector<size_t> order = occupation_order(E, w, nocc);
  Mat Cn(C.n_rows, norb);
  Vec En(norb);
  for (size_t o = 0; o < norb; o++) {
    En[o] = E[order[o]];
    for (size_t i = 0; i < C.n_rows; i++) Cn(i, o) = C(i, order[o]);
  }
  C = Cn;
  E = En;
}


namespace {
// the part of the drivers shared by the diatomic and atomic programs: everything from the one-electron matrices on
struct Problem {
  Mat S, T, Vnuc;
  std::vector<std::vector<size_t> > dsym;
  int symm = 1;
  int nel = 0;
  std::function<void()> compute_tei_and_prepare;
  ModelPotential guess1, guess2;  // screened nuclei of the guess (kind 0: core Hamiltonian, nothing to evaluate)
  // --maverage (atomic): groups of equally sized index lists whose diagonal blocks of F are averaged
  std::vector<std::vector<std::vector<size_t> > > avg_idx;
  // --readocc: the plan once the spin state is known
  std::function<OccupationPlan(int, int)> occupations;
  // --load from another basis: interbasis overlap (this x checkpoint), when the program can form it
  std::function<Mat()> guess_overlap;
};

// scf::fock_symmetry_average (src/general/scf_helpers.cpp:263-284)
Mat fock_symmetry_average(const Mat &Fin, const std::vector<std::vector<std::vector<size_t> > > &sym_idx) {
  Mat Fout(Fin);
  for (const auto &grp : sym_idx) {
    if (grp.empty()) continue;
    const size_t nn = grp[0].size();
    Mat Fmean(nn, nn);
    for (const auto &idx : grp)
      for (size_t j = 0; j < nn; j++)
        for (size_t i = 0; i < nn; i++) Fmean(i, j) += Fin(idx[i], idx[j]);
    for (double &v : Fmean.d) v /= (double)grp.size();
    for (const auto &idx : grp)
      for (size_t j = 0; j < nn; j++)
        for (size_t i = 0; i < nn; i++) Fout(idx[i], idx[j]) = Fmean(i, j);
  }
  return Fout;
}

// scf::ROHF_update (src/general/scf_helpers.cpp:470-523; Tsuchimochi & Scuseria, J. Chem. Phys. 134, 064101):
// natural orbitals of the total density, lambda = -Delta on the core-virtual blocks, Fa += lambda, Fb -= lambda.
// Sh is the partner of Sinvh (Sh^T Sinvh = 1; S Sinvh, which is S^{1/2} for the symmetric half-inverse).
void rohf_update(Backend &be, Mat &Fa, Mat &Fb, const Mat &P, const Mat &Sh, const Mat &Sinvh, size_t nocca, size_t noccb) {
  const size_t N = P.n_rows;
  Mat Porth = be.gemm(be.gemm(Sh, true, P, false), false, Sh, false);
  Vec occ;
  Mat Pvec;
  be.eig_sym(occ, Pvec, Porth);  // ascending occupations: virtual, active, core
  Mat A2N = be.gemm(Sinvh, false, Pvec, false);
  Mat ShPv = be.gemm(Sh, false, Pvec, false);
  Mat Delta = 0.5 * (Fa - Fb);
  Mat Dno = be.gemm(be.gemm(A2N, true, Delta, false), false, A2N, false);
  const size_t Nc = std::min(nocca, noccb), Na = std::max(nocca, noccb) - Nc, Nv = N - Na - Nc;
  Mat lam(N, N);
  for (size_t c = N - Nc; c < N; c++)
    for (size_t v = 0; v < Nv; v++) {
      lam(c, v) = -Dno(c, v);
      lam(v, c) = -Dno(v, c);
    }
  Mat lamAO = be.gemm(be.gemm(ShPv, false, lam, false), false, ShPv, true);
  Fa += lamAO;
  Fb -= lamAO;
}

Result scf_loop(const Options &opt, Backend &be, Problem &pb, Result res) {
  const bool verbose = opt.verbose;
  const bool dft = (opt.x_func > 0 || opt.c_func > 0);
  const int nel = pb.nel;
  // occupations (main.cpp:300-340): nela - nelb = M - 1
  int na = opt.nela, nb = opt.nelb, Qv = opt.Q, Mv = opt.multiplicity;
  parse_nela_nelb(na, nb, Qv, Mv, nel + opt.Q);  // nel = Ztot - Q
  const size_t nela = (size_t)na, nelb = (size_t)nb;
  const bool restr_req = (opt.restricted == -1) ? (nela == nelb) : (opt.restricted != 0);
  const bool rohf = restr_req && nela != nelb;  // restricted open shell: unrestricted machinery + CUHF constraint
  const bool restr = restr_req && !rohf;
  res.nela = (int)nela;
  res.nelb = (int)nelb;
  const OccupationPlan occ = pb.occupations ? pb.occupations((int)nela, (int)nelb) : OccupationPlan();
  const int symm = pb.symm;
  const Mat &S = pb.S, &T = pb.T, &Vnuc = pb.Vnuc;
  const std::vector<std::vector<size_t> > &dsym = pb.dsym;
  Mat H0(T + Vnuc);
  double t0 = wall();
  Mat Sinvh(be.Sinvh(S, !opt.diag, dsym));
  if (verbose) printf("Half-inverse formed in %.6f\n", wall() - t0);

  Mat Sh;
  if (rohf) Sh = be.gemm(S, false, Sinvh, false);
  // guess (main.cpp:650-712): core Hamiltonian, or T + model potential of the screened nuclei evaluated by
  // quadrature; the latter needs the backend's quadrature tables, so the integrals are prepared first
  Vec Ea, Eb;
  Mat Ca, Cb;
  bool prepared = false;
  Mat Hguess(H0);
  if (opt.have_guess) {
    // orbitals of a previous run, nothing to evaluate
  } else if (opt.iguess != 0) {
    if (verbose) printf("Computing two-electron integrals\n");
    t0 = wall();
    pb.compute_tei_and_prepare();
    if (verbose) printf("Done in %.6f\n", wall() - t0);
    prepared = true;
    if (verbose)
      printf("Guess orbitals from %s nucleus\n", opt.iguess == 1 ? "GSZ screened" : opt.iguess == 3 ? "Thomas-Fermi" : "screened");
    Hguess = T + be.model_potential(pb.guess1, pb.guess2);
  } else if (verbose)
    printf("Guess orbitals from core Hamiltonian\n");
  if (opt.have_guess) {
    if (verbose) printf("Guess orbitals from checkpoint\nGuess orbitals from previous calculation\n");
    guess_from_checkpoint(opt, S, Sinvh, pb.guess_overlap ? pb.guess_overlap() : Mat(), nela, nelb, Ca, Cb, Ea, Eb);
  } else {
    be.eig_gsym_sub(Ea, Ca, Hguess, Sinvh, dsym);
    if (!restr) {
      Eb = Ea;
      Cb = Ca;
    }
  }
  if (occ.until && !opt.have_guess) {  // main.cpp:716-722
    enforce_occupations(Ca, Ea, S, occ.na, occ.sym);
    if (!restr) enforce_occupations(Cb, Eb, S, occ.nb, occ.sym);
  }

  if (!prepared) {
    if (verbose) printf("Computing two-electron integrals\n");
    t0 = wall();
    pb.compute_tei_and_prepare();
    if (verbose) printf("Done in %.6f\n", wall() - t0);
  }

  FockHistory diis(opt, restr);
  double Eold = 0.0;
  Mat P, Fa, Fb;
  struct {
    Mat Pa, Pb, J, Ka, Kb, XCa, XCb;
  } last;  // of the last iteration, for the checkpoint
  const size_t Nb = S.n_rows;
  for (int it = 1; it <= opt.maxit; it++) {
    if (verbose) printf("\n**** Iteration %i ****\n\n", it);
    Mat Pa = form_density(Ca, nela);
    Mat Pb = restr ? Pa : form_density(Cb, nelb);
    P = Pa + Pb;
    if (verbose) {
      printf("Tr Pa = %f\n", trace_prod(Pa, S));
      if (!restr) printf("Tr Pb = %f\n", trace_prod(Pb, S));
    }
    res.Ekin = trace_prod(P, T);
    res.Epot = trace_prod(P, Vnuc);

    t0 = wall();
    Mat J(be.coulomb(P));
    res.tJ = wall() - t0;
    res.Ecoul = 0.5 * trace_prod(P, J);
    if (verbose) printf("Coulomb energy %.10e % .6f\n", res.Ecoul, res.tJ);

    Mat Ka, Kb;
    res.Exx = 0.0;
    if (opt.kfrac != 0.0 || opt.kshort != 0.0) {
      t0 = wall();
      // atomic/main.cpp:763-780: full-range and short-range exact exchange
      auto buildK = [&](const Mat &Ps) {
        Mat K(Nb, Nb);
        if (opt.kfrac != 0.0) K += opt.kfrac * be.exchange(Ps);
        if (opt.omega != 0.0) K += opt.kshort * be.rs_exchange(Ps);
        return K;
      };
      Ka = buildK(Pa);
      if (!restr) {
        if (nelb) Kb = buildK(Pb);
        else Kb.zeros(Nb, Nb);
      }
      res.tK = wall() - t0;
      // 0.5 Tr PaKa + 0.5 Tr PbKb (main.cpp:838-850)
      res.Exx = restr ? trace_prod(Pa, Ka) : 0.5 * trace_prod(Pa, Ka) + 0.5 * trace_prod(Pb, Kb);
      if (verbose) printf("Exchange energy %.10e % .6f\n", res.Exx, res.tK);
    }

    Mat XCa, XCb;
    res.Exc = 0.0;
    if (dft) {
      t0 = wall();
      double nelnum = 0, ekin = 0;
      if (restr) be.eval_Fxc(opt.x_func, opt.c_func, P, XCa, res.Exc, nelnum, ekin, opt.dftthr);
      else be.eval_Fxc_pol(opt.x_func, opt.c_func, Pa, Pb, XCa, XCb, res.Exc, nelnum, ekin, opt.dftthr);
      res.tXC = wall() - t0;
      if (verbose) {
        printf("DFT energy %.10e % .6f\n", res.Exc, res.tXC);
        printf("Error in integrated number of electrons % e\n", nelnum - nel);
      }
    }

    Fa = H0 + J;
    if (Ka.n_rows == Fa.n_rows) Fa += Ka;
    if (dft) Fa += XCa;
    if (!pb.avg_idx.empty()) Fa = fock_symmetry_average(Fa, pb.avg_idx);  // atomic/main.cpp:839-842
    if (symm) Fa = enforce_sym(Fa, dsym);
    if (!restr) {
      Fb = H0 + J;
      if (Kb.n_rows == Fb.n_rows) Fb += Kb;
      if (dft) Fb += XCb;
      if (!pb.avg_idx.empty()) Fb = fock_symmetry_average(Fb, pb.avg_idx);
      if (symm) Fb = enforce_sym(Fb, dsym);
      if (rohf) rohf_update(be, Fa, Fb, P, Sh, Sinvh, nela, nelb);  // main.cpp:903-904
    }

    if (opt.keep_matrices) {
      last.Pa = Pa;
      last.Pb = Pb;
      last.J = J;
      last.Ka = Ka;
      last.Kb = restr ? Ka : Kb;
      last.XCa = XCa;
      last.XCb = restr ? XCa : XCb;
    }
    res.Etot = res.Ekin + res.Epot + res.Ecoul + res.Exx + res.Exc + res.Enucr;
    double dE = res.Etot - Eold;
    if (verbose) {
      printf("Total energy is % .10f\n", res.Etot);
      if (it > 1) printf("Energy changed by %e\n", dE);
    }
    Eold = res.Etot;

    // DIIS error Sinvh^T (F P S - S P F) Sinvh per spin; unrestricted: the two spin blocks side by side
    // (uDIIS, diis.cpp:129-168)
    t0 = wall();
    auto diis_err = [&](const Mat &F, const Mat &Ps) {
      Mat FPS = be.gemm(be.gemm(F, false, Ps, false), false, S, false);
      Mat err = FPS - FPS.t();
      return be.gemm(be.gemm(Sinvh, true, err, false), false, Sinvh, false);
    };
    Mat err, Fcat, Pcat;
    if (restr) {
      err = diis_err(Fa, Pa);
      Fcat = Fa;
      Pcat = Pa;
    } else {
      Mat ea = diis_err(Fa, Pa), eb = diis_err(Fb, Pb);
      err.zeros(ea.n_rows, 2 * ea.n_cols);
      Fcat.zeros(Nb, 2 * Nb);
      Pcat.zeros(Nb, 2 * Nb);
      std::copy(ea.d.begin(), ea.d.end(), err.d.begin());
      std::copy(eb.d.begin(), eb.d.end(), err.d.begin() + ea.d.size());
      std::copy(Fa.d.begin(), Fa.d.end(), Fcat.d.begin());
      std::copy(Fb.d.begin(), Fb.d.end(), Fcat.d.begin() + Fa.d.size());
      std::copy(Pa.d.begin(), Pa.d.end(), Pcat.d.begin());
      std::copy(Pb.d.begin(), Pb.d.end(), Pcat.d.begin() + Pa.d.size());
    }
    double diiserr = 0.0;
    for (double v : err.d) diiserr = std::max(diiserr, fabs(v));
    diis.update(Fcat, Pcat, err, res.Etot, diiserr);
    if (verbose) printf("DIIS error is %e, update done in %.6f\n", diiserr, wall() - t0);
    t0 = wall();
    Mat Fd = diis.solve();
    if (verbose) printf("DIIS solution done in %.6f\n", wall() - t0);

    bool convd = (diiserr < opt.convthr) && (fabs(dE) < opt.convthr);

    // Fock damping of the atomic program (atomic/main.cpp:917-936): in the basis of the current orbitals the
    // occupied-virtual blocks are scaled by dampfock while the DIIS error is at least dampthr
    if (opt.dampfock != 1.0 && diiserr >= opt.dampthr) {
      if (verbose) printf("Damping off-diagonal elements of Fock matrix by % .3f\n", opt.dampfock);
      auto damp = [&](const Mat &F, const Mat &C, size_t nocc) {
        if (!nocc || F.n_rows <= nocc) return F;
        Mat fmo = be.gemm(be.gemm(C, true, F, false), false, C, false);
        for (size_t j = nocc; j < fmo.n_cols; j++)
          for (size_t i = 0; i < nocc; i++) {
            fmo(i, j) *= opt.dampfock;
            fmo(j, i) *= opt.dampfock;
          }
        Mat SC = be.gemm(S, false, C, false);
        return be.gemm(be.gemm(SC, false, fmo, false), false, SC, true);
      };
      if (restr)
        Fd = damp(Fd, Ca, nela);
      else {
        Mat Fda(Nb, Nb), Fdb(Nb, Nb);
        std::copy(Fd.d.begin(), Fd.d.begin() + Nb * Nb, Fda.d.begin());
        std::copy(Fd.d.begin() + Nb * Nb, Fd.d.end(), Fdb.d.begin());
        Fda = damp(Fda, Ca, nela);
        Fdb = damp(Fdb, Cb, nelb);
        std::copy(Fda.d.begin(), Fda.d.end(), Fd.d.begin());
        std::copy(Fdb.d.begin(), Fdb.d.end(), Fd.d.begin() + Nb * Nb);
      }
    }

    t0 = wall();
    if (restr) {
      be.eig_gsym_sub(Ea, Ca, Fd, Sinvh, dsym);
      if (occ.active(it)) enforce_occupations(Ca, Ea, S, occ.na, occ.sym);  // main.cpp:942-944
    } else {
      Mat Fda(Nb, Nb), Fdb(Nb, Nb);
      std::copy(Fd.d.begin(), Fd.d.begin() + Nb * Nb, Fda.d.begin());
      std::copy(Fd.d.begin() + Nb * Nb, Fd.d.end(), Fdb.d.begin());
      be.eig_gsym_sub(Ea, Ca, Fda, Sinvh, dsym);
      be.eig_gsym_sub(Eb, Cb, Fdb, Sinvh, dsym);
      if (occ.active(it)) {
        enforce_occupations(Ca, Ea, S, occ.na, occ.sym);
        enforce_occupations(Cb, Eb, S, occ.nb, occ.sym);  // main.cpp:956-958
      }
    }
    res.tdiag = wall() - t0;
    if (verbose) {
      printf("%s diagonalization done in %.6f\n", symm ? "Subspace" : "Full", res.tdiag);
      if (Ea.size() > nela && nela) printf("Alpha HOMO-LUMO gap is % .3f eV\n", (Ea[nela] - Ea[nela - 1]) * 27.211386);
      if (!restr && Eb.size() > nelb && nelb)
        printf("Beta  HOMO-LUMO gap is % .3f eV\n", (Eb[nelb] - Eb[nelb - 1]) * 27.211386);
      fflush(stdout);
    }
    res.iterations = it;
    if (convd) {
      res.converged = true;
      break;
    }
  }
  res.E = Ea;
  res.C = Ca;
  res.P = P;
  res.F = Fa;
  res.Eb = restr ? Ea : Eb;
  res.Cb = restr ? Ca : Cb;
  res.Fb = restr ? Fa : Fb;
  if (opt.keep_matrices) {
    std::map<std::string, Mat> &m = res.mats;
    m["S"] = S;
    m["T"] = T;
    m["Vnuc"] = Vnuc;
    m["H0"] = H0;
    m["Sinvh"] = Sinvh;
    m["P"] = P;
    m["Pa"] = last.Pa;
    m["Pb"] = last.Pb;
    m["J"] = last.J;
    m["Ka"] = last.Ka;
    m["Kb"] = last.Kb;
    m["XCa"] = last.XCa;
    m["XCb"] = last.XCb;
    m["Fa"] = res.F;
    m["Fb"] = res.Fb;
    m["Ca"] = res.C;
    m["Cb"] = res.Cb;
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
}  // namespace

void parse_nela_nelb(int &nela, int &nelb, int &Q, int &M, int Ztot) {
  if (nela == 0 && nelb == 0) {
    const int nel = Ztot - Q;
    if (M < 1) throw std::runtime_error("Invalid value for multiplicity, which must be >=1.\n");
    if ((nel % 2 == 0 && M % 2 != 1) || (nel % 2 == 1 && M % 2 != 0))
      throw std::runtime_error("Requested multiplicity " + std::to_string(M) + " with " + std::to_string(nel) + " electrons.\n");
    nela = (nel % 2 == 0) ? nel / 2 + (M - 1) / 2 : nel / 2 + M / 2;
    nelb = nel - nela;
    if (nela < 0) throw std::runtime_error("A multiplicity of " + std::to_string(M) + " would mean " + std::to_string(nela) + " alpha electrons!\n");
    if (nelb < 0) throw std::runtime_error("A multiplicity of " + std::to_string(M) + " would mean " + std::to_string(nelb) + " beta electrons!\n");
  } else {
    Q = Ztot - nela - nelb;
    M = 1 + nela - nelb;
    if (M < 1)
      throw std::runtime_error("nela=" + std::to_string(nela) + ", nelb=" + std::to_string(nelb) + " would mean multiplicity " +
                               std::to_string(M) + " which is not allowed!\n");
  }
}

static ModelPotential guess_potential(int iguess, int Z, double gsz_d) {
  ModelPotential p;
  p.Z = Z;
  if (iguess == 0) p.kind = 0;
  else if (iguess == 1) {
    p.kind = 1;
    p.d = gsz_d;
    if (!(gsz_d > 0.0)) throw std::logic_error("GSZ guess: the screening length d_Z must be given\n");
  } else if (iguess == 3) p.kind = 3;
  else throw std::logic_error("Unsupported guess\n");
  if (Z == 0) p.kind = 0;  // a ghost centre has no potential
  return p;
}

Result run_diatomic(const Options &opt, Backend &be) {
  if (opt.omega != 0.0) throw std::logic_error("Range separated functionals are not supported.\n");  // diatomic/main.cpp:393
  Result res;
  const bool verbose = opt.verbose;
  Problem pb;
  int nel = opt.Z1 + opt.Z2 - opt.Q;

  int Nquad = opt.nquad;
  if (Nquad == 0) Nquad = 5 * opt.nnodes;
  else if (Nquad < 2 * opt.nnodes) throw std::logic_error("Insufficient radial quadrature.\n");

  IVec lval, mval;
  diatomic::lm_to_l_m(opt.lmmax, lval, mval);
  double Rhalf = 0.5 * opt.Rbond;
  double mumax = arcosh(opt.Rmax / Rhalf);
  Vec bval = get_grid(mumax, opt.nelem, opt.igrid, opt.zexp);

  diatomic::TwoDBasis basis(opt.Z1, opt.Z2, Rhalf, opt.nnodes, Nquad, bval, lval, mval, opt.lpad);
  res.Nbf = basis.Nbf();
  if (verbose)
    printf("Basis set consists of %i angular shells composed of %i radial functions, totaling %i basis functions\n",
           (int)basis.Nang(), (int)basis.Nrad(), (int)basis.Nbf());
  res.Enucr = opt.Z1 * opt.Z2 / opt.Rbond;

  const bool dft = (opt.x_func > 0 || opt.c_func > 0);
  int ldft = opt.ldft, mdft = opt.mdft;
  if (dft || opt.iguess != 0) {  // the model-potential guess uses the same product grid (lquad of main.cpp:689)
    int lmaxmax = 0;
    for (int l : opt.lmmax) lmaxmax = std::max(lmaxmax, l);
    if (ldft == 0) ldft = 4 * lmaxmax + 12;
    if (ldft < 2 * lmaxmax + 2) throw std::logic_error("Increase ldft to guarantee accuracy of quadrature!\n");
    if (mdft == 0) mdft = 4 * (int)opt.lmmax.size() + 5;
    if (mdft < 2 * (int)opt.lmmax.size()) throw std::logic_error("Increase mdft to guarantee accuracy of quadrature!\n");
  }

  int symm = opt.symmetry;
  if (symm == 2 && opt.Z1 != opt.Z2) symm = 1;
  pb.symm = symm;
  pb.dsym = basis.get_sym_idx(symm);

  pb.nel = nel;
  pb.S = basis.overlap();
  pb.T = basis.kinetic();
  pb.Vnuc = basis.nuclear();
  pb.guess1 = guess_potential(opt.iguess, opt.Z1, opt.gsz_d1);
  pb.guess2 = guess_potential(opt.iguess, opt.Z2, opt.gsz_d2);
  pb.compute_tei_and_prepare = [&]() {
    basis.compute_tei(opt.kfrac != 0.0);
    be.prepare(basis, opt.kfrac != 0.0, ldft, mdft);
  };
  Options oo = opt;
  oo.symmetry = symm;
  pb.occupations = [oo, &basis](int na, int nb) { return occupation_plan(oo, basis, na, nb); };
  if (opt.guess_basis) pb.guess_overlap = [&opt, &basis]() { return basis.overlap(*opt.guess_basis); };
  return scf_loop(opt, be, pb, res);
}

// index lists of atomic/main.cpp:308-312: for every l the functions of the shells (l,m), all m
std::vector<std::vector<std::vector<size_t> > > atomic_average_groups(const atomic::TwoDBasis &basis) {
  int lmax = 0;
  for (int l : basis.lval) lmax = std::max(lmax, l);
  std::vector<std::vector<std::vector<size_t> > > grp(lmax + 1);
  for (int l = 0; l <= lmax; l++)
    for (size_t a = 0; a < basis.Nang(); a++)
      if (basis.lval[a] == l) grp[l].push_back(basis.lm_indices(l, basis.mval[a]));
  return grp;
}

Result run_atomic(const AtomicOptions &aopt, Backend &be) {
  const Options &opt = aopt.common;
  Result res;
  const bool verbose = opt.verbose;
  Problem pb;
  int nel = aopt.Z - aopt.Q;
  if (nel <= 0) throw std::logic_error("No electrons.\n");

  // atomic/main.cpp:245-251
  int Nquad = opt.nquad;
  if (Nquad == 0) Nquad = 5 * opt.nnodes;
  else if (Nquad < 2 * opt.nnodes) throw std::logic_error("Insufficient radial quadrature.\n");

  IVec lval, mval;
  atomic::angular_basis(aopt.lmax, aopt.mmax, lval, mval);
  Vec bval = get_grid(opt.Rmax, opt.nelem, opt.igrid, opt.zexp);
  atomic::TwoDBasis basis(aopt.Z, opt.nnodes, Nquad, bval, lval, mval);
  res.Nbf = basis.Nbf();
  if (verbose)
    printf("Basis set consists of %i angular shells composed of %i radial functions, totaling %i basis functions\n",
           (int)basis.Nang(), (int)basis.Nrad(), (int)basis.Nbf());
  res.Enucr = 0.0;

  const bool dft = (opt.x_func > 0 || opt.c_func > 0);
  int ldft = opt.ldft, mdft = opt.mdft;
  if (dft) {  // atomic/main.cpp:389-403
    if (ldft == 0) ldft = 4 * aopt.lmax + 10;
    if (ldft < 2 * aopt.lmax) throw std::logic_error("Increase ldft to guarantee accuracy of quadrature!\n");
    if (mdft == 0) mdft = 4 * aopt.mmax + 5;
    if (mdft < 2 * aopt.mmax) throw std::logic_error("Increase mdft to guarantee accuracy of quadrature!\n");
  }
  pb.symm = opt.symmetry;
  pb.dsym = basis.get_sym_idx(pb.symm);
  if (aopt.maverage) pb.avg_idx = atomic_average_groups(basis);
  pb.nel = nel;
  pb.S = basis.overlap();
  pb.T = basis.kinetic();
  pb.Vnuc = basis.nuclear();
  pb.guess1 = guess_potential(opt.iguess, aopt.Z, opt.gsz_d1);
  pb.guess2 = pb.guess1;
  if (verbose && opt.omega != 0.0) {  // atomic/main.cpp:363-370
    printf("\nUsing range-separated exchange with range-separation constant omega = % .3f.\n", opt.omega);
    printf("Using % .3f %% short-range and % .3f %% long-range exchange.\n", (opt.kfrac + opt.kshort) * 100, opt.kfrac * 100);
    printf("Range separation is done with the %s kernel.\n", opt.rs_kind == 1 ? "Yukawa" : "error function");
  }
  pb.compute_tei_and_prepare = [&]() {
    basis.compute_tei(opt.kfrac != 0.0);
    if (opt.omega != 0.0) {  // atomic/main.cpp:709-712
      if (opt.rs_kind == 1) basis.compute_yukawa(opt.omega);
      else basis.compute_erfc(opt.omega);
    }
    be.prepare_atomic(basis, opt.kfrac != 0.0, ldft, mdft);
  };
  pb.occupations = [&opt, &basis](int na, int nb) { return occupation_plan(opt, basis, na, nb); };
  if (opt.guess_basis_atomic) pb.guess_overlap = [&opt, &basis]() { return basis.overlap(*opt.guess_basis_atomic); };
  return scf_loop(opt, be, pb, res);
}

}  // namespace scf
}  // namespace helfem
