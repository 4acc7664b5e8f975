// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  C entry points for ctypes (tests, bench
// cpu_baseline, smoke).  Plain pointers, column-major doubles, int status (0 = ok).
#include "oracle.h"
#include "oracle_scf.h"
#include <cstdint>
#include <cstring>
#include <string>

using namespace oracle;
using helfem::diatomic::TwoDBasis;

static thread_local std::string g_err;
static thread_local int g_orc_readocc = 0;
static thread_local std::vector<std::vector<int> > g_orc_occs;
static thread_local int g_orc_iguess = 0;  // --iguess of the following orc_scf_* calls
static thread_local double g_orc_diiseps = 1e-2, g_orc_diisthr = 1e-3;  // --diiseps / --diisthr of the following calls
#define ORC_TRY try {
#define ORC_CATCH                   \
  }                                 \
  catch (const std::exception &e) { \
    g_err = e.what();               \
    return 1;                       \
  }                                 \
  return 0;

static Mat to_mat(const double *p, size_t r, size_t c) {
  Mat M(r, c);
  memcpy(M.memptr(), p, sizeof(double) * r * c);
  return M;
}
static std::vector<std::vector<size_t> > to_blocks(int nblk, const int64_t *ptr, const int64_t *idx) {
  std::vector<std::vector<size_t> > b(nblk);
  for (int i = 0; i < nblk; i++)
    for (int64_t k = ptr[i]; k < ptr[i + 1]; k++) b[i].push_back((size_t)idx[k]);
  return b;
}


extern "C" {

const char *orc_last_error() { return g_err.c_str(); }

/// test hook: evaluate P_L^M/Q_L^M through fn (e.g. the reference's Fortran library in oracle/_ref)
void orc_set_legendre_provider(helfem::legendre_provider_t fn) { helfem::set_legendre_provider(fn); }

int orc_basis_create(int Z1, int Z2, double Rhalf, int nnodes, int nquad, const double *bval, int nbval,
                     const int *lval, const int *mval, int nang, int lpad, void **out) {
  ORC_TRY
  *out = new TwoDBasis(Z1, Z2, Rhalf, nnodes, nquad, Vec(bval, bval + nbval), helfem::IVec(lval, lval + nang),
                       helfem::IVec(mval, mval + nang), lpad);
  ORC_CATCH
}
int orc_basis_destroy(void *h) {
  delete (TwoDBasis *)h;
  return 0;
}
int orc_basis_dims(void *h, int64_t *Nbf, int64_t *Ndummy, int64_t *Nrad, int64_t *Nang, int64_t *Nel) {
  TwoDBasis *b = (TwoDBasis *)h;
  *Nbf = b->Nbf();
  *Ndummy = b->Ndummy();
  *Nrad = b->Nrad();
  *Nang = b->Nang();
  *Nel = b->Nel();
  return 0;
}
int orc_compute_tei(void *h, int exchange) {
  ORC_TRY((TwoDBasis *)h)->compute_tei(exchange != 0);
  ORC_CATCH
}
int orc_coulomb(void *h, const double *P, double *J) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Jm = coulomb(*b, to_mat(P, N, N));
  memcpy(J, Jm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_exchange(void *h, const double *P, double *K) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Km = exchange(*b, to_mat(P, N, N));
  memcpy(K, Km.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_exchange_blocks(void *h, const double *P, double *K, int nsel, const int *jsel, const int *ksel) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  std::vector<std::pair<int, int> > only;
  for (int i = 0; i < nsel; i++) only.push_back(std::make_pair(jsel[i], ksel[i]));
  Mat Km = exchange(*b, to_mat(P, N, N), &only);
  memcpy(K, Km.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_coulomb_shard(void *h, const double *P, double *J, int shard_rank, int shard_n) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Jm = coulomb(*b, to_mat(P, N, N), shard_rank, shard_n);
  memcpy(J, Jm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_eval_fxc_shard(void *h, int lang, int mang, int x_func, int c_func, const double *P, double *H, double *Exc,
                       double *Nel, double *Ekin, double thr, int shard_rank, int shard_n) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Hm;
  eval_Fxc(*b, lang, mang, x_func, c_func, to_mat(P, N, N), Hm, *Exc, *Nel, *Ekin, thr, 0, -1, shard_rank, shard_n);
  memcpy(H, Hm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_eval_fxc(void *h, int lang, int mang, int x_func, int c_func, const double *P, double *H, double *Exc,
                 double *Nel, double *Ekin, double thr, long q_begin, long q_end) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Hm;
  eval_Fxc(*b, lang, mang, x_func, c_func, to_mat(P, N, N), Hm, *Exc, *Nel, *Ekin, thr, q_begin, q_end);
  memcpy(H, Hm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_eval_fxc_pol_range(void *h, int lang, int mang, int x_func, int c_func, const double *Pa, const double *Pb, double *Ha,
                           double *Hb, double *Exc, double *Nel, double *Ekin, double thr, long q_begin, long q_end) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Ham, Hbm;
  eval_Fxc_pol(*b, lang, mang, x_func, c_func, to_mat(Pa, N, N), to_mat(Pb, N, N), Ham, Hbm, *Exc, *Nel, *Ekin, thr, q_begin,
               q_end);
  memcpy(Ha, Ham.memptr(), sizeof(double) * N * N);
  memcpy(Hb, Hbm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_grid_overlap(void *h, int lang, int mang, double *S) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  Mat Sm = grid_overlap(*b, lang, mang);
  memcpy(S, Sm.memptr(), sizeof(double) * Sm.n_elem());
  ORC_CATCH
}
int orc_grid_kinetic(void *h, int lang, int mang, double *T) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  Mat Tm = grid_kinetic(*b, lang, mang);
  memcpy(T, Tm.memptr(), sizeof(double) * Tm.n_elem());
  ORC_CATCH
}
int orc_eig_sym(int64_t n, const double *A, double *E, double *C) {
  ORC_TRY
  Vec Ev;
  Mat Cm;
  eig_sym(Ev, Cm, to_mat(A, n, n));
  memcpy(E, Ev.data(), sizeof(double) * n);
  memcpy(C, Cm.memptr(), sizeof(double) * n * n);
  ORC_CATCH
}
int orc_eig_gsym(int64_t N, int64_t n, const double *F, const double *Sinvh, double *E, double *C) {
  ORC_TRY
  Vec Ev;
  Mat Cm;
  eig_gsym(Ev, Cm, to_mat(F, N, N), to_mat(Sinvh, N, n));
  memcpy(E, Ev.data(), sizeof(double) * n);
  memcpy(C, Cm.memptr(), sizeof(double) * N * n);
  ORC_CATCH
}
int orc_eig_gsym_sub(int64_t N, const double *F, const double *Sinvh, int nblk, const int64_t *blk_ptr,
                     const int64_t *blk_idx, double *E, double *C) {
  ORC_TRY
  Vec Ev;
  Mat Cm;
  eig_gsym_sub(Ev, Cm, to_mat(F, N, N), to_mat(Sinvh, N, N), to_blocks(nblk, blk_ptr, blk_idx));
  memcpy(E, Ev.data(), sizeof(double) * N);
  memcpy(C, Cm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_form_sinvh(int64_t N, const double *S, int chol, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                   double *Sinvh) {
  ORC_TRY
  Mat X = form_Sinvh(to_mat(S, N, N), chol != 0, to_blocks(nblk, blk_ptr, blk_idx));
  memcpy(Sinvh, X.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_form_density(int64_t N, int64_t ncols, const double *C, int64_t nocc, double *P) {
  ORC_TRY
  Mat Pm = form_density(to_mat(C, N, ncols), nocc);
  memcpy(P, Pm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_xc_unpolarized(int func_id, int64_t N, const double *rho, const double *sigma, double *exc, double *vrho,
                       double *vsigma, double thr) {
  ORC_TRY
  xc_unpolarized(func_id, N, rho, sigma, exc, vrho, vsigma, thr);
  ORC_CATCH
}

int orc_xc_unpolarized_mgga(int func_id, int64_t N, const double *rho, const double *sigma, const double *tau, double *exc,
                            double *vrho, double *vsigma, double *vtau, double thr) {
  ORC_TRY
  xc_unpolarized_mgga(func_id, N, rho, sigma, tau, exc, vrho, vsigma, vtau, thr);
  ORC_CATCH
}
int orc_xc_polarized_mgga(int func_id, int64_t n, const double *rho, const double *sigma, const double *tau, double *exc,
                          double *vrho, double *vsigma, double *vtau, double thr) {
  ORC_TRY
  xc_polarized_mgga(func_id, (size_t)n, rho, sigma, tau, exc, vrho, vsigma, vtau, thr);
  ORC_CATCH
}
int orc_xc_polarized(int func_id, int64_t N, const double *rho, const double *sigma, double *exc, double *vrho,
                     double *vsigma, double thr) {
  ORC_TRY
  xc_polarized(func_id, N, rho, sigma, exc, vrho, vsigma, thr);
  ORC_CATCH
}
int orc_eval_fxc_pol(void *h, int lang, int mang, int x_func, int c_func, const double *Pa, const double *Pb, double *Ha,
                     double *Hb, double *Exc, double *Nel, double *Ekin, double thr) {
  ORC_TRY
  TwoDBasis *b = (TwoDBasis *)h;
  size_t N = b->Nbf();
  Mat Ham, Hbm;
  eval_Fxc_pol(*b, lang, mang, x_func, c_func, to_mat(Pa, N, N), to_mat(Pb, N, N), Ham, Hbm, *Exc, *Nel, *Ekin, thr);
  memcpy(Ha, Ham.memptr(), sizeof(double) * N * N);
  memcpy(Hb, Hbm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}

/// Restricted closed-shell diatomic SCF on the CPU oracle.  out[0..7] = Etot, Ekin, Epot, Ecoul, Exx, Exc,
/// Enucr, iterations(+0.5 if converged)
int orc_scf_diatomic(int Z1, int Z2, double Rbond, const int *lmmax, int nlm, int nelem, int nnodes, int nquad,
                     double Rmax, int igrid, double zexp, int lpad, const char *method, int ldft, int mdft,
                     int symmetry, int multiplicity, int maxit, double convthr, int verbose, double *out) {
  ORC_TRY
  oracle::ScfIn o;
  o.multiplicity = multiplicity < 0 ? -multiplicity : multiplicity;  // negative: restricted open shell (ROHF)
  if (multiplicity < 0) o.restricted = 1;
  o.Z1 = Z1;
  o.Z2 = Z2;
  o.Rbond = Rbond;
  o.lmmax.assign(lmmax, lmmax + nlm);
  o.nelem = nelem;
  o.nnodes = nnodes;
  o.nquad = nquad;
  o.Rmax = Rmax;
  o.igrid = igrid;
  o.zexp = zexp;
  o.lpad = lpad;
  parse_xc_func(o.x_func, o.c_func, method);
  o.kfrac = (o.x_func == -1) ? 1.0 : (o.x_func == 406 ? 0.25 : (o.x_func == 402 ? 0.20 : 0.0));
  o.iguess = g_orc_iguess;
  o.readocc = g_orc_readocc;
  o.occs = g_orc_occs;
  o.diiseps = g_orc_diiseps;
  o.diisthr = g_orc_diisthr;
  o.ldft = ldft;
  o.mdft = mdft;
  o.symmetry = symmetry;
  o.maxit = maxit;
  o.convthr = convthr;
  o.verbose = verbose != 0;
  oracle::ScfOut r = oracle::scf_diatomic(o);
  out[0] = r.Etot;
  out[1] = r.Ekin;
  out[2] = r.Epot;
  out[3] = r.Ecoul;
  out[4] = r.Exx;
  out[5] = r.Exc;
  out[6] = r.Enucr;
  out[7] = r.iterations + (r.converged ? 0.5 : 0.0);
  ORC_CATCH
}


// ---- atomic program ----
typedef helfem::atomic::TwoDBasis ABasis;

int orc_atomic_basis_create(int Z, int nnodes, int nquad, const double *bval, int nbval, const int *lval,
                            const int *mval, int nang, void **out) {
  ORC_TRY
  *out = new ABasis(Z, nnodes, nquad, Vec(bval, bval + nbval), helfem::IVec(lval, lval + nang),
                    helfem::IVec(mval, mval + nang));
  ORC_CATCH
}
int orc_atomic_basis_destroy(void *h) {
  delete (ABasis *)h;
  return 0;
}
int orc_atomic_basis_dims(void *h, int64_t *Nbf, int64_t *Nrad, int64_t *Nang, int64_t *Nel) {
  ABasis *b = (ABasis *)h;
  *Nbf = b->Nbf();
  *Nrad = b->Nrad();
  *Nang = b->Nang();
  *Nel = b->Nel();
  return 0;
}
/// which: 0 overlap, 1 kinetic, 2 nuclear
int orc_atomic_onebody(void *h, int which, double *out) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  Mat M = which == 0 ? b->overlap() : which == 1 ? b->kinetic() : b->nuclear();
  memcpy(out, M.memptr(), sizeof(double) * M.n_elem());
  ORC_CATCH
}
int orc_atomic_compute_tei(void *h, int exchange) {
  ORC_TRY((ABasis *)h)->compute_tei(exchange != 0);
  ORC_CATCH
}
int orc_atomic_coulomb(void *h, const double *P, double *J) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  size_t N = b->Nbf();
  Mat Jm = atomic_coulomb(*b, to_mat(P, N, N));
  memcpy(J, Jm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_atomic_exchange(void *h, const double *P, double *K) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  size_t N = b->Nbf();
  Mat Km = atomic_exchange(*b, to_mat(P, N, N));
  memcpy(K, Km.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
/// rs_kind 1: compute_yukawa(omega), 2: compute_erfc(omega)   (TwoDBasis.cpp:741/780)
int orc_atomic_compute_rs(void *h, int rs_kind, double omega) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  if (rs_kind == 1) b->compute_yukawa(omega);
  else if (rs_kind == 2) b->compute_erfc(omega);
  else throw std::logic_error("unknown range-separation kernel");
  ORC_CATCH
}
int orc_atomic_rs_exchange(void *h, const double *P, double *K) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  size_t N = b->Nbf();
  Mat Km = atomic_rs_exchange(*b, to_mat(P, N, N));
  memcpy(K, Km.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
/// --readocc + the rows of occs.dat for the following orc_scf_* calls of this thread (readocc = 0 switches it off)
int orc_scf_set_occupations(int readocc, int nrows, int ncols, const int *rows) {
  g_orc_readocc = readocc;
  g_orc_occs.clear();
  for (int r = 0; r < nrows; r++) g_orc_occs.push_back(std::vector<int>(rows + (size_t)r * ncols, rows + (size_t)(r + 1) * ncols));
  return 0;
}
/// --iguess for the following orc_scf_* calls of this thread (0 core, 3 Thomas-Fermi)
int orc_scf_set_iguess(int iguess) {
  g_orc_iguess = iguess;
  return 0;
}
int orc_set_xc_params(int x_func, const double *x_pars, int nx, int c_func, const double *c_pars, int nc) {
  ORC_TRY
  oracle::set_xc_params(x_pars, nx, x_func, c_pars, nc, c_func);
  ORC_CATCH
}
/// --diiseps / --diisthr for the following orc_scf_* calls of this thread
int orc_scf_set_diis(double diiseps, double diisthr) {
  g_orc_diiseps = diiseps;
  g_orc_diisthr = diisthr;
  return 0;
}
/// kind/Z/d/H of the two centres; atomic bases ignore the second one
int orc_model_potential(void *h, int atomic, int lang, int mang, int kind1, int Z1, double d1, double H1, int kind2, int Z2,
                        double d2, double H2, double *out) {
  ORC_TRY
  helfem::ModelPotential p1, p2;
  p1.kind = kind1; p1.Z = Z1; p1.d = d1; p1.H = H1;
  p2.kind = kind2; p2.Z = Z2; p2.d = d2; p2.H = H2;
  Mat V = atomic ? ((ABasis *)h)->model_potential(p1) : model_potential(*(TwoDBasis *)h, lang, mang, p1, p2);
  memcpy(out, V.memptr(), sizeof(double) * V.n_elem());
  ORC_CATCH
}
/// test hook, see special.h: 1 = the reference's binomial helper inside the erfc short-range series
void orc_set_erfc_binomial_mode(int mode) { helfem::set_erfc_binomial_mode(mode); }
double orc_bessel_il(double x, int L) { return helfem::bessel_il(x, L); }
double orc_bessel_kl(double x, int L) { return helfem::bessel_kl(x, L); }
double orc_erfc_phi(int n, double Xi, double xi) { return helfem::erfc_Phi(n, Xi, xi); }
int orc_atomic_eval_fxc(void *h, int lang, int mang, int x_func, int c_func, const double *P, double *H, double *Exc,
                        double *Nel, double *Ekin, double thr) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  size_t N = b->Nbf();
  Mat Hm;
  atomic_eval_Fxc(*b, lang, mang, x_func, c_func, to_mat(P, N, N), Hm, *Exc, *Nel, *Ekin, thr);
  memcpy(H, Hm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
int orc_atomic_eval_fxc_pol(void *h, int lang, int mang, int x_func, int c_func, const double *Pa, const double *Pb,
                            double *Ha, double *Hb, double *Exc, double *Nel, double *Ekin, double thr) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  size_t N = b->Nbf();
  Mat Ham, Hbm;
  atomic_eval_Fxc_pol(*b, lang, mang, x_func, c_func, to_mat(Pa, N, N), to_mat(Pb, N, N), Ham, Hbm, *Exc, *Nel, *Ekin,
                      thr);
  memcpy(Ha, Ham.memptr(), sizeof(double) * N * N);
  memcpy(Hb, Hbm.memptr(), sizeof(double) * N * N);
  ORC_CATCH
}
/// \int B_i B_j r^n dr over the whole radial basis (Nrad x Nrad) -- checked against the Maple rationals of
/// the reference's src/atomic/inttest.cpp
int orc_atomic_radial_integral(void *h, int n, double *out) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  size_t Nrad = b->Nrad();
  Mat M(Nrad, Nrad);
  for (size_t iel = 0; iel < b->Nel(); iel++) {
    size_t f, l;
    b->fem.get_idx(iel, f, l);
    Mat s = b->radial_integral(n, iel);
    for (size_t j = 0; j < s.n_cols; j++)
      for (size_t i = 0; i < s.n_rows; i++) M(f + i, f + j) += s(i, j);
  }
  memcpy(out, M.memptr(), sizeof(double) * M.n_elem());
  ORC_CATCH
}

/// twoe_integral on [0,R] for the full nnodes-node LIP set (nothing dropped) -- the set-up of the reference's
/// src/atomic/inttest.cpp; out is n^2 x n^2 with n = nnodes, without the 4 pi/(2L+1) factor
int orc_atomic_twoe_integral(double rmin, double rmax, int nnodes, int nquad, int L, double *out) {
  ORC_TRY
  Vec xq, wq;
  helfem::chebyshev_rule(nquad, xq, wq);
  helfem::LIPBasis poly(helfem::lobatto_nodes(nnodes));
  Mat m = helfem::atomic::twoe_integral(rmin, rmax, xq, wq, poly, L);
  memcpy(out, m.memptr(), sizeof(double) * m.n_elem());
  ORC_CATCH
}

/// in-element two-electron integral table of (L, iel), n^2 x n^2 without the 4 pi/(2L+1) factor
int orc_atomic_prim_tei(void *h, int L, int iel, double *out, int64_t *n) {
  ORC_TRY
  ABasis *b = (ABasis *)h;
  if (!b->have_tei) throw std::logic_error("Primitive teis have not been computed!\n");
  const Mat &m = b->prim_tei.at((size_t)L * b->Nel() + iel);
  *n = (int64_t)m.n_rows;
  memcpy(out, m.memptr(), sizeof(double) * m.n_elem());
  ORC_CATCH
}

/// Restricted closed-shell atomic SCF on the CPU oracle; out as for orc_scf_diatomic
int orc_scf_atomic(int Z, int Q, int lmax, int mmax, int nelem, int nnodes, int nquad, double Rmax, int igrid,
                   double zexp, const char *method, int ldft, int mdft, int symmetry, int multiplicity, int maverage,
                   int maxit, double convthr, int verbose, double *out) {
  ORC_TRY
  oracle::ScfIn o;
  o.maverage = maverage != 0;
  o.dampfock = 0.7;  // defaults of the atomic program (atomic/main.cpp:111-112)
  o.dampthr = 0.1;
  o.multiplicity = multiplicity < 0 ? -multiplicity : multiplicity;  // negative: restricted open shell (ROHF)
  if (multiplicity < 0) o.restricted = 1;
  o.Z1 = Z;
  o.Q = Q;
  o.lmax = lmax;
  o.mmax = mmax;
  o.nelem = nelem;
  o.nnodes = nnodes;
  o.nquad = nquad;
  o.Rmax = Rmax;
  o.igrid = igrid;
  o.zexp = zexp;
  parse_xc_func(o.x_func, o.c_func, method);
  o.kfrac = (o.x_func == -1) ? 1.0 : (o.x_func == 406 ? 0.25 : (o.x_func == 402 ? 0.20 : 0.0));
  if (o.x_func == 178) {  // hyb_lda_xc_cam_lda0: omega = 1/3, alpha = 1/2, beta = -1/4, erfc kernel
    o.kfrac = 0.5;
    o.kshort = -0.25;
    o.omega = 1.0 / 3.0;
    o.rs_kind = 2;
  }
  o.iguess = g_orc_iguess;
  o.readocc = g_orc_readocc;
  o.occs = g_orc_occs;
  o.diiseps = g_orc_diiseps;
  o.diisthr = g_orc_diisthr;
  o.ldft = ldft;
  o.mdft = mdft;
  o.symmetry = symmetry;
  o.maxit = maxit;
  o.convthr = convthr;
  o.verbose = verbose != 0;
  oracle::ScfOut r = oracle::scf_atomic(o);
  out[0] = r.Etot;
  out[1] = r.Ekin;
  out[2] = r.Epot;
  out[3] = r.Ecoul;
  out[4] = r.Exx;
  out[5] = r.Exc;
  out[6] = r.Enucr;
  out[7] = r.iterations + (r.converged ? 0.5 : 0.0);
  ORC_CATCH
}

}  // extern "C"
