/*
 * helfem_gpu.h — C ABI of the MI355X (gfx950) implementation of HelFEM's SCF hot path.
 *
 * The reference has no plugin/FFI layer: its hot path is a set of C++ member functions taking and
 * returning arma::mat (contiguous column-major double).  Each entry point below replaces one of
 * them; the C++ adapter with the reference's exact signatures is include/helfem_gpu_arma.hpp and
 * the binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all matrices: column-major double, leading dimension = number of rows, caller-owned buffers;
 *   - matrices live in the *boundary-cleaned* N x N index space of the reference (Nbf), exactly
 *     what TwoDBasis::coulomb & friends take/return; the expansion to the Ndummy layout
 *     (basis.cpp:1754 expand_boundaries / :1735 remove_boundaries) happens inside the kernels;
 *   - "host" entry points take host pointers and copy through pinned staging; "_dev" entry points
 *     take pointers to HBM (e.g. torch tensors' data_ptr()) and enqueue on the context's stream
 *     without synchronising;
 *   - return value: 0 = success, non-zero = error, text via hfg_last_error() (thread-local).  The code is the class of
 *     the exception the reference throws in the same situation: 1 = std::logic_error (e.g. "Primitive teis have not
 *     been computed!" basis.cpp:1361), 2 = std::runtime_error (functional, shape and device errors, dftgrid.cpp:54),
 *     3 = any other; include/helfem_gpu_arma.hpp turns them back into those exceptions;
 *   - a context is bound to one device + one stream; one context per host thread.
 *   - there is NO CPU fallback: every compute entry point fails with an error when no gfx950
 *     device is usable.
 */
#ifndef HELFEM_GPU_H
#define HELFEM_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfg_ctx hfg_ctx;
typedef struct hfg_basis hfg_basis;

/* ---- context ------------------------------------------------------------------------------ */
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), HFG_NULL_STREAM for the
 * device's default (null) stream -- which is what a framework's "current stream" handle 0 means: kernels must be
 * enqueued THERE to stay ordered with the framework's own operations and collectives -- or NULL to let the context
 * create its own non-blocking stream. */
#define HFG_NULL_STREAM ((void *)(intptr_t)-1)
int hfg_ctx_create(hfg_ctx **ctx, int device, void *stream);
int hfg_ctx_destroy(hfg_ctx *ctx);
int hfg_ctx_synchronize(hfg_ctx *ctx);
const char *hfg_last_error(void);
const char *hfg_version(void);
/* number of visible HIP devices (0 when there is none or the runtime is unusable) */
int hfg_device_count(void);

/* Multi-GPU sharding of the Fock build: this context computes only the contributions of its
 * shard (rank of nranks); the caller sums the partial matrices over ranks (one all-reduce over
 * RCCL).  Default (0,1) = everything. */
int hfg_ctx_set_shard(hfg_ctx *ctx, int rank, int nranks);

/* The caller's promise that the DEVICE matrix at dSinvh keeps its contents until the next call of this function
 * (NULL withdraws it): S^{-1/2} is fixed through an SCF run (diatomic/main.cpp:470-479 forms it once), so
 * hfg_eig_blocks_dev / hfg_eig_gsym_sub_dev derive the column supports of the symmetry blocks
 * (scf_helpers.cpp:150-157) from it once instead of in every iteration.  Without it every call re-derives them. */
int hfg_ctx_fix_sinvh(hfg_ctx *ctx, const double *dSinvh);

/* ---- basis (host-side setup; no GPU needed) ------------------------------------------------
 * Constructor arguments of diatomic::basis::TwoDBasis (src/diatomic/basis.cpp:307):
 * Z1,Z2,Rhalf, primitive basis (only primbas 4 = LIP on Gauss-Lobatto nodes), n_quad, element
 * boundaries bval (mu values), angular shells (lval,mval), lpad. */
typedef struct {
  int Z1, Z2;
  double Rhalf;
  int primbas; /* must be 4 */
  int nnodes;
  int nquad;
  const double *bval;
  int nbval;
  const int *lval;
  const int *mval;
  int nang;
  int lpad;
} hfg_diatomic_desc;

int hfg_diatomic_basis_create(const hfg_diatomic_desc *desc, hfg_basis **basis);

/* Constructor arguments of helfem::atomic::basis::TwoDBasis (src/atomic/TwoDBasis.cpp:38-76) for the
 * point-nucleus case the atomic driver builds at src/atomic/main.cpp:268-273 (finitenuc = 0, no off-centre
 * charges, zeroder = 0).  The handle is used with every hfg_basis_* / hfg_coulomb / hfg_exchange / hfg_xc_fock
 * entry below exactly like a diatomic one (replacing TwoDBasis::coulomb / exchange, TwoDBasis.cpp:817/957,
 * and atomic::dftgrid::DFTGrid::eval_Fxc, src/atomic/dftgrid.cpp:810). */
typedef struct {
  int Z;
  int primbas;   /* 4 = LIP on Gauss-Lobatto nodes (the default); others are rejected */
  int nnodes;
  int nquad;
  const double *bval; /* element boundaries in r, bval[0] = 0 */
  int nbval;
  const int *lval;
  const int *mval;
  int nang;
} hfg_atomic_desc;
int hfg_atomic_basis_create(const hfg_atomic_desc *desc, hfg_basis **basis);
/* atomic::basis::angular_basis (src/atomic/basis.cpp:174): shell list for --lmax/--mmax */
int hfg_angular_basis(int lmax, int mmax, int *lval, int *mval, int *nang /* in: capacity, out: count */);
int hfg_basis_destroy(hfg_basis *basis);
/* Nbf, Ndummy, Nrad, Nang, Nel (TwoDBasis::Nbf/Ndummy/Nrad/Nang, basis.cpp:457-480) */
int hfg_basis_dims(const hfg_basis *basis, int64_t *Nbf, int64_t *Ndummy, int64_t *Nrad, int64_t *Nang,
                   int64_t *Nel);
/* one-electron matrices, N x N  (TwoDBasis::overlap / kinetic / nuclear, basis.cpp:677/752/780) */
int hfg_basis_overlap(const hfg_basis *basis, double *S);
int hfg_basis_kinetic(const hfg_basis *basis, double *T);
int hfg_basis_nuclear(const hfg_basis *basis, double *V);
/* symmetry blocks (TwoDBasis::get_sym_idx, basis.cpp:561): blk_ptr has nblk+1 entries, blk_idx Nbf.
 * Pass blk_ptr = NULL to query nblk only. */
int hfg_basis_sym_blocks(const hfg_basis *basis, int symm, int *nblk, int64_t *blk_ptr, int64_t *blk_idx);
/* TwoDBasis::compute_tei (basis.cpp:1166): primitive two-electron integral tables (host, threaded) */
int hfg_compute_tei(hfg_basis *basis, int exchange);
/* Range-separated exchange tables of the atomic program: rs_kind 1 = TwoDBasis::compute_yukawa(omega)
 * (src/atomic/TwoDBasis.cpp:741), 2 = TwoDBasis::compute_erfc(omega) (:780).  Call before hfg_basis_upload, which
 * then also places the screened-kernel tables in HBM.  A diatomic basis fails like the reference driver
 * ("Range separated functionals are not supported.", src/diatomic/main.cpp:393). */
int hfg_compute_rs_tei(hfg_basis *basis, int rs_kind, double omega);
/* the same tables built on the GPU (diatomic: host computes quadrature points and Legendre values, the
 * O(Nlm nq p^4) sums run on the device and the tables stay there); follow with hfg_basis_upload */
int hfg_compute_tei_dev(hfg_ctx *ctx, hfg_basis *basis, int exchange);
/* helpers of main.cpp:276-277: mu grid for --grid/--zexp, and (l,m) shell list for lmmax */
/* DIIS::get_w + solve_F (src/general/diis.cpp:214-290, 392-412): ADIIS + CDIIS weights of the Fock extrapolation from the
 * inner products of a history of n entries, oldest first: B[i*n+j] = err_i . err_j, T[i*n+j] = Tr(Pa_i Fa_j) + Tr(Pb_i Fb_j),
 * energies E[n], max |err| of the newest entry.  mode 0: mixed as the drivers run it (--diiseps / --diisthr), 1: CDIIS only,
 * 2: ADIIS only.  Host-side arithmetic (no GPU). */
int hfg_diis_weights(int n, const double *B, const double *T, const double *E, double maxerr, double diiseps, double diisthr,
                     int mode, double *w, int *dropped);

/* Read-back of the setup tables (test and diagnostic access; basis.cpp:1166-1302 fills them):
 *   hfg_basis_lm_map: the sorted (L,|M|) channel list of the constructor (basis.cpp:333-375); n in: capacity, out: count
 *   hfg_basis_get_prim: which 0-3 prim_tei00/02/20/22, 4-7 prim_ktei00/02/20/22, 8-11 disjoint_P0/P2/Q0/Q2 of channel ilm and
 *   element iel, column-major in the reference's shape (rows/cols returned; out may be NULL to query the shape).  Tables built
 *   by hfg_compute_tei_dev are copied back from the device (ctx required).  Atomic handles: ilm = L; which 0 prim_tei[L],
 *   4 prim_ktei[L], 8 disjoint_L, 10 disjoint_m1L, and after hfg_compute_rs_tei 12 disjoint_iL, 13 disjoint_kL, 14 rs_tei,
 *   15 rs_ktei (erfc tables, one per element pair: iel * Nel + kel in place of iel). */
/* arma::mat TwoDBasis::overlap(const TwoDBasis &rh)   basis.cpp:713 and atomic/TwoDBasis.cpp:330: interbasis overlap
 * <a|b>, Nbf(a) x Nbf(b), of two bases of the same program (the projection of a checkpoint's orbitals onto another
 * basis, --load) */
int hfg_basis_interbasis_overlap(const hfg_basis *a, const hfg_basis *b, double *S12);
int hfg_basis_lm_map(const hfg_basis *basis, int *L, int *M, int *n);
int hfg_basis_get_prim(hfg_ctx *ctx, const hfg_basis *basis, int which, int ilm, int iel, double *out, int64_t *rows,
                       int64_t *cols);
int hfg_radial_grid(double mumax, int nelem, int igrid, double zexp, double *bval /* nelem+1 */);
int hfg_lm_list(const int *lmmax, int nlm, int *lval, int *mval, int *nang /* in: capacity, out: count */);

/* host-side special functions (exposed for the parity tests against the reference's
 * gaunt_test / legendre_test / sphtest known answers) */
double hfg_gaunt_coefficient(int L, int M, int l, int m, int lp, int mp);          /* gaunt.cpp:35 */
double hfg_modified_gaunt_coefficient(int lj, int mj, int L, int M, int li, int mi); /* gaunt.cpp:55 */
void hfg_legendre_PQ(int Lmax, int Mmax, double xi, double *P, double *Q);           /* Legendre_Wrapper.f90:135,173 */
double hfg_theta_lm(int l, int m, double cth);                                       /* spherical_harmonics.cpp:25 */
double hfg_bessel_il(double x, int L);              /* utils::bessel_il, libhelfem/src/utils.cpp:47 */
double hfg_bessel_kl(double x, int L);              /* utils::bessel_kl, libhelfem/src/utils.cpp:59 */
double hfg_erfc_phi(int n, double Xi, double xi);   /* atomic::erfc_expn::Phi, libhelfem/src/erfc_expn.cpp:181 */
void hfg_chebyshev_rule(int n, double *x, double *w);                                /* chebyshev.cpp:22 */
void hfg_lobatto_nodes(int n, double *x);                                            /* lobatto.cpp:588 */

/* ---- tables -> HBM ------------------------------------------------------------------------- */
/* Uploads the Gaunt-coupling lists, disjoint/in-element integral tables and (ldft>0) the XC grid
 * tables (DFTGrid ctor, dftgrid.cpp:760) of this basis to the context's device. */
int hfg_basis_upload(hfg_ctx *ctx, hfg_basis *basis, int ldft, int mdft);

/* ---- per-iteration hot path, host-pointer API (drop-in for the arma::mat signatures) -------- */
/* arma::mat TwoDBasis::coulomb(const arma::mat & P) const            basis.h:247, basis.cpp:1359 */
int hfg_coulomb(hfg_ctx *ctx, hfg_basis *basis, const double *P, double *J);
/* arma::mat TwoDBasis::exchange(const arma::mat & P) const           basis.h:249, basis.cpp:1532 */
int hfg_exchange(hfg_ctx *ctx, hfg_basis *basis, const double *P, double *K);
/* arma::mat atomic::basis::TwoDBasis::rs_exchange(const arma::mat & P) const   TwoDBasis.h:184, TwoDBasis.cpp:1142 */
int hfg_rs_exchange(hfg_ctx *ctx, hfg_basis *basis, const double *P, double *K);
/* void DFTGrid::eval_Fxc(x_func,x_pars,c_func,c_pars,P,H,Exc,Nel,Ekin,thr)   dftgrid.h:179 (restricted).
 * Functional ids are libxc's: 1 lda_x, 7 lda_c_vwn, 12 lda_c_pw, 101 gga_x_pbe, 130 gga_c_pbe, 406 hyb_gga_xc_pbeh
 * (its 0.25 exact exchange is the caller's K), 202 mgga_x_tpss, 231 mgga_c_tpss (Ekin returns the integral of tau),
 * 13 lda_c_pw_mod, 546 lda_x_erf, 641 lda_x_yukawa (omega = 0.3, libxc's default), 178 hyb_lda_xc_cam_lda0 (DFT part;
 * the caller adds 0.5 K - 0.25 K_erfc(omega = 1/3)); <=0 none.  The spin-polarised entry takes the same ids. */
int hfg_xc_fock(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *P, double *H, double *Exc,
                double *Nel, double *Ekin, double dens_thr);
/* void DFTGrid::eval_Fxc(x_func,x_pars,c_func,c_pars,Pa,Pb,Ha,Hb,Exc,Nel,Ekin,beta,thr)   dftgrid.h:181,
 * dftgrid.cpp:812 (unrestricted; both spin matrices are always formed, i.e. beta = true) */
int hfg_xc_fock_pol(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *Pa, const double *Pb,
                    double *Ha, double *Hb, double *Exc, double *Nel, double *Ekin, double dens_thr);
/* The same with the reference's full argument list (dftgrid.h:179/181): x_pars / c_pars are the external functional
 * parameters of --x_pars / --c_pars (xc_func_set_ext_params in the reference, dftgrid.cpp:405-410).  Supported: gga_x_pbe
 * (kappa, mu), gga_c_pbe (beta, gamma, B), lda_x (alpha); NULL / 0 keeps the functional's defaults; any other
 * combination is refused with an error rather than ignored. */
int hfg_xc_fock_ext(hfg_ctx *ctx, hfg_basis *basis, int x_func, const double *x_pars, int n_x_pars, int c_func,
                    const double *c_pars, int n_c_pars, const double *P, double *H, double *Exc, double *Nel, double *Ekin,
                    double dens_thr);
int hfg_xc_fock_pol_ext(hfg_ctx *ctx, hfg_basis *basis, int x_func, const double *x_pars, int n_x_pars, int c_func,
                        const double *c_pars, int n_c_pars, const double *Pa, const double *Pb, double *Ha, double *Hb,
                        double *Exc, double *Nel, double *Ekin, double dens_thr);
/* Initial-guess model potential: arma::mat TwoDGrid::model_potential(p1, p2) (src/diatomic/twodquadrature.cpp:351) on the
 * quadrature grid of hfg_basis_upload(ldft, mdft), or atomic::basis::TwoDBasis::model_potential(pot)
 * (src/atomic/TwoDBasis.cpp:458; the second centre is ignored).  kind: 0 point nucleus, 1 Green-Sellin-Zachor with the
 * screening length d (H = d (Z-1)^0.4 when H <= 0), 3 Thomas-Fermi (model_potential.cpp / gsz.cpp of the reference);
 * 2 (SAP) needs the reference's tabulation and fails with "Unsupported guess". */
typedef struct {
  int kind;
  int Z;
  double d, H;
} hfg_model_pot;
int hfg_model_potential(hfg_ctx *ctx, hfg_basis *basis, const hfg_model_pot *p1, const hfg_model_pot *p2, double *H);
/* --iguess of the drivers (0 core, 3 Thomas-Fermi) for the following hfg_scf_* calls of this thread */
int hfg_scf_set_iguess(int iguess);
/* void scf::eig_gsym(E,C,F,Sinvh): Sinvh is N x n                     scf_helpers.h:34, .cpp:131 */
int hfg_eig_gsym(hfg_ctx *ctx, int64_t N, int64_t n, const double *F, const double *Sinvh, double *E, double *C);
/* void scf::eig_gsym_sub(E,C,F,Sinvh,m_idx)                           scf_helpers.h:36, .cpp:142 */
int hfg_eig_gsym_sub(hfg_ctx *ctx, int64_t N, const double *F, const double *Sinvh, int nblk,
                     const int64_t *blk_ptr, const int64_t *blk_idx, double *E, double *C);
/* the two consecutive scf::eig_gsym_sub calls of an unrestricted iteration (diatomic/main.cpp:936-958: same Sinvh, same
 * blocks, Fa then Fb) as ONE batch: the blocks of both matrices share the tridiagonalisation's chain of launches */
int hfg_eig_gsym_sub_pair(hfg_ctx *ctx, int64_t N, const double *Fa, const double *Fb, const double *Sinvh, int nblk,
                          const int64_t *blk_ptr, const int64_t *blk_idx, double *Ea, double *Ca, double *Eb, double *Cb);
/* arma::eig_sym(E,C,A) as used by utils::invh                         libhelfem/src/utils.cpp:172 */
int hfg_eig_sym(hfg_ctx *ctx, int64_t n, const double *A, double *E, double *C);
/* arma::mat TwoDBasis::Sinvh(bool chol, int sym) -> block-structured S^{-1/2}   basis.cpp:627 */
int hfg_form_sinvh(hfg_ctx *ctx, int64_t N, const double *S, int chol, int nblk, const int64_t *blk_ptr,
                   const int64_t *blk_idx, double *Sinvh);
/* arma::mat scf::form_density(C, nocc): P = C(:,0:nocc) C(:,0:nocc)^T   scf_helpers.cpp:22 */
int hfg_form_density(hfg_ctx *ctx, int64_t N, int64_t ncols, const double *C, int64_t nocc, double *P);
/* C = op(A) op(B), FP64 MFMA (the dense products around the eigensolver and in DIIS) */
int hfg_gemm(hfg_ctx *ctx, int transA, int transB, int64_t m, int64_t n, int64_t k, const double *A, int64_t lda,
             const double *B, int64_t ldb, double *C, int64_t ldc);

/* ---- device-resident API (pointers into HBM, asynchronous on the context's stream) ---------- */
int hfg_coulomb_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dJ);
int hfg_exchange_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dK);
int hfg_rs_exchange_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, double *dK);
/* TwoDBasis::exchange(Pa) as the SCF driver calls it (src/diatomic/main.cpp:822, src/atomic/main.cpp:767): the caller has
 * just formed dP = C_occ C_occ^T with scf::form_density from the nocc occupied orbitals dC (N x nocc, column-major, ld N)
 * and hands them over as the factors of dP, which saves the pivoted factorisation of dP and its verification. */
int hfg_exchange_occ_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dP, const double *dC, int64_t nocc, double *dK);
/* dScal: 3 doubles in HBM receiving Exc, Nel, Ekin */
int hfg_xc_fock_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dH,
                    double *dScal, double dens_thr);
int hfg_xc_fock_pol_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dPa, const double *dPb,
                        double *dHa, double *dHb, double *dScal /* 3: Exc, Nel, Ekin */, double dens_thr);
/* Fused, shardable Fock build for the SCF loop (main.cpp:808-900): J and the XC matrix are block-banded in
 * the radial index, so each rank produces its shard's contribution in a compact layout of
 * hfg_fock_compact_size() doubles; the caller all-reduces that buffer (and dScal) over ranks and
 * hfg_fock_finish_dev() forms F = enforce_fock_symmetry(H0 + J + XC) (dBlockId: symmetry block of
 * every basis function, or NULL). */
int64_t hfg_fock_compact_size(hfg_basis *basis);
int hfg_fock_compact_dev(hfg_ctx *ctx, hfg_basis *basis, int x_func, int c_func, const double *dP, double *dFc,
                         double *dScal, double dens_thr);
int hfg_fock_finish_dev(hfg_ctx *ctx, hfg_basis *basis, const double *dFc, const double *dH0, const int *dBlockId,
                        double *dF);
int hfg_eig_gsym_sub_dev(hfg_ctx *ctx, int64_t N, const double *dF, const double *dSinvh, int nblk,
                         const int64_t *blk_ptr, const int64_t *blk_idx, double *dE, double *dC);
/* Multi-GPU form of eig_gsym_sub: symmetry blocks are independent (scf_helpers.cpp:148-175), block ib is
 * solved by rank ib % nranks into a buffer of hfg_eig_block_buf_size() doubles (other slots zero); after a
 * sum all-reduce of that buffer every rank calls hfg_eig_assemble_dev() for the global sort (:183-185). */
int64_t hfg_eig_block_buf_size(int nblk, const int64_t *blk_ptr);
int hfg_eig_blocks_dev(hfg_ctx *ctx, int64_t N, const double *dF, const double *dSinvh, int nblk,
                       const int64_t *blk_ptr, const int64_t *blk_idx, double *dBlockBuf);
int hfg_eig_assemble_dev(hfg_ctx *ctx, int64_t N, int nblk, const int64_t *blk_ptr, const int64_t *blk_idx,
                         const double *dBlockBuf, double *dE, double *dC);
int hfg_form_density_dev(hfg_ctx *ctx, int64_t N, int64_t ncols, const double *dC, int64_t nocc, double *dP);
int hfg_gemm_dev(hfg_ctx *ctx, int transA, int transB, int64_t m, int64_t n, int64_t k, const double *dA,
                 int64_t lda, const double *dB, int64_t ldb, double *dC, int64_t ldc);

/* ---- SCF driver, the loop of src/diatomic/main.cpp:780-995: restricted closed shell (multiplicity 1) or
 * unrestricted (multiplicity = 2S+1 > 1: nela - nelb = multiplicity - 1, --M of main.cpp:100); a NEGATIVE
 * multiplicity selects the restricted open-shell run of `--restricted 1` (scf::ROHF_update, main.cpp:903) --------- */
/* out[0..7] = Etot, Ekin, Epot, Ecoul, Exx, Exc, Enucr, iterations(+0.5 if converged); out[8..11] =
 * seconds of the last iteration's J, K, XC and diagonalisation steps (the reference's Timer prints). */
int hfg_scf_diatomic(hfg_ctx *ctx, int Z1, int Z2, double Rbond, const int *lmmax, int nlm, int nelem, int nnodes,
                     int nquad, double Rmax, int igrid, double zexp, int lpad, const char *method, int ldft, int mdft,
                     int symmetry, int multiplicity, int maxit, double convthr, int verbose, double *out /* 12 */);

/* ---- complete runs: what `diatomic` (src/diatomic/main.cpp) and `atomic` (src/atomic/main.cpp) do between parsing their
 * command line and printing the energy table.  Field names are the reference's flag names (main.cpp:89-133 /
 * atomic/main.cpp:63-119); hfg_scf_options_default() fills in the reference's defaults.  Flags of features outside the
 * hot-path scope (external fields, finite nuclei, confinement, non-LIP primitive bases) are carried so
 * that the drivers can reject non-default values with the reference's wording instead of ignoring them. -------------- */
#define HFG_MAX_LMMAX 16
typedef struct hfg_scf_options {
  int program;               /* 0: diatomic, 1: atomic */
  int Z1, Z2;                /* diatomic: --Z1 --Z2; atomic: --Z in Z1 (Z2 unused) */
  double Rbond;              /* --Rbond in bohr (the drivers convert --angstrom input) */
  int nela, nelb, Q, M;      /* --nela --nelb --Q --M (scf::parse_nela_nelb); M = 0 means "not given": 1 */
  int lmmax[HFG_MAX_LMMAX];  /* diatomic: l_max for |m| = 0 .. nlm-1 (--lmax list, or --lmax with --mmax) */
  int nlm;
  int lmax, mmax;            /* atomic: --lmax --mmax */
  int lpad;                  /* --lpad = 10 */
  double Rmax;               /* --Rmax = 40 */
  int grid;                  /* --grid = 4 */
  double zexp;               /* --zexp = 1 (diatomic), 2 (atomic) */
  int nelem, nnodes, nquad;  /* --nelem, --nnodes = 15, --nquad = 0 (5 per primitive) */
  int maxit;                 /* --maxit = 50 */
  double convthr;            /* --convthr = 1e-7 */
  int diag;                  /* --diag = 1: S^-1/2 by diagonalisation, 0: Cholesky */
  char method[128];          /* --method = HF */
  int ldft, mdft;            /* --ldft --mdft = 0 (automatic) */
  double dftthr;             /* --dftthr = 1e-12 */
  int restricted;            /* --restricted = -1 (restricted iff nela == nelb) */
  int symmetry;              /* --symmetry = 1 */
  int primbas;               /* --primbas = 4 (LIP); anything else is rejected */
  double diiseps, diisthr;   /* --diiseps = 1e-2, --diisthr = 1e-3 */
  int diisorder;             /* --diisorder = 5 */
  int iguess;                /* --iguess = 2 in the reference (SAP, a data table that is out of scope): 0 core, 3 Thomas-Fermi */
  const double *x_pars;      /* --x_pars / --c_pars: external functional parameters (scf::parse_xc_params), or NULL */
  int n_x_pars;
  const double *c_pars;
  int n_c_pars;
  int maverage;              /* --maverage = false */
  double dampfock, dampthr;  /* atomic: --dampfock = 0.7, --dampthr = 0.1 (damping of the occupied-virtual Fock blocks,
                                atomic/main.cpp:917-936); 1.0 switches it off */
  char save[512];            /* --save = helfem.chk ("" to skip); HDF5, src/general/checkpoint.cpp */
  char load[512];            /* --load = "" */
  /* out of scope, must keep their defaults: */
  double Ez, Qzz, Bz;        /* external fields */
  int finitenuc;             /* finite nuclear model */
  int readocc;               /* --readocc = 0: forced occupations from occs (the drivers read occs.dat), enforced after the
                                guess and after the eigensolves of the iterations i < readocc; negative: always */
  const int *occs;           /* occ_rows x occ_cols, row-major: nalpha, nbeta, m [, parity +-1 | atomic --symmetry 2: l, m] */
  int occ_rows, occ_cols;
  double perturb;            /* random perturbation of the guess */
  int iconf;                 /* atomic: confinement potential */
  int zeroder;               /* atomic: zero derivative at Rmax */
  int verbose;               /* print the reference's per-iteration lines to stdout */
} hfg_scf_options;

typedef struct hfg_scf_result {
  double Etot, Ekin, Epot, Enucr, Ecoul, Exx, Exc;
  int iterations, converged;
  int nela, nelb;
  int64_t Nbf;
  double tJ, tK, tXC, tdiag; /* seconds of the last iteration's steps (the reference's Timer prints) */
} hfg_scf_result;

int hfg_scf_options_default(hfg_scf_options *opt, int program);
/* --readocc for the following hfg_scf_diatomic / hfg_scf_atomic calls of this thread (the short entry points have no
 * options structure); readocc = 0 switches it off */
int hfg_scf_set_occupations(int readocc, int nrows, int ncols, const int *rows);
/* the validation hfg_scf_run performs before it touches the device (no GPU needed): 0 = acceptable */
int hfg_scf_options_check(const hfg_scf_options *opt);
/* runs the calculation on ctx's device; E / C (alpha orbital energies Nbf, orbitals Nbf x Nbf) may be NULL */
int hfg_scf_run(hfg_ctx *ctx, const hfg_scf_options *opt, hfg_scf_result *res, double *E, double *C);
/* scf::parse_xc_params (src/general/scf_helpers.cpp): one number per line of a text file; n in: capacity, out: count */
int hfg_parse_xc_params(const char *path, double *pars, int *n);
/* element symbol or number -> nuclear charge (get_Z of src/general/elements.h as the drivers use it); < 0: unknown */
int hfg_get_Z(const char *symbol_or_number);

/* ---- checkpoint files in the reference's HDF5 layout (src/general/checkpoint.cpp: matrices as 2-D datasets with swapped
 * dimensions :117-144, integer vectors :220-257, scalars :627/:701, the basis as its constructor arguments :477-507,
 * :560-584).  libhdf5 is loaded at run time ($HELFEM_HDF5_LIB, libhdf5.so, ...); every call fails with an error text
 * when none can be loaded.  hfg_scf_run writes one when hfg_scf_options::save is set. -------------------------------- */
typedef struct hfg_chk hfg_chk;
int hfg_chk_available(void);                                    /* 1 when a libhdf5 could be loaded */
int hfg_chk_open(const char *path, int write, hfg_chk **chk);   /* write != 0 truncates / creates */
int hfg_chk_close(hfg_chk *chk);
int hfg_chk_exist(hfg_chk *chk, const char *name);              /* 1 / 0 */
int hfg_chk_write_mat(hfg_chk *chk, const char *name, const double *m, int64_t rows, int64_t cols); /* arma::mat / arma::vec */
int hfg_chk_write_ivec(hfg_chk *chk, const char *name, const int *v, int64_t n);                    /* arma::ivec */
int hfg_chk_write_double(hfg_chk *chk, const char *name, double v);
int hfg_chk_write_int(hfg_chk *chk, const char *name, int v);
int hfg_chk_write_basis(hfg_chk *chk, const hfg_basis *basis);  /* Checkpoint::write(basis) */
/* m == NULL queries the shape */
int hfg_chk_read_mat(hfg_chk *chk, const char *name, double *m, int64_t *rows, int64_t *cols);
int hfg_chk_read_ivec(hfg_chk *chk, const char *name, int *v, int64_t *n);
int hfg_chk_read_double(hfg_chk *chk, const char *name, double *v);
int hfg_chk_read_int(hfg_chk *chk, const char *name, int *v);
/* Checkpoint::read(diatomic::basis::TwoDBasis &): a basis object from the stored constructor arguments */
int hfg_chk_read_diatomic_basis(hfg_chk *chk, int lpad, hfg_basis **basis);

/* ---- measurement --------------------------------------------------------------------------- */
/* When enabled, every kernel family is bracketed by hipEvents on the context's stream; the
 * accumulated device time (ms) and launch count per family can be read back after a synchronize.
 * names: "coulomb", "xc", "exchange", "eig_reduce", "eig_tridiag", "eig_tridiag_solve",
 * "eig_backtransform", "gemm", "density", "scatter". */
int hfg_profile_enable(hfg_ctx *ctx, int on);
int hfg_profile_reset(hfg_ctx *ctx);
int hfg_profile_get(hfg_ctx *ctx, const char *name, double *ms, int64_t *launches);
/* the names seen since the last reset, separated by '\n' (besides the families above: one name per tile shape of the
 * persistent tridiagonalisation, "k_trdp<R, U>", and "k_trdp" for all its launches) */
int hfg_profile_names(hfg_ctx *ctx, char *buf, size_t cap);

/* Atomic SCF, restricted closed shell or unrestricted (driver loop of src/atomic/main.cpp:760-1005); out as for hfg_scf_diatomic */
int hfg_scf_atomic(hfg_ctx *ctx, int Z, int Q, int lmax, int mmax, int nelem, int nnodes, int nquad, double Rmax,
                   int igrid, double zexp, const char *method, int ldft, int mdft, int symmetry, int multiplicity,
                   int maverage /* --maverage: scf::fock_symmetry_average over m */, int maxit, double convthr,
                   int verbose, double *out /* 12 */);

/* Replays every launch of the named kernel of the last eigensolve back to back between two HIP events on the
 * context's stream (the roofline leg of bench.py).  Supported: "k_trdf" (the tridiagonalisation sweep of the default
 * path; "k_trdb_gemv" is accepted as an alias and replays the same launches -- with HELFEM_TRD=twokernel those are the
 * launches of k_trdb_gemv).  Any other name: status 1. */
int hfg_measure_kernel(hfg_ctx *ctx, const char *name, double *ms, int64_t *launches);

/* pinned host memory for arma-owned buffers ("Armadillo matrices pinned and mirrored to HBM") */
int hfg_pin(void *host_ptr, size_t bytes);
int hfg_unpin(void *host_ptr);

#ifdef __cplusplus
}
#endif
#endif /* HELFEM_GPU_H */
