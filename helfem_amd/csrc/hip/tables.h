// Flattened, HBM-resident form of a diatomic or atomic TwoDBasis.
//
// The atomic basis (B_n(r)/r Y_lm, src/atomic/TwoDBasis.cpp) is stored in the same layout: its radial functions
// are numbered n = 1..E(p-1)-1 with the dropped first primitive as n = 0 of every shell (shell_skip = 1
// everywhere), its Coulomb kernel r_<^L / r_>^{L+1} fills the "0" slots of the tables below (P0 := int r^L,
// Q0 := int r^{-L-1}, tei00 := in-element integral, c0 := Gaunt coefficient) and the "2" slots are absent
// (ntt = 1, ndt = 2); the primitive tables depend on L only (lm_tab maps (L,|M|) -> L).
//
// Layouts (all doubles, "fastest index last" written as C arrays):
//   shells keep the reference order (basis.cpp:287-302); radial function n of shell a has dummy
//   index a*R+n and boundary-cleaned ("pure") index shell_off[a]+n, absent when shell_skip[a] && n==0.
//   Element e holds primitives i=0..p-1 <-> radial functions e*(p-1)+i; the last element has p-1
//   functions, its missing primitive p-1 is zero padding everywhere.
//
//   disj[t][tab][e][j][i]        t: 0=P0 1=P2 2=Q0 3=Q2   (disjoint_*  basis.cpp:1178-1187)
//   tei[t][tab][e][c][r]         t: 0=00 1=02 2=20 3=22   (prim_tei**, p^2 x p^2, col-major, r=j*p+i)
//   pair tables: for the ordered shell pair (x,y): M=m_x-m_y, L=Lmin..Lmax,
//        c0 = mod_coeff(l_x,m_x,L,M,l_y,m_y), c2 = coeff(...)      (basis.cpp:1395-1396, 1516-1521)
//   "compact" matrices X_c[x][y][e][j][i]: the element-diagonal p x p blocks of shell block (x,y);
//        these are the only non-zero blocks of J and of the XC matrix, and the only blocks of P they read.
#pragma once
#include "common.h"

struct hfg_dev_tables {
  int A = 0, R = 0, E = 0, p = 0, nq = 0, N = 0, Nd = 0, NLM = 0, Nlm = 0;
  int T = 0;  // number of (pair,L) coupling entries
  int geom = 0;        // 0 prolate spheroidal (diatomic), 1 spherical (atomic)
  int Ntab = 0;        // primitive-table slots (diatomic: Nlm, atomic: number of L)
  int ntt = 4;         // tei types present (diatomic 00,02,20,22; atomic 00)
  int ndt = 4;         // disjoint types present (diatomic P0,P2,Q0,Q2; atomic P0,Q0)
  int dQ0 = 2;         // index of the Q0 type inside disj (P0 = 0, P2 = 1 when present, Q2 = 3)
  int Lp1 = 0;         // size of the L axis of c0tab/c2tab
  double Rhalf = 0.0;
  bool have_tei = false, have_xc = false;
  // Range-separated exchange of the atomic program (TwoDBasis::rs_exchange, src/atomic/TwoDBasis.cpp:1142): a second
  // table set (hfg_basis::dev_rs) with the same couplings whose radial slots hold the screened kernel.  Yukawa
  // (rs_kind 1): P0 := int i_L(lambda r), Q0 := int k_L(lambda r), tei00 := in-element integral, LMfac = 4 pi lambda.
  // erfc (rs_kind 2): the kernel does not factorise over elements -- pair_tei = 1, no disjoint tables, and
  // tei[tab][e][f][c][r] holds one p^2 x p^2 block per ELEMENT PAIR (rows: primitives of e, columns: primitives of f),
  // LMfac = 4 pi mu/(2L+1).
  int rs_kind = 0;
  int pair_tei = 0;

  hfg::DevBuf<int> shell_l, shell_m, shell_off, shell_skip;
  // couplings by pair (x*A+y): entries [pair_off, pair_off+1)
  hfg::DevBuf<int> pair_off, ent_iLM;
  hfg::DevBuf<double> ent_c0, ent_c2;
  // couplings by iLM: entries [lm_off, lm_off+1) -> (x,y,c0,c2)
  hfg::DevBuf<int> lm_off, lm_x, lm_y;
  hfg::DevBuf<double> lm_c0, lm_c2;
  hfg::DevBuf<int> LM_ilm, LM_partner;  // ilm of each (L,M); index of (L,-M) (or -1)
  hfg::DevBuf<int> lm_tab;              // [Nlm] primitive-table slot of (L,|M|)
  hfg::DevBuf<int> LM_tab;              // [NLM] = lm_tab[LM_ilm[.]]
  hfg::DevBuf<double> LM_fac;
  hfg::DevBuf<double> disj, tei;

  // ---- XC grid ----
  int ntheta = 0, nphi = 0, G = 0;  // G = number of distinct m values
  hfg::DevBuf<double> rad_B, rad_dB;       // [E][nq][p]
  hfg::DevBuf<double> rad_w, rad_sh;       // [E][nq]  radial weight (wq*len/2), sinh(mu)
  hfg::DevBuf<double> th_c, th_s, th_w;    // [ntheta] cos, sin, Chebyshev weight
  hfg::DevBuf<double> Th, dTh;             // [A][ntheta]
  hfg::DevBuf<int> grp_m, grp_off, grp_shell;  // m of group g; shells of group g = grp_shell[grp_off[g]..]
  hfg::DevBuf<int> shell_grp;
  hfg::DevBuf<double> cosd, sind;  // [2*Dmax+1][nphi]  cos(D phi_j), sin(D phi_j), D=-Dmax..Dmax
  int Dmax = 0;

  // host copies needed by launch logic
  std::vector<int> h_LM_L, h_LM_M, h_LM_ilm, h_grp_off, h_shell_l, h_shell_m, h_shell_skip, h_lm_tab;
  std::vector<double> h_LM_fac;
  std::vector<double> h_c0tab, h_c2tab;  // [A][A][Lp1] coupling of the ordered shell pair with L
};

namespace hfg {
/// build (or rebuild) the device tables of basis on the context's device
void upload_tables(hfg_ctx *ctx, hfg_basis *basis, int ldft, int mdft);
/// build (or rebuild) basis->dev_rs from the host tables of compute_yukawa / compute_erfc
void upload_rs_tables(hfg_ctx *ctx, hfg_basis *basis);
}  // namespace hfg
