// Cross-lane sums of doubles on gfx950 through DPP (data-parallel primitives: the operand of a VALU move comes from
// another lane of the same row of 16, or lane 15 / 31 of the previous rows) instead of __shfl_*, which hipcc compiles to
// ds_bpermute_b32: one trip through the LDS crossbar (> 100 clocks) per step and per 32-bit half, in a dependent chain of
// six steps for a wave sum.  A DPP step is two v_mov_dpp and one v_add_f64.  Fixed summation order, so results are
// reproducible (but not the order of the __shfl_down tree).
#pragma once
#include <hip/hip_runtime.h>

namespace hfg {

// dpp_ctrl encodings (GFX9): quad_perm 0x00-0xFF, row_shr:n 0x110+n, row_ror:n 0x120+n, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast:15 0x142, row_bcast:31 0x143
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // old = 0: lanes of rows masked out, and lanes whose source lane does not exist (bound_ctrl), receive 0
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}

/// sum over the 64 lanes of the wave, returned in EVERY lane (read from lane 63: bitwise the same everywhere).
/// All 64 lanes must be active.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0x111>(v);        // row_shr:1   lane i: x[i] + x[i-1]
  v += dpp_f64<0x112>(v);        // row_shr:2   ... x[i-3..i]
  v += dpp_f64<0x114>(v);        // row_shr:4
  v += dpp_f64<0x118>(v);        // row_shr:8   lane 15 of each row: the row's sum
  v += dpp_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3: lanes 31 and 63 hold two rows
  v += dpp_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

/// dpp_f64 with a fill value: lanes of masked rows and lanes without a source lane receive `fill`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64_fill(double x, double fill) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

/// product over the 64 lanes, in every lane
__device__ __forceinline__ double wave_prod(double v) {
  v *= dpp_f64_fill<0x111>(v, 1.0);
  v *= dpp_f64_fill<0x112>(v, 1.0);
  v *= dpp_f64_fill<0x114>(v, 1.0);
  v *= dpp_f64_fill<0x118>(v, 1.0);
  v *= dpp_f64_fill<0x142, 0xa>(v, 1.0);
  v *= dpp_f64_fill<0x143, 0xc>(v, 1.0);
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

/// maximum over the 64 lanes, in every lane
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64_fill<0x111>(v, v));
  v = fmax(v, dpp_f64_fill<0x112>(v, v));
  v = fmax(v, dpp_f64_fill<0x114>(v, v));
  v = fmax(v, dpp_f64_fill<0x118>(v, v));
  v = fmax(v, dpp_f64_fill<0x142, 0xa>(v, v));
  v = fmax(v, dpp_f64_fill<0x143, 0xc>(v, v));
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

/// sum over each aligned group of 16 lanes (one DPP row), returned in every lane of the group (a butterfly of
/// commutative additions: bitwise the same in all 16 lanes)
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0xb1>(v);   // quad_perm [1,0,3,2]   lane ^ 1
  v += dpp_f64<0x4e>(v);   // quad_perm [2,3,0,1]   lane ^ 2
  v += dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the 8 (all four lanes of a quad hold the same value)
  v += dpp_f64<0x140>(v);  // row_mirror: the other half of the row
  return v;
}

/// sum over each aligned group of 8 lanes, in every lane of the group
__device__ __forceinline__ double oct_sum(double v) {
  v += dpp_f64<0xb1>(v);
  v += dpp_f64<0x4e>(v);
  v += dpp_f64<0x141>(v);
  return v;
}
/// sum over each aligned group of 4 lanes, in every lane of the group
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_f64<0xb1>(v);
  v += dpp_f64<0x4e>(v);
  return v;
}

/// One halving step of a transposing butterfly across the two halves of the wave: lanes 0-31 return x[l] + x[l + 32],
/// lanes 32-63 return y[l - 32] + y[l]  (v_permlane32_swap_b32, gfx950; lane maps printed by
/// tests/gpu_probe/permlane_probe.hip: swap(X, Y)[0] = {X.lo, Y.lo}, [1] = {X.hi, Y.hi})
__device__ __forceinline__ double swap32_sum(double x, double y) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
/// the same across the two rows of every 32 lanes: lanes with bit 4 clear return x[l] + x[l + 16], the others
/// y[l - 16] + y[l]  (v_permlane16_swap_b32)
__device__ __forceinline__ double swap16_sum(double x, double y) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

}  // namespace hfg
