// mvo_wave.h -- wavefront-level (64 lanes) reduction helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

// ---------------------------------------------------------------------------------------
// 8 values x 64 lanes -> 8 totals with 7 + 3 exchange steps instead of 8 x 6: every halving step
// trades half of the values with the partner lane (xor 32, 16 as v_permlane32/16_swap; xor 8 and
// the in-group steps through DPP).  On return lane L holds the total of value
// ((L>>5)&1)*4 + ((L>>4)&1)*2 + ((L>>3)&1), identical in the 8 lanes of its group.
// Fixed pairing order => bitwise reproducible.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ---- 64-lane reductions without LDS (gfx950: v_permlane32_swap / v_permlane16_swap) ------------
// swap32(a, b): a keeps its lanes 0..31 and receives b's lanes 0..31 in its lanes 32..63; b receives
// a's lanes 32..63 in its lanes 0..31 and keeps its lanes 32..63.  Hence (a + b) after the swap holds,
// in lanes 0..31, a(l) + a(l + 32) and, in lanes 32..63, b(l - 32) + b(l): the halving step of a
// butterfly that trades half of the VALUES for half of the LANES.  swap16 does the same with the odd
// and even 16-lane rows.
__device__ __forceinline__ double swap_add32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

__device__ __forceinline__ double wave_reduce8(const double (&v)[8], int lane) {
  const bool b3 = lane & 8;
  double w4[4], w2[2], w1;
  // the two halving steps across rows as register swaps (the same operands in the same order as the
  // shuffle-and-select form they replace -- bit-identical totals --, a third of its instructions and no
  // trip through the LDS crossbar)
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = swap_add32(v[j], v[j + 4]);
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = swap_add16(w4[j], w4[j + 2]);
  {
    const double keep = b3 ? w2[1] : w2[0], send = b3 ? w2[0] : w2[1];
    w1 = keep + dpp_mov_f64<0x128>(send);  // row_ror:8 == lane xor 8
  }
  w1 += dpp_mov_f64<0xB1>(w1);   // quad_perm [1,0,3,2]
  w1 += dpp_mov_f64<0x4E>(w1);   // quad_perm [2,3,0,1]
  w1 += dpp_mov_f64<0x141>(w1);  // row_half_mirror: joins the two quads of the 8-lane group
  return w1;
}

// lane L (L % 8 == 0) owns value index (L >> 3) in the bit order above
__device__ __forceinline__ int reduce8_value_of_lane(int lane) {
  return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
}

// all 8 totals of wave_reduce8 made wave-uniform (v_readlane: they land in SGPRs)
__device__ __forceinline__ void wave_reduce8_all(const double (&v)[8], int lane, double (&out)[8]) {
  const double w = wave_reduce8(v, lane);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int src = ((k >> 2) & 1) * 32 + ((k >> 1) & 1) * 16 + (k & 1) * 8;
    const int lo = __builtin_amdgcn_readlane(__double2loint(w), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(w), src);
    out[k] = __hiloint2double(hi, lo);
  }
}
