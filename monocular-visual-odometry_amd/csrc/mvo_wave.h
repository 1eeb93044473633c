// mvo_wave.h -- wavefront-level (64 lanes) reduction helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

// ---------------------------------------------------------------------------------------
// 8 values x 64 lanes -> 8 totals with 7 + 3 exchange steps instead of 8 x 6: every halving step
// trades half of the values with the partner lane (xor 32, 16 through the LDS crossbar; xor 8 and
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

__device__ __forceinline__ double wave_reduce8(const double (&v)[8], int lane) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
  double w4[4], w2[2], w1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double keep = b5 ? v[j + 4] : v[j], send = b5 ? v[j] : v[j + 4];
    w4[j] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const double keep = b4 ? w4[j + 2] : w4[j], send = b4 ? w4[j] : w4[j + 2];
    w2[j] = keep + __shfl_xor(send, 16, 64);
  }
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
