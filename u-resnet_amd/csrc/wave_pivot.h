// Wave-pivoted BatchNorm moments for the lane-per-voxel fp32 kernels (tconv, tdeconv, pconv).
//
// Those kernels sum z and z^2 per lane in fp32 registers.  Plain sums lose the variance once |mean| >> std (the one-pass
// S2/N - mean^2 cancels), which the finalise kernel used to repair by walking the tensor again with one block per channel.
// Here every lane of a wave subtracts the SAME pivot K[c] -- the first value of the channel that the wave produces, held in
// SGPRs, so it costs no vector register -- and sums d = z - K, d^2.  A wave's voxels are neighbours of one tile, so K sits
// within a few std of the wave's mean and the sums stay small; lane 0 folds them back to plain moments in fp64,
//     S1 = U + n K,   S2 = W + 2 K U + n K^2          (U = sum d, W = sum d^2, n = voxels of the wave)
// and the block / grid reduction stays the fp64 one of bn_stats_final_kernel.  No second pass over z is ever needed.
#pragma once
#include <hip/hip_runtime.h>

// value of lane `src` (wave-uniform index) broadcast as a scalar
__device__ __forceinline__ float wave_lane_value(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// first lane with `valid` set, -1 if none (wave-uniform)
__device__ __forceinline__ int wave_first_valid(bool valid) {
  const unsigned long long m = __ballot(valid);
  return m ? __ffsll((long long)m) - 1 : -1;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// fold a wave's pivoted sums back to plain moments (fp64)
__device__ __forceinline__ void wave_unpivot(float U, float W, float n, float K, double& S1, double& S2) {
  const double k = (double)K, u = (double)U, nn = (double)n;
  S1 = u + nn * k;
  S2 = (double)W + 2.0 * k * u + nn * k * k;
}
