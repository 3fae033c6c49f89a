// LDS-tiled direct convolution for the full-resolution, small-channel layers (k3, stride 1) -- gfx950.
//
// These layers carry ~70 % of the network's FLOPs with Cin/Cout in {8,16,32}: too narrow for the 16- or
// 32-wide MFMA shapes (half of every tile would be padding).  v_mfma_f32_4x4x1_16b_f32 has the right
// granularity: 16 independent 4x4 outer products per instruction, at the full fp32 rate (64 FLOP/clk/SIMD).
//
//   lane  = one output voxel (B operand: x[voxel + tap][ci], one float per lane)
//   A     = 4 output channels of one (tap, ci) weight row, BROADCAST from one 4-lane block to all 16 blocks
//           (cbsz = 4, abid = block): one VGPR holds 16 (tap,ci) rows x 4 cout, so ALL weights of a layer
//           (27*Cin*Cout/64 VGPRs) stay in registers for the whole kernel -- no weight traffic in the loop
//   D     = 4 cout of the lane's voxel per accumulator; Cout/4 accumulators per lane
//
// A workgroup (4 waves) owns a TYxTX tile of the two fastest axes and marches along the slowest axis,
// keeping a ring of 4 input planes (with halo) in LDS as [cin/4][y][x] float4 so every ds_read_b128 of a
// wave is 1 KiB contiguous (bank-conflict free).  Plane z+2 is fetched into registers before the
// MFMA block of plane z and written to LDS after it (one barrier per plane).  BatchNorm statistics
// (sum, sum of squares per channel) are accumulated per lane in the epilogue and reduced once per
// workgroup, which removes the separate statistics pass over the conv output.
//
// The same kernel evaluates the stride-1 data gradient (taps flipped, weight matrix transposed at
// register-load time).
#pragma once
#include <utility>

#include "ursn_common.h"
#include "wave_pivot.h"
#include "buffer_stage.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: the MFMA broadcast selector (abid) must be an integer constant expression
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#ifndef URSN_TCONV_NT_STORE
#define URSN_TCONV_NT_STORE 0   // measured: non-temporal output stores cost 3-5 % here
#endif
struct TConvArgs {
  const float* in;
  const float* w;
  float* out;
  double* stats_partial;  // [grid][2][cout] or null
  int N, Z, Y, X;         // 2-D problems: Z = H, Y = 1, X = W
  int in_cs, out_cs;
  int zseg;               // planes per workgroup
  int nzseg, nty, ntx;
  int accumulate;
  int cin_w, cout_w;      // dims of the weight tensor as stored: [t][cin_w][cout_w]
  // data gradient only: fused pointwise term of the parallel 1x1 shortcut (lib/resnet_module.py:25-33),
  //   out[v][ci] += sum_co pw_in[v][co] * pw_w[ci][co]     (pw_in = shortcut dz, CIN channels; null = none)
  const float* pw_in;
  const float* pw_w;      // rows = produced channels of this launch, row stride pw_ws
  int pw_in_cs, pw_ws;
  // forward only: normalise-on-load, in = raw z of the preceding conv: staged value = (z - mean) * rstd + beta
  const float* aff_mean;
  const float* aff_rstd;
  const float* aff_beta;
  // data gradient only (BS instantiations): BatchNorm-backward reductions of the layer(s) whose input gradient this launch
  // FINISHES, fused into the epilogue (the separate bn_bwd_reduce pass read this tensor and z again):
  //   g = out * mask;  partial[block][0][c] = sum g, [1][c] = sum g * xhat(z), [2][c] = sum g * xhat2(z2)
  // mask: none (bs_relu 0) | bn(z) > 0 recomputed with bs_beta (1) | the join's bit mask as written by bn_act (2)
  const float* bs_z;  const float* bs_mean;  const float* bs_rstd;  const float* bs_beta;
  const float* bs_z2; const float* bs_mean2; const float* bs_rstd2;
  const unsigned long long* bs_mask;
  double* bs_partial;      // [grid][3][COUT]
  int bs_z_cs, bs_z2_cs, bs_relu;
  // data gradient only (DZL instantiations): `in` is NOT dz but g, the gradient at the output of this layer's BatchNorm
  // (slim.batch_norm backward, lib/resnet_module.py:49 / lib/uresnet.py:109): the kernel forms
  //     dz = A g' + B (z - mu) + C,   g' = g [* (fma(z, S, T) > 0)]        per channel, while it stages the operand,
  // and WRITES the interior of every staged plane to dz_out (the weight gradient reads it there) -- the separate
  // bn_bwd_apply pass (read g, z; write dz) disappears.  dz_coef: [6][8] floats A, B, C, mu, S, T (bn_bwd_final_kernel).
  const float* dz_z;
  const float* dz_coef;
  float* dz_out;
  int dz_relu;
};

template <int MODE> struct Tile;
template <> struct Tile<3> { static constexpr int TX = 32, TY = 8, NTY = 3, NT = 27; };
template <> struct Tile<2> { static constexpr int TX = 256, TY = 1, NTY = 1, NT = 9; };

// partial accumulators per produced channel quad, used round-robin along the contraction.  Two for a single quad (the 3|4-
// channel logits layer would otherwise be ONE chain of dependent MFMAs); one otherwise: 2 or 4 per quad remove the
// s_nop between dependent MFMAs (148 per plane at 8 -> 8) but measure the same at 8 -> 8 and 15 % slower at 16 -> 16
// (registers, occupancy) -- the nops sit in the shadow of the previous MFMA
template <int CQ, int BS> struct TConvAcc { static constexpr int N = CQ == 1 ? 2 : 1; };

// CIN = contraction channels, COUT = produced channels (already swapped for the data gradient).
template <int CIN, int COUT, int MODE, bool FLIP, bool AFF = false, int BS = 0, bool DZL = false>
__global__ __launch_bounds__(256, (BS == 2 && CIN == 8 && !DZL) ? 3 : 2) void tconv_kernel(TConvArgs a) {
  static_assert(BS == 0 || (FLIP && COUT == 8 && MODE == 3), "fused BatchNorm-backward reductions: 3-D data gradients producing 8 channels");
  static_assert(!DZL || (FLIP && CIN == 8 && MODE == 3 && !AFF), "dz-on-load: 3-D data gradients contracting 8 channels");
  constexpr bool STATS = !FLIP;  // the data gradient never feeds a BatchNorm
  using TL = Tile<MODE>;
  constexpr int TX = TL::TX, TY = TL::TY, NTY = TL::NTY, NT = TL::NT;
  constexpr int PX = TX + 2, PY = TY + (NTY == 3 ? 2 : 0), PS = PX * PY;
  constexpr int NQ = CIN / 4, CQ = COUT / 4;
  constexpr int KTOT = NT * CIN, R = (KTOT + 15) / 16;
  // staging slots: slot i covers LDS elements tid + 256 i of the [quad][y][x] plane image; with DZL a slot stays inside ONE
  // channel quad (NS2 slots per quad), so that its BatchNorm coefficients are compile-time selected scalars
  constexpr int NS2 = (PS + 255) / 256;
  constexpr int NSTAGE = DZL ? NQ * NS2 : (NQ * PS + 255) / 256;
  constexpr int NACC = TConvAcc<CQ, BS>::N;
  auto acc_sum = [](const f32x4 (&p)[NACC]) {
    f32x4 v = p[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) v += p[i];
    return v;
  };
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];  // [4 ring slots][NQ][PS]

  const int tid = threadIdx.x, lane = tid & 63;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int xt = bid % a.ntx; bid /= a.ntx;
  const int yt = bid % a.nty; bid /= a.nty;
  const int zs = bid % a.nzseg;
  const int n = bid / a.nzseg;
  const int x0 = xt * TX, y0 = yt * TY;
  const int z0 = zs * a.zseg;
  const int z1 = (z0 + a.zseg < a.Z) ? z0 + a.zseg : a.Z;
  const int ty = tid / TX, tx = tid % TX;
  const int gy = y0 + ty, gx = x0 + tx;
  const bool vox_ok = gy < a.Y && gx < a.X;

  // ---- weights -> registers: wreg[cq][r], lane l holds W'[k = 16r + (l>>2)][co = 4cq + (l&3)] -------------
  float wreg[CQ][R];
  {
    const int kl = lane >> 2, cl = lane & 3;
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int k = 16 * r + kl;
        int t = k / CIN, kk = k - t * CIN;
        int co = 4 * cq + cl;
        float v = 0.f;
        if (k < KTOT) {  // channel counts padded to 4 in registers: rows/cols beyond the stored tensor are 0
          if (!FLIP) { if (kk < a.cin_w && co < a.cout_w) v = a.w[((size_t)t * a.cin_w + kk) * a.cout_w + co]; }
          else if (co < a.cin_w && kk < a.cout_w) v = a.w[((size_t)(NT - 1 - t) * a.cin_w + co) * a.cout_w + kk];
        }
        wreg[cq][r] = v;
      }
  }

  float wpw[CQ];  // fused pointwise term (FLIP): lane l holds Ws[ci = 4cq + (l&3)][co = l>>2]
#pragma unroll
  for (int cq = 0; cq < CQ; ++cq) wpw[cq] = 0.f;
  if constexpr (FLIP && CIN <= 16) {
    if (a.pw_in) {
      const int kl = lane >> 2, cl = lane & 3;
#pragma unroll
      for (int cq = 0; cq < CQ; ++cq) wpw[cq] = (kl < CIN) ? a.pw_w[(size_t)(4 * cq + cl) * a.pw_ws + kl] : 0.f;
    }
  }

  __shared__ f32x4 aff_s[AFF ? NQ : 1], aff_t[AFF ? NQ : 1];   // per-channel scale / shift of the normalise-on-load
  constexpr bool aff = AFF && !FLIP;   // separate instantiation: the plain kernels keep their register allocation
  if constexpr (aff) {
    if (tid < CIN) {
      const float r = a.aff_rstd[tid];
      ((float*)aff_s)[tid] = r;
      ((float*)aff_t)[tid] = a.aff_beta[tid] - a.aff_mean[tid] * r;
    }
    __syncthreads();
  }

  // scale / shift of the elements this thread stages (their channel quad is fixed per staging slot)
  f32x4 sreg[aff ? NSTAGE : 1], treg[aff ? NSTAGE : 1];
  if constexpr (aff) {
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
      int q = (tid + i * 256) / PS;
      if (q >= NQ) q = NQ - 1;
      sreg[i] = aff_s[q];
      treg[i] = aff_t[q];
    }
  }

  // ---- plane staging --------------------------------------------------------------------------------------
  // Loads go through buffer instructions (buffer_stage.h): byte offsets inside a z plane are tabulated once (pinned in
  // registers -- as plain arithmetic on tid the compiler re-derives them every plane), elements outside the image carry
  // URSN_OOB_OFFSET and a plane outside the volume gets num_records = 0: both read as 0 without a branch or a zero fill.
  f32x4 stage[NSTAGE];
  f32x4 stage2[DZL ? NSTAGE : 1];   // DZL: z beside g
  unsigned stage_inb = 0;   // which staged elements are real voxels (the affine must not touch the zero padding)
  unsigned soff[NSTAGE], sin_mask = 0;
  unsigned woff[DZL ? NSTAGE : 1];  // DZL: store offset of the formed dz element: interior voxels of the tile only
  int sidx[DZL ? NSTAGE : 1];       // DZL: LDS element index of the slot (-1: none)
#pragma unroll
  for (int i = 0; i < NSTAGE; ++i) {
    int idx = tid + i * 256, q = idx / PS, s = idx - q * PS;
    bool have = idx < NQ * PS;
    if constexpr (DZL) { q = i / NS2; s = tid + (i % NS2) * 256; have = s < PS; idx = q * PS + s; sidx[i] = have ? idx : -1; }
    const int yy = s / PX, xx = s - yy * PX;
    const int py = y0 + yy - (NTY == 3 ? 1 : 0), px = x0 + xx - 1;
    const bool ok = have && py >= 0 && py < a.Y && px >= 0 && px < a.X;
    soff[i] = ok ? (unsigned)((py * a.X + px) * a.in_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
    if (ok) sin_mask |= 1u << i;
    asm volatile("" : "+v"(soff[i]));
    if constexpr (DZL) {
      const bool interior = ok && yy >= (NTY == 3 ? 1 : 0) && yy < (NTY == 3 ? 1 : 0) + TY && xx >= 1 && xx <= TX;
      woff[i] = interior ? soff[i] : URSN_OOB_OFFSET;
      asm volatile("" : "+v"(woff[i]));
    }
  }
  asm volatile("" : "+v"(sin_mask));
  if constexpr (aff) {   // an element outside the image arrives as 0 and must stay 0: its shift is zeroed once, the store has no per-element test
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i)
      if (!((sin_mask >> i) & 1u)) treg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const ptrdiff_t in_plane = (ptrdiff_t)a.Y * a.X * a.in_cs;
  const unsigned in_plane_bytes = (unsigned)in_plane * 4u;
  const float* in_img = a.in + (size_t)n * a.Z * in_plane;
  const float* dzz_img = DZL ? a.dz_z + (size_t)n * a.Z * in_plane : nullptr;   // z and dz_out share g's layout (host-checked)
  float* dzo_img = DZL ? a.dz_out + (size_t)n * a.Z * in_plane : nullptr;
  const bool scalar_in = a.cin_w == 1 && !FLIP;   // single-channel input (conv0): scalar fetch, lanes 1..3 stay 0
  int stage_z = 0;   // DZL: the plane the staging registers hold
  float cf[DZL ? 6 : 1][8];   // DZL: A, B, C, mu, S, T per channel -- uniform, read once: scalar registers
  if constexpr (DZL) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 8; ++c) cf[k][c] = ((const __attribute__((address_space(4))) float*)a.dz_coef)[k * 8 + c];   // constant address space: s_load
  }
  auto stage_load = [&](int zin) {
    const bool zok = zin >= 0 && zin < a.Z;
    if constexpr (aff || DZL) stage_inb = zok ? 1u : 0u;   // wave-uniform: a plane outside the volume skips the affine
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(in_img + (ptrdiff_t)zin * in_plane, zok ? in_plane_bytes : 0u);
    if (scalar_in) {
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) stage[i] = (f32x4){ursn_buffer_load_f1(r, soff[i]), 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) stage[i] = ursn_buffer_load_f4(r, soff[i]);
    }
    if constexpr (DZL) {
      stage_z = zin;
      const __amdgpu_buffer_rsrc_t rz = ursn_plane_rsrc(dzz_img + (ptrdiff_t)zin * in_plane, zok ? in_plane_bytes : 0u);
#pragma unroll
      for (int i = 0; i < NSTAGE; ++i) stage2[i] = ursn_buffer_load_f4(rz, soff[i]);
    }
  };
  auto stage_store = [&](int slot) {
    if constexpr (DZL) {
      // planes z0 .. z1 - 1 are this workgroup's own: their interior goes to dz_out (the halo planes belong to the neighbours)
      const bool own = stage_z >= z0 && stage_z < z1;
      const __amdgpu_buffer_rsrc_t ro = ursn_plane_rsrc(dzo_img + (ptrdiff_t)stage_z * in_plane, own ? in_plane_bytes : 0u);
      static_for<NSTAGE>([&](auto I) {
        constexpr int i = decltype(I)::value, q = i / NS2;
        // elements outside the image in y / x keep the zeros written once before the march (below); a plane outside the
        // volume is all zeros
        if ((sin_mask >> i) & 1u) {
          const f32x4 g = stage[i], zv = stage2[i];
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (stage_inb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = 4 * q + j;
              float gj = g[j];
              if (a.dz_relu) { if (!(__builtin_fmaf(zv[j], cf[4][c], cf[5][c]) > 0.f)) gj = 0.f; }   // bn_act's expression
              v[j] = __builtin_fmaf(cf[0][c], gj, __builtin_fmaf(cf[1][c], zv[j] - cf[3][c], cf[2][c]));
            }
          }
          lds[(size_t)slot * NQ * PS + sidx[i]] = v;
          ursn_buffer_store_f4(ro, woff[i], v);   // out of range (halo or foreign plane): dropped
        }
      });
      return;
    }
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
      int idx = tid + i * 256;
      if (idx < NQ * PS) {
        f32x4 v = stage[i];
        // normalise-on-load is applied here, not at the load: the loads stay in flight during the MFMA block
        if constexpr (aff) {
          if (stage_inb) v = v * sreg[i] + treg[i];
        }
        lds[(size_t)slot * NQ * PS + idx] = v;
      }
    }
  };

  // BatchNorm moments around a wave-uniform pivot (wave_pivot.h): piv[] lives in SGPRs
  float s1[STATS ? COUT : 1], s2[STATS ? COUT : 1], piv[STATS ? COUT : 1];
#pragma unroll
  for (int c = 0; c < (STATS ? COUT : 1); ++c) s1[c] = s2[c] = piv[c] = 0.f;
  // fused BatchNorm-backward reductions.  Per lane (its <= zseg voxels of one column) fp32 in packed-math form -- the
  // MFMA-bound loop has few VALU slots to spare (fp64 per-lane sums: +0.9 ms/step); fp64 from the cross-lane step on, where
  // the cancellation of sum g (~1e-3 of sum |g|) happens
  f32x4 bg[BS ? CQ : 1], bgx[BS ? CQ : 1], bgx2[BS == 2 ? CQ : 1];
#pragma unroll
  for (int c = 0; c < (BS ? CQ : 1); ++c) bg[c] = bgx[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < (BS == 2 ? CQ : 1); ++c) bgx2[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (DZL) {   // staged elements outside the image (fixed per thread): zero in all four ring slots, never written again
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i)
      if (sidx[i] >= 0 && !((sin_mask >> i) & 1u)) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) lds[(size_t)sl * NQ * PS + sidx[i]] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  }
  // prologue: planes z0-1, z0, z0+1
  for (int p = -1; p <= 1; ++p) {
    stage_load(z0 + p);
    stage_store((z0 + p) & 3);
  }
  __syncthreads();

  const int lane_slot = ty * PX + tx;  // top-left of the lane's 3x3 window in a plane
  // the lane's voxel index in plane z, carried through the loop (rebuilt from (n, z, y, x) it costs a chain of 64-bit
  // multiplies per plane); lanes outside the image never dereference it
  const size_t plane_vox = (size_t)a.Y * a.X;
  size_t zvox = (((size_t)n * a.Z + z0) * a.Y + gy) * a.X + gx;
  for (int z = z0; z < z1; ++z, zvox += plane_vox) {
    stage_load(z + 2);
    f32x4 acc[CQ][NACC];   // NACC partial accumulators per quad (TConvAcc)
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
      for (int p = 0; p < NACC; ++p) acc[cq][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 pv[(FLIP && CIN <= 16) ? NQ : 1];
    if constexpr (FLIP && CIN <= 16) {
      if (a.pw_in) {  // this lane's voxel of the shortcut gradient (clamped for out-of-range lanes: MFMA needs all lanes)
        const int cy = gy < a.Y ? gy : a.Y - 1, cx = gx < a.X ? gx : a.X - 1;
        const float* pp = a.pw_in + (zvox - (size_t)(gy - cy) * a.X - (size_t)(gx - cx)) * a.pw_in_cs;
#pragma unroll
        for (int q = 0; q < NQ; ++q) pv[q] = *(const f32x4*)(pp + 4 * q);
      }
    }
    // fused BatchNorm-backward reductions: everything the epilogue reads is requested here, a whole MFMA block ahead of
    // its use (loaded in the epilogue itself the latency sat between the MFMAs and the plane barrier: +40 % kernel time)
    f32x4 bz[BS ? CQ : 1], bz2[BS == 2 ? CQ : 1], bold[BS ? CQ : 1];
    unsigned bmw[BS ? 4 : 1], bvlo = 0;
    if constexpr (BS != 0) {
      if (vox_ok) {
        const size_t vox = zvox;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
          bz[cq] = *(const f32x4*)(a.bs_z + vox * a.bs_z_cs + 4 * cq);
          if constexpr (BS == 2) bz2[cq] = *(const f32x4*)(a.bs_z2 + vox * a.bs_z2_cs + 4 * cq);
          if (a.accumulate) bold[cq] = *(const f32x4*)(a.out + vox * a.out_cs + 4 * cq);
        }
        if (a.bs_relu == 2) {   // the 32-bit half of each mask word that holds this voxel's two bits (see below)
          const unsigned* mw = (const unsigned*)(a.bs_mask + (vox >> 5) * 4) + (((unsigned)vox & 31u) >> 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) bmw[j] = mw[2 * j];
          bvlo = (unsigned)vox & 15u;
        }
      }
    }
    static_for<3>([&](auto TZ) {
      constexpr int tz = decltype(TZ)::value;
      const f32x4* plane = lds + (size_t)((z - 1 + tz) & 3) * NQ * PS + lane_slot;
      static_for<NTY * 3>([&](auto TYX) {
        constexpr int tyy = decltype(TYX)::value / 3, txx = decltype(TYX)::value % 3;
        constexpr int t = (tz * NTY + tyy) * 3 + txx;
        static_for<NQ>([&](auto Q) {
          constexpr int q = decltype(Q)::value;
          f32x4 xv = plane[q * PS + tyy * PX + txx];
          static_for<4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int k = t * CIN + 4 * q + j;
            if constexpr (CIN == 4 && !FLIP && j > 0) {
              if (a.cin_w == 1) return;   // conv0: channels 1..3 of the padded input quad are zeros (uniform branch)
            }
            if constexpr (CIN == 4 && FLIP && j > 0) {
              if (4 * q + j >= a.cout_w) return;   // logits layer dgrad: padded class channels carry zero weights
            }
            static_for<CQ>([&](auto C) {
              constexpr int cq = decltype(C)::value;
              acc[cq][k % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[cq][k / 16], xv[j], acc[cq][k % NACC], 4, k % 16, 0);
            });
          });
        });
      });
    });
    if constexpr (FLIP && CIN <= 16) {
      if (a.pw_in) {
        static_for<CIN>([&](auto K) {
          constexpr int k = decltype(K)::value;
          static_for<CQ>([&](auto C) {
            constexpr int cq = decltype(C)::value;
            acc[cq][k % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wpw[cq], pv[k / 4][k % 4], acc[cq][k % NACC], 4, k, 0);
          });
        });
      }
    }
    if constexpr (STATS) {
      if (z == z0 && a.stats_partial) {   // wave-uniform, once: the pivots = the first valid lane's values of this plane
        const int src = wave_first_valid(vox_ok);
        if (src >= 0) {
          const float* op = a.out + zvox * a.out_cs;
#pragma unroll
          for (int cq = 0; cq < CQ; ++cq) {
            f32x4 v = acc_sum(acc[cq]);
            if (a.accumulate && vox_ok) v += *(const f32x4*)(op + 4 * cq);
#pragma unroll
            for (int j = 0; j < 4; ++j) piv[4 * cq + j] = wave_lane_value(v[j], src);
          }
        }
      }
    }
    if (vox_ok) {
      float* op = a.out + zvox * a.out_cs;
#pragma unroll
      for (int cq = 0; cq < CQ; ++cq) {
        f32x4 v = acc_sum(acc[cq]);
        if constexpr (BS != 0) {
          if (a.accumulate) v += bold[cq];
        } else {
          if (a.accumulate) v += *(f32x4*)(op + 4 * cq);
        }
#if URSN_TCONV_NT_STORE
        __builtin_nontemporal_store(v, (f32x4*)(op + 4 * cq));
#else
        *(f32x4*)(op + 4 * cq) = v;
#endif
        if constexpr (STATS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = v[j] - piv[4 * cq + j];
            s1[4 * cq + j] += d;
            s2[4 * cq + j] = __builtin_fmaf(d, d, s2[4 * cq + j]);
          }
        }
        if constexpr (BS != 0) {
          const f32x4 zv = bz[cq];
          const f32x4 mu = *(const f32x4*)(a.bs_mean + 4 * cq), rs = *(const f32x4*)(a.bs_rstd + 4 * cq);
          f32x4 g = v;
          if (a.bs_relu == 1) {        // single conv-BN-ReLU layer: the mask is bn(z) > 0, same expression as bn_act
            const f32x4 be = *(const f32x4*)(a.bs_beta + 4 * cq);
            const f32x4 t = __builtin_elementwise_fma(zv, rs, be - mu * rs);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!(t[j] > 0.f)) g[j] = 0.f;
          } else if (a.bs_relu == 2) { // residual join: bit (v & 31) * 2 + cq of word j of the voxel's 256-element group
            const int bit = (int)(bvlo * 2u) + cq;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!((bmw[j] >> bit) & 1u)) g[j] = 0.f;
          }
          bg[cq] += g;
          bgx[cq] = __builtin_elementwise_fma(g, (zv - mu) * rs, bgx[cq]);
          if constexpr (BS == 2) {
            const f32x4 mu2 = *(const f32x4*)(a.bs_mean2 + 4 * cq), rs2 = *(const f32x4*)(a.bs_rstd2 + 4 * cq);
            bgx2[cq] = __builtin_elementwise_fma(g, (bz2[cq] - mu2) * rs2, bgx2[cq]);
          }
        }
      }
    }
    stage_store((z + 2) & 3);
    __syncthreads();
  }

  if constexpr (STATS) if (a.stats_partial) {  // workgroup partial sums (double) -> finalised by bn_stats_final_kernel
    __shared__ double red[4][2 * COUT];
    const float nw = wave_sum(vox_ok ? (float)(z1 - z0) : 0.f);   // voxels this wave summed
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      const float u = wave_sum(s1[c]), v = wave_sum(s2[c]);
      if (lane == 0) wave_unpivot(u, v, nw, piv[c], red[tid >> 6][c], red[tid >> 6][COUT + c]);
    }
    __syncthreads();
    if (tid < 2 * COUT) a.stats_partial[(size_t)blockIdx.x * 2 * COUT + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
  if constexpr (BS != 0) {   // [block][3][COUT] doubles, the layout bn_bwd_final_kernel reads
    __shared__ double bred[4][3 * COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      double u = (double)bg[c / 4][c % 4], w = (double)bgx[c / 4][c % 4], x2 = BS == 2 ? (double)bgx2[c / 4][c % 4] : 0.0;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        u += __shfl_xor(u, o);
        w += __shfl_xor(w, o);
        if constexpr (BS == 2) x2 += __shfl_xor(x2, o);
      }
      if (lane == 0) { bred[tid >> 6][c] = u; bred[tid >> 6][COUT + c] = w; bred[tid >> 6][2 * COUT + c] = x2; }
    }
    __syncthreads();
    if (tid < 3 * COUT)
      a.bs_partial[(size_t)blockIdx.x * 3 * COUT + tid] = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
  }
}


struct TPlan {
  int mode, cin, cout;  // kernel-view channels (swapped for the data gradient)
  bool flip;
  int Z, Y, X, zseg, nzseg, nty, ntx;
  size_t lds;
  int grid;
};

template <int CIN, int COUT, int MODE, bool FLIP, bool AFF = false, int BS = 0, bool DZL = false>
static int launch_t(const TPlan& p, const TConvArgs& a, hipStream_t s) {
  auto kern = tconv_kernel<CIN, COUT, MODE, FLIP, AFF, BS, DZL>;
  static size_t attr_lds = 48 * 1024;  // dynamic LDS above the default limit must be opted into per kernel
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

#define URSN_TC(ci, co)                                  \
  if (p.cin == ci && p.cout == co) {                     \
    ursn_note_kernel(flip ? "tconv_dgrad<" #ci "," #co ">" : "tconv<" #ci "," #co ">"); \
    if (!flip && a.aff_mean) {                           \
      if constexpr (ci == co && (ci == 8 || ci == 16)) return launch_t<ci, co, MODE, false, true>(p, a, s); \
      ursn_set_error("tiled conv: no normalise-on-load instantiation for %d->%d", ci, co); \
      return 3;                                          \
    }                                                    \
    if (flip && a.dz_z) {                                \
      if constexpr (MODE == 3 && co == 8 && ci == 8) {   \
        if (!a.bs_partial) return launch_t<ci, co, MODE, true, false, 0, true>(p, a, s);                \
        return a.bs_z2 ? launch_t<ci, co, MODE, true, false, 2, true>(p, a, s) : launch_t<ci, co, MODE, true, false, 1, true>(p, a, s); \
      }                                                  \
      ursn_set_error("tiled conv: no dz-on-load instantiation for %d->%d", ci, co);                     \
      return 3;                                          \
    }                                                    \
    if (flip && a.bs_partial) {                          \
      if constexpr (MODE == 3 && co == 8 && (ci == 8 || ci == 4))                                       \
        return a.bs_z2 ? launch_t<ci, co, MODE, true, false, 2>(p, a, s) : launch_t<ci, co, MODE, true, false, 1>(p, a, s); \
      ursn_set_error("tiled conv: no fused BatchNorm-backward instantiation for %d->%d", ci, co);      \
      return 3;                                          \
    }                                                    \
    return flip ? launch_t<ci, co, MODE, true>(p, a, s) : launch_t<ci, co, MODE, false>(p, a, s); \
  }

int tconv_dispatch_3d(const TPlan& p, const TConvArgs& a, hipStream_t s);
int tconv_dispatch_2d(const TPlan& p, const TConvArgs& a, hipStream_t s);
