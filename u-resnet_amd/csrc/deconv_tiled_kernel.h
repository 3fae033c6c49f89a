// LDS-tiled stride-2 "scatter-type" convolution for the full-resolution levels -- gfx950.
//
// Evaluates   out[2q + par][p] (+)= sum_{k : k == par (mod 2) per axis} X[q + (par - k)/2][c] * W[k][p][c]
// i.e. the transposed convolution of lib/uresnet.py:68-88 (k3 s2 SAME, W = [k..,Cout,Cin]) AND the data gradient
// of the stride-2 k3 convolutions of lib/resnet_module.py:43-51 (W = [k..,Cin,Cout]: same memory pattern
// [k][produced][contracted]); SURVEY.md Appendix B-2 (the two are the same operator).
//
// lane = one LOW-resolution voxel q; it owns the 2^d high-resolution outputs 2q + par (one accumulator set per
// parity class) and reads only its 2^d neighbours q + {0,-1}^d from a 2(+1)-plane LDS ring (halo of one on the
// low side).  Each of the 27 (9) filter taps feeds exactly one class, so the MFMA count equals the dense count
// (27*Cc*Cp/4 v_mfma_f32_4x4x1_16b per voxel), and weights stay in registers via the cbsz/abid A-broadcast
// exactly as in conv_tiled_kernel.h.  Every lane stores 2 x-adjacent output voxels per (pz,py): 64 contiguous
// bytes at Cp = 8, instead of the eight stride-2 scatter launches of the generic path.
//
// Round 4: those 64 bytes used to leave as four 16-byte stores with a 64-byte lane stride -- every store instruction touched
// all 64 half-lines of the wave's row segment with 16 bytes each, and the ablation (tools/td_ablate.sh) priced the stores at
// 0.17 of the layer's 0.43 ms (stores + loads alone: 0.32 ms for 906 MB = 2.8 TB/s).  Now the four lanes of a quad transpose
// their 4 x 4 sixteen-byte units in registers (two DPP butterfly stages, VALU work in the shadow of the MFMAs) and store
// instruction i of lane q carries unit q of lane i: 64 contiguous bytes per quad and instruction.  (A per-wave LDS exchange
// that makes whole 1 KB runs was measured first: 0.427 -> 0.375 ms, its ds_write / ds_read round trips cost the MFMA phase
// 0.10 of the 0.15 ms it saves.)
#pragma once
#include "conv_tiled_kernel.h"

// 2 x 2 block exchange between lanes l and l ^ 1 (CTRL 0xB1 = quad_perm [1,0,3,2]) or l ^ 2 (0x4E = [2,3,0,1]): lanes with the bit
// clear keep `lo` and receive the partner's `lo` into `hi`; lanes with the bit set keep `hi` and receive the partner's `hi` into `lo`
template <int CTRL>
__device__ __forceinline__ void td_quad_swap(f32x4& lo, f32x4& hi, bool bit) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float send = bit ? lo[j] : hi[j];
    const float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xf, 0xf, true));
    if (bit) lo[j] = r; else hi[j] = r;
  }
}
// r[i] of lane q (= lane & 3) becomes r[q] of lane i of the same quad
__device__ __forceinline__ void td_quad_transpose(f32x4 (&r)[4], int q) {
  td_quad_swap<0xB1>(r[0], r[1], q & 1);
  td_quad_swap<0xB1>(r[2], r[3], q & 1);
  td_quad_swap<0x4E>(r[0], r[2], q & 2);
  td_quad_swap<0x4E>(r[1], r[3], q & 2);
}

struct TDeconvArgs {
  const float* in;   // low-res [N][Z][Y][X][in_cs]
  const float* w;    // [taps][cp_w][ck_w]
  float* out;        // high-res [N][2Z][2Y][2X][out_cs]
  double* stats_partial;
  int N, Z, Y, X;    // LOW-res dims (2-D: Z = H, Y = 1, X = W)
  int in_cs, out_cs;
  int zseg, nzseg, nty, ntx;
  int accumulate;
  int cp_w, ck_w;    // stored weight dims
  // PW instantiations (data gradient of a unit's stride-2 resnet_conv1, lib/resnet_module.py:25-43): + pw_in[q] . pw_w^T into the
  // even-even-even output of low-res voxel q -- the data gradient of the unit's 1x1 stride-2 shortcut, which touches exactly
  // those voxels (as a separate pass it re-read and re-wrote them: pconv_dgrad, 0.37 ms at 192^3)
  const float* pw_in;   // [N][Z][Y][X][pw_in_cs], CK channels (the shortcut's dz)
  const float* pw_w;    // [produced][contracted], row stride pw_ws
  int pw_in_cs, pw_ws;
};

// CK = contracted channels, CP = produced channels; ACC: a.accumulate, compiled in (with a run-time flag the compiler joined the
// two paths in front of an s_waitcnt vmcnt(0) in the middle of every plane: all stores and the staged plane drained there)
template <int CK, int CP, int MODE, bool STATS, bool ACC, bool PW = false>
__global__ __launch_bounds__(256, (CP <= 8 && MODE == 3) ? 2 : 1) void tdeconv_kernel(TDeconvArgs a) {
  using TL = Tile<MODE>;
  constexpr int TX = TL::TX, TY = TL::TY, NTY = TL::NTY, NT = TL::NT;
  constexpr int HY = (NTY == 3) ? 1 : 0;                       // halo rows on the low side
  constexpr int PX = TX + 1, PY = TY + HY, PS = PX * PY;
  constexpr int NQ = CK / 4, CQ = CP / 4;
  constexpr int KTOT = NT * CK, R = (KTOT + 15) / 16;
  constexpr int NSTAGE = (NQ * PS + 255) / 256;
  constexpr int NCLS = (NTY == 3) ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) f32x4 dlds[];  // [3 ring slots][NQ][PS]

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

  // weights -> registers: lane l holds W[k = 16r + (l>>2) -> (tap, c)][p = 4cq + (l&3)]
  float wreg[CQ][R];
  {
    const int kl = lane >> 2, cl = lane & 3;
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int k = 16 * r + kl;
        int t = k / CK, c = k - t * CK;
        int p = 4 * cq + cl;
        float v = 0.f;
        if (k < KTOT && p < a.cp_w && c < a.ck_w) v = a.w[((size_t)t * a.cp_w + p) * a.ck_w + c];
        wreg[cq][r] = v;
      }
  }

  float wpw[PW ? CQ : 1];   // lane l holds Wsc[p = 4cq + (l&3)][c = l>>2]
  if constexpr (PW) {
    static_assert(CK == 16, "fused shortcut term: one 16-channel contraction block");
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq) {
      const int pch = 4 * cq + (lane & 3);
      wpw[cq] = pch < a.cp_w ? a.pw_w[(size_t)pch * a.pw_ws + (lane >> 2)] : 0.f;
    }
  }
  // staging through buffer loads (buffer_stage.h): tabulated byte offsets, out-of-range elements read as 0
  f32x4 stage[NSTAGE];
  unsigned soff[NSTAGE];
#pragma unroll
  for (int i = 0; i < NSTAGE; ++i) {
    const int idx = tid + i * 256;
    const int q = idx / PS, s = idx - q * PS;
    const int yy = s / PX, xx = s - yy * PX;
    const int py = y0 + yy - HY, px = x0 + xx - 1;
    const bool ok = idx < NQ * PS && py >= 0 && py < a.Y && px >= 0 && px < a.X;
    soff[i] = ok ? (unsigned)((py * a.X + px) * a.in_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
    asm volatile("" : "+v"(soff[i]));
  }
  const ptrdiff_t in_plane = (ptrdiff_t)a.Y * a.X * a.in_cs;
  const unsigned in_plane_bytes = (unsigned)in_plane * 4u;
  const float* in_img = a.in + (size_t)n * a.Z * in_plane;
  auto stage_load = [&](int zin) {
    const bool zok = zin >= 0 && zin < a.Z;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(in_img + (ptrdiff_t)zin * in_plane, zok ? in_plane_bytes : 0u);
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) stage[i] = ursn_buffer_load_f4(r, soff[i]);
  };
  auto stage_store = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
      int idx = tid + i * 256;
      if (idx < NQ * PS) dlds[(size_t)slot * NQ * PS + idx] = stage[i];
    }
  };

  f32x4 pv[PW ? NQ : 1];   // the lane's own voxel of the shortcut gradient, requested one plane ahead
  const ptrdiff_t pw_plane = PW ? (ptrdiff_t)a.Y * a.X * a.pw_in_cs : 0;
  const unsigned pw_off = (PW && vox_ok) ? (unsigned)((gy * a.X + gx) * a.pw_in_cs) * 4u : URSN_OOB_OFFSET;
  auto pw_load = [&](int zin) {
    if constexpr (PW) {
      const bool zok = zin >= 0 && zin < a.Z;
      const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(a.pw_in + ((ptrdiff_t)n * a.Z + (zok ? zin : 0)) * pw_plane, zok ? (unsigned)pw_plane * 4u : 0u);
#pragma unroll
      for (int q = 0; q < NQ; ++q) pv[q] = ursn_buffer_load_f4(r, pw_off + 16u * q);
    }
  };
  // BatchNorm moments around a wave-uniform pivot (wave_pivot.h): piv[] lives in SGPRs
  float s1[STATS ? CP : 1], s2[STATS ? CP : 1], piv[STATS ? CP : 1];
#pragma unroll
  for (int c = 0; c < (STATS ? CP : 1); ++c) s1[c] = s2[c] = piv[c] = 0.f;

  // prologue: planes z0-1, z0  (ring slot of plane z = (z + 3) % 3)
  stage_load(z0 - 1);
  stage_store((z0 + 2) % 3);
  stage_load(z0);
  stage_store(z0 % 3);
  stage_load(z0 + 1 < z1 ? z0 + 1 : -1);
  stage_store((z0 + 1) % 3);
  stage_load(z0 + 2 < z1 ? z0 + 2 : -1);
  pw_load(z0);
  __syncthreads();

  const int lane_slot = (ty + HY) * PX + tx + 1;   // the lane's own voxel inside a plane

  // line-contiguous stores: a lane owns 2 * CQ sixteen-byte units per output row class (2 x-adjacent voxels x CQ quads).  After
  // the quad transpose, register i of lane q holds unit q of lane i (CQ = 2: unit = (px, cq) = (q >> 1, q & 1) of one transpose;
  // CQ = 4: unit = cq of one transpose per px)
  static_assert(CQ == 2 || CQ == 4, "quad transpose of the stores: 8 or 16 produced channels");
  constexpr int NXP = CQ / 2;                       // transposes per row class
  const int qd = lane & 3;
  unsigned xoff[4];   // byte offsets inside the (pz, py) row class of a fine plane; out-of-range owners: the marker, dropped by the
                      // bounds check -- no lane-validity branch around the stores, so the compiler counts them (see the loop's end)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int sx = gx - qd + i;                     // the owner's voxel (quads never straddle a tile row)
    const int e = ((NTY == 3 ? 2 * gy * (2 * a.X) : 0) + 2 * sx + (CQ == 2 ? qd >> 1 : 0)) * a.out_cs + 4 * (CQ == 2 ? qd & 1 : qd);
    xoff[i] = (gy < a.Y && sx < a.X) ? (unsigned)e * 4u : URSN_OOB_OFFSET;
  }
  // bytes from the first row of a (pz, py) class to the end of its fine plane (the resource of a class); the lane's own voxel pair
  const unsigned out_rows_bytes = (unsigned)(((NTY == 3 ? 2 * a.Y : 1) * (2 * a.X)) * a.out_cs) * 4u;
  const unsigned own_off = vox_ok ? (unsigned)(((NTY == 3 ? 2 * gy * (2 * a.X) : 0) + 2 * gx) * a.out_cs) * 4u : URSN_OOB_OFFSET;
  for (int z = z0; z < z1; ++z) {
    f32x4 acc[NCLS][CQ];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
      for (int cq = 0; cq < CQ; ++cq) acc[c][cq] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4* p_cur = dlds + (size_t)(z % 3) * NQ * PS + lane_slot;
    const f32x4* p_prev = dlds + (size_t)((z + 2) % 3) * NQ * PS + lane_slot;
    // loop over the 2^d source offsets; every tap with that offset reuses the loaded voxel
    // the offsets with z-1 (second half) only feed the even output planes: the odd planes are complete, and stored, after the
    // first half -- their stores drain under the remaining third of the MFMAs
    auto run_offsets = [&](auto HALF) {
    static_for<NCLS / 2>([&](auto OFF) {
      constexpr int off = decltype(HALF)::value * (NCLS / 2) + decltype(OFF)::value;   // bit2: z-1, bit1: y-1, bit0: x-1 (3-D); 2-D: bit1: row-1
      constexpr int oz = (NTY == 3) ? (off >> 2) & 1 : (off >> 1) & 1;
      constexpr int oy = (NTY == 3) ? (off >> 1) & 1 : 0;
      constexpr int ox = off & 1;
      const f32x4* pl = oz ? p_prev : p_cur;
      static_for<NQ>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        f32x4 xv = pl[q * PS - oy * PX - ox];
        static_for<NT>([&](auto T) {
          constexpr int t = decltype(T)::value;
          constexpr int kz = t / (NTY * 3), ky = (t / 3) % NTY, kx = t % 3;   // 2-D: ky == 0 always, kz = row tap
          // per axis: k = 0 -> (par 0, source q); k = 1 -> (par 1, source q); k = 2 -> (par 0, source q-1)
          constexpr int toz = (kz == 2), tox = (kx == 2), toy = (NTY == 3) ? (ky == 2) : 0;
          if constexpr (toz == oz && toy == oy && tox == ox) {
            constexpr int pz = (kz == 1), px_ = (kx == 1), py_ = (NTY == 3) ? (ky == 1) : 0;
            constexpr int cls = (NTY == 3) ? (pz * 4 + py_ * 2 + px_) : (pz * 2 + px_);
            static_for<4>([&](auto J) {
              constexpr int j = decltype(J)::value;
              constexpr int k = t * CK + 4 * q + j;
              static_for<CQ>([&](auto C) {
                constexpr int cq = decltype(C)::value;
                acc[cls][cq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[cq][k / 16], xv[j], acc[cls][cq], 4, k % 16, 0);
              });
            });
          }
        });
      });
    });
    };
    run_offsets(std::integral_constant<int, 0>{});
    if constexpr (STATS) {
      if (z == z0 && a.stats_partial) {   // wave-uniform, once: the pivots = the first valid lane's all-odd outputs of this plane
        const int src = wave_first_valid(vox_ok);
        if (src >= 0) {
          const float* op = a.out + ((((size_t)n * (2 * a.Z) + 2 * z + 1) * ((NTY == 3) ? 2 * a.Y : 1) + 2 * gy + (NTY == 3 ? 1 : 0)) * (2 * a.X) + 2 * gx + 1) * a.out_cs;
#pragma unroll
          for (int cq = 0; cq < CQ; ++cq) {
            f32x4 v = acc[NCLS - 1][cq];
            if (ACC && vox_ok) v += *(const f32x4*)(op + 4 * cq);
#pragma unroll
            for (int j = 0; j < 4; ++j) piv[4 * cq + j] = wave_lane_value(v[j], src);
          }
        }
      }
    }
    auto store_rows = [&](auto RC) {
      constexpr int rc = decltype(RC)::value;
      constexpr int pz = (NTY == 3) ? (rc >> 1) & 1 : rc & 1;
      constexpr int py = (NTY == 3) ? rc & 1 : 0;
      float* rb = a.out + (((size_t)n * (2 * a.Z) + (2 * z + pz)) * ((NTY == 3) ? 2 * a.Y : 1) + py) * (size_t)(2 * a.X) * a.out_cs;
      const __amdgpu_buffer_rsrc_t rr = ursn_plane_rsrc(rb, out_rows_bytes);
      if constexpr (STATS) {
        // moments on the owner lanes, before the transpose (an accumulated output: the old values in owner layout)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int cq = 0; cq < CQ; ++cq) {
            f32x4& v = acc[2 * rc + px][cq];
            if constexpr (ACC) v += ursn_buffer_load_f4(rr, own_off + (unsigned)(px * a.out_cs + 4 * cq) * 4u);
            if (vox_ok) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float d = v[j] - piv[4 * cq + j];
                s1[4 * cq + j] += d;
                s2[4 * cq + j] = __builtin_fmaf(d, d, s2[4 * cq + j]);
              }
            }
          }
      }
#pragma unroll
      for (int h = 0; h < NXP; ++h) {               // CQ = 4: h = px
        const unsigned hoff = (CQ == 4) ? (unsigned)(h * a.out_cs) * 4u : 0u;
        f32x4 old[4];
        if (!STATS && ACC) {
#pragma unroll
          for (int i = 0; i < 4; ++i) old[i] = ursn_buffer_load_f4(rr, hoff + xoff[i]);
        }
        f32x4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (CQ == 2) ? acc[2 * rc + (i >> 1)][i & 1] : acc[2 * rc + h][i];
        td_quad_transpose(r, qd);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!STATS && ACC) r[i] += old[i];
          ursn_buffer_store_f4(rr, hoff + xoff[i], r[i]);
        }
      }
    };
    static_for<NCLS / 4>([&](auto I) { store_rows(std::integral_constant<int, NCLS / 4 + decltype(I)::value>{}); });
    if constexpr (PW) {
      static_for<CK>([&](auto K) {
        constexpr int k = decltype(K)::value;
        static_for<CQ>([&](auto C) {
          constexpr int cq = decltype(C)::value;
          acc[0][cq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wpw[cq], pv[k / 4][k % 4], acc[0][cq], 4, k, 0);
        });
      });
    }
    run_offsets(std::integral_constant<int, 1>{});
    static_for<NCLS / 4>([&](auto I) { store_rows(I); });
    __syncthreads();
    // a plane ENDS by putting the plane requested one plane ago into the slot of the plane it has just finished with (z - 1) and
    // requesting the next one: with the staging here the loop header sees the same pending accesses from the prologue and from the
    // back edge (the staged loads, youngest), and the compiler waits for them with vmcnt(this plane's stores) instead of
    // vmcnt(0) -- staged at the top of the body the two paths disagree and every plane drained its 16 stores (round 3; with one
    // wave per SIMD that was 0.1 of the pass's 0.43 ms)
    stage_store((z + 2) % 3);
    stage_load(z + 3 < z1 ? z + 3 : -1);
    pw_load(z + 1 < z1 ? z + 1 : -1);
  }

  if constexpr (STATS) if (a.stats_partial) {
    __shared__ double red[4][2 * CP];
    const float nw = wave_sum(vox_ok ? (float)(NCLS * (z1 - z0)) : 0.f);   // produced voxels this wave summed
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float u = wave_sum(s1[c]), v = wave_sum(s2[c]);
      if (lane == 0) wave_unpivot(u, v, nw, piv[c], red[tid >> 6][c], red[tid >> 6][CP + c]);
    }
    __syncthreads();
    if (tid < 2 * CP) a.stats_partial[(size_t)blockIdx.x * 2 * CP + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

struct TDPlan {
  int mode, ck, cp;
  int Z, Y, X, zseg, nzseg, nty, ntx;
  size_t lds;
  int grid;
};

template <int CK, int CP, int MODE, bool STATS, bool ACC, bool PW = false>
static int launch_td(const TDPlan& p, const TDeconvArgs& a, hipStream_t s) {
  auto kern = tdeconv_kernel<CK, CP, MODE, STATS, ACC, PW>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

#define URSN_TD(ck_, cp_)                                                                        \
  if (p.ck == ck_ && p.cp == cp_) {                                                              \
    ursn_note_kernel("tdeconv<" #ck_ "," #cp_ ">");                                              \
    if (a.pw_in) {                                                                               \
      ursn_note_kernel("tdeconv<" #ck_ "," #cp_ ">+pw");                                         \
      return a.accumulate ? launch_td<ck_, cp_, MODE, false, true, true>(p, a, s) : launch_td<ck_, cp_, MODE, false, false, true>(p, a, s); \
    }                                                                                            \
    if (a.accumulate) return a.stats_partial ? launch_td<ck_, cp_, MODE, true, true>(p, a, s) : launch_td<ck_, cp_, MODE, false, true>(p, a, s); \
    return a.stats_partial ? launch_td<ck_, cp_, MODE, true, false>(p, a, s) : launch_td<ck_, cp_, MODE, false, false>(p, a, s); \
  }

int tdeconv_dispatch_3d(const TDPlan& p, const TDeconvArgs& a, hipStream_t s);
int tdeconv_dispatch_2d(const TDPlan& p, const TDeconvArgs& a, hipStream_t s);
