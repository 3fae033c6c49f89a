// LDS-staged implicit-GEMM convolution (k3, stride 1) for the mid/deep levels: Cin, Cout multiples of 16,
// too many channels for the register-resident weights of conv_tiled_kernel.h -- gfx950.
//
// Workgroup = 4 waves, output box of 256 voxels (3-D: 4z x 4y x 16x, one z-slab per wave; 2-D: 16y x 16x,
// four rows per wave) x BM output channels.  K loop: Cin in chunks of 16; per chunk the input box + halo is
// staged ONCE into LDS as [ci/4][slot] float4 and reused by all 27 (9) taps (the gather kernel re-fetches it per
// tap through L1); per tap the 16 x BM weight slab goes through a double-buffered LDS tile shared by the 4 waves.
// v_mfma_f32_16x16x4_f32: A = W[ci][co] (row stride BM+16 floats -> conflict-free ds_read_b32), B = x[ci][voxel]
// (16 x-adjacent voxels -> 64 consecutive floats), D[co][voxel]: each lane ends with 4 consecutive channels of
// one voxel (16-byte stores).  Per tap and wave: 8 LDS operand reads feed 16*BM/64 MFMAs.
// FLIP evaluates the data gradient (taps mirrored, weight matrix transposed while staging).
#pragma once
#include <utility>

#include "ursn_common.h"
#include "buffer_stage.h"

typedef float ig_f32x4 __attribute__((ext_vector_type(4)));

template <int... Is, class F>
__device__ __forceinline__ void ig_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ig_static_for(F&& f) {
  ig_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct IGemmArgs {
  const float* in;
  const float* w;
  float* out;
  double* stats_partial;  // [grid.y][grid.x][2][BM], or null
  int N, Z, Y, X;         // 2-D: Z = 1
  int cin, cout;          // kernel view (contraction / produced)
  int in_cs, out_cs;
  int cin_w, cout_w;      // stored weight dims [t][cin_w][cout_w]
  int nbz, nby, nbx;
  int accumulate;
  // data gradient (igemm_at_kernel, FLIP) only: fused 1x1 shortcut term  out[v][ci] += sum_co pw_in[v][co] * pw_w[ci][co]
  const float* pw_in;   // shortcut dz, same channel count as the contraction; null = none
  const float* pw_w;    // [produced channels][contraction channels], row stride pw_ws
  int pw_cs, pw_ws;
};

template <int MODE> struct IBox;
template <> struct IBox<3> { static constexpr int BZ = 4, BY = 4, BX = 16, NT = 27, KZ = 3; };
template <> struct IBox<2> { static constexpr int BZ = 1, BY = 16, BX = 16, NT = 9, KZ = 1; };
// igemm_at_kernel only: boxes for the small deep levels.  The 16 lanes of an MFMA column tile are 16 consecutive voxels
// of the flattened box (per-lane slot table), so a 12-wide or 6-wide row does not leave lanes idle.
template <int MODE, int VAR> struct ABox : IBox<MODE> {};
template <> struct ABox<3, 1> { static constexpr int BZ = 4, BY = 4, BX = 12, NT = 27, KZ = 3; };   // 192 voxels = 12 tiles
template <> struct ABox<3, 2> { static constexpr int BZ = 3, BY = 6, BX = 6, NT = 27, KZ = 3; };    // 108 voxels -> 7 of 8 tiles

template <int MODE, int BM, bool FLIP, bool STATS>
__global__ __launch_bounds__(256, 2) void igemm_conv_kernel(IGemmArgs a) {
  using B = IBox<MODE>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ;
  constexpr int HZ = BZ + (KZ - 1), HY = BY + 2, HX = BX + 2, PS = HZ * HY * HX;   // halo box slots
  constexpr int KC = 16, WS = BM + 16;                                              // ci chunk, weight row stride
  constexpr int MT = BM / 16;
  constexpr int NH = (4 * PS + 255) / 256;                                          // halo float4 per thread
  extern __shared__ __attribute__((aligned(16))) float ilds[];  // [4][PS][4] halo, then [2][KC][WS] weights
  float* hal = ilds;
  float* wl = ilds + 4 * PS * 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int bx = bid % a.nbx; bid /= a.nbx;
  const int by = bid % a.nby; bid /= a.nby;
  const int bz = bid % a.nbz;
  const int n = bid / a.nbz;
  const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
  const int co0 = blockIdx.y * BM;

  // this wave's 4 voxel tiles (16 x each): 3-D: z = wave, y = 0..3; 2-D: y = 4*wave + 0..3
  ig_f32x4 acc[4][MT];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[v][m] = (ig_f32x4){0.f, 0.f, 0.f, 0.f};

  // weight staging: one float4 of the [KC][BM] slab per thread (threads beyond KC*BM/4 idle)
  auto load_w = [&](int t, int ci0) -> ig_f32x4 {
    ig_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tid < KC * BM / 4) {
      if (!FLIP) {
        int k = tid / (BM / 4), c4 = (tid % (BM / 4)) * 4;
        if (co0 + c4 < a.cout) v = *(const ig_f32x4*)(a.w + ((size_t)t * a.cin_w + ci0 + k) * a.cout_w + co0 + c4);
      } else {
        int nn = tid / (KC / 4), k4 = (tid % (KC / 4)) * 4;   // produced channel nn, contraction k4..k4+3
        if (co0 + nn < a.cout) v = *(const ig_f32x4*)(a.w + ((size_t)(NT - 1 - t) * a.cin_w + co0 + nn) * a.cout_w + ci0 + k4);
      }
    }
    return v;
  };
  auto store_w = [&](int buf, ig_f32x4 v) {
    if (tid < KC * BM / 4) {
      float* dst = wl + (size_t)buf * KC * WS;
      if (!FLIP) {
        int k = tid / (BM / 4), c4 = (tid % (BM / 4)) * 4;
        *(ig_f32x4*)(dst + k * WS + c4) = v;
      } else {
        int nn = tid / (KC / 4), k4 = (tid % (KC / 4)) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(k4 + j) * WS + nn] = v[j];
      }
    }
  };

  const int nchunks = a.cin / KC;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int ci0 = ch * KC;
    __syncthreads();  // previous chunk's MFMAs are done with the halo / weight tiles
    // ---- input box + halo for 16 channels -> LDS [q][slot] float4 ----
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      int idx = tid + i * 256;
      if (idx < 4 * PS) {
        int s = idx >> 2, q = idx & 3;   // consecutive threads: the 4 quads of one voxel (64 contiguous bytes)
        int hx = s % HX, r = s / HX;
        int hy = r % HY, hz = r / HY;
        int pz = z0 + hz - (KZ == 3 ? 1 : 0), py = y0 + hy - 1, px = x0 + hx - 1;
        ig_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pz >= 0 && pz < a.Z && py >= 0 && py < a.Y && px >= 0 && px < a.X)
          v = *(const ig_f32x4*)(a.in + ((((size_t)n * a.Z + pz) * a.Y + py) * a.X + px) * a.in_cs + ci0 + 4 * q);
        *(ig_f32x4*)(hal + ((size_t)q * PS + s) * 4) = v;
      }
    }
    ig_f32x4 wv = load_w(0, ci0);
    store_w(0, wv);
    __syncthreads();

    for (int t = 0; t < NT; ++t) {
      const int buf = t & 1;
      if (t + 1 < NT) wv = load_w(t + 1, ci0);
      const int tz = (KZ == 3) ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
      const float* wb = wl + (size_t)buf * KC * WS;
      // halo slot of this lane's voxel for tile v: 3-D (z = wave, y = v), 2-D (y = 4*wave + v)
      const int zz = (MODE == 3) ? wave + tz : 0;
      const int yb = (MODE == 3) ? ty : 4 * wave + ty;
      const float* hb = hal + ((size_t)(zz * HY + yb) * HX + il + tx) * 4 + kl;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float bv[4], av[MT];
#pragma unroll
        for (int v = 0; v < 4; ++v) bv[v] = hb[((size_t)s * PS + v * HX) * 4];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wb[(4 * s + kl) * WS + m * 16 + il];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[v][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[v], acc[v][m], 0, 0, 0);
      }
      if (t + 1 < NT) store_w(buf ^ 1, wv);
      __syncthreads();
    }
  }

  // ---- epilogue: lane (il = x, kl) holds channels co0 + 16m + 4kl + r of voxel (z, y, x0 + il) ----
  // BatchNorm moments around a per-lane pivot (ursn_common.h: shifted one-pass moments); re-centred in fp64 below
  float s1[MT][4], s2[MT][4], piv[MT][4], nacc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    nacc[m] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = piv[m][r] = 0.f;
  }
  const int gx = x0 + il;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int gz = (MODE == 3) ? z0 + wave : 0;
    const int gy = (MODE == 3) ? y0 + v : y0 + 4 * wave + v;
    const bool ok = gz < a.Z && gy < a.Y && gx < a.X;
    if (!ok) continue;
    float* op = a.out + ((((size_t)n * a.Z + gz) * a.Y + gy) * a.X + gx) * a.out_cs + co0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      int c = 16 * m + 4 * kl;
      if (co0 + c >= a.cout) continue;
      ig_f32x4 val = acc[v][m];
      if (a.accumulate) val += *(ig_f32x4*)(op + c);
      *(ig_f32x4*)(op + c) = val;
      if constexpr (STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nacc[m] == 0.f) piv[m][r] = val[r];
          ursn_sacc(piv[m][r], s1[m][r], s2[m][r], val[r]);
        }
        nacc[m] += 1.f;
      }
    }
  }
  if constexpr (STATS) if (a.stats_partial) {
    __shared__ double red[4][2 * BM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double u, w2;
        ursn_sacc_final(piv[m][r], s1[m][r], s2[m][r], nacc[m], u, w2);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
        if (il == 0) {
          red[wave][16 * m + 4 * kl + r] = u;
          red[wave][BM + 16 * m + 4 * kl + r] = w2;
        }
      }
    __syncthreads();
    if (tid < 2 * BM)
      a.stats_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * BM + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// All-taps variant (BM = 16 with 16-channel chunks, BM = 32 with 8-channel chunks): the per-tap barrier + dependent
// weight-tile load of igemm_conv_kernel is latency the MFMAs cannot hide when there are few boxes or narrow co tiles.
// Here ALL taps of the KC x BM weight slab of a chunk are staged at once (27.6 KB, one 16-column plane per co tile ->
// conflict-free operand reads), the next chunk's halo and weights are prefetched into registers during the MFMAs, and
// the inner loop runs 27 x 16 MFMAs per wave between barriers.
template <int MODE, int BM, int KC, bool FLIP, bool STATS, int VAR = 0>
__global__ __launch_bounds__(256, 2) void igemm_at_kernel(IGemmArgs a) {
  using B = ABox<MODE, VAR>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ;
  constexpr int NV = BZ * BY * BX, TPW = (NV + 63) / 64;     // voxels of the box, 16-voxel tiles per wave
  constexpr int HZ = BZ + (KZ - 1), HY = BY + 2, HX = BX + 2, PS = HZ * HY * HX;
  constexpr int MT = BM / 16, NQ = KC / 4;
  constexpr int NH = (NQ * PS + 255) / 256;
  constexpr int NWF = NT * KC * BM / 4;          // float4 of the weight slab
  constexpr int NW = (NWF + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float ilds[];  // [NQ][PS][4] halo, then [MT][NT][KC][16] weights
  float* hal = ilds;
  float* wl = ilds + NQ * PS * 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int bx = bid % a.nbx; bid /= a.nbx;
  const int by = bid % a.nby; bid /= a.nby;
  const int bz = bid % a.nbz;
  const int n = bid / a.nbz;
  const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
  const int co0 = blockIdx.y * BM;

  ig_f32x4 acc[TPW][MT];
#pragma unroll
  for (int v = 0; v < TPW; ++v)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[v][m] = (ig_f32x4){0.f, 0.f, 0.f, 0.f};

  // Halo and weight-slab loads go through buffer instructions with byte offsets tabulated ONCE (buffer_stage.h): rebuilt per
  // chunk from idx they were ~40 instructions per element (divisions by HX / HY, five bounds tests, a 64-bit address), and every
  // issued instruction costs this SIMD's matrix pipe its slot.  The chunk's first channel moves the resource base.
  unsigned hoff[NH], woff[NW];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int idx = tid + i * 256;
    const int s = idx / NQ, q = idx % NQ;
    const int hx = s % HX, r = s / HX;
    const int hy = r % HY, hz = r / HY;
    const int pz = z0 + hz - (KZ == 3 ? 1 : 0), py = y0 + hy - 1, px = x0 + hx - 1;
    const bool ok = idx < NQ * PS && pz >= 0 && pz < a.Z && py >= 0 && py < a.Y && px >= 0 && px < a.X;
    hoff[i] = ok ? (unsigned)(((pz * a.Y + py) * a.X + px) * a.in_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
    asm volatile("" : "+v"(hoff[i]));
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int idx = tid + i * 256;
    const int t = idx / (KC * BM / 4), rem = idx % (KC * BM / 4);
    bool ok = idx < NWF;
    unsigned off;
    if (!FLIP) {
      const int k = rem / (BM / 4), c4 = (rem % (BM / 4)) * 4;
      ok = ok && co0 + c4 < a.cout;
      off = (unsigned)((t * a.cin_w + k) * a.cout_w + co0 + c4) * 4u;
    } else {
      const int nn = rem / (KC / 4), k4 = (rem % (KC / 4)) * 4;
      ok = ok && co0 + nn < a.cout;
      off = (unsigned)(((NT - 1 - t) * a.cin_w + co0 + nn) * a.cout_w + k4) * 4u;
    }
    woff[i] = ok ? off : URSN_OOB_OFFSET;
    asm volatile("" : "+v"(woff[i]));
  }
  const float* in_img = a.in + (size_t)n * a.Z * a.Y * a.X * a.in_cs;
  const unsigned in_img_bytes = (unsigned)a.Z * a.Y * a.X * a.in_cs * 4u, w_bytes = (unsigned)NT * a.cin_w * a.cout_w * 4u;
  auto load_h = [&](int ci0, ig_f32x4 (&hv)[NH]) {
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(in_img + ci0, in_img_bytes);
#pragma unroll
    for (int i = 0; i < NH; ++i) hv[i] = ursn_buffer_load_f4(r, hoff[i]);
  };
  auto store_h = [&](const ig_f32x4 (&hv)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = tid + i * 256;
      if (idx < NQ * PS) *(ig_f32x4*)(hal + ((size_t)(idx % NQ) * PS + idx / NQ) * 4) = hv[i];
    }
  };
  // weights: float4 index of the [NT][KC][BM] slab; FLIP loads along the contraction (contiguous in memory)
  auto load_w = [&](int ci0, ig_f32x4 (&wv)[NW]) {
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(a.w + (FLIP ? (size_t)ci0 : (size_t)ci0 * a.cout_w), w_bytes);
#pragma unroll
    for (int i = 0; i < NW; ++i) wv[i] = ursn_buffer_load_f4(r, woff[i]);
  };
  auto store_w = [&](const ig_f32x4 (&wv)[NW]) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * 256;
      if (idx < NWF) {
        const int t = idx / (KC * BM / 4), rem = idx % (KC * BM / 4);
        if (!FLIP) {
          const int k = rem / (BM / 4), c4 = (rem % (BM / 4)) * 4;
          *(ig_f32x4*)(wl + ((((size_t)(c4 >> 4) * NT + t) * KC + k) * 16 + (c4 & 15))) = wv[i];
        } else {
          const int nn = rem / (KC / 4), k4 = (rem % (KC / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) wl[(((size_t)(nn >> 4) * NT + t) * KC + k4 + j) * 16 + (nn & 15)] = wv[i][j];
        }
      }
    }
  };

  // fused pointwise term (FLIP): the box's own voxels of the shortcut gradient [NQ][NV][4] and the KC x BM slab of Ws^T
  constexpr int NPX = (NQ * NV + 255) / 256, NPW = (BM * KC / 4 + 255) / 256;
  float* pwx = wl + (size_t)MT * NT * KC * 16;
  float* pww = pwx + (size_t)NQ * NV * 4;
  const bool pw = FLIP && a.pw_in != nullptr;
  auto load_pw = [&](int ci0, ig_f32x4 (&px)[NPX], ig_f32x4 (&pq)[NPW]) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int idx = tid + i * 256;
      const int vox = idx / NQ, q = idx % NQ;
      const int lx = vox % BX, r = vox / BX;
      const int gx = x0 + lx, gy = y0 + r % BY, gz = z0 + r / BY;
      ig_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NQ * NV && gz < a.Z && gy < a.Y && gx < a.X)
        v = *(const ig_f32x4*)(a.pw_in + ((((size_t)n * a.Z + gz) * a.Y + gy) * a.X + gx) * a.pw_cs + ci0 + 4 * q);
      px[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = tid + i * 256;
      const int nn = idx / (KC / 4), k4 = (idx % (KC / 4)) * 4;
      ig_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < BM * KC / 4 && co0 + nn < a.cout) v = *(const ig_f32x4*)(a.pw_w + (size_t)(co0 + nn) * a.pw_ws + ci0 + k4);
      pq[i] = v;
    }
  };
  auto store_pw = [&](const ig_f32x4 (&px)[NPX], const ig_f32x4 (&pq)[NPW]) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int idx = tid + i * 256;
      if (idx < NQ * NV) *(ig_f32x4*)(pwx + ((size_t)(idx % NQ) * NV + idx / NQ) * 4) = px[i];
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = tid + i * 256;
      if (idx < BM * KC / 4) {
        const int nn = idx / (KC / 4), k4 = (idx % (KC / 4)) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) pww[(((size_t)(nn >> 4)) * KC + k4 + j) * 16 + (nn & 15)] = pq[i][j];
      }
    }
  };

  // halo slot of this lane's voxel for tile v (tap 0,0,0): voxel (wave*TPW + v)*16 + il of the flattened box
  // (standard boxes: 3-D z = wave, y = v; 2-D y = 4*wave + v); lanes past the box read slot 0 and store nothing
  int hb[TPW], vox[TPW];
#pragma unroll
  for (int v = 0; v < TPW; ++v) {
    const int id = (wave * TPW + v) * 16 + il;
    vox[v] = id < NV ? id : -1;
    const int lx = id % BX, r = id / BX;
    const int ly = r % BY, lz = r / BY;
    hb[v] = (id < NV ? ((lz * HY + ly) * HX + lx) * 4 : 0) + kl;
  }
  const int wb = kl * 16 + il;

  ig_f32x4 hv[NH], wv[NW], px[NPX], pq[NPW];
  const int nchunks = a.cin / KC;
  load_h(0, hv);
  load_w(0, wv);
  if (pw) load_pw(0, px, pq);
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch) __syncthreads();
    store_h(hv);
    store_w(wv);
    if (pw) store_pw(px, pq);
    __syncthreads();
    if (ch + 1 < nchunks) {
      load_h((ch + 1) * KC, hv);
      load_w((ch + 1) * KC, wv);
      if (pw) load_pw((ch + 1) * KC, px, pq);
    }
    // all taps unrolled: every tap offset is an immediate of the ds_read.  Unrolled by 3 with runtime t the loop spent 45
    // scalar instructions (t / 9, t % 3 by multiply-and-shift) and 12 address adds per 48 MFMAs, and every instruction issued
    // costs the matrix pipe its slot (DESIGN.md s3 "Round 3")
    ig_static_for<NT>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int tz = (KZ == 3) ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
      constexpr int toff = ((tz * HY + ty) * HX + tx) * 4;
#pragma unroll
      for (int s = 0; s < NQ; ++s) {
        float av[MT], bv[TPW];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wl[wb + ((m * NT + t) * KC + 4 * s) * 16];
#pragma unroll
        for (int v = 0; v < TPW; ++v) bv[v] = hal[hb[v] + toff + s * PS * 4];
#pragma unroll
        for (int v = 0; v < TPW; ++v)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[v][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[v], acc[v][m], 0, 0, 0);
      }
    });
    if (pw) {   // one more "tap": the shortcut's 1x1 weights on the box's own voxels
#pragma unroll
      for (int s = 0; s < NQ; ++s) {
        float av[MT], bv[TPW];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = pww[wb + ((size_t)m * KC + 4 * s) * 16];
#pragma unroll
        for (int v = 0; v < TPW; ++v) bv[v] = pwx[((size_t)s * NV + (vox[v] < 0 ? 0 : vox[v])) * 4 + kl];
#pragma unroll
        for (int v = 0; v < TPW; ++v)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[v][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[v], acc[v][m], 0, 0, 0);
      }
    }
  }

  // epilogue (as igemm_conv_kernel)
  // BatchNorm moments around a per-lane pivot (ursn_common.h: shifted one-pass moments); re-centred in fp64 below
  float s1[MT][4], s2[MT][4], piv[MT][4], nacc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    nacc[m] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = piv[m][r] = 0.f;
  }
#pragma unroll
  for (int v = 0; v < TPW; ++v) {
    if (vox[v] < 0) continue;
    const int lx = vox[v] % BX, r = vox[v] / BX;
    const int gx = x0 + lx, gy = y0 + r % BY, gz = z0 + r / BY;
    if (!(gz < a.Z && gy < a.Y && gx < a.X)) continue;
    float* op = a.out + ((((size_t)n * a.Z + gz) * a.Y + gy) * a.X + gx) * a.out_cs + co0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int c = 16 * m + 4 * kl;
      if (co0 + c >= a.cout) continue;
      ig_f32x4 val = acc[v][m];
      if (a.accumulate) val += *(ig_f32x4*)(op + c);
      *(ig_f32x4*)(op + c) = val;
      if constexpr (STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nacc[m] == 0.f) piv[m][r] = val[r];
          ursn_sacc(piv[m][r], s1[m][r], s2[m][r], val[r]);
        }
        nacc[m] += 1.f;
      }
    }
  }
  if constexpr (STATS) if (a.stats_partial) {
    __shared__ double red[4][2 * BM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double u, w2;
        ursn_sacc_final(piv[m][r], s1[m][r], s2[m][r], nacc[m], u, w2);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
        if (il == 0) {
          red[wave][16 * m + 4 * kl + r] = u;
          red[wave][BM + 16 * m + 4 * kl + r] = w2;
        }
      }
    __syncthreads();
    if (tid < 2 * BM)
      a.stats_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * BM + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

struct IGPlan {
  bool alltaps;   // BM = 16 | 32: igemm_at_kernel
  int var;        // igemm_at box variant (ABox): 0 standard, 1 = 4x4x12, 2 = 3x6x6
  int mode, bm;
  bool flip;
  int Z, Y, X, nbz, nby, nbx;
  size_t lds;
  int gridx, gridy;
};

template <int MODE, int BM, int KC, bool FLIP, bool STATS, int VAR = 0>
static int launch_ig_at(const IGPlan& p, const IGemmArgs& a, hipStream_t s) {
  auto kern = igemm_at_kernel<MODE, BM, KC, FLIP, STATS, VAR>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.gridy), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

template <int MODE, int BM, bool FLIP, bool STATS>
static int launch_ig(const IGPlan& p, const IGemmArgs& a, hipStream_t s) {
  auto kern = igemm_conv_kernel<MODE, BM, FLIP, STATS>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.gridy), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}
