// LDS-staged stride-2 gather convolutions and their weight gradients -- gfx950.
//
// Gather-type stride-2 passes (conv_api.hip build_geoms):
//     conv k3 s2 forward:         y[o]  = sum_t x [2o + t - pb] . W [t][ci][co]       (hi-res x  -> lo-res y)
//     transposed conv k3 s2 dgrad: dx[i] = sum_t dy[2i + t]      . Wd[t][co][ci]       (hi-res dy -> lo-res dx)
// and the matching weight gradients   dW[t][k][n] = sum_o S[2o + t - pb][k] * C[o][n]   (S hi-res, C lo-res).
// Both read the weight tensor in its stored [t][K][N] order (K = channels of the hi-res tensor).
//
// s2conv_kernel: workgroup = 4 waves, a box of lo-res output voxels (3-D 2z x 4y x 16x, 2-D 16y x 16x) x 16 produced
// channels.  Per 8-channel chunk of K the hi-res halo box ((2B+1) per axis) is staged ONCE into LDS, one plane
// per channel with the even and odd x positions split (a lane's 16 outputs then read 16 consecutive floats
// for every tap; plane stride = 16 mod 32 -> conflict-free ds_read_b32), and ALL taps of the 8 x 16 weight slab
// sit beside it, so the inner loop is 2 LDS reads per v_mfma_f32_16x16x4_f32 with no barrier.
// s2wgrad_kernel: same boxes; S halo staged as [slot][8 ci] (16-byte writes), C as [voxel][16]; a 16-row MFMA
// tile carries two taps x 8 ci; the waves split the voxel quads and are summed in LDS in fixed order.
#include <stdlib.h>
#include <utility>

#include "ursn_common.h"

typedef float s2_f32x4 __attribute__((ext_vector_type(4)));

template <int MODE> struct SBox;
template <> struct SBox<3> { static constexpr int BZ = 2, BY = 4, BX = 16, NT = 27, KZ = 3; };
template <> struct SBox<2> { static constexpr int BZ = 1, BY = 16, BX = 16, NT = 9, KZ = 1; };

struct S2Geom {
  int N, IZ, IY, IX, OZ, OY, OX, pz, py, px;  // hi-res dims, lo-res dims, pad-before (2-D: IZ = OZ = 1)
  int nbz, nby, nbx;
};

struct S2Args {
  const float* in;
  const float* w;
  float* out;
  double* stats_partial;  // [grid.y][grid.x][2][16] or null
  S2Geom g;
  int cin, cout;          // contraction channels K (hi-res tensor), produced channels (lo-res tensor)
  int in_cs, out_cs;
  int accumulate;
  int nboxes, bpw;         // boxes per workgroup (grid.x = ceil(nboxes / bpw))
};

template <int N, class F, int... I>
__device__ __forceinline__ void s2_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void s2_static_for(F&& f) {
  s2_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// per-thread staging table of the hi-res halo box: element offset from the box origin and LDS offset of every
// float4 this thread moves -- box independent, so the per-box work is one add per load (interior boxes)
template <int MODE, int NTHR = 256>
struct HaloGeom {
  using B = SBox<MODE>;
  static constexpr int HZ = (B::KZ == 3) ? 2 * B::BZ + 1 : 1, HY = 2 * B::BY + 1, HX = 2 * B::BX + 1, HXH = B::BX + 1;
  static constexpr int PS = HZ * HY * 2 * HXH;   // slots: [hz][hy][x parity][x / 2]
  static constexpr int NLD = HZ * HY * HX * 2;   // float4 loads (2 channel quads per voxel)
  static constexpr int NH = (NLD + NTHR - 1) / NTHR;
};

struct BoxPos { int n, z0, y0, x0; };
__device__ __forceinline__ BoxPos s2_box(const S2Geom& g, int box, int BZ, int BY, int BX) {
  BoxPos b;
  const int bx = box % g.nbx; box /= g.nbx;
  const int by = box % g.nby; box /= g.nby;
  const int bz = box % g.nbz;
  b.n = box / g.nbz;
  b.x0 = bx * BX; b.y0 = by * BY; b.z0 = bz * BZ;
  return b;
}

// loads the halo box of `bp` (channels c0..c0+7 of tensor `src`) into hv[]
template <int MODE, int NTHR = 256>
__device__ __forceinline__ void s2_load_halo(const float* __restrict__ src, int cs, int c0, const S2Geom& g, const BoxPos& bp,
                                             const int (&goff)[HaloGeom<MODE, NTHR>::NH], int tid,
                                             s2_f32x4 (&hv)[HaloGeom<MODE, NTHR>::NH]) {
  using H = HaloGeom<MODE, NTHR>;
  constexpr int KZ = SBox<MODE>::KZ;
  const int oz0 = (KZ == 3) ? 2 * bp.z0 - g.pz : 0, oy0 = 2 * bp.y0 - g.py, ox0 = 2 * bp.x0 - g.px;
  const int64_t base = ((((int64_t)bp.n * g.IZ + oz0) * g.IY + oy0) * g.IX + ox0) * cs + c0;
  const bool interior = oz0 >= 0 && oz0 + H::HZ <= g.IZ && oy0 >= 0 && oy0 + H::HY <= g.IY && ox0 >= 0 && ox0 + H::HX <= g.IX;
  if (interior) {
#pragma unroll
    for (int i = 0; i < H::NH; ++i) {
      s2_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((i + 1) * NTHR <= H::NLD || tid + i * NTHR < H::NLD) v = *(const s2_f32x4*)(src + base + goff[i]);
      hv[i] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < H::NH; ++i) {
      const int idx = tid + i * NTHR;
      const int sv = idx >> 1;
      const int hx = sv % H::HX, r = sv / H::HX;
      const int hy = r % H::HY, hz = r / H::HY;
      const int pz = oz0 + hz, py = oy0 + hy, px = ox0 + hx;
      s2_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < H::NLD && pz >= 0 && pz < g.IZ && py >= 0 && py < g.IY && px >= 0 && px < g.IX)
        v = *(const s2_f32x4*)(src + base + goff[i]);
      hv[i] = v;
    }
  }
}

template <int MODE, bool STATS>
__global__ __launch_bounds__(512, 4) void s2conv_kernel(S2Args a) {
  // 8 waves share one staged box (one 16-voxel row each in 3-D): 4 waves per SIMD with two workgroups per CU hide the
  // barrier / staging phases of a kernel with only ~100 MFMAs per wave and box
  constexpr int NTHR = 512, NWAVE = NTHR / 64;
  using B = SBox<MODE>;
  using H = HaloGeom<MODE, NTHR>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ;
  constexpr int HY = H::HY, HX = H::HX, HXH = H::HXH, PS = H::PS, NLD = H::NLD, NH = H::NH;
  constexpr int PSP = PS + ((16 - PS % 32) + 32) % 32;        // channel-plane stride == 16 (mod 32)
  constexpr int NR = (BZ * BY) / NWAVE;                       // 16-voxel rows per wave
  extern __shared__ __attribute__((aligned(16))) float s2l[];  // [8][PSP] halo, then [NT][8][16] weights
  float* hal = s2l;
  float* wl = s2l + 8 * PSP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  const S2Geom& g = a.g;
  const int co0 = blockIdx.y * 16;

  int goff[NH], loff[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int idx = tid + i * NTHR;
    const int sv = idx >> 1, q = idx & 1;
    const int hx = sv % HX, r = sv / HX;
    const int hy = r % HY, hz = r / HY;
    goff[i] = ((hz * g.IY + hy) * g.IX + hx) * a.in_cs + 4 * q;
    loff[i] = 4 * q * PSP + ((hz * HY + hy) * 2 + (hx & 1)) * HXH + (hx >> 1);
  }
  int hb[NR];
#pragma unroll
  for (int v = 0; v < NR; ++v) {
    const int row = NR * wave + v;
    const int lz = (MODE == 3) ? row / BY : 0, ly = (MODE == 3) ? row % BY : row;
    hb[v] = kl * PSP + ((2 * lz) * HY + 2 * ly) * 2 * HXH + il;
  }
  const int wb = kl * 16 + il;

  auto store_halo = [&](const s2_f32x4 (&hv)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      if ((i + 1) * NTHR <= NLD || tid + i * NTHR < NLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) hal[loff[i] + j * PSP] = hv[i][j];
      }
    }
  };
  constexpr int NWL = (NT * 32 + NTHR - 1) / NTHR;
  auto load_w = [&](int ci0, s2_f32x4 (&wv)[NWL]) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = tid + i * NTHR;
      const int t = idx >> 5, rem = idx & 31;
      const int k = rem >> 2, c4 = (rem & 3) * 4;
      s2_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NT * 32) v = *(const s2_f32x4*)(a.w + ((size_t)t * a.cin + ci0 + k) * a.cout + co0 + c4);
      wv[i] = v;
    }
  };
  auto store_w = [&](const s2_f32x4 (&wv)[NWL]) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < NT * 32) *(s2_f32x4*)(wl + idx * 4) = wv[i];
    }
  };

  // work items = (box, 8-channel chunk); the next item's global loads are in flight during this item's MFMAs
  const int nchunks = a.cin / 8;
  const int box_begin = ursn_xcd_block(blockIdx.x, gridDim.x) * a.bpw;
  int box_end = box_begin + a.bpw;
  if (box_end > a.nboxes) box_end = a.nboxes;
  const int nitems = (box_end - box_begin) * nchunks;

  s2_f32x4 acc[NR];
  // BatchNorm moments around a per-lane pivot (ursn_common.h: shifted one-pass moments); re-centred in fp64 below
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, piv[4] = {0.f, 0.f, 0.f, 0.f}, nacc = 0.f;
  s2_f32x4 hv[NH], wv[NWL];
  BoxPos cur = s2_box(g, box_begin, BZ, BY, BX);
  s2_load_halo<MODE, NTHR>(a.in, a.in_cs, 0, g, cur, goff, tid, hv);
  load_w(0, wv);
  int box = box_begin, ch = 0;
  for (int it = 0; it < nitems; ++it) {
    if (it) __syncthreads();
    store_halo(hv);
    if (it == 0 || nchunks > 1) store_w(wv);
    __syncthreads();
    int nbox = box, nch = ch + 1;
    if (nch == nchunks) { nch = 0; ++nbox; }
    BoxPos nxt = cur;
    if (it + 1 < nitems) {
      if (nbox != box) nxt = s2_box(g, nbox, BZ, BY, BX);
      s2_load_halo<MODE, NTHR>(a.in, a.in_cs, 8 * nch, g, nxt, goff, tid, hv);
      if (nchunks > 1) load_w(8 * nch, wv);
    }
    if (ch == 0) {
#pragma unroll
      for (int v = 0; v < NR; ++v) acc[v] = (s2_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    s2_static_for<NT>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int tz = (KZ == 3) ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
      constexpr int toff = ((tz * HY + ty) * 2 + (tx & 1)) * HXH + (tx >> 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float av = wl[wb + (t * 8 + 4 * s) * 16];
#pragma unroll
        for (int v = 0; v < NR; ++v) {
          const float bv = hal[hb[v] + 4 * s * PSP + toff];
          acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[v], 0, 0, 0);
        }
      }
    });
    if (ch == nchunks - 1) {
      // lane (il, kl) holds channels co0 + 4kl + r of lo-res voxel (row v, x0 + il)
      const int gx = cur.x0 + il;
#pragma unroll
      for (int v = 0; v < NR; ++v) {
        const int row = NR * wave + v;
        const int gz = (MODE == 3) ? cur.z0 + row / BY : 0, gy = (MODE == 3) ? cur.y0 + row % BY : cur.y0 + row;
        if (!(gz < g.OZ && gy < g.OY && gx < g.OX)) continue;
        float* op = a.out + ((((size_t)cur.n * g.OZ + gz) * g.OY + gy) * g.OX + gx) * a.out_cs + co0 + 4 * kl;
        s2_f32x4 val = acc[v];
        if (a.accumulate) val += *(s2_f32x4*)op;
        *(s2_f32x4*)op = val;
        if constexpr (STATS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (nacc == 0.f) piv[r] = val[r];
            ursn_sacc(piv[r], s1[r], s2[r], val[r]);
          }
          nacc += 1.f;
        }
      }
    }
    box = nbox; ch = nch; cur = nxt;
  }
  if constexpr (STATS) {
    __shared__ double red[NWAVE][32];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double u, w2;
      ursn_sacc_final(piv[r], s1[r], s2[r], nacc, u, w2);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (il == 0) {
        red[wave][4 * kl + r] = u;
        red[wave][16 + 4 * kl + r] = w2;
      }
    }
    __syncthreads();
    if (tid < 32)
      a.stats_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + tid] =
          ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
          ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
  }
}

// ------------------------------------------------------------------------------------------------------------
// s2conv2_kernel (round 4): the same boxes, but the operands are read as 16-byte pieces.  s2conv_kernel issues two
// ds_read_b32 per v_mfma_f32_16x16x4_f32 (one weight, one input value per lane) and is bound by instruction issue
// (MFMA-busy 0.35-0.46, 54 % of its wave cycles waiting to issue: profiles/r04_pmc_mfma_busy_cfg3.json).  Here a
// lane's float4 holds FOUR k values and element j feeds MFMA j of a group of four (any bijection between a lane
// quad's 16 k values and the four MFMAs' k indices is a valid contraction as long as A and B use the same one):
//   C16  k = 16 channels of one tap:  lane (il, kl) holds channels 4 kl .. 4 kl + 3
//   C8   k = 8 channels of two taps:  lanes kl = 0, 1 hold tap 2 p, lanes kl = 2, 3 tap 2 p + 1 (K = 8 layers)
// -> one b128 weight read and one b128 input read per FOUR MFMAs (NR + 1 per 4 NR).  Halo layout [piece][position]
// with the x parity split as before, so the 16 lanes of a k group read 256 contiguous bytes; weights are staged in
// A-operand order [tap][piece][co].
template <int MODE, bool C16, int NTHR = 512>
struct Halo2 {
  using B = SBox<MODE>;
  static constexpr int NPC = C16 ? 4 : 2;                      // 16-byte pieces per staged voxel
  static constexpr int HZ = (B::KZ == 3) ? 2 * B::BZ + 1 : 1, HY = 2 * B::BY + 1, HX = 2 * B::BX + 1, HXH = B::BX + 1;
  static constexpr int PS = HZ * HY * 2 * HXH;
  static constexpr int PSP = PS + ((4 - PS % 16) + 16) % 16;   // piece-plane stride == 4 (mod 16) float4: planes 64 B apart mod 256
  static constexpr int NLD = HZ * HY * HX * NPC;
  static constexpr int NH = (NLD + NTHR - 1) / NTHR;
  static constexpr int WST = 20;                               // float4 per (tap, piece) row of the staged weights (16 + pad)
};

template <int MODE, bool C16, bool STATS>
__global__ __launch_bounds__(512) void s2conv2_kernel(S2Args a) {
  constexpr int NTHR = 512, NWAVE = 8;
  using B = SBox<MODE>;
  using H = Halo2<MODE, C16, NTHR>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ;
  constexpr int HY = H::HY, HX = H::HX, HXH = H::HXH, PSP = H::PSP, NLD = H::NLD, NH = H::NH, NPC = H::NPC, WST = H::WST;
  constexpr int CC = C16 ? 16 : 8;                            // channels per chunk
  constexpr int NR = (BZ * BY) / NWAVE;                       // 16-voxel rows per wave
  constexpr int NG = C16 ? NT : (NT + 1) / 2;                 // groups of four MFMAs: taps | tap pairs
  extern __shared__ __attribute__((aligned(16))) float s2l[];
  s2_f32x4* hal = (s2_f32x4*)s2l;                             // [NPC][PSP]
  s2_f32x4* wl = hal + NPC * PSP;                             // [NT][NPC][WST]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  const S2Geom& g = a.g;
  const int co0 = blockIdx.y * 16;

  int goff[NH], loff[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int idx = tid + i * NTHR;
    const int sv = idx / NPC, q = idx - sv * NPC;
    const int hx = sv % HX, r = sv / HX;
    const int hy = r % HY, hz = r / HY;
    goff[i] = ((hz * g.IY + hy) * g.IX + hx) * a.in_cs + 4 * q;
    loff[i] = q * PSP + ((hz * HY + hy) * 2 + (hx & 1)) * HXH + (hx >> 1);
  }
  // B operand: lane (il, kl) reads piece plane (C16: kl | C8: kl & 1) at the lane's voxel + the tap's offset
  int hb[NR];
#pragma unroll
  for (int v = 0; v < NR; ++v) {
    const int row = NR * wave + v;
    const int lz = (MODE == 3) ? row / BY : 0, ly = (MODE == 3) ? row % BY : row;
    hb[v] = (C16 ? kl : (kl & 1)) * PSP + ((2 * lz) * HY + 2 * ly) * 2 * HXH + il;
  }
  const int wb = (C16 ? kl : (kl & 1)) * WST + il;
  const bool second = !C16 && kl >= 2;   // C8: this lane's k values belong to the pair's second tap

  auto load_halo = [&](const BoxPos& bp, int c0, s2_f32x4 (&hv)[NH]) {
    const int oz0 = (KZ == 3) ? 2 * bp.z0 - g.pz : 0, oy0 = 2 * bp.y0 - g.py, ox0 = 2 * bp.x0 - g.px;
    const int64_t base = ((((int64_t)bp.n * g.IZ + oz0) * g.IY + oy0) * g.IX + ox0) * a.in_cs + c0;
    const bool interior = oz0 >= 0 && oz0 + H::HZ <= g.IZ && oy0 >= 0 && oy0 + HY <= g.IY && ox0 >= 0 && ox0 + HX <= g.IX;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = tid + i * NTHR;
      bool ok = (i + 1) * NTHR <= NLD || idx < NLD;
      if (!interior) {
        const int sv = idx / NPC;
        const int hx = sv % HX, r = sv / HX;
        const int hy = r % HY, hz = r / HY;
        const int pz = oz0 + hz, py = oy0 + hy, px = ox0 + hx;
        ok = ok && pz >= 0 && pz < g.IZ && py >= 0 && py < g.IY && px >= 0 && px < g.IX;
      }
      s2_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *(const s2_f32x4*)(a.in + base + goff[i]);
      hv[i] = v;
    }
  };
  auto store_halo = [&](const s2_f32x4 (&hv)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if ((i + 1) * NTHR <= NLD || tid + i * NTHR < NLD) hal[loff[i]] = hv[i];
  };
  // weights of a chunk in A-operand order: element (t, piece q, co) = W[t][c0 + 4 q + (0..3)][co0 + co]
  constexpr int NWE = NT * NPC * 16, NWL = (NWE + NTHR - 1) / NTHR;
  auto load_w = [&](int c0, s2_f32x4 (&wv)[NWL]) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = tid + i * NTHR;
      s2_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NWE) {
        const int co = idx & 15, q = (idx >> 4) % NPC, t = idx / (16 * NPC);
        const float* wp = a.w + ((size_t)t * a.cin + c0 + 4 * q) * a.cout + co0 + co;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = wp[(size_t)j * a.cout];
      }
      wv[i] = v;
    }
  };
  auto store_w = [&](const s2_f32x4 (&wv)[NWL]) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < NWE) wl[(idx >> 4) * WST + (idx & 15)] = wv[i];
    }
  };

  const int nchunks = a.cin / CC;
  const int box_begin = ursn_xcd_block(blockIdx.x, gridDim.x) * a.bpw;
  int box_end = box_begin + a.bpw;
  if (box_end > a.nboxes) box_end = a.nboxes;
  const int nitems = (box_end - box_begin) * nchunks;

  s2_f32x4 acc[NR];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, piv[4] = {0.f, 0.f, 0.f, 0.f}, nacc = 0.f;
  s2_f32x4 hv[NH], wv[NWL];
  BoxPos cur = s2_box(g, box_begin, BZ, BY, BX);
  load_halo(cur, 0, hv);
  load_w(0, wv);
  int box = box_begin, ch = 0;
  for (int it = 0; it < nitems; ++it) {
    if (it) __syncthreads();
    store_halo(hv);
    if (it == 0 || nchunks > 1) store_w(wv);
    __syncthreads();
    int nbox = box, nch = ch + 1;
    if (nch == nchunks) { nch = 0; ++nbox; }
    BoxPos nxt = cur;
    if (it + 1 < nitems) {
      if (nbox != box) nxt = s2_box(g, nbox, BZ, BY, BX);
      load_halo(nxt, CC * nch, hv);
      if (nchunks > 1) load_w(CC * nch, wv);
    }
    if (ch == 0) {
#pragma unroll
      for (int v = 0; v < NR; ++v) acc[v] = (s2_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    s2_static_for<NG>([&](auto Gi) {
      constexpr int gi = decltype(Gi)::value;
      constexpr int t0 = C16 ? gi : 2 * gi, t1 = C16 ? gi : (2 * gi + 1 < NT ? 2 * gi + 1 : NT - 1);   // (phantom tap of the last pair: zero weights below)
      constexpr int tz0 = (KZ == 3) ? t0 / 9 : 0, ty0 = (t0 / 3) % 3, tx0 = t0 % 3;
      constexpr int tz1 = (KZ == 3) ? t1 / 9 : 0, ty1 = (t1 / 3) % 3, tx1 = t1 % 3;
      constexpr int toff0 = ((tz0 * HY + ty0) * 2 + (tx0 & 1)) * HXH + (tx0 >> 1);
      constexpr int toff1 = ((tz1 * HY + ty1) * 2 + (tx1 & 1)) * HXH + (tx1 >> 1);
      const int toff = second ? toff1 : toff0;
      s2_f32x4 av = wl[wb + (second ? t1 : t0) * (NPC * WST)];
      if (!C16 && 2 * gi + 1 >= NT && second) av = (s2_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NR; ++v) {
        const s2_f32x4 bv = hal[hb[v] + toff];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[v], 0, 0, 0);
      }
    });
    if (ch == nchunks - 1) {
      const int gx = cur.x0 + il;
#pragma unroll
      for (int v = 0; v < NR; ++v) {
        const int row = NR * wave + v;
        const int gz = (MODE == 3) ? cur.z0 + row / BY : 0, gy = (MODE == 3) ? cur.y0 + row % BY : cur.y0 + row;
        if (!(gz < g.OZ && gy < g.OY && gx < g.OX)) continue;
        float* op = a.out + ((((size_t)cur.n * g.OZ + gz) * g.OY + gy) * g.OX + gx) * a.out_cs + co0 + 4 * kl;
        s2_f32x4 val = acc[v];
        if (a.accumulate) val += *(s2_f32x4*)op;
        *(s2_f32x4*)op = val;
        if constexpr (STATS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (nacc == 0.f) piv[r] = val[r];
            ursn_sacc(piv[r], s1[r], s2[r], val[r]);
          }
          nacc += 1.f;
        }
      }
    }
    box = nbox; ch = nch; cur = nxt;
  }
  if constexpr (STATS) {
    __shared__ double red[NWAVE][32];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double u, w2;
      ursn_sacc_final(piv[r], s1[r], s2[r], nacc, u, w2);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (il == 0) {
        red[wave][4 * kl + r] = u;
        red[wave][16 + 4 * kl + r] = w2;
      }
    }
    __syncthreads();
    if (tid < 32)
      a.stats_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + tid] =
          ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
          ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
  }
}

// ------------------------------------------------------------------------------------------------------------
struct S2WArgs {
  const float* S;   // hi-res tensor (x for a conv, dy for a transposed conv)
  const float* C;   // lo-res tensor
  float* slab;      // [grid.z][grid.y][grid.x][NT][8][16]
  S2Geom g;
  int s_cs, c_cs;
  int nboxes, bpg;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void s2wgrad_kernel(S2WArgs a) {
  using B = SBox<MODE>;
  using H = HaloGeom<MODE>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ;
  constexpr int HY = H::HY, HX = H::HX, HXH = H::HXH, PS = H::PS, NLD = H::NLD, NH = H::NH;
  constexpr int DS = BZ * BY * BX, NLC = DS * 4, NC = (NLC + 255) / 256;
  constexpr int NP = (NT + 1) / 2;             // tap pairs: 16 MFMA rows = 2 taps x 8 ci
  constexpr int NQW = DS / 16;                 // voxel quads per wave: rows NQW/4*wave .. (4 quads per 16-voxel row)
  extern __shared__ __attribute__((aligned(16))) float s2w[];  // [PS][8] S halo, then [DS][16] C
  float* sl = s2w;
  float* cl = s2w + PS * 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  const S2Geom& g = a.g;
  const int ci0 = blockIdx.y * 8, co0 = blockIdx.z * 16;

  int goff[NH], loff[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int idx = tid + i * 256;
    const int sv = idx >> 1, q = idx & 1;
    const int hx = sv % HX, r = sv / HX;
    const int hy = r % HY, hz = r / HY;
    goff[i] = ((hz * g.IY + hy) * g.IX + hx) * a.s_cs + 4 * q;
    loff[i] = (((hz * HY + hy) * 2 + (hx & 1)) * HXH + (hx >> 1)) * 8 + 4 * q;
  }
  int coff[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = tid + i * 256;
    const int vox = idx >> 2, q = idx & 3;
    const int vx = vox % BX, r = vox / BX;
    const int vy = r % BY, vz = r / BY;
    coff[i] = ((vz * g.OY + vy) * g.OX + vx) * a.c_cs + 4 * q;
  }

  // this wave's first row of the box (rows of 16 lo-res voxels; NQW/4 rows per wave) and the lane's operand bases
  const int row0 = (NQW / 4) * wave;
  const int lz0 = (MODE == 3) ? row0 / BY : 0, ly0 = (MODE == 3) ? row0 % BY : row0;
  s2_f32x4 acc[NP];
  int t_off[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    acc[m] = (s2_f32x4){0.f, 0.f, 0.f, 0.f};
    int t = 2 * m + (il >> 3);
    if (t >= NT) t = NT - 1;   // phantom tap of the last pair: finite operand, its rows are dropped below
    const int tz = (KZ == 3) ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
    t_off[m] = ((((2 * lz0 + tz) * HY + 2 * ly0 + ty) * 2 + (tx & 1)) * HXH + (tx >> 1) + kl) * 8 + (il & 7);
  }
  const int b_off = ((row0 * BX) + kl) * 16 + il;

  auto load_box = [&](const BoxPos& bp, s2_f32x4 (&sv4)[NH], s2_f32x4 (&cv4)[NC]) {
    s2_load_halo<MODE>(a.S, a.s_cs, ci0, g, bp, goff, tid, sv4);
    const int64_t cbase = ((((int64_t)bp.n * g.OZ + bp.z0) * g.OY + bp.y0) * g.OX + bp.x0) * a.c_cs + co0;
    const bool interior = bp.z0 + BZ <= g.OZ && bp.y0 + BY <= g.OY && bp.x0 + BX <= g.OX;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int idx = tid + i * 256;
      bool ok = (i + 1) * 256 <= NLC || idx < NLC;
      if (!interior) {
        const int vox = idx >> 2;
        const int vx = vox % BX, r = vox / BX;
        const int vy = r % BY, vz = r / BY;
        ok = ok && bp.z0 + vz < g.OZ && bp.y0 + vy < g.OY && bp.x0 + vx < g.OX;
      }
      s2_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *(const s2_f32x4*)(a.C + cbase + coff[i]);
      cv4[i] = v;
    }
  };
  auto store_box = [&](const s2_f32x4 (&sv4)[NH], const s2_f32x4 (&cv4)[NC]) {
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if ((i + 1) * 256 <= NLD || tid + i * 256 < NLD) *(s2_f32x4*)(sl + loff[i]) = sv4[i];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int idx = tid + i * 256;
      if ((i + 1) * 256 <= NLC || idx < NLC) *(s2_f32x4*)(cl + idx * 4) = cv4[i];
    }
  };

  const int box_begin = ursn_xcd_block(blockIdx.x, gridDim.x) * a.bpg;
  int box_end = box_begin + a.bpg;
  if (box_end > a.nboxes) box_end = a.nboxes;
  s2_f32x4 sv4[NH], cv4[NC];
  if (box_begin < box_end) load_box(s2_box(g, box_begin, BZ, BY, BX), sv4, cv4);
  for (int box = box_begin; box < box_end; ++box) {
    __syncthreads();
    store_box(sv4, cv4);
    __syncthreads();
    if (box + 1 < box_end) load_box(s2_box(g, box + 1, BZ, BY, BX), sv4, cv4);
    s2_static_for<NQW>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int dr = j >> 2, xq = (j & 3) * 4;                 // row within the wave's rows, x of the quad
      const float bv = cl[b_off + (dr * BX + xq) * 16];
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const float av = sl[t_off[m] + ((2 * dr) * 2 * HXH + xq) * 8];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
      }
    });
  }

  // fixed-order sum of the 4 waves in LDS, one slab per workgroup.  D row = 4kl + r = (tap in pair, ci), col = co
  __syncthreads();
  float* red = s2w;
  for (int i = tid; i < NT * 128; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < NP; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * kl + r, t = 2 * m + (row >> 3);
          if (t < NT) red[(t * 8 + (row & 7)) * 16 + il] += acc[m][r];
        }
    }
    __syncthreads();
  }
  float* slab = a.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)(NT * 128);
  for (int i = tid; i < NT * 128; i += 256) slab[i] = red[i];
}

// dw[t][k][n] += sum_g slab[(cz*ncy + cy)*ng + g][t][k%8][n%16]
__global__ __launch_bounds__(256) void s2w_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slab, int taps,
                                                         int K, int Nn, int ng) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t total = (int64_t)taps * K * Nn;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f;
  if (i < total) {
    const int co = (int)(i % Nn);
    const int64_t r = i / Nn;
    const int ci = (int)(r % K);
    const int t = (int)(r / K);
    const int cy = ci / 8, cz = co / 16, ncy = K / 8;
    const int64_t per = (int64_t)taps * 128;
    const float* base = slab + ((int64_t)(cz * ncy + cy) * ng) * per + ((int64_t)t * 8 + (ci & 7)) * 16 + (co & 15);
    int gi = cg;
    for (; gi + 4 < ng; gi += 8) {
      s0 += base[(int64_t)gi * per];
      s1 += base[(int64_t)(gi + 4) * per];
    }
    for (; gi < ng; gi += 4) s0 += base[(int64_t)gi * per];
  }
  sm[cg][e] = s0 + s1;
  __syncthreads();
  if (cg == 0 && i < total) dw[i] += (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}

// ------------------------------------------------------------------------------------------------------------
struct S2Plan {
  int mode, K, Nn, in_cs, out_cs;   // hi-res tensor channels / stride, lo-res tensor channels / stride
  S2Geom g;
  int nboxes, bpw, gridx;   // conv: boxes per workgroup, workgroups per produced-channel tile
  size_t lds_conv, lds_wgrad;
  int ngroups, bpg;
  size_t scratch;
};

static bool s2_enabled(const ursn_conv_desc& d) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("URSN_DISABLE_TILED");
    const char* f = getenv("URSN_STRIDE2");
    off = ((e && e[0] == '1') || (f && f[0] == '0')) ? 1 : 0;
  }
  return !off || d.algo == 6;
}

// gather-type stride-2 geometry of a descriptor (conv s2: x hi-res; transposed: dy hi-res)
static bool make_s2plan(const ursn_conv_desc& d, S2Plan& p) {
  if (!s2_enabled(d)) return false;
  if (d.k != 3 || d.stride != 2 || d.in_split || d.in_mean) return false;
  if (d.ndim != 2 && d.ndim != 3) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  p.mode = d.ndim;
  p.K = d.transposed ? d.cout : d.cin;
  p.Nn = d.transposed ? d.cin : d.cout;
  p.in_cs = d.transposed ? ocs : ics;
  p.out_cs = d.transposed ? ics : ocs;
  if ((p.K % 8) || (p.Nn % 16) || (p.in_cs & 3) || (p.out_cs & 3)) return false;
  int hi[3] = {1, 1, 1}, lo[3] = {1, 1, 1}, pb[3] = {0, 0, 0};
  const int lead = 3 - d.ndim;
  for (int j = 0; j < d.ndim; ++j) {
    const int sz = d.in_sp[j];
    if (sz < 1) return false;
    if (d.transposed) { lo[lead + j] = sz; hi[lead + j] = 2 * sz; pb[lead + j] = 0; }
    else {
      const int o = (sz + 1) / 2;
      int tot = (o - 1) * 2 + 3 - sz;
      if (tot < 0) tot = 0;
      hi[lead + j] = sz; lo[lead + j] = o; pb[lead + j] = tot / 2;
    }
  }
  S2Geom& g = p.g;
  g.N = d.n;
  g.IZ = hi[0]; g.IY = hi[1]; g.IX = hi[2];
  g.OZ = lo[0]; g.OY = lo[1]; g.OX = lo[2];
  g.pz = pb[0]; g.py = pb[1]; g.px = pb[2];
  static const int minx = getenv("URSN_S2_MINX") ? atoi(getenv("URSN_S2_MINX")) : 8;
  if (g.OX < minx && d.algo != 6) return false;   // 16-wide lo-res x tiles: below that the gather kernel wastes less
  const int BZ = p.mode == 3 ? 2 : 1, BY = p.mode == 3 ? 4 : 16, BX = 16;
  g.nbz = (g.OZ + BZ - 1) / BZ;
  g.nby = (g.OY + BY - 1) / BY;
  g.nbx = (g.OX + BX - 1) / BX;
  const int64_t nb = (int64_t)d.n * g.nbz * g.nby * g.nbx;
  if (nb > (1 << 30)) return false;
  p.nboxes = (int)nb;
  const int NT = p.mode == 3 ? 27 : 9;
  const int HZ = p.mode == 3 ? 2 * BZ + 1 : 1, HY = 2 * BY + 1, HXH = BX + 1;
  const int PS = HZ * HY * 2 * HXH;
  const int PSP = PS + ((16 - PS % 32) + 32) % 32;
  p.lds_conv = ((size_t)8 * PSP + (size_t)NT * 128) * sizeof(float);
  // several boxes per workgroup so the next box's loads overlap the MFMAs; keep >= ~2048 workgroups
  p.bpw = (int)(((int64_t)p.nboxes * (p.Nn / 16)) / 2048);
  if (p.bpw < 1) p.bpw = 1;
  if (p.bpw > 8) p.bpw = 8;
  p.gridx = (p.nboxes + p.bpw - 1) / p.bpw;
  p.lds_wgrad = ((size_t)PS * 8 + (size_t)BZ * BY * BX * 16) * sizeof(float);
  const int blocks = (p.K / 8) * (p.Nn / 16);
  int want = 1024 / blocks;
  if (want < 1) want = 1;
  if (want > p.nboxes) want = p.nboxes;
  p.bpg = (p.nboxes + want - 1) / want;
  p.ngroups = (p.nboxes + p.bpg - 1) / p.bpg;
  p.scratch = (size_t)blocks * p.ngroups * NT * 128 * sizeof(float);
  return p.scratch <= ((size_t)1 << 30);
}

int stride2_conv_supported(const ursn_conv_desc& d, ConvPass pass) {
  if (!((!d.transposed && pass == PASS_FWD) || (d.transposed && pass == PASS_DGRAD))) return 0;
  S2Plan p;
  return make_s2plan(d, p) ? 1 : 0;
}

size_t stride2_stats_scratch_doubles(const ursn_conv_desc& d) {
  S2Plan p;
  if (d.transposed || !make_s2plan(d, p)) return 0;
  return (size_t)p.gridx * (p.Nn / 16) * 32;
}

template <int MODE, bool STATS>
static int launch_s2c(const S2Plan& p, const S2Args& a, hipStream_t s) {
  auto kern = s2conv_kernel<MODE, STATS>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds_conv > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_conv));
    attr_lds = p.lds_conv;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.Nn / 16), dim3(512), p.lds_conv, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

template <int MODE, bool C16, bool STATS>
static int launch_s2c2(const S2Plan& p, const S2Args& a, hipStream_t s) {
  using H = Halo2<MODE, C16, 512>;
  constexpr int NT = SBox<MODE>::NT;
  const size_t lds = ((size_t)H::NPC * H::PSP + (size_t)NT * H::NPC * H::WST) * 16;
  auto kern = s2conv2_kernel<MODE, C16, STATS>;
  static size_t attr_lds = 48 * 1024;
  if (lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.Nn / 16), dim3(512), lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_stride2_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                        int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s) {
  S2Plan p;
  URSN_REQUIRE(stride2_conv_supported(d, pass) && make_s2plan(d, p), "stride-2 conv: unsupported shape");
  S2Args a;
  a.in = in; a.w = w; a.out = out; a.stats_partial = stats_partial;
  a.g = p.g;
  a.cin = p.K; a.cout = p.Nn; a.in_cs = p.in_cs; a.out_cs = p.out_cs;
  a.accumulate = accumulate;
  a.nboxes = p.nboxes; a.bpw = p.bpw;
  ursn_note_kernel(d.transposed ? "s2conv(deconv dgrad)" : "s2conv");
  int rc;
  // 16-byte operand reads (s2conv2_kernel): one weight and one input read per four MFMAs instead of one each per MFMA.  Measured
  // (tools/s2_v2_sweep.sh, forward of the stride-2 conv / data gradient of the transposed conv, ms): the issue diet alone buys
  // nothing -- 3-D 192^3 8 -> 16: 0.376 vs 0.336, 96^3 16 -> 32: 0.198 vs 0.180, 24^3 64 -> 128: 0.088 vs 0.078; 2-D 512^2 x 16
  // 16 -> 32: 0.182 vs 0.175, 64^2 x 16 128 -> 256: 0.156 vs 0.131 -- except where the old kernel's four / eight 8-channel chunk
  // items per box stall it: 2-D 256^2 x 16 32 -> 64: 0.173 vs 0.321, 128^2 x 16 64 -> 128: 0.173 vs 0.231 (and their transposed
  // twins).  Default: those shapes only.  URSN_S2CONV_V2=0 never, =2 wherever the channel count allows
  static const int v2mode = getenv("URSN_S2CONV_V2") ? atoi(getenv("URSN_S2CONV_V2")) : 1;
  const bool v2 = v2mode == 2 || (v2mode == 1 && p.mode == 2 && (p.K == 32 || p.K == 64));
  if (v2) {
    const bool c16 = (p.K % 16) == 0;
#define S2V2(m_, c_) (stats_partial ? launch_s2c2<m_, c_, true>(p, a, s) : launch_s2c2<m_, c_, false>(p, a, s))
    if (p.mode == 3) rc = c16 ? S2V2(3, true) : S2V2(3, false);
    else rc = c16 ? S2V2(2, true) : S2V2(2, false);
#undef S2V2
  } else if (p.mode == 3) rc = stats_partial ? launch_s2c<3, true>(p, a, s) : launch_s2c<3, false>(p, a, s);
  else rc = stats_partial ? launch_s2c<2, true>(p, a, s) : launch_s2c<2, false>(p, a, s);
  if (rc) return rc;
  if (stats_partial) {
    const int64_t V = (int64_t)d.n * p.g.OZ * p.g.OY * p.g.OX;
    URSN_TRY(launch_bn_stats_final_blocked(stats_partial, p.gridx, p.Nn, 16, 16, (size_t)p.gridx * 32, V, eps, mean, rstd, s));
  }
  return 0;
}

int stride2_wgrad_supported(const ursn_conv_desc& d) {
  S2Plan p;
  return make_s2plan(d, p) ? 1 : 0;
}
size_t stride2_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  S2Plan p;
  return make_s2plan(d, p) ? p.scratch : 0;
}

template <int MODE>
static int launch_s2w(const S2Plan& p, const S2WArgs& a, dim3 grid, hipStream_t s) {
  auto kern = s2wgrad_kernel<MODE>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds_wgrad > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_wgrad));
    attr_lds = p.lds_wgrad;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), p.lds_wgrad, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_stride2_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                         size_t scratch_bytes, hipStream_t s) {
  S2Plan p;
  URSN_REQUIRE(make_s2plan(d, p), "stride-2 wgrad: unsupported shape");
  URSN_REQUIRE(scratch && scratch_bytes >= p.scratch, "stride-2 wgrad: scratch too small (%zu < %zu)", scratch_bytes, p.scratch);
  S2WArgs a;
  a.S = d.transposed ? dy : x;
  a.C = d.transposed ? x : dy;
  a.slab = (float*)scratch;
  a.g = p.g;
  a.s_cs = p.in_cs; a.c_cs = p.out_cs;
  a.nboxes = p.nboxes; a.bpg = p.bpg;
  dim3 grid(p.ngroups, p.K / 8, p.Nn / 16);
  ursn_note_kernel("s2wgrad");
  URSN_TRY(p.mode == 3 ? launch_s2w<3>(p, a, grid, s) : launch_s2w<2>(p, a, grid, s));
  const int taps = p.mode == 3 ? 27 : 9;
  const int64_t total = (int64_t)taps * p.K * p.Nn;
  hipLaunchKernelGGL(s2w_reduce_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(256), 0, s, dw, (const float*)scratch, taps,
                     p.K, p.Nn, p.ngroups);
  URSN_HIP(hipGetLastError());
  return 0;
}
