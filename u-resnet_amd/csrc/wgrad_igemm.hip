// LDS-staged weight gradient for k3 s1 layers with >= 32 channels (mid/deep levels, most of the 2-D net).
//
//   dW[t][ci][co] = sum_v x[v + t - 1][ci] * dz[v][co]       GEMM: M = ci, N = co, K = voxels
//
// A workgroup owns a (16 ci) x (16*MT co) block of dW and a group of 256-voxel boxes.  Per box the input box +
// halo (16 channels) and the dz box (16*MT channels) are staged once into LDS as [channel quad][slot] float4
// (plane stride padded to 2 mod 8 slots -> the 16-channel x 2-voxel operand reads of v_mfma_f32_16x16x4_f32
// hit 32 distinct banks).  The 27 (9) taps are split over the 4 waves (7+7+7+6), every wave sweeps all 64
// voxel quads of the box for its taps: per quad MT reads of dz feed 7*MT MFMAs, and no cross-wave reduction
// is needed.  One slab per workgroup; a single reduce kernel sums the box groups and ADDS into the gradient.
#include <stdlib.h>
#include <utility>

#include "ursn_common.h"
#include "buffer_stage.h"

typedef float iw_f32x4 __attribute__((ext_vector_type(4)));

template <int... Is, class F>
__device__ __forceinline__ void iw_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void iw_static_for(F&& f) {
  iw_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct IGWArgs {
  const float* x;
  const float* dz;
  float* slab;  // [grid.z][grid.y][grid.x][taps][16][BN]
  int N, Z, Y, X;
  int x_cs, dz_cs;
  int nbz, nby, nbx, nboxes, boxes_per_group;
};

template <int MODE, int VAR = 0> struct WBox;
template <> struct WBox<3, 0> { static constexpr int BZ = 4, BY = 4, BX = 16, NT = 27, KZ = 3; };
template <> struct WBox<3, 1> { static constexpr int BZ = 4, BY = 4, BX = 12, NT = 27, KZ = 3; };   // 12- and 24-wide rows
template <> struct WBox<2, 0> { static constexpr int BZ = 1, BY = 16, BX = 16, NT = 9, KZ = 1; };

template <int MODE, int MT, int VAR = 0>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_kernel(IGWArgs a) {
  using B = WBox<MODE, VAR>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ;
  constexpr int HZ = BZ + (KZ - 1), HY = BY + 2, HX = BX + 2, PS = HZ * HY * HX;
  constexpr int PSP = PS + ((2 - PS % 8) + 8) % 8;          // plane stride == 2 (mod 8) slots
  constexpr int DS = BZ * BY * BX, DSP = DS + 2;            // 256 voxels, padded likewise
  constexpr int BN = 16 * MT;
  // 3-D: the 27 taps are split over the waves (7+7+7+6); 2-D: 9 taps do not split evenly (3+3+3+0), so every
  // wave keeps all 9 taps and the waves split the 64 voxel quads instead (fixed-order LDS sum at the end)
  constexpr bool SPLITK = (MODE == 2);
  constexpr int TPW = SPLITK ? NT : (NT + 3) / 4;
  constexpr int NHX = (4 * PS + 255) / 256, NHD = (BN / 4 * DS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float iwl[];  // [4][PSP][4] x, then [BN/4][DSP][4] dz
  float* xl = iwl;
  float* dl = iwl + 4 * PSP * 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * BN;

  iw_f32x4 acc[TPW][MT];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[i][m] = (iw_f32x4){0.f, 0.f, 0.f, 0.f};

  // this wave's taps: halo-slot offset of the tap shift.  A tap index past the last one (wave 3 holds 6 of its 7
  // slots) reads tap 0's (finite) data; its accumulator is never written out.
  int abase[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int t = SPLITK ? i : wave * TPW + i;
    if (t >= NT) t = 0;
    int tz = (KZ == 3) ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
    abase[i] = ((il >> 2) * PSP + kl) * 4 + (il & 3) + ((tz * HY + ty) * HX + tx + (SPLITK ? 4 * wave : 0)) * 4;
  }
  const int b_lane = ((il >> 2) * DSP + kl + (SPLITK ? 4 * wave : 0)) * 4 + (il & 3);   // + (m*4*DSP + voxel slot of quad start)*4

  // staging in two phases (loads of box b+1 are in flight during the MFMAs of box b).  Buffer loads (buffer_stage.h): the
  // element's offset is box origin + a per-thread constant tabulated once; planes outside the volume fall outside the
  // image's byte range by themselves (an origin below the image wraps to a huge unsigned offset), rows and columns outside it
  // are two range compares per element -- instead of rebuilding (z, y, x) from idx, five bounds tests and a 64-bit address
  // per element and box.
  unsigned xrel[NHX], xyx[NHX], drel[NHD], dyx[NHD];   // byte offset relative to the box origin; (row | column << 16) in the box
#pragma unroll
  for (int i = 0; i < NHX; ++i) {
    const int idx = tid + i * 256;
    const int s = idx >> 2, q = idx & 3;
    const int hx = s % HX, r = s / HX;
    const int hy = r % HY, hz = r / HY;
    xrel[i] = (unsigned)(((hz * a.Y + hy) * a.X + hx) * a.x_cs + 4 * q) * 4u;
    xyx[i] = idx < 4 * PS ? (unsigned)hy | ((unsigned)hx << 16) : 0xffffffffu;   // past the halo: never in range
    asm volatile("" : "+v"(xrel[i]));
    asm volatile("" : "+v"(xyx[i]));
  }
#pragma unroll
  for (int i = 0; i < NHD; ++i) {
    const int idx = tid + i * 256;
    const int s = idx / (BN / 4), q = idx % (BN / 4);
    const int vx = s % BX, r = s / BX;
    const int vy = r % BY, vz = r / BY;
    drel[i] = (unsigned)(((vz * a.Y + vy) * a.X + vx) * a.dz_cs + 4 * q) * 4u;
    dyx[i] = idx < BN / 4 * DS ? (unsigned)vy | ((unsigned)vx << 16) : 0xffffffffu;
    asm volatile("" : "+v"(drel[i]));
    asm volatile("" : "+v"(dyx[i]));
  }
  const unsigned x_img_bytes = (unsigned)a.Z * a.Y * a.X * a.x_cs * 4u, d_img_bytes = (unsigned)a.Z * a.Y * a.X * a.dz_cs * 4u;
  auto load_box = [&](int box, iw_f32x4 (&xv)[NHX], iw_f32x4 (&dv)[NHD]) {
    int bid = box;
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby; bid /= a.nby;
    const int bz = bid % a.nbz;
    const int n = bid / a.nbz;
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
    const __amdgpu_buffer_rsrc_t rx = ursn_plane_rsrc(a.x + (size_t)n * a.Z * a.Y * a.X * a.x_cs + ci0, x_img_bytes);
    const __amdgpu_buffer_rsrc_t rd = ursn_plane_rsrc(a.dz + (size_t)n * a.Z * a.Y * a.X * a.dz_cs + co0, d_img_bytes);
    // halo origin (z0 - 1 | z0, y0 - 1, x0 - 1); rows hy in [ylo, yhi) and columns hx in [xlo, xhi) of the halo are inside the image
    const unsigned xorg = (unsigned)((((z0 - (KZ == 3 ? 1 : 0)) * a.Y + y0 - 1) * a.X + x0 - 1) * a.x_cs) * 4u;
    const unsigned ylo = y0 == 0 ? 1u : 0u, yn = (unsigned)(a.Y - y0 + 1 < HY ? a.Y - y0 + 1 : HY) - ylo;
    const unsigned xlo = x0 == 0 ? 1u : 0u, xn = (unsigned)(a.X - x0 + 1 < HX ? a.X - x0 + 1 : HX) - xlo;
#pragma unroll
    for (int i = 0; i < NHX; ++i) {
      const bool ok = ((xyx[i] & 0xffffu) - ylo) < yn && ((xyx[i] >> 16) - xlo) < xn;
      xv[i] = ursn_buffer_load_f4(rx, ok ? xorg + xrel[i] : URSN_OOB_OFFSET);
    }
    const unsigned dorg = (unsigned)(((z0 * a.Y + y0) * a.X + x0) * a.dz_cs) * 4u;
    const unsigned dyn = (unsigned)(a.Y - y0), dxn = (unsigned)(a.X - x0);
#pragma unroll
    for (int i = 0; i < NHD; ++i) {
      const bool ok = (dyx[i] & 0xffffu) < dyn && (dyx[i] >> 16) < dxn;
      dv[i] = ursn_buffer_load_f4(rd, ok ? dorg + drel[i] : URSN_OOB_OFFSET);
    }
  };
  auto store_box = [&](const iw_f32x4 (&xv)[NHX], const iw_f32x4 (&dv)[NHD]) {
#pragma unroll
    for (int i = 0; i < NHX; ++i) {
      const int idx = tid + i * 256;
      if (idx < 4 * PS) *(iw_f32x4*)(xl + ((size_t)(idx & 3) * PSP + (idx >> 2)) * 4) = xv[i];
    }
#pragma unroll
    for (int i = 0; i < NHD; ++i) {
      const int idx = tid + i * 256;
      if (idx < BN / 4 * DS) *(iw_f32x4*)(dl + ((size_t)(idx % (BN / 4)) * DSP + idx / (BN / 4)) * 4) = dv[i];
    }
  };

  const int box_begin = ursn_xcd_block(blockIdx.x, gridDim.x) * a.boxes_per_group;
  int box_end = box_begin + a.boxes_per_group;
  if (box_end > a.nboxes) box_end = a.nboxes;
  iw_f32x4 xv[NHX], dv[NHD];
  if (box_begin < box_end) load_box(box_begin, xv, dv);
  for (int box = box_begin; box < box_end; ++box) {
    __syncthreads();
    store_box(xv, dv);
    __syncthreads();
    if (box + 1 < box_end) load_box(box + 1, xv, dv);
    // 3-D: every wave sweeps the 16 rows x 4 quads of the box for its taps; 2-D (SPLITK): wave w takes quad w of every row
    // all rows unrolled: every LDS offset is an immediate (by 2 with runtime row the loop paid row / BY, row % BY and the
    // address adds per row; every instruction issued costs the matrix pipe its slot, DESIGN.md s3 "Round 3")
    iw_static_for<BZ * BY>([&](auto ROW) {
      constexpr int row = decltype(ROW)::value;
      constexpr int vz = (MODE == 3) ? row / BY : 0, vy = (MODE == 3) ? row % BY : row;
      const float* xr = xl + ((vz * HY + vy) * HX) * 4;
      const float* dr = dl + (row * BX) * 4;
#pragma unroll
      for (int qx = 0; qx < (SPLITK ? 1 : BX / 4); ++qx) {
        float bv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) bv[m] = dr[b_lane + ((size_t)m * 4 * DSP + 4 * qx) * 4];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const float av = xr[abase[i] + 16 * qx];
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[m], acc[i][m], 0, 0, 0);
        }
      }
    });
  }

  // D: row = ci (4*kl + r), col = co (il)
  float* slab = a.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)(NT * 16 * BN);
  if constexpr (SPLITK) {
    __syncthreads();
    float* red = iwl;  // NT*16*BN floats fit in the halo region
    for (int i = tid; i < NT * 16 * BN; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)i * 16 + 4 * kl + r) * BN + 16 * m + il] += acc[i][m][r];
      }
      __syncthreads();
    }
    for (int i = tid; i < NT * 16 * BN; i += 256) slab[i] = red[i];
  } else {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave * TPW + i;
      if (t >= NT) continue;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[((size_t)t * 16 + 4 * kl + r) * BN + 16 * m + il] = acc[i][m][r];
    }
  }
}

// dw[t][ci][co] += sum_g slab[(cz*ncy + cy)*ng + g][t][ci%16][co%BN]
__global__ __launch_bounds__(256) void igw_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slab, int taps,
                                                         int cin, int cout, int BN, int ng) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t total = (int64_t)taps * cin * cout;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f;
  if (i < total) {
    int co = (int)(i % cout);
    int64_t r = i / cout;
    int ci = (int)(r % cin);
    int t = (int)(r / cin);
    int cy = ci / 16, cz = co / BN, ncy = cin / 16;
    const int64_t per = (int64_t)taps * 16 * BN;
    const float* base = slab + ((int64_t)(cz * ncy + cy) * ng) * per + ((int64_t)t * 16 + (ci & 15)) * BN + (co % BN);
    int g = cg;
    for (; g + 4 < ng; g += 8) {
      s0 += base[(int64_t)g * per];
      s1 += base[(int64_t)(g + 4) * per];
    }
    for (; g < ng; g += 4) s0 += base[(int64_t)g * per];
  }
  sm[cg][e] = s0 + s1;
  __syncthreads();
  if (cg == 0 && i < total) dw[i] += (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}

struct IGWPlan {
  int var;   // 1: 4x4x12 boxes (3-D rows of 12 / 24 voxels)
  int mode, mt, Z, Y, X, nbz, nby, nbx, nboxes, ngroups, bpg;
  size_t lds, scratch;
};

static bool make_igwplan(const ursn_conv_desc& d, IGWPlan& p) {
  {
    static int off = -1;
    if (off < 0) {
      const char* e = getenv("URSN_DISABLE_TILED");
      const char* f = getenv("URSN_IGEMM");
      off = ((e && e[0] == '1') || (f && f[0] == '0')) ? 1 : 0;
    }
    if (off && d.algo != 4) return false;
  }
  if (d.transposed || d.k != 3 || d.stride != 1 || d.in_split || d.in_mean) return false;
  if ((d.cin % 16) || (d.cout % 16)) return false;
  if (d.cin <= 16 && d.cout <= 16 && d.algo != 4) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || (ocs & 3)) return false;
  p.mode = d.ndim;
  if (d.ndim == 3) { p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2]; }
  else { p.Z = 1; p.Y = d.in_sp[0]; p.X = d.in_sp[1]; }
  // buffer-path staging (buffer_stage.h): 32-bit byte offsets inside one image, below the out-of-range marker
  if ((int64_t)p.Z * p.Y * p.X * (ics > ocs ? ics : ocs) * 4 >= (int64_t)0x80000000ll) return false;
  static const int min_x = getenv("URSN_IGEMM_MINX") ? atoi(getenv("URSN_IGEMM_MINX")) : 12;
  if (p.X < min_x && d.algo != 4) return false;
  p.var = (p.mode == 3 && (p.X % 16) != 0 && (p.X % 16) <= 12 && (p.X % 12) == 0) ? 1 : 0;
  const int BZ = p.mode == 3 ? 4 : 1, BY = p.mode == 3 ? 4 : 16, BX = p.var ? 12 : 16;
  p.nbz = (p.Z + BZ - 1) / BZ;
  p.nby = (p.Y + BY - 1) / BY;
  p.nbx = (p.X + BX - 1) / BX;
  p.nboxes = d.n * p.nbz * p.nby * p.nbx;
  p.mt = (d.cout % 32 == 0) ? 2 : 1;
  const int blocks = (d.cin / 16) * (d.cout / (16 * p.mt));
  int want = 1024 / blocks;
  if (want < 1) want = 1;
  if (want > p.nboxes) want = p.nboxes;
  p.bpg = (p.nboxes + want - 1) / want;
  p.ngroups = (p.nboxes + p.bpg - 1) / p.bpg;
  const int taps = p.mode == 3 ? 27 : 9;
  const int HZ = p.mode == 3 ? 6 : 1, PS = HZ * (BY + 2) * (BX + 2);
  const int PSP = PS + ((2 - PS % 8) + 8) % 8;
  p.lds = ((size_t)4 * PSP * 4 + (size_t)(4 * p.mt) * (BZ * BY * BX + 2) * 4) * sizeof(float);
  p.scratch = (size_t)blocks * p.ngroups * taps * 16 * (16 * p.mt) * sizeof(float);
  return p.scratch <= ((size_t)1 << 30);
}

int igemm_wgrad_supported(const ursn_conv_desc& d) {
  IGWPlan p;
  return make_igwplan(d, p) ? 1 : 0;
}
size_t igemm_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  IGWPlan p;
  return make_igwplan(d, p) ? p.scratch : 0;
}

template <int MODE, int MT, int VAR = 0>
static int launch_igw(const IGWPlan& p, const IGWArgs& a, dim3 grid, hipStream_t s) {
  auto kern = igemm_wgrad_kernel<MODE, MT, VAR>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_igemm_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                       size_t scratch_bytes, hipStream_t s) {
  IGWPlan p;
  URSN_REQUIRE(make_igwplan(d, p), "igemm wgrad: unsupported shape");
  URSN_REQUIRE(scratch && scratch_bytes >= p.scratch, "igemm wgrad: scratch too small (%zu < %zu)", scratch_bytes, p.scratch);
  IGWArgs a;
  a.x = x; a.dz = dy; a.slab = (float*)scratch;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.x_cs = d.in_cstride > 0 ? d.in_cstride : d.cin;
  a.dz_cs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.nboxes = p.nboxes; a.boxes_per_group = p.bpg;
  dim3 grid(p.ngroups, d.cin / 16, d.cout / (16 * p.mt));
  ursn_note_kernel(p.mt == 2 ? "igemm_wgrad<32>" : "igemm_wgrad<16>");
  int rc;
  if (p.mode == 3 && p.var == 1) rc = (p.mt == 2) ? launch_igw<3, 2, 1>(p, a, grid, s) : launch_igw<3, 1, 1>(p, a, grid, s);
  else if (p.mode == 3) rc = (p.mt == 2) ? launch_igw<3, 2>(p, a, grid, s) : launch_igw<3, 1>(p, a, grid, s);
  else rc = (p.mt == 2) ? launch_igw<2, 2>(p, a, grid, s) : launch_igw<2, 1>(p, a, grid, s);
  if (rc) return rc;
  const int taps = p.mode == 3 ? 27 : 9;
  const int64_t total = (int64_t)taps * d.cin * d.cout;
  hipLaunchKernelGGL(igw_reduce_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(256), 0, s, dw, (const float*)scratch, taps,
                     d.cin, d.cout, 16 * p.mt, p.ngroups);
  URSN_HIP(hipGetLastError());
  return 0;
}
