// bf16 weight gradient of the 3x3x3 stride-1 layers of the DEEP levels (>= 64 produced channels, contraction channels a multiple
// of 32: levels 3-5 of an F = 8 network, lib/resnet_module.py:43-66 as built by lib/uresnet.py:56-64,95-100) --
//     dW[t][ci][co] += sum_v S[v + d_t][ci] * C[v][co]          (S = the layer's input, C = dz; fp32 accumulation, fp32 dW)
// on v_mfma_f32_32x32x16_bf16, a workgroup of SEVEN waves.
//
// The generic kernel (bf16_conv.hip::bwgrad_kernel) ran these layers at 0.15 of their roofline: 16 x 16 x 32 tiles with 14 + 2
// operands per 28 MFMAs are LDS-read-bound (every operand is two transposing reads), its 120 KB single-buffered box leaves one
// workgroup per CU alternating between staging and computing.  Here
//   * a workgroup owns 32 contraction channels x 64 produced channels x ALL 27 taps; wave w owns taps 4 w .. 4 w + 3 (28 slots
//     for 27 taps) and a 128 x 64 block of D as 4 x 2 tiles of 32 x 32: a B operand (dz, shared by every tap) feeds four
//     MFMAs, an A operand (the tap-shifted input) two -- 12 transposing reads per 8 MFMAs of 32 cycles, against 32 per 28 of 16;
//   * operands are read straight out of [voxel][channel] images with ds_read_b64_tr_b16 (no transposed copy): the lane's
//     address picks its voxel, so the tap shift is a constant added to the address;
//   * the halo image of a 256-voxel box (32 channels) and the box of dz (64 channels) are DMA'd into one of TWO buffers while
//     the other is being multiplied (74 KB each);
//   * a workgroup walks `per` boxes and leaves one fp32 slab [27][32][64]; the slabs of a (ci, co) block are summed in slice
//     order by bdwgrad_reduce_kernel into dW (+=): bitwise reproducible.
#include <stdlib.h>

#include "bf16_common.h"
#include "buffer_stage.h"

namespace {

constexpr int DW_THREADS = 448, DW_BOXV = 256;
typedef short dw_s16x4 __attribute__((ext_vector_type(4)));
typedef float dw_f32x16 __attribute__((ext_vector_type(16)));

struct DWArgs {
  const bf16_t* S; const bf16_t* C; float* slab;
  int N, Z, Y, X;
  int s_cs, c_cs;
  int bq[3], nb[3], lbx, lby;   // box of 256 voxels: bq[2] in {8, 16}, powers of two
  int hy, hx, hvox;             // halo image: rows per plane, row length, voxels (padded to 16)
  int nboxes, per, nslices;     // boxes (all images), boxes per workgroup, workgroups per (ci, co) block
  int ncob;                     // blocks of 64 produced channels
  int toff[28];                 // LDS byte offset of tap slot t inside the halo image, relative to the voxel's own halo position
};

__device__ __forceinline__ dw_s16x4 dw_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw_s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bfx8 dw_operand(const unsigned char* p0, const unsigned char* p1) {
  const dw_s16x4 lo = dw_tr16(p0), hi = dw_tr16(p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bfx8, v);
}

__global__ __launch_bounds__(DW_THREADS) void bdwgrad_kernel(DWArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4, tq = li >> 2, tp = li & 3;
  const int cob = blockIdx.y % a.ncob, cib = blockIdx.y / a.ncob;
  const int s_bytes = a.hvox * 64, buf_bytes = s_bytes + DW_BOXV * 128;
  const int hz = a.bq[0] + 2;

  // ---- staging geometry (box independent) ----
  // S: piece q = i * 448 + tid -> halo voxel q >> 2, 16-byte piece q & 3; C: piece q -> box voxel q >> 3, piece q & 7
  constexpr int NSI = 6, NCI = 5;
  unsigned s_rel[NSI], s_pos[NSI], c_rel[NCI], c_pos[NCI];
#pragma unroll
  for (int i = 0; i < NSI; ++i) {
    const int q = i * DW_THREADS + tid, hv = q >> 2, pc = q & 3;
    const int pz = hv / (a.hy * a.hx), r2 = hv - pz * a.hy * a.hx, py = r2 / a.hx, px = r2 - py * a.hx;
    s_pos[i] = hv < hz * a.hy * a.hx ? (unsigned)(pz | (py << 8) | (px << 16)) : 0xffffffffu;
    s_rel[i] = (unsigned)(((pz * a.Y + py) * a.X + px) * a.s_cs + pc * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < NCI; ++i) {
    const int q = i * DW_THREADS + tid, cv = q >> 3, pc = q & 7;
    const int vx = cv & (a.bq[2] - 1), r2 = cv >> a.lbx, vy = r2 & (a.bq[1] - 1), vz = r2 >> a.lby;
    c_pos[i] = cv < DW_BOXV ? (unsigned)(vz | (vy << 8) | (vx << 16)) : 0xffffffffu;
    c_rel[i] = (unsigned)(((vz * a.Y + vy) * a.X + vx) * a.c_cs + pc * 8) * 2u;
  }
  const size_t vox_img = (size_t)a.Z * a.Y * a.X;
  auto stage = [&](int box, int slot) {
    int b = box;
    const int bx_ = b % a.nb[2]; b /= a.nb[2];
    const int by_ = b % a.nb[1]; b /= a.nb[1];
    const int bz_ = b % a.nb[0];
    const int n = b / a.nb[0];
    const int z0 = bz_ * a.bq[0], y0 = by_ * a.bq[1], x0 = bx_ * a.bq[2];
    const __amdgpu_buffer_rsrc_t rs = ursn_rsrc(a.S + (size_t)n * vox_img * a.s_cs, (unsigned)(vox_img * a.s_cs * 2));
    const __amdgpu_buffer_rsrc_t rc = ursn_rsrc(a.C + (size_t)n * vox_img * a.c_cs, (unsigned)(vox_img * a.c_cs * 2));
    const unsigned sbase = (unsigned)((((z0 - 1) * a.Y + (y0 - 1)) * a.X + (x0 - 1)) * a.s_cs + cib * 32) * 2u;   // (wraps for border boxes: masked below)
    const unsigned cbase = (unsigned)(((z0 * a.Y + y0) * a.X + x0) * a.c_cs + cob * 64) * 2u;
    unsigned char* sdst = lds + (size_t)slot * buf_bytes;
    unsigned char* cdst = sdst + s_bytes;
#pragma unroll
    for (int i = 0; i < NSI; ++i) {
      if ((i * 7 + wave) * 64 < a.hvox * 4) {   // whole wave instructions
        unsigned off = URSN_OOB_BYTES;
        const unsigned ps = s_pos[i];
        if (ps != 0xffffffffu) {
          const int gz = z0 - 1 + (int)(ps & 255), gy = y0 - 1 + (int)((ps >> 8) & 255), gx = x0 - 1 + (int)(ps >> 16);
          if (gz >= 0 && gz < a.Z && gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X) off = sbase + s_rel[i];
        }
        ursn_bload_lds_b128(rs, sdst + (size_t)((i * 7 + wave) * 64) * 16, off);
      }
    }
#pragma unroll
    for (int i = 0; i < NCI; ++i) {
      if ((i * 7 + wave) * 64 < DW_BOXV * 8) {
        unsigned off = URSN_OOB_BYTES;
        const unsigned pc_ = c_pos[i];
        if (pc_ != 0xffffffffu) {
          const int gz = z0 + (int)(pc_ & 255), gy = y0 + (int)((pc_ >> 8) & 255), gx = x0 + (int)(pc_ >> 16);
          if (gz < a.Z && gy < a.Y && gx < a.X) off = cbase + c_rel[i];
        }
        ursn_bload_lds_b128(rc, cdst + (size_t)((i * 7 + wave) * 64) * 16, off);
      }
    }
  };

  // ---- operand geometry: transposing read r of a lane covers voxels j = 8 (gq >> 1) + 4 r + tq of the 16-voxel k step and
  // channels 16 (gq & 1) + 4 tp .. + 3 of the 32-channel tile; the lane ends with channel 16 (gq & 1) + li of those 4 voxels ----
  const int rows_per_ks = 16 >> a.lbx;   // 16-voxel k step = 1 | 2 rows of the box
  unsigned a_lane[2], b_lane[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j = 8 * (gq >> 1) + 4 * r + tq;
    a_lane[r] = (unsigned)((((j >> a.lbx) * a.hx + (j & (a.bq[2] - 1))) * 32 + 16 * (gq & 1) + 4 * tp) * 2);
    b_lane[r] = (unsigned)((j * 64 + 16 * (gq & 1) + 4 * tp) * 2);
  }
  int tof[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) tof[k] = __builtin_amdgcn_readfirstlane(a.toff[4 * wave + k]);

  dw_f32x16 acc[4][2];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[k][nt][i] = 0.f;

  const int slice = blockIdx.x;
  const int box0 = slice * a.per;
  int box1 = box0 + a.per;
  if (box1 > a.nboxes) box1 = a.nboxes;
  if (box0 < box1) stage(box0, 0);
  for (int box = box0; box < box1; ++box) {
    const int slot = (box - box0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the box has landed
    __syncthreads();                                    // ... everyone's; and nobody reads the other buffer any more
    if (box + 1 < box1) stage(box + 1, slot ^ 1);
    const unsigned char* sb = lds + (size_t)slot * buf_bytes;
    const unsigned char* cb = sb + s_bytes;
    for (int ks = 0; ks < DW_BOXV / 16; ++ks) {
      const int r0 = ks * rows_per_ks, jy = r0 & (a.bq[1] - 1), jz = r0 >> a.lby;
      const unsigned kss = (unsigned)(((jz * a.hy + jy) * a.hx) * 64), ksc = (unsigned)(ks * 16 * 128);
      bfx8 B[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) B[nt] = dw_operand(cb + ksc + b_lane[0] + nt * 64, cb + ksc + b_lane[1] + nt * 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bfx8 A = dw_operand(sb + kss + a_lane[0] + tof[k], sb + kss + a_lane[1] + tof[k]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[nt], acc[k][nt], 0, 0, 0);
      }
    }
  }

  // ---- slab [27][32][64]: lane (column n = lane & 31, half h = lane >> 5) holds rows 8 (i >> 2) + 4 h + (i & 3) ----
  float* sl = a.slab + ((size_t)blockIdx.y * a.nslices + slice) * (size_t)(27 * 32 * 64);
  const int n32 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = 4 * wave + k;
    if (t >= 27) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) sl[(size_t)(t * 32 + 8 * (i >> 2) + 4 * h + (i & 3)) * 64 + 32 * nt + n32] = acc[k][nt][i];
  }
}

// dw[tap_w[t]][ci][co] += sum over the slices of the element's block, slice order; block = 64 consecutive elements x 4 quarters of
// the slices (threadIdx.y), the quarters are added in a fixed order
struct DWRedArgs {
  const float* slab; float* dw;
  int nslices, ncob, Kw, Nw, w_tap_stride, w_sk, w_sn, nblk;
  int tap_w[27];
};
__global__ __launch_bounds__(256) void bdwgrad_reduce_kernel(DWRedArgs a) {
  __shared__ float part[4][64];
  const int el = threadIdx.x, sl = threadIdx.y;
  const int64_t per = 27 * 32 * 64;
  const int64_t e = (int64_t)blockIdx.x * 64 + el;   // (block, tap, row, column)
  float sum = 0.f;
  const bool ok = e < (int64_t)a.nblk * per;
  int blk = 0, t = 0, row = 0, col = 0;
  if (ok) {
    blk = (int)(e / per);
    const int rc = (int)(e - (int64_t)blk * per);
    t = rc / 2048; row = (rc >> 6) & 31; col = rc & 63;
    const float* p = a.slab + (size_t)blk * a.nslices * per + rc;
    const int k0 = (a.nslices * sl) / 4, k1 = (a.nslices * (sl + 1)) / 4;
    float s0 = 0.f, s1 = 0.f;
    int k = k0;
    for (; k + 1 < k1; k += 2) { s0 += p[(size_t)k * per]; s1 += p[(size_t)(k + 1) * per]; }
    if (k < k1) s0 += p[(size_t)k * per];
    sum = s0 + s1;
  }
  part[sl][el] = sum;
  __syncthreads();
  if (sl == 0 && ok) {
    const int cib = blk / a.ncob, cob = blk - cib * a.ncob;
    const int ci = cib * 32 + row, co = cob * 64 + col;
    if (ci < a.Kw && co < a.Nw)
      a.dw[(size_t)a.tap_w[t] * a.w_tap_stride + (size_t)ci * a.w_sk + (size_t)co * a.w_sn] += (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
  }
}

struct DWPlan { int bq[3], nb[3], hy, hx, hvox, nboxes, per, nslices, ncib, ncob; size_t lds, scratch; };

bool dw_plan(const GatherGeom& g, DWPlan& p) {
  static const bool off = getenv("URSN_BDWGRAD") && getenv("URSN_BDWGRAD")[0] == '0';
  if (off) return false;
  if (g.ntaps != 27 || g.K < 32 || (g.K & 31) || g.Nn < 64 || (g.Nn & 63) || (g.in_cs & 7) || (g.out_cs & 7)) return false;
  for (int j = 0; j < 3; ++j) {
    if (g.so[j] != 1 || g.si[j] != 1 || g.po[j] != 0) return false;
    if (g.in_d[j] != g.q_d[j]) return false;
  }
  for (int t = 0; t < 27; ++t)
    for (int j = 0; j < 3; ++j)
      if (g.tap_d[t][j] < -1 || g.tap_d[t][j] > 1) return false;
  const int Z = g.in_d[0], Y = g.in_d[1], X = g.in_d[2];
  const int64_t vox = (int64_t)Z * Y * X;
  if (vox * (g.in_cs > g.out_cs ? g.in_cs : g.out_cs) * 2 >= (int64_t)0x40000000) return false;   // one buffer resource per image
  static const int64_t maxvox = getenv("URSN_BDWGRAD_MAXVOX") ? atoll(getenv("URSN_BDWGRAD_MAXVOX")) : (1 << 18);
  if ((int64_t)g.N * vox > maxvox || X < 4 || Y < 2) return false;
  p.bq[2] = X >= 16 ? 16 : 8;
  p.bq[1] = p.bq[2] == 16 ? 4 : 8;
  p.bq[0] = 4;
  for (int j = 0; j < 3; ++j) p.nb[j] = (g.in_d[j] + p.bq[j] - 1) / p.bq[j];
  p.hy = p.bq[1] + 2; p.hx = p.bq[2] + 2;
  p.hvox = ((p.bq[0] + 2) * p.hy * p.hx + 15) & ~15;
  if (p.hvox * 4 > 6 * DW_THREADS) return false;
  p.lds = 2 * ((size_t)p.hvox * 64 + DW_BOXV * 128);
  if (p.lds > 156 * 1024) return false;
  p.ncib = g.K / 32; p.ncob = g.Nn / 64;
  const int64_t boxes = (int64_t)g.N * p.nb[0] * p.nb[1] * p.nb[2];
  if (boxes < 1 || boxes > (1 << 24)) return false;
  p.nboxes = (int)boxes;
  const int nblk = p.ncib * p.ncob;
  static const int target = getenv("URSN_BDWGRAD_WGS") ? atoi(getenv("URSN_BDWGRAD_WGS")) : 256;
  int64_t ns = (target + nblk - 1) / nblk;
  if (ns < 1) ns = 1;
  if (ns > boxes) ns = boxes;
  p.per = (int)((boxes + ns - 1) / ns);
  p.nslices = (int)((boxes + p.per - 1) / p.per);
  p.scratch = (size_t)nblk * p.nslices * 27 * 32 * 64 * sizeof(float) + 256;
  return p.scratch < ((size_t)1 << 31);
}

}  // namespace

bool bdwgrad_ok(const GatherGeom& g) { DWPlan p; return dw_plan(g, p); }
size_t bdwgrad_scratch_bytes(const GatherGeom& g) { DWPlan p; return dw_plan(g, p) ? p.scratch : 0; }

int launch_bdwgrad(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch, size_t scratch_bytes,
                   hipStream_t s) {
  DWPlan p;
  URSN_REQUIRE(dw_plan(g, p), "bf16 deep weight gradient: unsupported geometry");
  URSN_REQUIRE(scratch && scratch_bytes >= p.scratch, "bf16 deep weight gradient: scratch too small");
  DWArgs a;
  a.S = S; a.C = C; a.slab = (float*)scratch;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2];
  a.s_cs = g.in_cs; a.c_cs = g.out_cs;
  for (int j = 0; j < 3; ++j) { a.bq[j] = p.bq[j]; a.nb[j] = p.nb[j]; }
  a.lbx = __builtin_ctz(p.bq[2]); a.lby = __builtin_ctz(p.bq[1]);
  a.hy = p.hy; a.hx = p.hx; a.hvox = p.hvox;
  a.nboxes = p.nboxes; a.per = p.per; a.nslices = p.nslices; a.ncob = p.ncob;
  for (int t = 0; t < 28; ++t) {
    const int tt = t < 27 ? t : 26;
    a.toff[t] = (((g.tap_d[tt][0] + 1) * p.hy + (g.tap_d[tt][1] + 1)) * p.hx + (g.tap_d[tt][2] + 1)) * 64;
  }
  static size_t attr = 48 * 1024;
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)bdwgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  ursn_note_kernel("bdwgrad_bf16<32,64>");
  hipLaunchKernelGGL(bdwgrad_kernel, dim3(p.nslices, p.ncib * p.ncob), dim3(DW_THREADS), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  DWRedArgs r;
  r.slab = a.slab; r.dw = dw; r.nslices = p.nslices; r.ncob = p.ncob; r.Kw = Kw > 0 ? Kw : g.K; r.Nw = Nw > 0 ? Nw : g.Nn;
  r.w_tap_stride = g.w_tap_stride; r.w_sk = g.w_sk; r.w_sn = g.w_sn; r.nblk = p.ncib * p.ncob;
  for (int t = 0; t < 27; ++t) r.tap_w[t] = g.tap_w[t];
  const int64_t total = (int64_t)r.nblk * 27 * 32 * 64;
  hipLaunchKernelGGL(bdwgrad_reduce_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(64, 4), 0, s, r);
  URSN_HIP(hipGetLastError());
  return 0;
}
