// fp32 weight gradient of the 3^d stride-1 convolutions of the DEEPEST levels (>= 64 channels on both sides, a few thousand
// voxels: lib/resnet_module.py:43-66 at levels 4-5 under tf.gradients) -- OPERANDS STRAIGHT FROM L2, no LDS in the loop, on
// v_mfma_f32_16x16x4_f32.
//
//     dW[t][ci][co] += sum_v x[v + d_t][ci] * dz[v][co]
//
// is a GEMM with the VOXELS as the contraction and 27 x Cin x Cout outputs (7 MB at 256 channels): at 6^3 / 12^3 there are
// 864 / 6912 voxels to contract and the box kernels (wgrad_igemm.hip: halo boxes in LDS, taps split over waves, a slab per
// workgroup; conv_generic.hip at the 6-wide level) spend their time staging and reducing: 84-117 us per layer against 20-40 us
// of matrix time.  Here a workgroup owns ONE tap x 64 input channels x 64 produced channels (16 accumulator tiles) and a slice
// of the voxels; its four waves deal the slice's voxel quads round-robin.  Per quad a lane loads ONE float4 of x (its voxel
// shifted by the tap, 4 consecutive channels) and ONE float4 of dz: element j of the first feeds row tile j, element j' of the
// second column tile j' -- 2 loads for 16 MFMAs, 1 KB per wave instruction, both straight from L2 (the tensors of these levels
// fit there many times over), image borders through the buffer bounds check.  The four partial tiles are summed through LDS in
// wave order; with one slice the result is added to dW in place (every element has one owner: no atomics), else slabs are
// summed in slice order by dwgrad_reduce_kernel.
#include <stdlib.h>

#include "ursn_common.h"
#include "buffer_stage.h"

namespace {

typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
#define DW_OOB 0x80000000u

struct DWArgs {
  const float* x;
  const float* dz;
  float* dw;       // [taps][Cin][Cout]
  float* slab;     // nslice > 1: [slice][taps][Cin][Cout]
  int N, Z, Y, X;  // 2-D problems: Z = 1
  int x_cs, dz_cs, Cin, Cout;
  int ntaps, ncib, ncob, nslice;
  int qpp;         // voxel quads per z plane: ceil(Y * X / 4)
  int nsteps;      // N * Z * qpp quads in all
  int spw;         // quads per slice
};

__global__ __launch_bounds__(256, 2) void dwgrad_kernel(DWArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 4 waves x 16 tiles x 1 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, kq = lane >> 4;
  // blockIdx.x = ((slice * ntaps + tap) * ncib + cib) * ncob + cob
  int r = blockIdx.x;
  const int cob = r % a.ncob; r /= a.ncob;
  const int cib = r % a.ncib; r /= a.ncib;
  const int tap = r % a.ntaps;
  const int slice = r / a.ntaps;
  const int tz = a.ntaps == 27 ? tap / 9 - 1 : 0, ty = (tap / 3) % 3 - 1, tx = tap % 3 - 1;
  const int plane = a.Y * a.X;
  const size_t ximg = (size_t)a.Z * plane * a.x_cs, dimg = (size_t)a.Z * plane * a.dz_cs;
  const float rX = 1.0f / (float)a.X;

  dw_f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[j][c] = (dw_f32x4){0.f, 0.f, 0.f, 0.f};

  const int s0 = slice * a.spw;
  int s1 = s0 + a.spw;
  if (s1 > a.nsteps) s1 = a.nsteps;
  auto fetch = [&](int st, dw_f32x4& xa, dw_f32x4& dzb) {
    xa = (dw_f32x4){0.f, 0.f, 0.f, 0.f};
    dzb = xa;
    if (st >= s1) return;
    const int pl = st / a.qpp, q = st - pl * a.qpp;          // wave-uniform
    const int n = pl / a.Z, z = pl - n * a.Z;
    const int p = q * 4 + kq;                                // the lane's voxel inside the plane
    int y = (int)((float)p * rX);
    int xx = p - y * a.X;
    if (xx < 0) { --y; xx += a.X; } else if (xx >= a.X) { ++y; xx -= a.X; }
    const bool inpl = p < plane;
    const int zi = z + tz, yi = y + ty, xi = xx + tx;
    const bool okx = inpl && (unsigned)zi < (unsigned)a.Z && (unsigned)yi < (unsigned)a.Y && (unsigned)xi < (unsigned)a.X;
    const __amdgpu_buffer_rsrc_t rx = ursn_plane_rsrc(a.x + (size_t)n * ximg, (unsigned)(ximg * 4));
    const __amdgpu_buffer_rsrc_t rd = ursn_plane_rsrc(a.dz + (size_t)n * dimg, (unsigned)(dimg * 4));
    const unsigned xo = okx ? (unsigned)(((zi * a.Y + yi) * a.X + xi) * a.x_cs + cib * 64 + 4 * m16) * 4u : DW_OOB;
    const unsigned dofs = inpl ? (unsigned)((z * plane + p) * a.dz_cs + cob * 64 + 4 * m16) * 4u : DW_OOB;
    xa = ursn_buffer_load_f4(rx, xo);
    dzb = ursn_buffer_load_f4(rd, dofs);
  };
  // PF quads in flight per wave: with one quad ahead the loop ran at the L2 round trip (1.2 us per quad against 0.25 us of
  // MFMAs: 27 TFLOP/s); a ring of PF register pairs, each refilled right after its MFMAs
  constexpr int PF = 8;
  dw_f32x4 xa[PF], db[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) fetch(s0 + wave + 4 * i, xa[i], db[i]);
  for (int st = s0 + wave; st < s1; st += 4 * PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (st + 4 * i < s1) {   // wave-uniform; a slot past the end holds zeros
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][j], db[i][c], acc[j][c], 0, 0, 0);
        fetch(st + 4 * (i + PF), xa[i], db[i]);
      }
    }
  }

  // ---- the four waves' partial tiles through LDS, summed in wave order; wave w finishes row tile j = w ----
  {
    unsigned char* mine = lds + (size_t)(wave * 16) * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == wave) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) *(dw_f32x4*)(mine + (j * 4 + c) * 1024) = acc[j][c];
    }
  }
  __syncthreads();
  dw_f32x4 fin[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dw_f32x4 s = (dw_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      dw_f32x4 p;
      if (w == wave) {
        p = acc[0][c];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (wave == q) p = acc[q][c];
      } else {
        p = *(const dw_f32x4*)(lds + (size_t)(w * 16 + wave * 4 + c) * 1024 + lane * 16);
      }
      if (w == 0) s = p; else s += p;
    }
    fin[c] = s;
  }
  // lane (n16 = m16, kq): rows 4 kq + r of row tile j = wave -> input channel cib * 64 + 4 (4 kq + r) + wave; column n16 of column
  // tile c -> produced channel cob * 64 + 4 n16 + c: the four column tiles of a row are 4 consecutive produced channels
  float* base = (a.nslice > 1 ? a.slab + (size_t)slice * a.ntaps * a.Cin * a.Cout : a.dw) + (size_t)tap * a.Cin * a.Cout;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int ci = cib * 64 + 4 * (4 * kq + r4) + wave, co = cob * 64 + 4 * m16;
    if (ci >= a.Cin || co >= a.Cout) continue;
    dw_f32x4 v = (dw_f32x4){fin[0][r4], fin[1][r4], fin[2][r4], fin[3][r4]};
    dw_f32x4* o = (dw_f32x4*)(base + (size_t)ci * a.Cout + co);
    if (a.nslice == 1) v += *o;
    *o = v;
  }
}

// dw[i] += sum over slices (slice order)
__global__ __launch_bounds__(256) void dwgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slab, int64_t total4,
                                                            int nslice) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    dw_f32x4 s = ((const dw_f32x4*)slab)[i];
    for (int k = 1; k < nslice; ++k) s += ((const dw_f32x4*)slab)[(int64_t)k * total4 + i];
    ((dw_f32x4*)dw)[i] += s;
  }
}

struct DWPlan {
  int Z, Y, X, ntaps, ncib, ncob, nslice, qpp, nsteps, spw;
  size_t scratch;
};

bool dw_plan(const ursn_conv_desc& d, DWPlan& p) {
  static const bool off = getenv("URSN_DWGRAD") && getenv("URSN_DWGRAD")[0] == '0';
  if (off || d.dtype != 0 || (d.ndim != 2 && d.ndim != 3) || d.transposed || d.k != 3 || d.stride != 1 || d.in_split || d.in_mean) return false;
  if (d.cin < 64 || d.cout < 64 || (d.cin & 3) || (d.cout & 3)) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || (ocs & 3)) return false;
  if (d.ndim == 3) { p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2]; p.ntaps = 27; }
  else { p.Z = 1; p.Y = d.in_sp[0]; p.X = d.in_sp[1]; p.ntaps = 9; }
  const int64_t V = (int64_t)d.n * p.Z * p.Y * p.X;
  // measured against the box kernels (tools/op_bench.py; this kernel sits at ~60 TFLOP/s whatever the size -- ~40 VALU of address
  // work per 16 MFMAs and the L2 round trips of 8 quads in flight -- the box kernels climb from 16-26 at the last level to 100+):
  // 6^3 x 4 x 256 channels 0.117 -> 0.054 ms, 2-D 8^2 x 4 x 512 0.075 -> 0.030; 12^3 0.084 -> 0.103, 2-D 16^2 0.031 -> 0.040: only
  // the last level takes it
  static const int64_t maxv = getenv("URSN_DWGRAD_MAXVOX") ? atoll(getenv("URSN_DWGRAD_MAXVOX")) : 1024;
  if (V * (d.ndim == 3 ? 1 : 4) > maxv || p.X < 2) return false;
  if ((int64_t)p.Z * p.Y * p.X * (ics > ocs ? ics : ocs) * 4 >= ((int64_t)1 << 31)) return false;
  p.ncib = (d.cin + 63) / 64;
  p.ncob = (d.cout + 63) / 64;
  p.qpp = (p.Y * p.X + 3) / 4;
  p.nsteps = d.n * p.Z * p.qpp;
  // slices of the voxel quads: enough workgroups for the chip, at least ~48 quads per wave
  const int64_t tasks = (int64_t)p.ntaps * p.ncib * p.ncob;
  int ns = 1;
  while (tasks * ns < 384 && p.nsteps / (ns * 2) >= 4 * 48) ns *= 2;
  p.nslice = ns;
  p.spw = (p.nsteps + ns - 1) / ns;
  p.scratch = ns > 1 ? (size_t)ns * p.ntaps * d.cin * d.cout * sizeof(float) : 0;
  return tasks * ns < ((int64_t)1 << 30);
}

}  // namespace

int deep_wgrad_supported(const ursn_conv_desc& d) {
  DWPlan p;
  return dw_plan(d, p) ? 1 : 0;
}
size_t deep_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  DWPlan p;
  return dw_plan(d, p) ? p.scratch + 256 : 0;
}

int launch_deep_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch, size_t scratch_bytes,
                      hipStream_t s) {
  DWPlan p;
  URSN_REQUIRE(dw_plan(d, p), "deep wgrad: unsupported shape");
  URSN_REQUIRE(p.nslice == 1 || (scratch && scratch_bytes >= p.scratch), "deep wgrad: scratch too small");
  DWArgs a;
  a.x = x; a.dz = dy; a.dw = dw; a.slab = (float*)scratch;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.x_cs = d.in_cstride > 0 ? d.in_cstride : d.cin;
  a.dz_cs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.Cin = d.cin; a.Cout = d.cout;
  a.ntaps = p.ntaps; a.ncib = p.ncib; a.ncob = p.ncob; a.nslice = p.nslice;
  a.qpp = p.qpp; a.nsteps = p.nsteps; a.spw = p.spw;
  ursn_note_kernel(p.nslice > 1 ? "dwgrad+splitk" : "dwgrad");
  static bool attr = false;
  if (!attr) {   // 64 KB of dynamic LDS (the four waves' partial tiles) is above the default limit
    URSN_HIP(hipFuncSetAttribute((const void*)dwgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(dwgrad_kernel, dim3(p.nslice * p.ntaps * p.ncib * p.ncob), dim3(256), 64 * 1024, s, a);
  URSN_HIP(hipGetLastError());
  if (p.nslice > 1) {
    const int64_t total4 = (int64_t)p.ntaps * d.cin * d.cout / 4;
    const int blocks = (int)(cdiv64(total4, 256) < 2048 ? cdiv64(total4, 256) : 2048);
    hipLaunchKernelGGL(dwgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, dw, (const float*)scratch, total4, p.nslice);
    URSN_HIP(hipGetLastError());
  }
  return 0;
}
