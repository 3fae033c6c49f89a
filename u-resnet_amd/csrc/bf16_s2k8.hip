// bf16 stride-2 GATHER passes between the two finest levels of an F = 8 network: the forward of the stride-2 conv 8 -> 16
// (lib/resnet_module.py:25-43 as called by lib/uresnet.py:56-64) -- optionally with the unit's 1x1 stride-2 shortcut conv
// (lib/resnet_module.py:25-33) in the SAME pass -- and the data gradient of the transposed conv 16 -> 8 (lib/uresnet.py:72-79).
//     out[q][co] = sum_t in[2 q + d_t][0..7] . W_t[0..7][co]            (in: the fine tensor, 8 channels; out: the coarse one, 16)
//
// The generic box kernel ran these at 0.47 of their (HBM) roofline and the shortcut re-read the fine tensor (1.07 GB at 256^3 x 4)
// in a pass of its own.  Here a workgroup marches z over a 16 x 8 tile of the COARSE grid; the fine planes (33 x 17 voxels of
// 16 bytes) travel global -> LDS by DMA into a ring of five (output plane Z reads fine planes 2 Z + dmin .. + 2, the next two are in
// flight); v_mfma_f32_16x16x32_bf16 with k = 4 taps x 8 channels: 27 taps are 7 MFMAs per 16 coarse voxels with ONE k slot idle
// -- that slot carries the shortcut's tap (the voxel 2 q itself) into a second accumulator with the shortcut's weights, so the 1x1
// conv costs one more MFMA and no memory pass.  All weights (8 fragments) stay in registers.
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

constexpr int S2_TX = 16, S2_TY = 8, S2_FX = 2 * S2_TX + 1, S2_FY = 2 * S2_TY + 1;
constexpr int S2_PIECES = S2_FX * S2_FY, S2_PLANE = ((S2_PIECES + 63) / 64) * 64 * 16, S2_NST = (S2_PIECES + 255) / 256;

struct S2KArgs {
  const bf16_t* in;        // fine tensor (N, Zf, Yf, Xf, in_cs), 8 channels read
  const bf16_t* wp;        // [8 fragments][lane][8]: 7 of the conv (k slot 27 zero), 1 of the shortcut (k slot 27 only)
  bf16_t* out;             // coarse tensor (N, Zc, Yc, Xc, out_cs), 16 channels
  bf16_t* out2;            // shortcut output (same shape, out2_cs) or null
  double* stats_partial;   // [grid][2][16] doubles (+ [grid][2][16] of the shortcut behind them) or null
  int N, Zf, Yf, Xf, Zc, Yc, Xc;
  int in_cs, out_cs, out2_cs;
  int dmin[3];
  int zseg, nzseg, nty, ntx;
  int accumulate;
  int toff[28];            // in-plane LDS byte offset of k slot s = tap (27: the shortcut's voxel 2 q) ; bits 28-29: its plane dz
};

template <bool SC, bool STATS>
__global__ __launch_bounds__(256, 2) void bs2k8_kernel(S2KArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[5 * S2_PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.ntx; b /= a.ntx;
  const int ty = b % a.nty; b /= a.nty;
  const int zs = b % a.nzseg, n = b / a.nzseg;
  const int x0 = tx * S2_TX, y0 = ty * S2_TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Zc ? z0 + a.zseg : a.Zc;

  bfx8 A[7], A2;
#pragma unroll
  for (int j = 0; j < 7; ++j) A[j] = *(const bfx8*)(a.wp + ((size_t)j * 64 + lane) * 8);
  A2 = *(const bfx8*)(a.wp + ((size_t)7 * 64 + lane) * 8);

  // staging: piece idx = i * 256 + tid of a fine plane tile; byte offset inside the plane or the out-of-range marker
  unsigned soff[S2_NST];
#pragma unroll
  for (int i = 0; i < S2_NST; ++i) {
    const int idx = i * 256 + tid, fy = idx / S2_FX, fx = idx - fy * S2_FX;
    const int gy = 2 * y0 + a.dmin[1] + fy, gx = 2 * x0 + a.dmin[2] + fx;
    soff[i] = (idx < S2_PIECES && gy >= 0 && gy < a.Yf && gx >= 0 && gx < a.Xf) ? (unsigned)((gy * a.Xf + gx) * a.in_cs) * 2u : URSN_OOB_BYTES;
  }
  const size_t fplane = (size_t)a.Yf * a.Xf * a.in_cs;
  const bf16_t* img = a.in + (size_t)n * a.Zf * fplane;
  const int pbase = 2 * z0 + a.dmin[0];   // fine plane of ring index 0
  auto dma = [&](int fp) {                // ring index fp >= 0 -> fine plane pbase + fp
    const int p = pbase + fp;
    const bool ok = p >= 0 && p < a.Zf;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(img + (size_t)(ok ? p : 0) * fplane, ok ? (unsigned)(fplane * 2) : 0u);
    unsigned char* dst = ring + (size_t)(fp % 5) * S2_PLANE + wave * 1024;
#pragma unroll
    for (int i = 0; i < S2_NST; ++i)
      if (i * 256 + wave * 64 < S2_PLANE / 16) ursn_bload_lds_b128(r, dst + i * 4096, soff[i]);
  };

  // B operand of MFMA j: this lane's k slot s = 4 j + g -> in-plane offset + plane index dz (0..2)
  unsigned lo[7];
  int ldz[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int t = a.toff[4 * j + g];
    lo[j] = (unsigned)(t & 0x0fffffff) + (unsigned)(2 * n16 * 16);
    ldz[j] = (t >> 28) & 3;
  }

  float piv[4], s1[4], s2[4], piv2[4], t1[4], t2[4], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) piv[k] = s1[k] = s2[k] = piv2[k] = t1[k] = t2[k] = 0.f;
  const size_t cplane = (size_t)a.Yc * a.Xc;

  dma(0); dma(1); dma(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int z = z0; z < z1; ++z) {
    const int k0 = 2 * (z - z0);
    if (z + 1 < z1) { dma(k0 + 3); dma(k0 + 4); }
    const unsigned sl0 = (unsigned)((k0 % 5) * S2_PLANE), sl1 = (unsigned)(((k0 + 1) % 5) * S2_PLANE), sl2 = (unsigned)(((k0 + 2) % 5) * S2_PLANE);
    unsigned base[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) base[j] = lo[j] + (ldz[j] == 0 ? sl0 : (ldz[j] == 1 ? sl1 : sl2));
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int qy = 2 * wave + rr, gy = y0 + qy, gx = x0 + n16;
      const unsigned rowoff = (unsigned)(2 * qy * S2_FX * 16);
      bf_f32x4 acc = (bf_f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const bfx8 B = *(const bfx8*)(ring + base[j] + rowoff);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j], B, acc, 0, 0, 0);
        if (SC && j == 6) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B, acc2, 0, 0, 0);
      }
      const bool ok = gy < a.Yc && gx < a.Xc;
      const size_t vox = ((size_t)n * a.Zc + z) * cplane + (size_t)gy * a.Xc + gx;
      if (ok) {
        u32x2* o = (u32x2*)(a.out + vox * a.out_cs + 4 * g);
        if (a.accumulate) {
          const u32x2 e = *o;
          acc[0] += __uint_as_float(e[0] << 16); acc[1] += __uint_as_float(e[0] & 0xffff0000u);
          acc[2] += __uint_as_float(e[1] << 16); acc[3] += __uint_as_float(e[1] & 0xffff0000u);
        }
        u32x2 pk;
        pk[0] = pack_bf2(acc[0], acc[1]);
        pk[1] = pack_bf2(acc[2], acc[3]);
        *o = pk;
        u32x2 pk2 = {0u, 0u};
        if constexpr (SC) {
          pk2[0] = pack_bf2(acc2[0], acc2[1]);
          pk2[1] = pack_bf2(acc2[2], acc2[3]);
          *(u32x2*)(a.out2 + vox * a.out2_cs + 4 * g) = pk2;
        }
        if constexpr (STATS) {
          const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                               __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
          const float rw[4] = {__uint_as_float(pk2[0] << 16), __uint_as_float(pk2[0] & 0xffff0000u),
                               __uint_as_float(pk2[1] << 16), __uint_as_float(pk2[1] & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (nacc == 0.f) { piv[r] = rv[r]; piv2[r] = rw[r]; }
            ursn_sacc(piv[r], s1[r], s2[r], rv[r]);
            if constexpr (SC) ursn_sacc(piv2[r], t1[r], t2[r], rw[r]);
          }
          nacc += 1.f;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the two planes in flight and this plane's stores)
    __syncthreads();
  }

  if constexpr (STATS) {
    __shared__ double red[4][2][32];   // [wave][conv | shortcut][sum 16 | sumsq 16]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double u, w2, u2 = 0.0, v2 = 0.0;
      ursn_sacc_final(piv[r], s1[r], s2[r], nacc, u, w2);
      if constexpr (SC) ursn_sacc_final(piv2[r], t1[r], t2[r], nacc, u2, v2);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o);
        if constexpr (SC) { u2 += __shfl_xor(u2, o); v2 += __shfl_xor(v2, o); }
      }
      if (n16 == 0) {
        red[wave][0][4 * g + r] = u; red[wave][0][16 + 4 * g + r] = w2;
        red[wave][1][4 * g + r] = u2; red[wave][1][16 + 4 * g + r] = v2;
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, e = tid & 31;
      if (which == 0 || SC)
        a.stats_partial[(size_t)which * gridDim.x * 32 + (size_t)blockIdx.x * 32 + e] =
            (red[0][which][e] + red[1][which][e]) + (red[2][which][e] + red[3][which][e]);
    }
  }
}

struct S2KPlan { int zseg, nzseg, nty, ntx, grid, dmin[3]; };
bool s2k_plan(const GatherGeom& g, S2KPlan& p) {
  static const bool off = getenv("URSN_BS2K8") && getenv("URSN_BS2K8")[0] == '0';
  if (off) return false;
  if (g.ntaps != 27 || g.K != 8 || g.Nn != 16 || (g.in_cs & 7) || (g.out_cs & 3)) return false;
  for (int j = 0; j < 3; ++j) {
    if (g.so[j] != 1 || g.si[j] != 2 || g.po[j] != 0 || g.q_d[j] != g.out_d[j]) return false;
    p.dmin[j] = 1 << 20;
    for (int t = 0; t < 27; ++t) if (g.tap_d[t][j] < p.dmin[j]) p.dmin[j] = g.tap_d[t][j];
    for (int t = 0; t < 27; ++t) if (g.tap_d[t][j] - p.dmin[j] > 2) return false;
  }
  if ((int64_t)g.in_d[1] * g.in_d[2] * g.in_cs * 2 >= (int64_t)0x40000000) return false;
  const int Zc = g.q_d[0];
  p.ntx = (g.q_d[2] + S2_TX - 1) / S2_TX;
  p.nty = (g.q_d[1] + S2_TY - 1) / S2_TY;
  if (g.q_d[2] < 8 || g.q_d[1] < 4) return false;
  const int64_t tiles = (int64_t)g.N * p.nty * p.ntx;
  int zseg = Zc;
  while (zseg > 8 && tiles * ((Zc + zseg - 1) / zseg) < 2048) zseg = (zseg + 1) / 2;
  p.zseg = zseg;
  p.nzseg = (Zc + zseg - 1) / zseg;
  if (tiles * p.nzseg > (1 << 20)) return false;
  p.grid = (int)(tiles * p.nzseg);
  return true;
}

}  // namespace

bool bs2k8_ok(const GatherGeom& g) { S2KPlan p; return s2k_plan(g, p); }
int bs2k8_grid_blocks(const GatherGeom& g) { S2KPlan p; return s2k_plan(g, p) ? p.grid : 0; }
size_t bs2k8_pack_elems() { return 8 * 64 * 8 + 8; }

// sc_w: the shortcut's weights [8][16] (stored [ci][co]) or null; out2 / out2_cs: its output; stats_partial: [grid][2][16] doubles
// for the conv followed by the same for the shortcut
int launch_bs2k8(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out, double* stats_partial,
                 int accumulate, const float* sc_w, bf16_t* out2, int out2_cs, hipStream_t s) {
  S2KPlan p;
  URSN_REQUIRE(s2k_plan(g, p), "bf16 stride-2 gather (8 -> 16): unsupported geometry");
  URSN_REQUIRE(!sc_w || (out2 && (out2_cs & 3) == 0 && !accumulate), "bf16 stride-2 gather (8 -> 16): bad shortcut arguments");
  BPackJob k = bpack_job(BPK_S2K8);
  k.w = w; k.wp = wpack; k.pw_w = sc_w; k.Kw = Kw > 0 ? Kw : g.K; k.Nw = Nw > 0 ? Nw : g.Nn;
  k.w_tap_stride = g.w_tap_stride; k.w_sk = g.w_sk; k.w_sn = g.w_sn;
  for (int t = 0; t < 27; ++t) k.tap[t] = g.tap_w[t];
  k.blocks = (8 * 64 * 8 + 255) / 256;
  URSN_TRY(bpack_submit(k, s));
  S2KArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.wp = wpack; a.out = out; a.out2 = out2; a.stats_partial = stats_partial;
  a.N = g.N; a.Zf = g.in_d[0]; a.Yf = g.in_d[1]; a.Xf = g.in_d[2]; a.Zc = g.q_d[0]; a.Yc = g.q_d[1]; a.Xc = g.q_d[2];
  a.in_cs = g.in_cs; a.out_cs = g.out_cs; a.out2_cs = out2_cs;
  for (int j = 0; j < 3; ++j) a.dmin[j] = p.dmin[j];
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.accumulate = accumulate;
  for (int t = 0; t < 28; ++t) {
    int dz = 0, dy = 0, dx = 0;   // k slot 27: the shortcut's voxel 2 q (a 1x1 stride-2 conv reads offset 0)
    if (t < 27) { dz = g.tap_d[t][0] - p.dmin[0]; dy = g.tap_d[t][1] - p.dmin[1]; dx = g.tap_d[t][2] - p.dmin[2]; }
    else { dz = -p.dmin[0]; dy = -p.dmin[1]; dx = -p.dmin[2]; }
    URSN_REQUIRE(dz >= 0 && dz <= 2 && dy >= 0 && dy <= 2 && dx >= 0 && dx <= 2, "bf16 stride-2 gather (8 -> 16): tap outside the staged planes");
    a.toff[t] = ((dy * S2_FX + dx) * 16) | (dz << 28);
  }
  ursn_note_kernel(sc_w ? "bs2k8_bf16<8,16>+sc" : "bs2k8_bf16<8,16>");
  const bool st = stats_partial != nullptr;
  if (sc_w) {
    if (st) hipLaunchKernelGGL((bs2k8_kernel<true, true>), dim3(p.grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bs2k8_kernel<true, false>), dim3(p.grid), dim3(256), 0, s, a);
  } else {
    if (st) hipLaunchKernelGGL((bs2k8_kernel<false, true>), dim3(p.grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bs2k8_kernel<false, false>), dim3(p.grid), dim3(256), 0, s, a);
  }
  URSN_HIP(hipGetLastError());
  return 0;
}
