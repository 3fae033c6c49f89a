// bf16 stride-2 SCATTER-type passes between the two finest levels of an F = 8 network in ONE launch: the forward of
// slim.conv3d_transpose k3 s2 (lib/uresnet.py:72-79, 16 -> 8 channels) and the data gradient of the stride-2 convs
// (lib/resnet_module.py:25-43, 8 -> 16 forward).  v_mfma_f32_32x32x16_bf16, input-stationary.
//
// Evaluated as gathers these passes are eight output-parity classes (conv_api.hip::build_geoms) and ran as eight launches of
// the generic kernel, each writing every second voxel along x: 16-byte pieces with 16-byte holes, i.e. partial sectors that
// the next class completes a whole-tensor pass later.  Here a workgroup owns a 32 x 8 tile of the COARSE grid and walks z:
// for a coarse voxel q the eight fine voxels 2q + r are 64 MFMA rows (parity class r, produced channel) over a contraction
// of 8 neighbours q + e (e in {0,1}^3, relative to the smallest tap offset) x 16 channels; class r simply has zero weights
// for the neighbours it does not touch (27 of 64 blocks are non-zero -- the MFMA has the time: the pass is HBM-bound).  The
// eight classes of a coarse voxel are stored back-to-back by the same wave, so the fine tensor is written in whole sectors.
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

typedef float d3_f32x16 __attribute__((ext_vector_type(16)));

constexpr int D3_TY = 8, D3_RPW = 2;                      // coarse tile 32 x 8, two rows per wave
constexpr int D3_PX = 33, D3_PY = D3_TY + 1;              // staged coarse plane: tile + one neighbour column / row
constexpr int D3_PIECES = D3_PX * D3_PY * 2;              // 16 channels = 2 pieces per voxel
constexpr int D3_PLANE = D3_PIECES * 16;
constexpr int D3_NST = (D3_PIECES + 255) / 256;
constexpr int D3_WPACK = 8 * 2 * 64 * 8;                  // [neighbour][row tile][lane][8] bf16

struct D3Args {
  const bf16_t* in;      // coarse tensor (N, Zc, Yc, Xc, in_cs), 16 channels
  const bf16_t* wp;
  bf16_t* out;           // fine tensor (N, 2 Zc, 2 Yc, 2 Xc, out_cs), 8 channels
  double* stats_partial; // [grid][2][16] doubles or null
  int N, Zc, Yc, Xc;
  int in_cs, out_cs;
  int dmin[3];           // smallest tap offset per axis (-1 | 0): neighbour e reads coarse voxel q + dmin + e
  int zseg, nzseg, nty, ntx;
  int accumulate;
};

// ACC: a.accumulate, compiled in (a run-time flag puts uniform branches with loads between the counted accesses)
template <bool STATS, bool ACC>
__global__ __launch_bounds__(256, 2) void bdeconv_kernel(D3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * D3_PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tx = bid % a.ntx;
  int r_ = bid / a.ntx;
  const int ty = r_ % a.nty;
  r_ /= a.nty;
  const int zs = r_ % a.nzseg, n = r_ / a.nzseg;
  const int x0 = tx * 32, y0 = ty * D3_TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Zc ? z0 + a.zseg : a.Zc;

  bfx8 A[8][2];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) A[e][mt] = *(const bfx8*)(a.wp + ((size_t)((e * 2 + mt) * 64 + lane)) * 8);

  // Every global access of the plane loop is a buffer instruction with a per-lane byte offset (out-of-range elements: the
  // URSN_OOB_BYTES marker, read as zeros / dropped by the bounds check) -- no lane-validity branch around any of them, so the
  // compiler counts them and waits for the staged plane with vmcnt(stores issued since), not vmcnt(0).  Round 3 had the branches:
  // each plane (32 MFMAs = 0.4 us) ended by draining its 16 output stores before the next plane's 10 KB could go to LDS.
  unsigned soffb[D3_NST];
#pragma unroll
  for (int i = 0; i < D3_NST; ++i) {
    const int idx = tid + 256 * i;
    soffb[i] = URSN_OOB_BYTES;
    if (idx < D3_PIECES) {
      const int vi = idx >> 1, hp = idx & 1;
      const int yy = vi / D3_PX, xx = vi - yy * D3_PX;
      const int gy = y0 + yy + a.dmin[1], gx = x0 + xx + a.dmin[2];
      if (gy >= 0 && gy < a.Yc && gx >= 0 && gx < a.Xc) soffb[i] = (unsigned)((gy * a.Xc + gx) * a.in_cs + hp * 8) * 2u;
    }
  }
  const size_t in_plane = (size_t)a.Yc * a.Xc * a.in_cs;
  const unsigned in_plane_bytes = (unsigned)(in_plane * 2);
  u32x4 st[D3_NST];
  auto stage_load = [&](int p) {   // coarse plane p (outside the volume or the segment's reach: zeros, no traffic)
    const bool pz = p >= 0 && p < a.Zc;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(a.in + ((size_t)n * a.Zc + (pz ? p : 0)) * in_plane, pz ? in_plane_bytes : 0u);
#pragma unroll
    for (int i = 0; i < D3_NST; ++i) st[i] = ursn_bload_b128(r, soffb[i]);
  };
  auto stage_store = [&](int slot) {
#pragma unroll
    for (int i = 0; i < D3_NST; ++i) {
      const int idx = tid + 256 * i;
      if (idx < D3_PIECES) *(u32x4*)(lds + slot * D3_PLANE + idx * 16) = st[i];
    }
  };

  // B operand of neighbour e = (ez, ey, ex): lane (column c = coarse voxel, channel half h)
  unsigned be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) be[e] = (unsigned)(((((e >> 1) + D3_RPW * wave) * D3_PX + (e & 1) + c) * 2 + h) * 16);

  float piv[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, nacc = 0.f;

  // coarse plane P lives in slot P mod 3 (P >= -1): iteration qz reads planes pl = qz + dmin_z and pl + 1.  Before the loop planes
  // pl .. pl + 2 are in LDS and pl + 3 is requested; an iteration ENDS by putting the plane requested one iteration ago into the
  // slot it has just finished with and requesting the next one.  With the staging at the end of the body the loop header sees the
  // same pending accesses from the prologue and from the back edge (three loads, youngest), so the compiler waits for them with
  // vmcnt(16): the iteration's 16 output stores stay in flight.  (Staged at the top -- round 3 -- the two paths disagree and
  // the wait is vmcnt(0): every plane, 32 MFMAs = 0.4 us, drained its stores before 10 KB went to LDS.)
  const int pz0 = z0 + a.dmin[0];
  stage_load(pz0);
  stage_store((pz0 + 3) % 3);
  stage_load(pz0 + 1);
  stage_store((pz0 + 4) % 3);
  stage_load(z0 + 1 < z1 ? pz0 + 2 : -1);
  stage_store((pz0 + 5) % 3);
  stage_load(z0 + 2 < z1 ? pz0 + 3 : -1);
  __syncthreads();
  // fine-plane resources and the lane's offsets inside a fine plane, per row of the wave
  const size_t out_plane = (size_t)(2 * a.Yc) * (2 * a.Xc) * a.out_cs;
  const unsigned out_plane_bytes = (unsigned)(out_plane * 2);
  unsigned ooff[D3_RPW];
#pragma unroll
  for (int rr = 0; rr < D3_RPW; ++rr) {
    const int gy = y0 + D3_RPW * wave + rr, gx = x0 + c;
    ooff[rr] = (gy < a.Yc && gx < a.Xc) ? (unsigned)((2 * gy * (2 * a.Xc) + 2 * gx) * a.out_cs + h * 4) * 2u : URSN_OOB_BYTES;
  }
  const unsigned oy = (unsigned)(2 * a.Xc * a.out_cs) * 2u, ox = (unsigned)a.out_cs * 2u;
  auto out_rsrc = [&](int qz_, int pz) {
    return ursn_rsrc(a.out + ((size_t)n * (2 * a.Zc) + 2 * qz_ + pz) * out_plane, out_plane_bytes);
  };
  // accumulate: the old values of a row are requested one row ahead of their use (read inside the epilogue -- a load behind every
  // store of the previous class -- the data gradient of the stride-2 conv took 0.58 ms instead of 0.26 + the 0.18 of the extra read)
  u32x2 exn[8];
  auto old_load = [&](int qz_, int rr_) {
    const __amdgpu_buffer_rsrc_t r0 = out_rsrc(qz_, 0), r1 = out_rsrc(qz_, 1);
#pragma unroll
    for (int cl = 0; cl < 8; ++cl)
      exn[cl] = ursn_bload_b64((cl >> 2) ? r1 : r0, ooff[rr_] + ((cl >> 1) & 1) * oy + (cl & 1) * ox);
  };
  if constexpr (ACC) old_load(z0, 0);
  for (int qz = z0; qz < z1; ++qz) {
    const int pl = qz + a.dmin[0];
    const unsigned sb0 = (unsigned)(((pl + 3) % 3) * D3_PLANE), sb1 = (unsigned)(((pl + 4) % 3) * D3_PLANE);
    const __amdgpu_buffer_rsrc_t ro0 = out_rsrc(qz, 0), ro1 = out_rsrc(qz, 1);
#pragma unroll
    for (int rr = 0; rr < D3_RPW; ++rr) {
      u32x2 exo[8];
#pragma unroll
      for (int cl = 0; cl < 8; ++cl) exo[cl] = exn[cl];
      if constexpr (ACC) {
        if (rr + 1 < D3_RPW) old_load(qz, rr + 1);
        else if (qz + 1 < z1) old_load(qz + 1, 0);
      }
      d3_f32x16 cc[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) cc[0][i] = cc[1][i] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bfx8 b = *(const bfx8*)(lds + ((e & 4) ? sb1 : sb0) + be[e & 3] + rr * (D3_PX * 32));
        cc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[e][0], b, cc[0], 0, 0, 0);
        cc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[e][1], b, cc[1], 0, 0, 0);
      }
      const bool lane_ok = ooff[rr] != URSN_OOB_BYTES;
#pragma unroll
      for (int cl = 0; cl < 8; ++cl) {   // class (pz, py, px) = bits of cl: register block (cl & 3) of tile cl >> 2, channels 4 h ..
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = cc[cl >> 2][(cl & 3) * 4 + i];
        if constexpr (ACC) {
          const u32x2 ex = exo[cl];
          v[0] += __uint_as_float(ex[0] << 16); v[1] += __uint_as_float(ex[0] & 0xffff0000u);
          v[2] += __uint_as_float(ex[1] << 16); v[3] += __uint_as_float(ex[1] & 0xffff0000u);
        }
        u32x2 pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        ursn_bstore_b64(pk, (cl >> 2) ? ro1 : ro0, ooff[rr] + ((cl >> 1) & 1) * oy + (cl & 1) * ox);
        if constexpr (STATS) {
          if (lane_ok) {
            const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                 __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (nacc == 0.f) piv[k] = rv[k];
              ursn_sacc(piv[k], s1[k], s2[k], rv[k]);
            }
            nacc += 1.f;
          }
        }
      }
    }
    __syncthreads();
    stage_store((pl + 3) % 3);   // plane pl + 3 (zeros past the segment) into the slot of plane pl
    stage_load(qz + 3 < z1 ? pl + 4 : -1);
  }

  if constexpr (STATS) {
    __shared__ double red[4][16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (c == 0) {
        red[wave][h * 4 + k] = u;
        red[wave][8 + h * 4 + k] = w2;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int ch = tid & 15, which = tid >> 4;
      double t = 0.0;
      if (ch < 8) t = (red[0][which * 8 + ch] + red[1][which * 8 + ch]) + (red[2][which * 8 + ch] + red[3][which * 8 + ch]);
      a.stats_partial[(size_t)blockIdx.x * 32 + tid] = t;
    }
  }
}

// (weight packing: BPK_D3 in bf16_pack.hip)

struct D3Plan { int zseg, nzseg, nty, ntx, grid, dmin[3]; };
bool d3_plan(const GatherGeom* g, int cnt, D3Plan& p) {
  static const bool off = getenv("URSN_BDECONV") && getenv("URSN_BDECONV")[0] == '0';
  if (off || cnt != 8) return false;
  const GatherGeom& g0 = g[0];
  if (g0.K != 16 || g0.Nn != 8 || (g0.in_cs & 7) || (g0.out_cs & 7)) return false;
  for (int j = 0; j < 3; ++j) {
    if (g0.out_d[j] != 2 * g0.in_d[j]) return false;   // even fine extents: every class iterates the whole coarse grid
    p.dmin[j] = 1 << 20;
  }
  for (int i = 0; i < cnt; ++i) {
    if (g[i].ntaps < 1) return false;
    for (int j = 0; j < 3; ++j) {
      if (g[i].so[j] != 2 || g[i].si[j] != 1 || g[i].q_d[j] != g0.in_d[j]) return false;
      for (int t = 0; t < g[i].ntaps; ++t)
        if (g[i].tap_d[t][j] < p.dmin[j]) p.dmin[j] = g[i].tap_d[t][j];
    }
  }
  for (int i = 0; i < cnt; ++i)
    for (int t = 0; t < g[i].ntaps; ++t)
      for (int j = 0; j < 3; ++j)
        if (g[i].tap_d[t][j] - p.dmin[j] > 1) return false;
  const int Zc = g0.in_d[0], Yc = g0.in_d[1], Xc = g0.in_d[2];
  // buffer path: a coarse plane and a fine plane stay below the out-of-range marker (and marker + an in-plane offset below 2^32)
  if ((int64_t)Yc * Xc * g0.in_cs * 2 >= (int64_t)URSN_OOB_BYTES || (int64_t)Yc * Xc * 4 * g0.out_cs * 2 >= (int64_t)URSN_OOB_BYTES) return false;
  p.ntx = (Xc + 31) / 32;
  p.nty = (Yc + D3_TY - 1) / D3_TY;
  const int64_t tiles = (int64_t)g0.N * p.nty * p.ntx;
  int zseg = Zc;
  while (zseg > 8 && tiles * ((Zc + zseg - 1) / zseg) < 2048) zseg = (zseg + 1) / 2;
  p.zseg = zseg;
  p.nzseg = (Zc + zseg - 1) / zseg;
  if (tiles * p.nzseg > (1 << 20)) return false;
  p.grid = (int)(tiles * p.nzseg);
  return true;
}

}  // namespace

bool bdeconv_ok(const GatherGeom* g, int cnt) { D3Plan p; return d3_plan(g, cnt, p); }
int bdeconv_grid_blocks(const GatherGeom* g, int cnt) { D3Plan p; return d3_plan(g, cnt, p) ? p.grid : 0; }
size_t bdeconv_pack_elems() { return (size_t)D3_WPACK + 8; }

int launch_bdeconv(const GatherGeom* g, int cnt, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                   double* stats_partial, int accumulate, hipStream_t s) {
  D3Plan p;
  URSN_REQUIRE(d3_plan(g, cnt, p), "bf16 stride-2 scatter pass: unsupported geometry");
  BPackJob k = bpack_job(BPK_D3);
  k.w = w; k.wp = wpack; k.Kw = Kw > 0 ? Kw : g[0].K; k.Nw = Nw > 0 ? Nw : g[0].Nn;
  k.w_tap_stride = g[0].w_tap_stride; k.w_sk = g[0].w_sk; k.w_sn = g[0].w_sn;
  for (int i = 0; i < 64; ++i) k.tap[i] = -1;
  for (int i = 0; i < cnt; ++i) {
    const int cl = g[i].po[0] * 4 + g[i].po[1] * 2 + g[i].po[2];
    for (int t = 0; t < g[i].ntaps; ++t) {
      const int e = (g[i].tap_d[t][0] - p.dmin[0]) * 4 + (g[i].tap_d[t][1] - p.dmin[1]) * 2 + (g[i].tap_d[t][2] - p.dmin[2]);
      k.tap[cl * 8 + e] = g[i].tap_w[t];
    }
  }
  k.p[0] = D3_WPACK; k.blocks = (D3_WPACK + 255) / 256;
  URSN_TRY(bpack_submit(k, s));
  D3Args a;
  a.in = in; a.wp = wpack; a.out = out; a.stats_partial = stats_partial;
  a.N = g[0].N; a.Zc = g[0].in_d[0]; a.Yc = g[0].in_d[1]; a.Xc = g[0].in_d[2];
  a.in_cs = g[0].in_cs; a.out_cs = g[0].out_cs;
  for (int j = 0; j < 3; ++j) a.dmin[j] = p.dmin[j];
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.accumulate = accumulate;
  ursn_note_kernel("bdeconv_bf16<16,8>");
  if (stats_partial && accumulate) hipLaunchKernelGGL((bdeconv_kernel<true, true>), dim3(p.grid), dim3(256), 0, s, a);
  else if (stats_partial) hipLaunchKernelGGL((bdeconv_kernel<true, false>), dim3(p.grid), dim3(256), 0, s, a);
  else if (accumulate) hipLaunchKernelGGL((bdeconv_kernel<false, true>), dim3(p.grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((bdeconv_kernel<false, false>), dim3(p.grid), dim3(256), 0, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}
