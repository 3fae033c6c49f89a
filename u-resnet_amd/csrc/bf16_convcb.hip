// bf16 3x3x3 stride-1 convolution with 16 / 32 contraction channels and 16 .. 64 produced channels (the residual units of
// spatial levels 1 and 2 of an F = 8 network: lib/resnet_module.py:43-66 as built by lib/uresnet.py:56-64,95-100; forward
// and data gradient) -- CHANNEL-BLOCK, INPUT-STATIONARY, Z-MARCHING on v_mfma_f32_32x32x16_bf16.
//
// The generic box kernel (bf16_conv.hip) stages a halo BOX per (box, channel chunk): 2.7x the input bytes of its 512 voxels,
// one buffer only (two do not fit beside 55 KB of weights), and every tap re-reads its B operand from LDS -- 4-5x over its
// roofline at 64^3 / 128^3.  Here a workgroup owns an 8 x 32 voxel column and walks it through z:
//   * the packed weights of its block of 32 produced channels, all 27 taps (27 | 54 KB), are DMA'd into LDS ONCE;
//   * every input plane (+ 1 halo ring in y / x: 1.33x) is DMA'd into LDS exactly once, into one of two slots, while the
//     previous plane computes (LDS-DMA through the buffer path: padding voxels are out of range and arrive as zeros);
//   * a staged input plane p feeds the THREE output planes p + 1, p, p - 1 (tap planes dz = -1, 0, +1): a B operand (32
//     voxels x 16 channels of one in-plane tap) is read from LDS once and used by three MFMAs whose A operands are the three
//     tap planes' weights; after a plane the complete accumulator set is rounded to bf16 once and stored and the other two
//     move up one role (64 register copies per wave against 108 MFMAs: hidden);
//   * MFMA rows = 32 produced channels, columns = 32 voxels of one x row; LDS images are piece-major ([16-byte channel piece]
//     [y][x]) so that a lane's B address is a per-lane base plus a compile-time immediate and 16-lane groups read 256
//     contiguous bytes (conflict-free); BatchNorm moments of the STORED tensor ride in the epilogue (per-lane pivots).
// LDS reads per MFMA: (3 A + 2 B) KB per 6 MFMAs = 0.83 KB against 2 KB in the box kernel.
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

typedef float cb_f32x16 __attribute__((ext_vector_type(16)));

template <int CI>
struct CB {
  static constexpr int CPV = CI / 8;                    // 16-byte pieces per voxel
  static constexpr int KS = CI / 16;                    // k steps of 16 channels per tap
  static constexpr int TY = 8, RPW = TY / 4;            // tile rows (32 columns); rows per wave
  static constexpr int PX = 34, PY = TY + 2;
  static constexpr int PVOX = PX * PY;                  // staged voxels per plane
  static constexpr int PIECES = PVOX * CPV;
  static constexpr int NST = (PIECES + 255) / 256;      // DMA instructions per thread per plane
  static constexpr int PLANE = NST * 256 * 16;          // bytes of one slot (padded to whole wave instructions)
  static constexpr int WBYTES = 27 * KS * 1024;         // one block of 32 produced channels: [tap][k step][lane][8] bf16
  // fused shortcut term of a data gradient (PW): + pw[v] . Wsc^T, the data gradient of the unit's parallel 1x1 shortcut
  // (lib/resnet_module.py:25-33): the shortcut's dz plane (interior voxels, piece-major) rides behind the x plane of every
  // slot and KS more A fragments behind the 27 taps
  static constexpr int PWPLANE = TY * 32 * CPV * 16;    // bytes, a multiple of 4096
  static constexpr int NSTPW = PWPLANE / 4096;          // DMA instructions per thread
  static constexpr int WPW = KS * 1024;
  static constexpr int lds(bool pw) { return 2 * (PLANE + (pw ? PWPLANE : 0)) + WBYTES + (pw ? WPW : 0); }
};

struct CBArgs {
  const bf16_t* in;
  const bf16_t* wp;        // [cout block][27 taps][KS][64 lanes][8], then one zero piece
  const bf16_t* zero;      // 16 zero bytes in device memory (source of the padding pieces)
  bf16_t* out;
  double* stats_partial;   // [grid.y][stats_total][2][32] doubles or null
  int N, Z, Y, X;
  int in_cs, out_cs, Cout;
  int zseg, nzseg, nty, ntx;
  int accumulate;
  int stats_total;
  const bf16_t* pw;        // PW: the shortcut's dz, CI channels per voxel (the data gradient's contraction channels)
  int pw_cs;
};

// One input plane (LDS slot L) against the three tap planes.  accN / accM / accO: the accumulator sets of output planes
// p + 1 (first contribution: starts from zero), p, p - 1 (last contribution).  NEW / MID / OLD: which of them exist inside the
// z segment (uniform; instantiated, not branched on, so that the MFMA stream stays one basic block).
template <int CI, bool NEW, bool MID, bool OLD, bool PW = false>
__device__ __forceinline__ void cb_plane(const unsigned char* __restrict__ L, const unsigned char* __restrict__ W, unsigned boff,
                                         unsigned aoff, cb_f32x16 (&accN)[CB<CI>::RPW], cb_f32x16 (&accM)[CB<CI>::RPW],
                                         cb_f32x16 (&accO)[CB<CI>::RPW], unsigned pwoff = 0) {
  using G = CB<CI>;
  constexpr int KS = G::KS, RPW = G::RPW;
  if constexpr (NEW) {
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) accN[nt][i] = 0.f;
  }
  // in-plane taps, software-pipelined by hand: the operands of tap t + 1 are requested before the MFMAs of tap t are issued
  // (left to itself the scheduler hoists every LDS read of the unrolled plane to the top: 500 registers of fragments)
  bfx8 A[2][3][KS], B[2][KS][RPW];
  auto fetch = [&](int buf, int t) {
    const int dy = t / 3, dx = t % 3;   // in-plane tap (dy - 1, dx - 1)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // tap plane tz of the weights multiplies input plane p into output plane p + 1 - tz
      if constexpr (NEW) A[buf][0][ks] = *(const bfx8*)(W + aoff + ((0 * 9 + t) * KS + ks) * 1024);
      if constexpr (MID) A[buf][1][ks] = *(const bfx8*)(W + aoff + ((1 * 9 + t) * KS + ks) * 1024);
      if constexpr (OLD) A[buf][2][ks] = *(const bfx8*)(W + aoff + ((2 * 9 + t) * KS + ks) * 1024);
#pragma unroll
      for (int nt = 0; nt < RPW; ++nt) B[buf][ks][nt] = *(const bfx8*)(L + boff + (((2 * ks) * G::PY + nt + dy) * G::PX + dx) * 16);
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int cur = t & 1;
    if (t + 1 < 9) fetch(cur ^ 1, t + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < RPW; ++nt) {
        if constexpr (NEW) accN[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][0][ks], B[cur][ks][nt], accN[nt], 0, 0, 0);
        if constexpr (MID) accM[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][1][ks], B[cur][ks][nt], accM[nt], 0, 0, 0);
        if constexpr (OLD) accO[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][2][ks], B[cur][ks][nt], accO[nt], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (PW && MID) {   // the shortcut's term of output plane p: its own voxel only, KS more k steps
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bfx8 Ap = *(const bfx8*)(W + G::WBYTES + aoff + ks * 1024);
#pragma unroll
      for (int nt = 0; nt < RPW; ++nt) {
        const bfx8 Bp = *(const bfx8*)(L + G::PLANE + pwoff + ((2 * ks) * G::TY * 32 + nt * 32) * 16);
        accM[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ap, Bp, accM[nt], 0, 0, 0);
      }
    }
  }
}

template <int CI, bool STATS, bool PW = false>
__global__ __launch_bounds__(256, 1) void bcbconv_kernel(CBArgs a) {
  static_assert(!(PW && STATS), "the fused shortcut term belongs to a data gradient");
  using G = CB<CI>;
  constexpr int RPW = G::RPW, PX = G::PX, PY = G::PY, CPV = G::CPV;
  constexpr int SLOT = G::PLANE + (PW ? G::PWPLANE : 0), WALL = G::WBYTES + (PW ? G::WPW : 0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const W = lds + 2 * SLOT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // neighbouring columns (shared halos) on one XCD's L2
  const int tx = bid % a.ntx;
  int r_ = bid / a.ntx;
  const int ty = r_ % a.nty;
  r_ /= a.nty;
  const int zs = r_ % a.nzseg, n = r_ / a.nzseg;
  const int x0 = tx * 32, y0 = ty * G::TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;
  const int cob = blockIdx.y;

  // ---- weights of this block of produced channels: global (L2-resident) -> LDS, once ----
  {
    const bf16_t* wsrc = a.wp + (size_t)cob * (WALL / 2);
    for (int base = 0; base < WALL / 16; base += 256)   // WALL / 16 pieces, a multiple of 64
      if (base + wave * 64 < WALL / 16)
        __builtin_amdgcn_global_load_lds((const void*)(wsrc + (size_t)(base + tid) * 8),
                                         (__attribute__((address_space(3))) void*)(W + (size_t)(base + wave * 64) * 16), 16, 0, 0);
  }

  // ---- staging geometry of this thread's pieces of a plane (fixed over z): LDS piece index i * 256 + tid = (piece, y, x) ----
  // byte offsets inside a z plane; a piece outside the image carries URSN_OOB_BYTES and arrives as zeros through the buffer
  // bounds check (buffer_stage.h): no per-piece select between the voxel and a zero piece, no 64-bit address
  unsigned srel[G::NST];
#pragma unroll
  for (int i = 0; i < G::NST; ++i) {
    const int idx = tid + 256 * i;
    srel[i] = URSN_OOB_BYTES;
    if (idx < G::PIECES) {
      const int hp = idx / G::PVOX, vi = idx - hp * G::PVOX;
      const int yy = vi / PX, xx = vi - yy * PX;
      const int gy = y0 + yy - 1, gx = x0 + xx - 1;
      if (gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X) srel[i] = (unsigned)((gy * a.X + gx) * a.in_cs + hp * 8) * 2u;
    }
    asm volatile("" : "+v"(srel[i]));
  }
  const size_t plane_stride = (size_t)a.Y * a.X * a.in_cs;
  const bf16_t* in_img = a.in + (size_t)n * a.Z * plane_stride;
  // PW: this thread's pieces of the shortcut plane (interior voxels): LDS piece index i * 256 + tid = (piece, y, x)
  unsigned prel[PW ? G::NSTPW : 1];
  if constexpr (PW) {
#pragma unroll
    for (int i = 0; i < G::NSTPW; ++i) {
      const int idx = tid + 256 * i, hp = idx / (G::TY * 32), vi = idx - hp * (G::TY * 32);
      const int gy = y0 + vi / 32, gx = x0 + (vi & 31);
      prel[i] = (gy < a.Y && gx < a.X) ? (unsigned)((gy * a.X + gx) * a.pw_cs + hp * 8) * 2u : URSN_OOB_BYTES;
      asm volatile("" : "+v"(prel[i]));
    }
  }
  auto stage = [&](int p, int slot) {   // plane p -> LDS slot; planes outside the tensor are zero planes
    const bool pz = p >= 0 && p < a.Z;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(in_img + (ptrdiff_t)p * (ptrdiff_t)plane_stride, pz ? (unsigned)plane_stride * 2u : 0u);
    unsigned char* dst = lds + slot * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < G::NST; ++i) ursn_bload_lds_b128(r, dst + i * 4096, srel[i]);
    if constexpr (PW) {   // the shortcut's dz of the same plane (used by the middle role only: planes of the segment)
      const bool pq = p >= z0 && p < z1;
      const size_t pplane = (size_t)a.Y * a.X * a.pw_cs;
      const __amdgpu_buffer_rsrc_t rp = ursn_rsrc(a.pw + ((ptrdiff_t)n * a.Z + p) * (ptrdiff_t)pplane, pq ? (unsigned)pplane * 2u : 0u);
#pragma unroll
      for (int i = 0; i < G::NSTPW; ++i) ursn_bload_lds_b128(rp, dst + G::PLANE + i * 4096, prel[i]);
    }
  };

  // B operand of lane (column c, k half h): piece 2 ks + h of voxel (row RPW * wave + nt + dy, column c + dx) of the halo plane
  const unsigned boff = (unsigned)(((h * PY + RPW * wave) * PX + c) * 16);
  const unsigned aoff = (unsigned)(lane * 16);
  const unsigned pwoff = (unsigned)(((h * G::TY + RPW * wave) * 32 + c) * 16);   // shortcut plane: piece h of this lane's voxel

  cb_f32x16 accN[RPW], accM[RPW], accO[RPW];   // output planes p + 1 (new), p (middle), p - 1 (complete after this plane)
#pragma unroll
  for (int nt = 0; nt < RPW; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) accN[nt][i] = accM[nt][i] = accO[nt][i] = 0.f;
  // per-lane BatchNorm moments: channels 8 q + 4 h + r (q = 0..3, r = 0..3) over this lane's voxels
  float piv[16], s1[16], s2[16], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) piv[k] = s1[k] = s2[k] = 0.f;

  // ---- epilogue geometry (fixed over z): this lane's voxel of row nt, its four 8-byte channel groups 8 qd + 4 h ----
  // orel: byte offset of the lane's first group inside a z plane, URSN_OOB_BYTES for a lane outside the image (its stores are
  // dropped and its loads read 0: buffer_stage.h).  Channel groups at or beyond Cout (a multiple of 8) are skipped by a
  // scalar branch; omask keeps the per-lane validity for the statistics only.
  unsigned orel[RPW];
  unsigned omask = 0;   // bit nt * 4 + qd: the lane owns that group
#pragma unroll
  for (int nt = 0; nt < RPW; ++nt) {
    const int gy = y0 + RPW * wave + nt, gx = x0 + c;
    orel[nt] = URSN_OOB_BYTES;
    if (gy < a.Y && gx < a.X) {
      orel[nt] = (unsigned)((gy * a.X + gx) * a.out_cs + cob * 32 + 4 * h) * 2u;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        if (cob * 32 + 8 * qd + 4 * h < a.Cout) omask |= 1u << (nt * 4 + qd);
    }
    asm volatile("" : "+v"(orel[nt]));
  }
  const size_t oplane = (size_t)a.Y * a.X * a.out_cs;
  const bf16_t* out_img = a.out + (size_t)n * a.Z * oplane;
  auto out_rsrc = [&](int q) { return ursn_rsrc(out_img + (ptrdiff_t)q * (ptrdiff_t)oplane, (unsigned)oplane * 2u); };
  // A complete plane is rounded to bf16 right after its last MFMA (pend) but STORED at the start of the next plane
  // iteration, in front of that iteration's DMA: the wait that ends an iteration then finds the stores a whole MFMA block old
  // instead of paying the HBM write latency every plane.  accumulate: the old values are requested a plane ahead.
  u32x2 pend[RPW][4], oldv[RPW][4];
  int pend_q = -1;
  auto flush = [&]() {   // store the pending plane, take the moments of what was stored (what BatchNorm will normalise)
    if (pend_q < 0) return;
    const __amdgpu_buffer_rsrc_t ro = out_rsrc(pend_q);
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (cob * 32 + 8 * qd >= a.Cout) continue;   // wave-uniform
        const u32x2 pk = pend[nt][qd];
        ursn_bstore_b64(pk, ro, orel[nt] + 16 * qd);
        if ((omask >> (nt * 4 + qd)) & 1u) {
          if constexpr (STATS) {
            const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                 __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (nacc == 0.f) piv[4 * qd + k] = rv[k];
              ursn_sacc(piv[4 * qd + k], s1[4 * qd + k], s2[4 * qd + k], rv[k]);
            }
          }
        }
      }
      if constexpr (STATS) { if ((omask >> (nt * 4)) & 15u) nacc += 1.f; }
    }
    pend_q = -1;
  };
  auto prefetch_old = [&](int q) {
    if (!a.accumulate) return;
    const __amdgpu_buffer_rsrc_t ro = out_rsrc(q);
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        oldv[nt][qd] = (u32x2){0u, 0u};
        if (cob * 32 + 8 * qd < a.Cout) oldv[nt][qd] = ursn_bload_b64(ro, orel[nt] + 16 * qd);
      }
  };
  auto finish = [&](int q, cb_f32x16 (&acc)[RPW]) {   // output plane q has seen its three input planes
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {   // accumulator registers 4 qd .. 4 qd + 3 = rows (produced channels) 8 qd + 4 h + (0..3)
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[nt][4 * qd + i];
        if (a.accumulate) {
          const u32x2 e = oldv[nt][qd];
          v[0] += __uint_as_float(e[0] << 16); v[1] += __uint_as_float(e[0] & 0xffff0000u);
          v[2] += __uint_as_float(e[1] << 16); v[3] += __uint_as_float(e[1] & 0xffff0000u);
        }
        pend[nt][qd][0] = pack_bf2(v[0], v[1]);
        pend[nt][qd][1] = pack_bf2(v[2], v[3]);
      }
    pend_q = q;
  };
  auto plane_done = [&]() {   // the next plane has landed; every wave is done with this slot
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // Input plane p meets tap plane tz in output plane p + 1 - tz.  First plane (z0 - 1): only the "new" set; planes z0 .. z1 - 1:
  // all three (the sets that fall outside the segment at its two ends collect values that are never stored); plane z1: only
  // the set it completes.
  stage(z0 - 1, 0);
  plane_done();
  stage(z0, 1);
  cb_plane<CI, true, false, false>(lds, W, boff, aoff, accN, accM, accO);
#pragma unroll
  for (int nt = 0; nt < RPW; ++nt) accM[nt] = accN[nt];
  plane_done();
  int slot = 1;
  for (int p = z0; p < z1; ++p) {
    flush();                                 // plane p - 2
    if (p - 1 >= z0) prefetch_old(p - 1);    // completes at the end of this iteration
    stage(p + 1, slot ^ 1);                  // lands during the MFMA block
    cb_plane<CI, true, true, true, PW>(lds + slot * SLOT, W, boff, aoff, accN, accM, accO, pwoff);
    if (p - 1 >= z0) finish(p - 1, accO);
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt) { accO[nt] = accM[nt]; accM[nt] = accN[nt]; }
    plane_done();
    slot ^= 1;
  }
  flush();
  prefetch_old(z1 - 1);
  cb_plane<CI, false, false, true>(lds + slot * SLOT, W, boff, aoff, accN, accM, accO);
  finish(z1 - 1, accO);
  flush();

  if constexpr (STATS) {
    __shared__ double red[4][64];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (c == 0) {
        const int ch = 8 * (k >> 2) + 4 * h + (k & 3);
        red[wave][ch] = u;
        red[wave][32 + ch] = w2;
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int ch = tid & 31;
      double t = 0.0;
      if (cob * 32 + ch < a.Cout) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.stats_partial[((size_t)cob * a.stats_total + blockIdx.x) * 64 + tid] = t;
    }
  }
}

// (weight packing: BPK_CB in bf16_pack.hip)

struct CBPlan { int zseg, nzseg, nty, ntx, ncob, grid; };
CBPlan cb_plan(const GatherGeom& g) {
  CBPlan p;
  const int Z = g.in_d[0], Y = g.in_d[1], X = g.in_d[2];
  p.ntx = (X + 31) / 32;
  p.nty = (Y + 7) / 8;
  p.ncob = (g.Nn + 31) / 32;
  // one workgroup per CU: segments as long as possible (two halo planes each) while every CU still gets a column
  const int64_t cols = (int64_t)g.N * p.nty * p.ntx * p.ncob;
  static const int64_t minwg = getenv("URSN_BCB_MINWG") ? atoi(getenv("URSN_BCB_MINWG")) : 256;   // A/B
  int zseg = Z;
  while (zseg > 4 && cols * ((Z + zseg - 1) / zseg) < minwg) zseg = (zseg + 1) / 2;
  p.zseg = zseg;
  p.nzseg = (Z + zseg - 1) / zseg;
  p.grid = (int)((int64_t)g.N * p.nty * p.ntx * p.nzseg);
  return p;
}

}  // namespace

bool bcbconv_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_BCB") && getenv("URSN_BCB")[0] == '0';
  if (off) return false;
  {   // buffer-path staging (buffer_stage.h): a z plane of either tensor must stay below the out-of-range marker
    const int64_t pv = (int64_t)g.in_d[1] * g.in_d[2], qv = (int64_t)g.out_d[1] * g.out_d[2];
    const int64_t cs = g.in_cs > g.out_cs ? g.in_cs : g.out_cs;
    if ((pv > qv ? pv : qv) * cs * 2 >= (int64_t)0x40000000) return false;
  }
  if ((g.K != 16 && g.K != 32) || g.Nn < 16 || (g.Nn & 7) || g.ntaps != 27 || (g.in_cs & 7) || (g.out_cs & 7)) return false;
  if (b3conv_ok(g)) return false;   // 16 -> 16 and below: all weights in registers
  for (int j = 0; j < 3; ++j) {
    if (g.so[j] != 1 || g.si[j] != 1 || g.po[j] != 0) return false;
    if (g.in_d[j] != g.out_d[j] || g.in_d[j] != g.q_d[j]) return false;
  }
  for (int t = 0; t < 27; ++t)
    for (int j = 0; j < 3; ++j)
      if (g.tap_d[t][j] < -1 || g.tap_d[t][j] > 1) return false;
  if (g.in_d[0] < 2 || g.in_d[2] < 24 || g.in_d[1] < 6) return false;   // narrow volumes idle most of an 8 x 32 tile: box kernel
  if ((int64_t)g.in_d[1] * g.in_d[2] * (g.in_cs > g.out_cs ? g.in_cs : g.out_cs) >= ((int64_t)1 << 31)) return false;   // int plane offsets
  const CBPlan p = cb_plan(g);
  return (int64_t)p.grid * p.ncob < ((int64_t)1 << 30);
}

size_t bcbconv_pack_elems(const GatherGeom& g) { return (size_t)((g.Nn + 31) / 32) * 28 * (g.K / 16) * 512 + 8; }   // incl. a fused shortcut's fragments
int bcbconv_grid_blocks(const GatherGeom& g) { return cb_plan(g).grid; }   // rows of the statistics partials PER cout block
size_t bcbconv_stats_scratch_doubles(const GatherGeom& g) { const CBPlan p = cb_plan(g); return (size_t)p.grid * p.ncob * 64; }

template <int CI, bool STATS, bool PW>
static int cb_launch(const CBPlan& p, const CBArgs& a, hipStream_t s) {
  auto kern = bcbconv_kernel<CI, STATS, PW>;
  static bool attr = false;
  if (!attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CB<CI>::lds(PW)));
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid, p.ncob), dim3(256), CB<CI>::lds(PW), s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

bool bcbconv_pw_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_BCB_PW") && getenv("URSN_BCB_PW")[0] == '0';
  return !off && bcbconv_ok(g);
}

int launch_bcbconv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                   double* stats_partial, hipStream_t s, const bf16_t* pw, int pw_cs, const float* pw_w) {
  URSN_REQUIRE(bcbconv_ok(g), "bf16 channel-block conv: unsupported geometry");
  URSN_REQUIRE(!pw || (pw_w && !stats_partial && (pw_cs & 7) == 0 && pw_cs >= g.K), "bf16 channel-block conv: bad fused shortcut arguments");
  URSN_REQUIRE(!pw || ursn_bf16_plane_ok(g, pw_cs), "bf16 channel-block conv: a z plane of the fused shortcut operand (stride %d) reaches the buffer path's out-of-range marker", pw_cs);
  const CBPlan p = cb_plan(g);
  BPackJob k = bpack_job(BPK_CB);
  k.w = w; k.wp = wpack; k.Kw = Kw > 0 ? Kw : g.K; k.Nw = Nw > 0 ? Nw : g.Nn;
  k.w_tap_stride = g.w_tap_stride; k.w_sk = g.w_sk; k.w_sn = g.w_sn;
  k.pw_w = pw ? pw_w : nullptr;
  for (int i = 0; i < 27; ++i) k.tap[i] = -1;
  // in[q + d_t] W_t lands in out[q]: the staged plane p = q + dz feeds output plane p - dz, tap plane tz = dz + 1
  for (int t = 0; t < g.ntaps; ++t) k.tap[(g.tap_d[t][0] + 1) * 9 + (g.tap_d[t][1] + 1) * 3 + (g.tap_d[t][2] + 1)] = g.tap_w[t];
  const int total = p.ncob * (27 + (pw ? 1 : 0)) * (g.K / 16) * 512;
  k.p[0] = g.K / 16; k.p[1] = p.ncob; k.blocks = (total + 255) / 256;
  URSN_TRY(bpack_submit(k, s));
  CBArgs a;
  a.in = in; a.wp = wpack; a.zero = wpack + total; a.out = out; a.stats_partial = stats_partial;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2];
  a.in_cs = g.in_cs; a.out_cs = g.out_cs; a.Cout = g.Nn;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.accumulate = g.accumulate;
  a.stats_total = p.grid;
  a.pw = pw; a.pw_cs = pw_cs;
  if (pw) {
    ursn_note_kernel(g.K == 16 ? "bcbconv_bf16<16>+pw" : "bcbconv_bf16<32>+pw");
    return g.K == 16 ? cb_launch<16, false, true>(p, a, s) : cb_launch<32, false, true>(p, a, s);
  }
  ursn_note_kernel(g.K == 16 ? "bcbconv_bf16<16>" : "bcbconv_bf16<32>");
  if (g.K == 16) return stats_partial ? cb_launch<16, true, false>(p, a, s) : cb_launch<16, false, false>(p, a, s);
  return stats_partial ? cb_launch<32, true, false>(p, a, s) : cb_launch<32, false, false>(p, a, s);
}

int bcbconv_stats_finalize(const GatherGeom& g, const double* partial, int64_t V, float eps, float* mean, float* rstd, hipStream_t s) {
  const CBPlan p = cb_plan(g);
  // partials [cout block][grid][2][32]: one finalise launch walks the blocks of 32 channels
  return launch_bn_stats_final_blocked(partial, p.grid, g.Nn, 32, 32, (size_t)p.grid * 64, V, eps, mean, rstd, s);
}
