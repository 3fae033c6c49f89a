// bf16 3x3x3 stride-1 convolution of the 8-channel levels (lib/uresnet.py:31-36 conv0, :84-100 conv1 / conv2,
// lib/resnet_module.py:43-51 at spatial level 0; forward and data gradient) -- INPUT-STATIONARY on v_mfma_f32_32x32x16_bf16.
//
// The generic bconv kernel (bf16_conv.hip) reads every (voxel, tap) operand piece from LDS once per MFMA and, with 8 produced
// channels, fills half of the MFMA's rows: at 256^3 it ran 4-5x over its HBM time, bound by LDS reads and address
// arithmetic.  Here one staged input plane p serves the THREE output planes p+1, p, p-1 at once: the MFMA rows are
// (tap plane dz = 0, 1, 2; produced channel) -- 24 of 32 -- and the 16-deep contraction is 2 in-plane taps x 8 channels, so a
// B operand piece is read from LDS once per input plane (5 reads per 32 voxels instead of 14) at a fixed per-lane address
// plus an immediate, and each input plane is staged exactly once (two LDS slots).  An output plane's partial sums move one
// row group down per input plane -- row groups are register quads of the 32x32 accumulator (row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)), so the move is 8 register copies per 32 voxels, hidden in the MFMA gaps -- and leave from group 2.
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

typedef float b3_f32x16 __attribute__((ext_vector_type(16)));

// CI, CO in {8, 16} (spatial levels 0 and 1 of an F = 8 network).  CO = 16: two 32-row tiles -- tile 0 = tap planes 0 and 1,
// tile 1 = tap plane 2 (+ 16 idle rows) -- and a 32 x 8 voxel workgroup tile (the accumulators of 32 x 16 would not fit).
template <int CI, int CO>
struct B3 {
  static constexpr int CPV = CI / 8;                    // 16-byte pieces per staged voxel
  static constexpr int TY = CO == 8 ? 16 : 8;           // tile rows; 32 columns
  static constexpr int RPW = TY / 4;                    // rows (32-voxel MFMA column blocks) per wave
  static constexpr int PX = 34, PY = TY + 2;
  static constexpr int PIECES = PX * PY * CPV;
  static constexpr int PLANE = PIECES * 16;             // bytes
  static constexpr int NST = (PIECES + 255) / 256;      // staging pieces per thread
  static constexpr int KS = CI == 8 ? 5 : 9;            // k steps of 16: two in-plane taps x 8 channels | one tap x 16
  static constexpr int MT = CO == 8 ? 1 : 2;
  static constexpr int WPACK = KS * MT * 64 * 8;        // packed weights: [k step][row tile][lane][8] bf16
};

struct B3Args {
  const bf16_t* in;
  const bf16_t* wp;
  bf16_t* out;
  double* stats_partial;   // [stats_total][2][16] doubles (the layout bconv_stats_finalize reads) or null
  int N, Z, Y, X;
  int in_cs, out_cs;
  int zseg, nzseg, nty, ntx;
  int accumulate;
  int stats_off, stats_total;
  // PW instantiation (data gradient of a module's resnet_conv1, 8 -> 16): + pw[v] . Wsc^T, the data gradient of the parallel
  // 1x1 shortcut (lib/resnet_module.py:25-33), in the otherwise idle half of the last k step
  const bf16_t* pw;
  int pw_cs;
  // BS instantiations (data gradients producing 8 channels): the BatchNorm-backward reductions of the layer(s) whose output
  // gradient this launch completes, g = stored dx * mask: partial[block][0][c] = sum g, [1][c] = sum g * xhat(z),
  // [2][c] = sum g * xhat(z2) (BS = 2).  bs_mode: 0 no mask, 1 mask = y > 0 (bs_y), 2 mask = bn(z) > 0 (bs_beta)
  const bf16_t* bs_z; const bf16_t* bs_y; const bf16_t* bs_z2; const unsigned char* bs_maskb;   // bs_mode 3: relu mask bytes
  const float* bs_mean; const float* bs_rstd; const float* bs_beta; const float* bs_mean2; const float* bs_rstd2;
  double* bs_partial;
  int bs_z_cs, bs_y_cs, bs_z2_cs, bs_mode;
  // AFF instantiations (forward, CI = CO): the input is the RAW output z of the preceding conv; its BatchNorm (+ ReLU) is
  // applied while the plane is staged (lib/resnet_module.py:43-51: resnet_conv1's BatchNorm feeds resnet_conv2 only), so that
  // activation is never written.  Padding stays zero.
  const float* aff_mean; const float* aff_rstd; const float* aff_beta;
  int aff_relu;
  // CO = 16 only: produced channels 8..15 go to a second tensor (the two halves of a concat gradient have different consumers)
  bf16_t* out2;
  int out2_cs;
  // CI = 8 only: the input is ONE fp32 channel per voxel (the network's data tensor, lib/uresnet.py:31-36); staged as
  // (bf16(value), 0 x 7) -- no 8-channel bf16 copy of the input exists
  const float* in_f32;
  // accumulate with res != null (data gradient of a unit's resnet_conv1 behind an IDENTITY shortcut, lib/resnet_module.py:52-66):
  // out = conv + res * mask instead of out += conv -- res = the gradient of the unit's output, mask = the join's relu mask bytes
  // (bit j of byte [voxel * CO / 8 + cb] = channel 8 cb + j), i.e. the residual branch's share of d(input), which the join's
  // BatchNorm backward then does not have to write
  const bf16_t* res; int res_cs; const unsigned char* res_mask;
};

// DMA: the planes travel global -> LDS by LDS-DMA (global_load_lds, no staging registers) into a ring of FOUR slots, three
// planes in flight per workgroup.  A plane iteration is memory-latency-bound (its MFMA block is ~0.14 of ~0.5 us per plane at
// 256^3): with one register-staged plane ahead and three workgroups per CU only ~29 KB were in flight per CU, half of what
// 31 GB/s per CU x ~2 us of loaded HBM latency asks for (measured 2.9 - 3.9 TB/s); a second register set cost a workgroup
// per CU (URSN_B3CONV_PF2, round 2: slower).  The wait that ends an iteration counts this wave's own younger VM operations
// exactly (its stores of the plane just completed, the DMAs of the planes behind the next one), so it never drains the ring.
template <int CI, int CO, bool STATS, bool PW = false, int BS = 0, bool AFF = false, bool DMA = false>
// (BS == 1 on the DMA path -- opt-in, URSN_B3CONV_BS_DMA -- stays at two waves per SIMD: at three it spills 80 bytes per lane, and
// compiler-placed scratch traffic between a DMA and its hand-counted s_waitcnt vmcnt(N) would make that count too loose)
__global__ __launch_bounds__(256, ((BS == 1 && !DMA) || (CI == 8 && CO == 8 && BS == 0)) ? 3 : 2) void b3conv_kernel(B3Args a) {
  static_assert(!AFF || (!PW && BS == 0), "normalise-on-load: plain forward instantiations");
  static_assert(!DMA || (!AFF && !PW), "LDS-DMA staging: no transform on the way in (the BS operands belong to the epilogue)");
  using G = B3<CI, CO>;
  static_assert(!PW || (CI == 8 && CO == 16 && !STATS), "fused shortcut term: the 8 -> 16 data gradient");
  static_assert(BS == 0 || (CO == 8 && !STATS && !PW), "fused BatchNorm-backward reductions: data gradients producing 8 channels");
  constexpr int CPV = G::CPV, PX = G::PX, RPW = G::RPW, KS = G::KS, MT = G::MT, NCH = CO / 8;
  // the shortcut's plane rides behind the x plane; DMA slots are whole wave instructions (64 pieces each) long
  constexpr int PWPLANE = PW ? G::PX * G::PY * 16 : 0, SLOT = DMA ? G::NST * 4096 : G::PLANE + PWPLANE,
                NSLOT = !DMA ? 2 : (CI == 16 && CO == 8) ? 3 : 4;   // 16 -> 8: 20 KB planes, three slots keep two workgroups per CU
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // neighbouring tiles (shared halos) on one XCD's L2
  const int tx = bid % a.ntx;
  int r_ = bid / a.ntx;
  const int ty = r_ % a.nty;
  r_ /= a.nty;
  const int zs = r_ % a.nzseg, n = r_ / a.nzseg;
  const int x0 = tx * 32, y0 = ty * G::TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;

  bfx8 A[KS][MT];
#pragma unroll
  for (int m = 0; m < KS; ++m)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) A[m][mt] = *(const bfx8*)(a.wp + ((size_t)((m * MT + mt) * 64 + lane)) * 8);

  // staging geometry of this thread's pieces of a plane: fixed over z.  Byte offsets inside the plane (fp32 scalar input:
  // half the byte offset); a piece outside the image carries URSN_OOB_BYTES and reads as zeros through the buffer bounds
  // check (buffer_stage.h) -- no select, no 64-bit address per piece, and this kernel is bound by instruction issue
  unsigned sob[G::NST];
  unsigned sval = 0;
#pragma unroll
  for (int i = 0; i < G::NST; ++i) {
    const int idx = tid + 256 * i;
    sob[i] = URSN_OOB_BYTES;
    if (idx < G::PIECES) {
      const int vi = idx / CPV, hp = idx - vi * CPV;
      const int yy = vi / PX, xx = vi - yy * PX;
      const int gy = y0 + yy - 1, gx = x0 + xx - 1;
      if (gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X) {
        sval |= 1u << i;
        sob[i] = (unsigned)((gy * a.X + gx) * a.in_cs + hp * 8) * 2u;
      }
    }
    asm volatile("" : "+v"(sob[i]));
  }
  const size_t in_plane = (size_t)a.Y * a.X * a.in_cs;          // elements
  const unsigned in_plane_bytes = (unsigned)in_plane * 2u;
  const bf16_t* in_img = a.in + (size_t)n * a.Z * in_plane;
  u32x4 st[G::NST], stpw = {0u, 0u, 0u, 0u};
  unsigned stin = 0;   // AFF: which staged pieces are real voxels
  float asc[8], ash[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    asc[j] = 1.f; ash[j] = 0.f;
    if constexpr (AFF) {   // a thread always stages the same channel half (256 is even)
      const int ch = (CI == 16 ? (tid & 1) * 8 : 0) + j;
      asc[j] = a.aff_rstd[ch];
      ash[j] = fmaf(-a.aff_mean[ch], asc[j], a.aff_beta[ch]);   // ONE rounding: the staged value is bf16(fma(z, r, fma(-mu, r, beta)))
    }
  }
  // shortcut plane (PW): interior voxels only, one piece per thread (32 x 8 tile), kept at the x plane's coordinates
  const int pwy = tid >> 5, pwx = tid & 31;
  const bool pwok = PW && y0 + pwy < a.Y && x0 + pwx < a.X;
  const unsigned pwob = pwok ? (unsigned)(((y0 + pwy) * a.X + x0 + pwx) * a.pw_cs) * 2u : URSN_OOB_BYTES;
  auto stage_dma = [&](int p, int slot) {   // plane p -> LDS slot, straight from global memory; padding arrives as zeros
    const bool pz = p >= 0 && p < a.Z;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(in_img + (ptrdiff_t)p * (ptrdiff_t)in_plane, pz ? in_plane_bytes : 0u);
    unsigned char* dst = lds + slot * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < G::NST; ++i) ursn_bload_lds_b128(r, dst + i * 4096, sob[i]);
  };
  auto stage_load = [&](int p, u32x4 (&arr)[G::NST]) {
    const bool pz = p >= 0 && p < a.Z;
    if (CI == 8 && a.in_f32) {   // scalar fp32 input (in_cs = 1): uniform branch
      const __amdgpu_buffer_rsrc_t r = ursn_rsrc(a.in_f32 + ((ptrdiff_t)n * a.Z + p) * (ptrdiff_t)a.Y * a.X, pz ? (unsigned)a.Y * a.X * 4u : 0u);
#pragma unroll
      for (int i = 0; i < G::NST; ++i) {
        unsigned cv = (unsigned)f2bf(__uint_as_float(ursn_bload_b32(r, sob[i] * 2u)));
        asm volatile("" : "+v"(cv));   // keeps the vectoriser from pairing the conversions (it then fails to select the build_vector)
        arr[i] = (u32x4){cv, 0u, 0u, 0u};
      }
    } else {
      const __amdgpu_buffer_rsrc_t r = ursn_rsrc(in_img + (ptrdiff_t)p * (ptrdiff_t)in_plane, pz ? in_plane_bytes : 0u);
#pragma unroll
      for (int i = 0; i < G::NST; ++i) arr[i] = ursn_bload_b128(r, sob[i]);
    }
    if constexpr (AFF) stin = pz ? sval : 0u;
    if constexpr (PW) {
      const __amdgpu_buffer_rsrc_t r = ursn_rsrc(a.pw + ((ptrdiff_t)n * a.Z + p) * (ptrdiff_t)a.Y * a.X * a.pw_cs,
                                                 pz ? (unsigned)a.Y * a.X * a.pw_cs * 2u : 0u);
      stpw = ursn_bload_b128(r, pwob);
    }
  };
  auto stage_store = [&](int slot, u32x4 (&arr)[G::NST]) {
#pragma unroll
    for (int i = 0; i < G::NST; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (AFF) {   // applied at the store: the loads stay in flight during the MFMA block
        if ((stin >> i) & 1u) {
          float f[8];
          unpack8(arr[i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] = fmaf(f[j], asc[j], ash[j]);
            if (a.aff_relu) f[j] = fmaxf(f[j], 0.f);
          }
          arr[i] = pack8(f);
        }
      }
      if (idx < G::PIECES) *(u32x4*)(lds + slot * SLOT + idx * 16) = arr[i];
    }
    if constexpr (PW) *(u32x4*)(lds + slot * SLOT + G::PLANE + ((pwy + 1) * PX + pwx + 1) * 16) = stpw;
  };

  // B operand: lane (column c = voxel, k half h) reads in-plane tap 2 m + h (CI = 8) | channel half h of tap m (CI = 16)
  unsigned bm[KS];
#pragma unroll
  for (int m = 0; m < KS; ++m) {
    int t = CI == 8 ? 2 * m + h : m;
    if (t > 8) t = 8;
    bm[m] = (unsigned)(((((t / 3) + RPW * wave) * PX + (t % 3) + c) * CPV + (CI == 16 ? h : 0)) * 16);
    if (PW && m == KS - 1 && h == 1) bm[m] = (unsigned)(G::PLANE + ((1 + RPW * wave) * PX + 1 + c) * 16);   // the shortcut's voxel
  }

  b3_f32x16 acc[RPW][MT];   // tap-plane groups as register blocks: partial sums of output plane p + 1 - group
#pragma unroll
  for (int nt = 0; nt < RPW; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;
  float piv[4 * NCH], s1[4 * NCH], s2[4 * NCH], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 4 * NCH; ++k) piv[k] = s1[k] = s2[k] = 0.f;
  // BS: this lane's four channels (4 h ..); per-lane sums in fp32 (<= zseg x 4 terms of bf16 data), fp64 across lanes
  float bmu[4], brs[4], bsh[4], bmu2[4], brs2[4], bg[4], bgx[4], bgx2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bmu[k] = brs[k] = bsh[k] = bmu2[k] = brs2[k] = bg[k] = bgx[k] = bgx2[k] = 0.f;
    if constexpr (BS != 0) {
      bmu[k] = a.bs_mean[4 * h + k]; brs[k] = a.bs_rstd[4 * h + k];
      if (a.bs_mode == 2) bsh[k] = a.bs_beta[4 * h + k] - bmu[k] * brs[k];
      if constexpr (BS == 2) { bmu2[k] = a.bs_mean2[4 * h + k]; brs2[k] = a.bs_rstd2[4 * h + k]; }
    }
  }

  // DMA + accumulate: the old values of the plane that completes at the end of an iteration are requested at its start, in
  // front of that iteration's DMA (read inside the epilogue they would make it wait for every DMA in flight)
  // The lane's voxel of row nt inside a z plane (Y X for a lane outside the image: every offset built from it is out of
  // range, so its loads read 0 and its stores are dropped -- no bounds branch, no 64-bit address per row and plane)
  unsigned vrow[RPW];
  unsigned rowok = 0;
#pragma unroll
  for (int nt = 0; nt < RPW; ++nt) {
    const int gy = y0 + RPW * wave + nt, gx = x0 + c;
    const bool ok = gy < a.Y && gx < a.X;
    vrow[nt] = ok ? (unsigned)(gy * a.X + gx) : (unsigned)(a.Y * a.X);
    if (ok) rowok |= 1u << nt;
    asm volatile("" : "+v"(vrow[nt]));
  }
  const size_t plane_vox = (size_t)a.Y * a.X;
  const unsigned out_plane_bytes = (unsigned)plane_vox * a.out_cs * 2u;
  const unsigned out2_plane_bytes = (CO == 16 && a.out2) ? (unsigned)plane_vox * a.out2_cs * 2u : 0u;
  auto out_rsrc = [&](int q) { return ursn_rsrc(a.out + ((size_t)n * a.Z + q) * plane_vox * a.out_cs, out_plane_bytes); };
  auto out2_rsrc = [&](int q) { return ursn_rsrc(a.out2 + ((size_t)n * a.Z + q) * plane_vox * a.out2_cs, out2_plane_bytes); };
  u32x2 oldv[RPW][NCH];
  unsigned oldm[RPW][NCH];
  const unsigned res_plane_bytes = a.res ? (unsigned)plane_vox * a.res_cs * 2u : 0u;
  auto res_rsrc = [&](int q) { return ursn_rsrc(a.res + ((size_t)n * a.Z + q) * plane_vox * a.res_cs, res_plane_bytes); };
  auto resm_rsrc = [&](int q) { return ursn_rsrc(a.res_mask + ((size_t)n * a.Z + q) * plane_vox * NCH, (unsigned)plane_vox * NCH); };
  // the value a lane adds to its four produced channels: the tensor's old value, or the masked residual gradient
  auto old_fetch = [&](const __amdgpu_buffer_rsrc_t& ro, const __amdgpu_buffer_rsrc_t& rres, const __amdgpu_buffer_rsrc_t& rmask, int q,
                       int nt, int cb, u32x2& v, unsigned& m) {
    m = 0xffu;
    if (a.res) {
      v = ursn_bload_b64(rres, vrow[nt] * (unsigned)(a.res_cs * 2) + h * 8 + 16 * cb);
      m = ursn_bload_u8(rmask, vrow[nt] * (unsigned)NCH + cb);
    } else if (CO == 16 && cb == 1 && a.out2) v = ursn_bload_b64(out2_rsrc(q), vrow[nt] * (unsigned)(a.out2_cs * 2) + h * 8);
    else v = ursn_bload_b64(ro, vrow[nt] * (unsigned)(a.out_cs * 2) + h * 8 + 16 * cb);
  };
  auto old_load = [&](int q) {
    const __amdgpu_buffer_rsrc_t ro = out_rsrc(q);
    const __amdgpu_buffer_rsrc_t rres = a.res ? res_rsrc(q) : ro, rmask = a.res ? resm_rsrc(q) : ro;
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt) {
#pragma unroll
      for (int cb = 0; cb < NCH; ++cb) old_fetch(ro, rres, rmask, q, nt, cb, oldv[nt][cb], oldm[nt][cb]);
    }
  };
  auto plane_step = [&](int p, int slot) {
    const unsigned char* L = lds + slot * SLOT;
    // BS: everything the epilogue of output plane p - 1 reads is requested here, a whole MFMA block ahead (loaded inside the
    // epilogue the latency sat between the MFMAs and the plane barrier: measured slower than the separate reduce pass)
    u32x2 pz[BS ? RPW : 1], pz2[BS == 2 ? RPW : 1];
    unsigned pm[BS ? RPW : 1];
    if constexpr (BS != 0) {
      const int q = p - 1;
      if (q >= z0 && q < z1) {
        const size_t pv = ((size_t)n * a.Z + q) * plane_vox;
        const __amdgpu_buffer_rsrc_t rz = ursn_rsrc(a.bs_z + pv * a.bs_z_cs, (unsigned)plane_vox * a.bs_z_cs * 2u);
#pragma unroll
        for (int nt = 0; nt < RPW; ++nt) pz[nt] = ursn_bload_b64(rz, vrow[nt] * (unsigned)(a.bs_z_cs * 2) + 8 * h);
        if constexpr (BS == 2) {
          const __amdgpu_buffer_rsrc_t rz2 = ursn_rsrc(a.bs_z2 + pv * a.bs_z2_cs, (unsigned)plane_vox * a.bs_z2_cs * 2u);
#pragma unroll
          for (int nt = 0; nt < RPW; ++nt) pz2[nt] = ursn_bload_b64(rz2, vrow[nt] * (unsigned)(a.bs_z2_cs * 2) + 8 * h);
        }
        if (a.bs_mode == 3) {
          const __amdgpu_buffer_rsrc_t rm = ursn_rsrc(a.bs_maskb + pv, (unsigned)plane_vox);
#pragma unroll
          for (int nt = 0; nt < RPW; ++nt) pm[nt] = ursn_bload_u8(rm, vrow[nt]);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < RPW; ++nt) {
      b3_f32x16 cc[MT];
      // the plane that was p + 1 is now p, p becomes p - 1, a new one starts
      if constexpr (CO == 8) {
        cc[0] = acc[nt][0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          cc[0][8 + i] = acc[nt][0][4 + i];
          cc[0][4 + i] = acc[nt][0][i];
          cc[0][i] = 0.f;
        }
      } else {
        cc[1] = acc[nt][1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          cc[1][i] = acc[nt][0][8 + i];
          cc[0][8 + i] = acc[nt][0][i];
          cc[0][i] = 0.f;
        }
      }
#pragma unroll
      for (int m = 0; m < KS; ++m) {
        const bfx8 b = *(const bfx8*)(L + bm[m] + nt * (PX * CPV * 16));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) cc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][mt], b, cc[mt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = cc[mt];
    }
    const int q = p - 1;   // complete: it has seen input planes q - 1, q, q + 1
    if (q >= z0 && q < z1) {
      const __amdgpu_buffer_rsrc_t ro = out_rsrc(q);
      const __amdgpu_buffer_rsrc_t ro2 = (CO == 16 && a.out2) ? out2_rsrc(q) : ro;
#pragma unroll
      for (int nt = 0; nt < RPW; ++nt) {
        {
#pragma unroll
          for (int cb = 0; cb < NCH; ++cb) {   // channels 8 cb + 4 h + (0..3): registers 8 + i (CO = 8) | 4 cb + i of tile 1
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = CO == 8 ? acc[nt][0][8 + i] : acc[nt][1][4 * cb + i];
            const bool second = CO == 16 && cb == 1 && a.out2;
            const __amdgpu_buffer_rsrc_t rr = second ? ro2 : ro;
            const unsigned oo = second ? vrow[nt] * (unsigned)(a.out2_cs * 2) + h * 8 : vrow[nt] * (unsigned)(a.out_cs * 2) + h * 8 + 16 * cb;
            if (a.accumulate) {
              u32x2 e;
              unsigned mk;
              if constexpr (DMA) { e = oldv[nt][cb]; mk = oldm[nt][cb]; }
              else old_fetch(ro, a.res ? res_rsrc(q) : ro, a.res ? resm_rsrc(q) : ro, q, nt, cb, e, mk);
              mk >>= 4 * h;   // this lane's four channels
              v[0] += (mk & 1u) ? __uint_as_float(e[0] << 16) : 0.f; v[1] += (mk & 2u) ? __uint_as_float(e[0] & 0xffff0000u) : 0.f;
              v[2] += (mk & 4u) ? __uint_as_float(e[1] << 16) : 0.f; v[3] += (mk & 8u) ? __uint_as_float(e[1] & 0xffff0000u) : 0.f;
            }
            u32x2 pk;
            pk[0] = pack_bf2(v[0], v[1]);
            pk[1] = pack_bf2(v[2], v[3]);
            ursn_bstore_b64(pk, rr, oo);
            if (!((rowok >> nt) & 1u)) continue;   // sums below: real voxels only
            if constexpr (STATS) {   // moments of the STORED (rounded) tensor: that is what BatchNorm will normalise
              const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                   __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (nacc == 0.f) piv[4 * cb + k] = rv[k];
                ursn_sacc(piv[4 * cb + k], s1[4 * cb + k], s2[4 * cb + k], rv[k]);
              }
            }
            if constexpr (BS != 0) {   // g = the STORED gradient (what the BatchNorm backward will read) under the activation mask
              float gq[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                             __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
              const u32x2 zc = pz[nt];
              const float zf[4] = {__uint_as_float(zc[0] << 16), __uint_as_float(zc[0] & 0xffff0000u),
                                   __uint_as_float(zc[1] << 16), __uint_as_float(zc[1] & 0xffff0000u)};
              if (a.bs_mode == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (!(fmaf(zf[k], brs[k], bsh[k]) > 0.f)) gq[k] = 0.f;   // same expression as bbn_bwd
              } else if (a.bs_mode == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (!((pm[nt] >> (4 * h + k)) & 1u)) gq[k] = 0.f;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                bg[k] += gq[k];
                bgx[k] = fmaf(gq[k], (zf[k] - bmu[k]) * brs[k], bgx[k]);
              }
              if constexpr (BS == 2) {
                const u32x2 z2c = pz2[nt];
                const float z2f[4] = {__uint_as_float(z2c[0] << 16), __uint_as_float(z2c[0] & 0xffff0000u),
                                      __uint_as_float(z2c[1] << 16), __uint_as_float(z2c[1] & 0xffff0000u)};
#pragma unroll
                for (int k = 0; k < 4; ++k) bgx2[k] = fmaf(gq[k], (z2f[k] - bmu2[k]) * brs2[k], bgx2[k]);
              }
            }
          }
          if constexpr (STATS) if ((rowok >> nt) & 1u) nacc += 1.f;
        }
      }
    }
  };

  if constexpr (!DMA) {
    stage_load(z0 - 1, st);
    stage_store(0, st);
    __syncthreads();
    int slot = 0;
    for (int p = z0 - 1; p <= z1; ++p) {
      if (p < z1) stage_load(p + 1, st);
      plane_step(p, slot);
      if (p < z1) stage_store(slot ^ 1, st);
      __syncthreads();
      slot ^= 1;
    }
  } else {
    // VM operations retire in issue order.  At the end of iteration p everything up to the DMA of plane p + 1 must have landed;
    // younger than it are this wave's stores of the last AHEAD output planes (one per row and 8-channel group, none while
    // a plane lies outside the segment) and the DMAs of planes p + 2 .. p + AHEAD (NST each): exactly those may stay in flight.
    constexpr int krow = RPW * NCH;   // every row issues its stores (out-of-range ones are dropped by the bounds check, but count)
    auto wait_vm = [&](int keep) {   // s_waitcnt takes an immediate; a smaller count than necessary only waits for more
      switch (keep) {
#define B3W(n_) case n_: asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory"); break;
        B3W(0) B3W(1) B3W(2) B3W(3) B3W(4) B3W(5) B3W(6) B3W(7) B3W(8) B3W(9) B3W(10) B3W(11) B3W(12) B3W(13) B3W(14) B3W(15) B3W(16) B3W(17) B3W(18)
        B3W(19) B3W(20) B3W(21) B3W(22) B3W(23) B3W(24) B3W(25) B3W(26) B3W(27) B3W(28) B3W(29) B3W(30) B3W(31) B3W(32) B3W(33) B3W(34)
#undef B3W
        default: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
      }
    };
    constexpr int AHEAD = NSLOT - 1;   // planes in flight
    stage_dma(z0 - 1, 0);
    stage_dma(z0, 1);        // z1 >= z0 + 1: planes z0 - 1, z0, z0 + 1 always exist as iterations
    if (AHEAD > 2) stage_dma(z0 + 1, 2);
    wait_vm((AHEAD - 1) * G::NST);
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int p = z0 - 1; p <= z1; ++p) {
      if (a.accumulate && p - 1 >= z0 && p - 1 < z1) old_load(p - 1);   // older than this iteration's DMA
      // the slot behind the ring held plane p - 1: every wave left it before the last barrier
      if (p + AHEAD <= z1) stage_dma(p + AHEAD, slot == 0 ? NSLOT - 1 : slot - 1);
      plane_step(p, slot);
      if (p < z1) {
        // younger than the DMA of plane p + 1 (issued AHEAD - 1 iterations ago): the stores of output planes p - AHEAD .. p - 1
        // and the DMAs of planes p + 2 .. p + AHEAD
        // `accumulate`: the old-value loads of iterations p + 2 - AHEAD .. p (krow each, issued in front of that iteration's
        // DMA) are younger than the DMA of plane p + 1 as well.  The counts assume what the compiler emits today: ONE
        // buffer_store_b64 per (row, 8-channel group) and ONE buffer_load_b64 per old value.  More VM operations than counted
        // (a split store) would only make the wait stricter; fewer cannot happen (a 64-bit value is one instruction at most).
        if (p - AHEAD >= z0 && p + AHEAD <= z1) {   // steady state: every plane of the window exists -> a constant count
          constexpr int STEADY = AHEAD * krow + G::NST * (AHEAD - 1);
          constexpr int STEADY_ACC = STEADY + (AHEAD - 1) * krow < 63 ? STEADY + (AHEAD - 1) * krow : 63;   // a smaller count only waits for more
          static_assert(STEADY <= 63, "vmcnt immediate");
          if constexpr (BS != 0) {   // + the epilogue operand loads of iterations p + 1 - AHEAD .. p (issued behind that iteration's DMA)
            const int bsl = RPW * (1 + (BS == 2 ? 1 : 0)) + (a.bs_mode == 3 ? RPW : 0);
            wait_vm(STEADY + AHEAD * bsl + (a.accumulate ? (AHEAD - 1) * krow : 0));
          } else if (a.accumulate) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEADY_ACC) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEADY) : "memory");
        } else {
          int kst = 0, nd = 0;
#pragma unroll
          for (int k = 1; k <= AHEAD; ++k) kst += (p - k >= z0 && p - k < z1) ? krow : 0;
          if constexpr (BS != 0) {
            const int bsl = RPW * (1 + (BS == 2 ? 1 : 0)) + (a.bs_mode == 3 ? RPW : 0);
#pragma unroll
            for (int k = 1; k <= AHEAD; ++k) kst += (p - k >= z0 && p - k < z1) ? bsl : 0;
          }
#pragma unroll
          for (int k = 2; k <= AHEAD; ++k) nd += (p + k <= z1) ? 1 : 0;
          if (a.accumulate) {
#pragma unroll
            for (int k = 0; k <= AHEAD - 2; ++k) kst += (p - k - 1 >= z0 && p - k - 1 < z1) ? krow : 0;   // old_load of iteration p - k
          }
          wait_vm(kst + G::NST * nd);
        }
        __builtin_amdgcn_s_barrier();   // plane p + 1 is in LDS for every wave; nobody reads this slot any more
      }
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
  }

  if constexpr (BS != 0) {   // [block][3][8] doubles: the layout bn_bwd_final_kernel reads
    __shared__ double bred[4][24];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double u = (double)bg[k], w2 = (double)bgx[k], x2 = (double)bgx2[k];
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); x2 += __shfl_xor(x2, o); }
      if (c == 0) { bred[wave][4 * h + k] = u; bred[wave][8 + 4 * h + k] = w2; bred[wave][16 + 4 * h + k] = x2; }
    }
    __syncthreads();
    if (tid < 24) a.bs_partial[(size_t)blockIdx.x * 24 + tid] = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
  }
  if constexpr (STATS) {
    __shared__ double red[4][32];
#pragma unroll
    for (int k = 0; k < 4 * NCH; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (c == 0) {
        const int ch = 8 * (k >> 2) + 4 * h + (k & 3);
        red[wave][ch] = u;
        red[wave][16 + ch] = w2;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int ch = tid & 15;
      double t = 0.0;
      if (ch < CO) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.stats_partial[((size_t)a.stats_off + blockIdx.x) * 32 + tid] = t;
    }
  }
}

// (weight packing: BPK_B3 in bf16_pack.hip)

struct B3Plan { int zseg, nzseg, nty, ntx, grid; };
B3Plan b3_plan(const GatherGeom& g) {
  B3Plan p;
  const int Z = g.in_d[0], Y = g.in_d[1], X = g.in_d[2];
  const int TY = g.Nn == 8 ? 16 : 8;
  p.ntx = (X + 31) / 32;
  p.nty = (Y + TY - 1) / TY;
  // two halo planes per z segment: long segments, but enough workgroups to fill 256 CUs x 2 a few times over
  int zseg = Z;
  const int64_t tiles = (int64_t)g.N * p.nty * p.ntx;
  static const int64_t minwg = getenv("URSN_B3_MINWG") ? atoi(getenv("URSN_B3_MINWG")) : 2048;   // A/B
  while (zseg > 16 && tiles * ((Z + zseg - 1) / zseg) < minwg) zseg = (zseg + 1) / 2;
  static const int force_nz = getenv("URSN_B3_NZSEG") ? atoi(getenv("URSN_B3_NZSEG")) : 0;   // A/B
  if (force_nz > 0) zseg = (Z + force_nz - 1) / force_nz;
  p.zseg = zseg;
  p.nzseg = (Z + zseg - 1) / zseg;
  p.grid = (int)(tiles * p.nzseg);
  return p;
}

}  // namespace

bool b3conv_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_B3CONV") && getenv("URSN_B3CONV")[0] == '0';
  if (off) return false;
  {   // buffer-path staging (buffer_stage.h): a z plane of either tensor must stay below the out-of-range marker
    const int64_t pv = (int64_t)g.in_d[1] * g.in_d[2], qv = (int64_t)g.out_d[1] * g.out_d[2];
    const int64_t cs = g.in_cs > g.out_cs ? g.in_cs : g.out_cs;
    if ((pv > qv ? pv : qv) * cs * 2 >= (int64_t)0x40000000) return false;
  }
  if ((g.K != 8 && g.K != 16) || (g.Nn != 8 && g.Nn != 16) || g.ntaps != 27 || (g.in_cs & 7) || (g.out_cs & 7)) return false;
  for (int j = 0; j < 3; ++j) {
    if (g.so[j] != 1 || g.si[j] != 1 || g.po[j] != 0) return false;
    if (g.in_d[j] != g.out_d[j] || g.in_d[j] != g.q_d[j]) return false;
  }
  for (int t = 0; t < 27; ++t)
    for (int j = 0; j < 3; ++j)
      if (g.tap_d[t][j] < -1 || g.tap_d[t][j] > 1) return false;
  const int64_t tiles = (int64_t)g.N * ((g.in_d[1] + 7) / 8) * ((g.in_d[2] + 31) / 32) * g.in_d[0];
  if (tiles > (int64_t)1 << 30) return false;
  if ((int64_t)g.in_d[1] * g.in_d[2] * (g.in_cs > g.out_cs ? g.in_cs : g.out_cs) >= ((int64_t)1 << 31)) return false;   // int plane offsets
  return true;
}

bool b3conv_aff_ok(const GatherGeom& g) { return b3conv_ok(g) && g.K == g.Nn; }
bool b3conv_pw_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_B3CONV_PW") && getenv("URSN_B3CONV_PW")[0] == '0';
  return !off && b3conv_ok(g) && g.K == 8 && g.Nn == 16;
}
size_t b3conv_pack_elems() { return (size_t)B3<16, 16>::WPACK + 8; }
int b3conv_grid_blocks(const GatherGeom& g) { return b3_plan(g).grid; }

// Off unless URSN_BF16_FUSE_BN_BWD_REDUCE=1: measured at cfg5 (256^3 x 4) the five fused launches cost more than the five
// bbn_bwd_reduce passes they replace -- 65.1 vs 67.0 img/s with the z / y loads inside the epilogue, 75.9 vs 80.0 with the
// operands (z, z2, mask bytes) requested a whole MFMA block ahead: 168 VGPRs + 84 bytes of spills at three waves per SIMD, or
// 226 VGPRs at two, cost this latency-bound kernel more than the 16 bytes per voxel of the separate pass.  Parity-tested.
bool b3conv_bs_ok(const GatherGeom& g) {
  static const bool on = getenv("URSN_BF16_FUSE_BN_BWD_REDUCE") && getenv("URSN_BF16_FUSE_BN_BWD_REDUCE")[0] == '1';
  return on && b3conv_ok(g) && g.Nn == 8 && g.K == 8;
}

int launch_b3conv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                  double* stats_partial, int stats_off, int stats_total, hipStream_t s, const bf16_t* pw, int pw_cs,
                  const float* pw_w, const B3BnRed* bs, const B3Affine* aff, bf16_t* out2, int out2_cs, const float* in_f32,
                  const B3Residual* res) {
  URSN_REQUIRE(b3conv_ok(g), "bf16 3x3x3 conv: unsupported geometry");
  URSN_REQUIRE(!pw || (b3conv_pw_ok(g) && pw_w && !stats_partial && (pw_cs & 7) == 0), "bf16 3x3x3 conv: the fused shortcut term needs the 8 -> 16 data gradient");
  URSN_REQUIRE((!pw || ursn_bf16_plane_ok(g, pw_cs)) && (!out2 || ursn_bf16_plane_ok(g, out2_cs)) &&
               (!bs || (ursn_bf16_plane_ok(g, bs->z_cs) && ursn_bf16_plane_ok(g, bs->y_cs) && ursn_bf16_plane_ok(g, bs->z2_cs))),
               "bf16 3x3x3 conv: a z plane of an auxiliary operand (strides %d / %d) reaches the buffer path's out-of-range marker", pw_cs, out2_cs);
  BPackJob k = bpack_job(BPK_B3);
  k.pw_w = pw ? pw_w : nullptr;
  k.w = w; k.wp = wpack; k.Kw = Kw > 0 ? Kw : g.K; k.Nw = Nw > 0 ? Nw : g.Nn;
  k.w_tap_stride = g.w_tap_stride; k.w_sk = g.w_sk; k.w_sn = g.w_sn;
  for (int i = 0; i < 27; ++i) k.tap[i] = -1;
  for (int t = 0; t < g.ntaps; ++t) k.tap[(g.tap_d[t][0] + 1) * 9 + (g.tap_d[t][1] + 1) * 3 + (g.tap_d[t][2] + 1)] = g.tap_w[t];
  auto pack = [&](int ci, int co, int wpack_elems, int mt) {
    k.p[0] = ci; k.p[1] = co; k.p[2] = wpack_elems; k.p[3] = mt; k.blocks = (wpack_elems + 255) / 256;
    return bpack_submit(k, s);
  };
  const B3Plan p = b3_plan(g);
  B3Args a;
  a.in = in; a.wp = wpack; a.out = out; a.stats_partial = stats_partial;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2];
  a.in_cs = g.in_cs; a.out_cs = g.out_cs;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.accumulate = g.accumulate;
  a.stats_off = stats_off; a.stats_total = stats_total > 0 ? stats_total : p.grid;
  a.pw = pw; a.pw_cs = pw_cs;
  a.bs_partial = nullptr;
  a.aff_mean = a.aff_rstd = a.aff_beta = nullptr; a.aff_relu = 0;
  URSN_REQUIRE(!out2 || (g.Nn == 16 && (out2_cs & 7) == 0 && !stats_partial), "bf16 3x3x3 conv: a second output tensor needs 16 produced channels");
  a.out2 = out2; a.out2_cs = out2_cs;
  URSN_REQUIRE(!in_f32 || (g.K == 8 && !pw && !bs && !aff), "bf16 3x3x3 conv: the scalar fp32 input form needs the 8-channel forward kernel");
  a.in_f32 = in_f32;
  if (in_f32) a.in_cs = 1;
  a.res = nullptr; a.res_cs = 0; a.res_mask = nullptr;
  if (res) {
    URSN_REQUIRE(g.K == g.Nn && !pw && !bs && !aff && !stats_partial && !out2 && !in_f32 && res->g && res->mask && (res->cs & 3) == 0 &&
                 ursn_bf16_plane_ok(g, res->cs), "bf16 3x3x3 conv: the residual term belongs to a plain C -> C data gradient");
    a.res = res->g; a.res_cs = res->cs; a.res_mask = res->mask;
    a.accumulate = 1;
  }
  if (aff) {
    URSN_REQUIRE(b3conv_aff_ok(g) && !pw && !bs && aff->mean && aff->rstd && aff->beta, "bf16 3x3x3 conv: normalise-on-load needs a C -> C forward shape");
    a.aff_mean = aff->mean; a.aff_rstd = aff->rstd; a.aff_beta = aff->beta; a.aff_relu = aff->relu;
#define B3AFF(c_, label)                                                                                                  \
    if (g.K == c_) {                                                                                                      \
      URSN_TRY(pack(c_, c_, B3<c_, c_>::WPACK, B3<c_, c_>::MT));        \
      ursn_note_kernel(label);                                                                                            \
      if (stats_partial) hipLaunchKernelGGL((b3conv_kernel<c_, c_, true, false, 0, true>), dim3(p.grid), dim3(256), 0, s, a);  \
      else hipLaunchKernelGGL((b3conv_kernel<c_, c_, false, false, 0, true>), dim3(p.grid), dim3(256), 0, s, a);          \
    }
    B3AFF(8, "b3conv_bf16<8,8>+bn")
    B3AFF(16, "b3conv_bf16<16,16>+bn")
#undef B3AFF
    URSN_HIP(hipGetLastError());
    return 0;
  }
  if (bs) {
    URSN_REQUIRE(b3conv_bs_ok(g) && !pw && !stats_partial && bs->z && bs->mean && bs->rstd && bs->partial && bs->mode != 1 && (bs->mode != 3 || bs->maskb) &&
                 (bs->mode != 2 || bs->beta) && (!bs->z2 || (bs->mean2 && bs->rstd2)), "bf16 3x3x3 conv: bad fused BatchNorm-backward arguments");
    a.bs_z = bs->z; a.bs_y = bs->y; a.bs_z2 = bs->z2; a.bs_maskb = bs->maskb; a.bs_mean = bs->mean; a.bs_rstd = bs->rstd; a.bs_beta = bs->beta;
    a.bs_mean2 = bs->mean2; a.bs_rstd2 = bs->rstd2; a.bs_partial = bs->partial;
    a.bs_z_cs = bs->z_cs; a.bs_y_cs = bs->y_cs; a.bs_z2_cs = bs->z2_cs; a.bs_mode = bs->mode;
    URSN_TRY(pack(8, 8, B3<8, 8>::WPACK, B3<8, 8>::MT));
    ursn_note_kernel("b3conv_bf16<8,8>+bnred");
    // URSN_B3CONV_BS_DMA=1: planes through the LDS-DMA ring.  Measured at cfg5 with the fusion on: 0.91-1.53 ms per launch against
    // 0.73-1.30 on the register path (and 0.45 + 0.38 for the plain data gradient + the separate reduce pass): the epilogue's
    // arithmetic, not the latency of its operands, is what the fusion costs -- off
    static const bool bs_dma = getenv("URSN_B3CONV_BS_DMA") && getenv("URSN_B3CONV_BS_DMA")[0] == '1' &&
                               !(getenv("URSN_B3CONV_DMA") && getenv("URSN_B3CONV_DMA")[0] == '0');
    if (bs_dma) {
      if (bs->z2) hipLaunchKernelGGL((b3conv_kernel<8, 8, false, false, 2, false, true>), dim3(p.grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((b3conv_kernel<8, 8, false, false, 1, false, true>), dim3(p.grid), dim3(256), 0, s, a);
    } else if (bs->z2) hipLaunchKernelGGL((b3conv_kernel<8, 8, false, false, 2>), dim3(p.grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((b3conv_kernel<8, 8, false, false, 1>), dim3(p.grid), dim3(256), 0, s, a);
    URSN_HIP(hipGetLastError());
    return 0;
  }
  if (pw) {
    URSN_TRY(pack(8, 16, B3<8, 16>::WPACK, B3<8, 16>::MT));
    ursn_note_kernel("b3conv_bf16<8,16>+pw");
    hipLaunchKernelGGL((b3conv_kernel<8, 16, false, true>), dim3(p.grid), dim3(256), 0, s, a);
    URSN_HIP(hipGetLastError());
    return 0;
  }
  // LDS-DMA ring (three planes in flight) for every plain instantiation; URSN_B3CONV_DMA=0: one register-staged plane ahead.
  // The scalar fp32 input of conv0 is converted on the way in and keeps the register path.
  static const bool dma_off = getenv("URSN_B3CONV_DMA") && getenv("URSN_B3CONV_DMA")[0] == '0';
  // measured (256^3 x 4 / 128^3 x 4, tools/bf16_op_bench.py): forward 8 -> 8 0.81 -> 0.70 ms, 16 -> 8 1.12 -> 1.04, 16 -> 16 0.254 ->
  // 0.235; data gradients 16 -> 8 1.00 -> 0.96, 16 -> 16 0.244 -> 0.232, 8 -> 8 0.553 -> 0.572 (no statistics epilogue to hide
  // behind: stays on the register path)
  // ... unless it accumulates: on the register path the old values are read inside the epilogue (0.85 ms); the DMA path requests
  // them a whole iteration ahead
  static const bool acc_dma = !(getenv("URSN_B3CONV_ACC_DMA") && getenv("URSN_B3CONV_ACC_DMA")[0] == '0');
  const bool pf2 = !dma_off && !in_f32 && (!(g.K == 8 && g.Nn == 8 && !stats_partial) || (acc_dma && (g.accumulate || res)));
#define B3GO(ci, co, label)                                                                                              \
  if (g.K == ci && g.Nn == co) {                                                                                          \
    URSN_TRY(pack(ci, co, B3<ci, co>::WPACK, B3<ci, co>::MT));          \
    ursn_note_kernel(label);                                                                                              \
    if (pf2) {                                                                                                            \
      if (stats_partial) hipLaunchKernelGGL((b3conv_kernel<ci, co, true, false, 0, false, true>), dim3(p.grid), dim3(256), 0, s, a);  \
      else hipLaunchKernelGGL((b3conv_kernel<ci, co, false, false, 0, false, true>), dim3(p.grid), dim3(256), 0, s, a);   \
    } else if (stats_partial) hipLaunchKernelGGL((b3conv_kernel<ci, co, true>), dim3(p.grid), dim3(256), 0, s, a);        \
    else hipLaunchKernelGGL((b3conv_kernel<ci, co, false>), dim3(p.grid), dim3(256), 0, s, a);                            \
  }
  B3GO(8, 8, "b3conv_bf16<8,8>")
  B3GO(16, 8, "b3conv_bf16<16,8>")
  B3GO(8, 16, "b3conv_bf16<8,16>")
  B3GO(16, 16, "b3conv_bf16<16,16>")
#undef B3GO
  URSN_HIP(hipGetLastError());
  return 0;
}
