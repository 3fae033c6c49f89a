// bf16 convolution kernels (gfx950): ONE LDS-tiled implicit-GEMM kernel for every conv-like forward / data-gradient
// pass of the path and ONE weight-gradient kernel, both driven by the GatherGeom of conv_api.hip::build_geoms
//     out[n, q*so + po, :] (=|+=) sum_t in[n, q*si + d_t, :] . W_t            dW_t += sum_q S[q*si + d_t]^T C[q]
// (k3 s1, k3 s2, 1x1 s1|s2, transposed conv and scatter-type data gradients by output-parity class: lib/uresnet.py:37-121,
// lib/resnet_module.py:25-66).  bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// bconv: a workgroup (4 waves) owns a box of q points (BZ x BY x BX, BX in {8, 16, 32}) and one block of <= 32 produced
// channels.  Per chunk of <= 32 contraction channels the input halo box is staged ONCE in LDS as [voxel][channel] (NDHWC
// order: a lane's MFMA B operand = 8 consecutive channels of one tap of one voxel = one ds_read_b128) together with the
// chunk's weights in A-operand order; every tap re-reads the box from LDS, never from HBM.  D = W^T (M = produced
// channels) x X (N = 16 voxels along x): a lane ends with 4 consecutive channels of one voxel -> 8-byte bf16 stores.
// k index inside a chunk: slot s = tap * (chunk channels / 8) + channel block, four slots per MFMA.
// BatchNorm moment partials (per-lane pivots, fp64 re-centring: ursn_common.h) ride in the forward epilogue.
//
// bwgrad: contraction over voxels.  The S halo box and the C box are staged in the same [voxel][channel] order and read
// with ds_read_b64_tr_b16 (hardware transpose: a 16-lane group gets, per lane, one channel of 4 consecutive voxels), so no
// transposed copy of either tensor ever exists.  M rows = (tap, contraction channel), taps split over the 4 waves (no
// cross-wave sum); a workgroup walks many boxes and leaves ONE fp32 slab, slabs are summed in fixed order.
#include <stdio.h>
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"

#define BCONV_MAX_SLOTS 112

#define BCONV_PMAX 6   // in-plane staging pieces per thread (plane pieces <= 256 * BCONV_PMAX)

struct BConvArgs {
  const bf16_t* in;
  const bf16_t* wp;
  const bf16_t* zero;      // >= 16 zero bytes in device memory: source of the padding voxels of the LDS-DMA staging
  bf16_t* out;
  double* stats_partial;   // [grid.y][stats_total][2][cob] or null; this launch writes blocks stats_off + blockIdx.x
  int stats_off, stats_total;
  int N;
  int in_d[3], out_d[3], q_d[3], so[3], po[3], si[3];
  int Cin, Cout, in_cs, out_cs, accumulate;
  int bq[3], nb[3], hb[3], dmin[3];
  int cinc, nchunks, nj, cob;
  int nboxes, per;         // boxes in all; consecutive boxes (z fastest) per workgroup
  int nbuf;                // 2: the next stage is DMA'd while this one computes; 1: one buffer, latency hidden by co-resident workgroups
  int pp;                  // pieces per staged halo plane, padded to a multiple of 256
  int stage_bytes;         // one LDS input buffer: hb[0] planes of pp pieces
  int toff[BCONV_MAX_SLOTS];
};

// exact for 0 <= x < 2^22 and 1 <= d < 2^22
__device__ __forceinline__ int fdiv(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  const int r = x - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

// LDS layout: [input buffer 0][input buffer 1][weights 0][weights 1 (only when the contraction is chunked)].
// An input buffer holds the halo box of one (box, channel chunk) stage as hb[0] planes of [y][x][channel piece], each
// plane padded to a multiple of 256 pieces so that a thread stages the SAME in-plane positions of every plane: its global
// offsets and bounds are computed once per kernel, a stage costs one add per piece.
template <int VT, int COT>
__global__ __launch_bounds__(256) void bconv_kernel(BConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
  const int cpb = a.cinc >> 3;                                   // 16-byte pieces per staged voxel
  const int plane = a.hb[1] * a.hb[2] * cpb;                     // real pieces per plane
  const int wpieces = a.nj * COT * 64;
  const int co0 = blockIdx.y * a.cob;
  unsigned char* wbase = smem + (size_t)a.nbuf * a.stage_bytes;

  __shared__ int toff_s[BCONV_MAX_SLOTS];   // tap / channel-block offsets of the k slots, read per lane (slot 4 j + g)
  if (tid < BCONV_MAX_SLOTS) toff_s[tid] = tid < 4 * a.nj ? a.toff[tid] : 0;

  // ---- per-thread staging geometry (once): in-plane pieces k * 256 + tid ----
  int in_off[BCONV_PMAX];        // element offset of the piece relative to the halo plane origin (chunk offset excluded)
  int in_yx[BCONV_PMAX];         // hy | hx << 16, or -1: no such piece (pad)
  {
    const float inv_cpb = 1.0f / (float)cpb, inv_h2 = 1.0f / (float)a.hb[2];
#pragma unroll
    for (int k = 0; k < BCONV_PMAX; ++k) {
      const int idx = k * 256 + tid;
      in_off[k] = 0; in_yx[k] = -1;
      if (idx < plane) {
        const int vox = fdiv(idx, cpb, inv_cpb), cb = idx - vox * cpb;
        const int hy = fdiv(vox, a.hb[2], inv_h2), hx = vox - hy * a.hb[2];
        in_off[k] = (hy * a.in_d[2] + hx) * a.in_cs + cb * 8;
        in_yx[k] = hy | (hx << 16);
      }
    }
  }
  const int npk = a.pp >> 8;                                     // staging pieces per thread per plane
  const int64_t in_plane_stride = (int64_t)a.in_d[1] * a.in_d[2] * a.in_cs;

  // ---- per-lane compute / epilogue geometry (once) ----
  int vbase[VT];   // LDS byte offset of the input voxel under this lane's q point (tap offsets are added per k step)
  int orel[VT];    // element offset of the lane's output voxel relative to the box's first output voxel
  int qrel[VT];    // qz | qy << 8 | qx << 16 inside the box
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    // the 16 columns of an MFMA tile are 16 consecutive q points of the box in (z, y, x) order: a whole x run of a wide box,
    // two rows of an 8-wide one (the 8^3 / 16^3 levels: a 32-wide tile idled 75 % / 50 % of its lanes there)
    const int lin = (wave * VT + vt) * 16 + m, row = lin / a.bq[2], qx = lin - row * a.bq[2];
    const int qz = row / a.bq[1], qy = row % a.bq[1];
    vbase[vt] = (qz * a.si[0]) * a.pp * 16 + ((qy * a.si[1]) * a.hb[2] + qx * a.si[2]) * cpb * 16;
    orel[vt] = ((qz * a.so[0] * a.out_d[1] + qy * a.so[1]) * a.out_d[2] + qx * a.so[2]) * a.out_cs;
    qrel[vt] = qz | (qy << 8) | (qx << 16);
  }

  const int box_lo = blockIdx.x * a.per;
  int box_hi = box_lo + a.per;
  if (box_hi > a.nboxes) box_hi = a.nboxes;
  const int nstage = (box_hi > box_lo ? box_hi - box_lo : 0) * a.nchunks;
  const float inv_nb0 = 1.0f / (float)a.nb[0], inv_nb2 = 1.0f / (float)a.nb[2], inv_nb1 = 1.0f / (float)a.nb[1];
  const float inv_nch = 1.0f / (float)a.nchunks;

  // box index -> (image, first q point): z runs fastest so that consecutive boxes of a workgroup share their z halo in L2
  auto box_origin = [&](int box, int& n, int (&q0)[3]) {
    int b = box;
    int t = fdiv(b, a.nb[0], inv_nb0);
    const int bz = b - t * a.nb[0]; b = t;
    t = fdiv(b, a.nb[2], inv_nb2);
    const int bx = b - t * a.nb[2]; b = t;
    n = fdiv(b, a.nb[1], inv_nb1);
    const int by = b - n * a.nb[1];
    q0[0] = bz * a.bq[0]; q0[1] = by * a.bq[1]; q0[2] = bx * a.bq[2];
  };
  // LDS-DMA of one stage: global_load_lds_dwordx4 writes 64 consecutive 16-byte pieces per wave-instruction (destination
  // = wave-uniform base + lane * 16), the source address is per lane; voxels outside the tensor read the zero piece
  auto issue = [&](int st, int buf) {
    const int bi = fdiv(st, a.nchunks, inv_nch), ch = st - bi * a.nchunks;
    int n, q0[3];
    box_origin(box_lo + bi, n, q0);
    const int g0 = q0[0] * a.si[0] + a.dmin[0], g1 = q0[1] * a.si[1] + a.dmin[1], g2 = q0[2] * a.si[2] + a.dmin[2];
    const bool inner = g1 >= 0 && g1 + a.hb[1] <= a.in_d[1] && g2 >= 0 && g2 + a.hb[2] <= a.in_d[2];   // uniform
    const bf16_t* pl0 = a.in + ((((int64_t)n * a.in_d[0] + g0) * a.in_d[1] + g1) * a.in_d[2] + g2) * a.in_cs + ch * a.cinc;
    unsigned char* dst = smem + (size_t)buf * a.stage_bytes + (size_t)wave * 1024;
    for (int hz = 0; hz < a.hb[0]; ++hz) {
      const bool zok = (unsigned)(g0 + hz) < (unsigned)a.in_d[0];
      const bf16_t* pl = pl0 + hz * in_plane_stride;
#pragma unroll
      for (int k = 0; k < BCONV_PMAX; ++k) {
        if (k < npk) {
          const bf16_t* src = a.zero;
          const int yx = in_yx[k];
          bool ok = zok && yx >= 0;
          if (ok && !inner) {
            const int y = g1 + (yx & 0xffff), x = g2 + (yx >> 16);
            ok = (unsigned)y < (unsigned)a.in_d[1] && (unsigned)x < (unsigned)a.in_d[2];
          }
          if (ok) src = pl + in_off[k];
          __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(dst + ((size_t)hz * a.pp + k * 256) * 16), 16, 0, 0);
        }
      }
    }
    if (a.nchunks > 1 || st == 0) {   // a single chunk's weights stay resident
      const bf16_t* wsrc = a.wp + (((size_t)blockIdx.y * a.nchunks + ch) * wpieces) * 8;
      unsigned char* wdst = wbase + (size_t)((a.nchunks > 1 && a.nbuf > 1) ? buf : 0) * wpieces * 16;
      for (int base = 0; base < wpieces; base += 256)   // wpieces is a multiple of 64
        if (base + wave * 64 < wpieces)
          __builtin_amdgcn_global_load_lds((const void*)(wsrc + (size_t)(base + tid) * 8),
                                           (__attribute__((address_space(3))) void*)(wdst + (size_t)(base + wave * 64) * 16), 16, 0, 0);
    }
  };

  bf_f32x4 acc[VT][COT];
  float s1[COT][4], s2[COT][4], piv[COT][4], nacc = 0.f;   // BatchNorm moments over ALL boxes of this workgroup
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[c][r] = s2[c][r] = piv[c][r] = 0.f;

  if (nstage > 0 && a.nbuf > 1) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = a.nbuf > 1 ? (st & 1) : 0;
    if (a.nbuf > 1) {
      if (st + 1 < nstage) issue(st + 1, buf ^ 1);   // lands while this stage computes; buf ^ 1 was last read before the previous barrier
    } else {
      issue(st, 0);                                  // every wave passed the barrier that ended the previous stage's reads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int bi = fdiv(st, a.nchunks, inv_nch), ch = st - bi * a.nchunks;
    if (ch == 0) {
#pragma unroll
      for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[vt][c] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* xin = smem + (size_t)buf * a.stage_bytes;
    const unsigned char* wl = wbase + (size_t)((a.nchunks > 1 && a.nbuf > 1) ? buf : 0) * wpieces * 16;
    // k loop, software-pipelined by hand: the tap offset and the A (weight) fragments of step j + 1 are fetched while
    // step j runs, and ALL B fragments of a step are in flight before its first MFMA (the compiler's own schedule kept two
    // and exposed the LDS latency four times per step)
    int to_n = toff_s[g];
    bfx8 A_n[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) A_n[c] = *(const bfx8*)(wl + ((size_t)c * 64 + lane) * 16);
    for (int j = 0; j < a.nj; ++j) {
      const int to = to_n;
      bfx8 A[COT], B[VT];
#pragma unroll
      for (int c = 0; c < COT; ++c) A[c] = A_n[c];
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) B[vt] = *(const bfx8*)(xin + vbase[vt] + to);
      const int jn = j + 1 < a.nj ? j + 1 : j;
      to_n = toff_s[4 * jn + g];
#pragma unroll
      for (int c = 0; c < COT; ++c) A_n[c] = *(const bfx8*)(wl + ((size_t)(jn * COT + c) * 64 + lane) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[vt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[c], B[vt], acc[vt][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    bool counted = false;   // this stage ended with exactly VT * COT store instructions per wave
    if (ch == a.nchunks - 1) {
      // ---- epilogue: lane (m, g) holds channels co0 + 16c + 4g + r of the voxel under q point (tile vt, m) ----
      int n, q0[3];
      box_origin(box_lo + bi, n, q0);
      const bool full = q0[0] + a.bq[0] <= a.q_d[0] && q0[1] + a.bq[1] <= a.q_d[1] && q0[2] + a.bq[2] <= a.q_d[2];   // uniform
      bf16_t* ob = a.out + ((((int64_t)n * a.out_d[0] + q0[0] * a.so[0] + a.po[0]) * a.out_d[1] + q0[1] * a.so[1] + a.po[1]) * a.out_d[2] +
                            q0[2] * a.so[2] + a.po[2]) * a.out_cs + co0 + 4 * g;
      if (full && !a.accumulate) {
        // fast path (every box but those cut by the tensor edge): no per-lane bounds, one exec mask for the whole tile
        // loop, pivots taken once
#pragma unroll
        for (int c = 0; c < COT; ++c) {
          if (co0 + 16 * c + 4 * g < a.Cout) {
            u32x2 pk[VT];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
              pk[vt][0] = pack_bf2(acc[vt][c][0], acc[vt][c][1]);
              pk[vt][1] = pack_bf2(acc[vt][c][2], acc[vt][c][3]);
              *(u32x2*)(ob + orel[vt] + 16 * c) = pk[vt];
            }
            if (a.stats_partial) {
              if (nacc == 0.f) {
                piv[c][0] = __uint_as_float(pk[0][0] << 16); piv[c][1] = __uint_as_float(pk[0][0] & 0xffff0000u);
                piv[c][2] = __uint_as_float(pk[0][1] << 16); piv[c][3] = __uint_as_float(pk[0][1] & 0xffff0000u);
              }
#pragma unroll
              for (int vt = 0; vt < VT; ++vt) {
                ursn_sacc(piv[c][0], s1[c][0], s2[c][0], __uint_as_float(pk[vt][0] << 16));
                ursn_sacc(piv[c][1], s1[c][1], s2[c][1], __uint_as_float(pk[vt][0] & 0xffff0000u));
                ursn_sacc(piv[c][2], s1[c][2], s2[c][2], __uint_as_float(pk[vt][1] << 16));
                ursn_sacc(piv[c][3], s1[c][3], s2[c][3], __uint_as_float(pk[vt][1] & 0xffff0000u));
              }
            }
          }
        }
        nacc += (float)VT;
      } else
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        if (!full) {
          const int qr = qrel[vt];
          if (!(q0[0] + (qr & 0xff) < a.q_d[0] && q0[1] + ((qr >> 8) & 0xff) < a.q_d[1] && q0[2] + (qr >> 16) < a.q_d[2])) continue;
        }
        bf16_t* op = ob + orel[vt];
#pragma unroll
        for (int c = 0; c < COT; ++c) {
          if (co0 + 16 * c + 4 * g >= a.Cout) continue;
          bf_f32x4 v = acc[vt][c];
          u32x2* o = (u32x2*)(op + 16 * c);
          if (a.accumulate) {
            const u32x2 e = *o;
            v[0] += __uint_as_float(e[0] << 16); v[1] += __uint_as_float(e[0] & 0xffff0000u);
            v[2] += __uint_as_float(e[1] << 16); v[3] += __uint_as_float(e[1] & 0xffff0000u);
          }
          u32x2 pk;
          pk[0] = pack_bf2(v[0], v[1]);
          pk[1] = pack_bf2(v[2], v[3]);
          *o = pk;
          if (a.stats_partial) {   // moments of the STORED (rounded) tensor: that is what BatchNorm will normalise
            const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                 __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (nacc == 0.f) piv[c][r] = rv[r];
              ursn_sacc(piv[c][r], s1[c][r], s2[c][r], rv[r]);
            }
          }
        }
        nacc += 1.f;
      }
      counted = full && !a.accumulate && co0 + 16 * (COT - 1) < a.Cout;
    }
    // The next stage's DMA must have landed; this stage's output stores need not (waiting for their acknowledgement every
    // stage serialised the pipeline on the HBM write latency).  VM operations retire in issue order and the stores are the
    // youngest: in a full box every wave issued exactly VT * COT of them, so vmcnt(VT * COT) waits for the DMA alone.
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VT * COT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (a.stats_partial) {
    __shared__ double red[4][2 * 16 * COT];
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double u, w2;
        ursn_sacc_final(piv[c][r], s1[c][r], s2[c][r], nacc, u, w2);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
        if (m == 0) {
          red[wave][16 * c + 4 * g + r] = u;
          red[wave][16 * COT + 16 * c + 4 * g + r] = w2;
        }
      }
    __syncthreads();
    if (tid < 2 * 16 * COT)
      a.stats_partial[((size_t)blockIdx.y * a.stats_total + a.stats_off + blockIdx.x) * 2 * 16 * COT + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// (weight packing: BPK_GENERIC in bf16_pack.hip -- [co block][chunk][j][co tile][lane = 16 g + m][8] + the zero piece)

// ---- host side: box / chunk choice ----------------------------------------------------------------------------------
struct BPlan {
  int bq[3], nb[3], hb[3], dmin[3];
  int cinc, nchunks, nj, cot, ncob, vt;
  size_t lds;        // nbuf input buffers + one | two weight buffers
  int stage_bytes, pp, nbuf;
  int nboxes, per, gridx;   // gridx persistent workgroups, each walks `per` consecutive boxes
};

static bool bconv_plan(const GatherGeom& g, BPlan& p) {
  if ((g.K & 7) || (g.Nn & 7) || (g.in_cs & 7) || (g.out_cs & 3) || g.ntaps < 1) return false;
  int dmax[3];
  for (int j = 0; j < 3; ++j) {
    p.dmin[j] = g.tap_d[0][j]; dmax[j] = g.tap_d[0][j];
    for (int t = 1; t < g.ntaps; ++t) {
      if (g.tap_d[t][j] < p.dmin[j]) p.dmin[j] = g.tap_d[t][j];
      if (g.tap_d[t][j] > dmax[j]) dmax[j] = g.tap_d[t][j];
    }
  }
  static int cand[][3] = {{2, 8, 32}, {1, 8, 32}, {2, 8, 16}, {1, 4, 32}, {1, 8, 16}, {4, 8, 8}, {1, 2, 32}, {1, 4, 16}, {2, 8, 8}, {1, 8, 8}};
  static bool env_done = false;
  // first-pass LDS budget.  78 KB (two workgroups per CU) chose 64-voxel boxes for the 32..256-channel layers, whose packed
  // weights alone take 55 KB: 0.31 ms per 32 -> 32 conv at 64^3 x 4; one workgroup per CU on 256..512-voxel boxes: 0.11 ms
  static size_t lds_cap = 158 * 1024;
  static int nbuf_env = 2;
  if (!env_done) {   // A/B: URSN_BCONV_BOX="z,y,x" replaces the first candidate, URSN_BCONV_LDS_KB caps the first-pass LDS budget
    env_done = true;
    const char* e = getenv("URSN_BCONV_BOX");
    if (e) sscanf(e, "%d,%d,%d", &cand[0][0], &cand[0][1], &cand[0][2]);
    const char* l = getenv("URSN_BCONV_LDS_KB");
    if (l) lds_cap = (size_t)atoi(l) * 1024;
    const char* nb = getenv("URSN_BCONV_NBUF");
    if (nb) nbuf_env = atoi(nb) == 1 ? 1 : 2;
  }
  const int ncand = (int)(sizeof(cand) / sizeof(cand[0]));
  static const int64_t minwg = getenv("URSN_BCONV_MINWG") ? atoi(getenv("URSN_BCONV_MINWG")) : 256;   // one workgroup per CU is enough to take the larger box (512: 77.7, 256: 78.6 img/s at cfg5)
  // Contraction channels per staged chunk: 32 where some box fits the LDS, else 16, else 8; per box candidate (largest first: the
  // packed weights of a chunk are re-read per box) two stage buffers (the next chunk's halo box and weights land while this one
  // computes), else one.  Measured and rejected (round 3): preferring 16-channel chunks with two buffers over 32-channel chunks
  // with one -- 64 -> 32 @64^3 0.200 -> 0.229 ms, 64 -> 64 @32^3 unchanged -- and 8-channel chunks on the largest box for the
  // stride-2 layers (16 -> 32 s2 @128^3: 0.157 -> 0.228 ms).
  const int forms[6][2] = {{32, 2}, {32, 1}, {16, 2}, {16, 1}, {8, 2}, {8, 1}};
  const size_t limits[2] = {lds_cap, 158 * 1024};
  for (int f0 = 0; f0 < 6; f0 += 2) {   // chunk width
  if (f0 > 0 && (g.K < forms[f0 - 2][0])) break;   // the narrower chunk is the same chunk (K itself)
  for (size_t limit : limits) {
    BPlan fit;
    bool have = false;
    for (int ci = 0; ci < ncand; ++ci)
      for (int fi = f0; fi < f0 + 2; ++fi) {
        const int nbuf = forms[fi][1];
        if (nbuf == 2 && nbuf_env == 1) continue;
        BPlan c = p;
        c.cinc = g.K < forms[fi][0] ? g.K : forms[fi][0];
        if (g.K % c.cinc) continue;
        c.nchunks = g.K / c.cinc;
        const int slots = g.ntaps * (c.cinc / 8);
        c.nj = (slots + 3) / 4;
        if (c.nj * 4 > BCONV_MAX_SLOTS) continue;
        c.cot = g.Nn > 16 ? 2 : 1;
        c.ncob = (g.Nn + 16 * c.cot - 1) / (16 * c.cot);
        const size_t wbytes = (size_t)c.nj * c.cot * 64 * 16;
        int bq[3] = {cand[ci][0], cand[ci][1], cand[ci][2]};
        if (g.q_d[0] == 1) { bq[1] *= bq[0]; bq[0] = 1; }          // 2-D problems: all rows in y
        // a box wider than the volume computes padding: 8- and 16-wide levels take the 8- / 16-wide boxes
        if (bq[2] > 8 && bq[2] >= 2 * g.q_d[2]) continue;
        if (bq[2] == 8 && g.q_d[2] > 8) continue;                   // narrow boxes only where the volume is narrow (short DMA runs)
        if (bq[2] == 16 && g.q_d[2] > 16 && !(cand[ci][0] == 1 && cand[ci][1] == 4)) continue;   // (1 x 4 x 16 stays the last resort of wide volumes)
        for (int j = 0; j < 3; ++j) {
          c.bq[j] = bq[j];
          c.hb[j] = (bq[j] - 1) * g.si[j] + (dmax[j] - p.dmin[j]) + 1;
          c.nb[j] = (g.q_d[j] + bq[j] - 1) / bq[j];
        }
        const int nq = c.bq[0] * c.bq[1] * c.bq[2];
        c.vt = nq / 64;
        c.pp = (c.hb[1] * c.hb[2] * (c.cinc / 8) + 255) & ~255;   // pieces per plane, padded
        if (c.pp > 256 * BCONV_PMAX) continue;
        c.stage_bytes = c.hb[0] * c.pp * 16;
        c.nbuf = nbuf;
        c.lds = (size_t)c.nbuf * c.stage_bytes + ((c.nchunks > 1 && c.nbuf > 1) ? 2 : 1) * wbytes;
        const int64_t boxes = (int64_t)g.N * c.nb[0] * c.nb[1] * c.nb[2];
        if (c.lds > limit || boxes <= 0 || boxes >= (1ll << 30)) continue;
        c.nboxes = (int)boxes;
        int64_t occ = (int64_t)(160 * 1024) / (c.lds + 2048);
        if (occ > 8) occ = 8;
        if (occ < 1) occ = 1;
        int64_t wg = (int64_t)ursn_cu_count() * occ / c.ncob;   // persistent workgroups: every resident slot
        if (wg < 1) wg = 1;
        if (wg > boxes) wg = boxes;
        c.per = (int)((boxes + wg - 1) / wg);
        c.gridx = (int)((boxes + c.per - 1) / c.per);
        if (!have) { fit = c; have = true; }
        if (boxes * c.ncob >= minwg) { p = c; return true; }
      }
    if (have) { p = fit; return true; }
  }
  }   // next (narrower) chunk
  return false;
}

size_t bconv_pack_elems(const GatherGeom& g) {
  if (bpw_ok(g)) return 8;
  if (b3conv_ok(g)) return b3conv_pack_elems();
  if (bcbconv_ok(g)) return bcbconv_pack_elems(g);
  if (bdconv_ok(g)) return bdconv_pack_elems(g);
  BPlan p;
  if (!bconv_plan(g, p)) return 0;
  return (size_t)p.ncob * p.nchunks * p.nj * p.cot * 64 * 8 + 8;   // + the zero piece
}

size_t bconv_stats_scratch_doubles(const GatherGeom& g) {
  if (bpw_ok(g)) return (size_t)bpw_grid_blocks(g) * 2 * 16;
  if (b3conv_ok(g)) return (size_t)b3conv_grid_blocks(g) * 2 * 16;
  if (bcbconv_ok(g)) return bcbconv_stats_scratch_doubles(g);
  if (bdconv_ok(g)) return bdconv_stats_scratch_doubles(g);
  BPlan p;
  if (!bconv_plan(g, p)) return 0;
  return (size_t)p.ncob * p.gridx * 2 * 16 * p.cot;
}

template <int VT, int COT>
static int bconv_launch(const BPlan& p, const BConvArgs& a, hipStream_t s) {
  auto kern = bconv_kernel<VT, COT>;
  static size_t attr = 48 * 1024;
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.ncob), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int bconv_grid_blocks(const GatherGeom& g) {
  if (bpw_ok(g)) return bpw_grid_blocks(g);
  if (b3conv_ok(g)) return b3conv_grid_blocks(g);
  if (bcbconv_ok(g)) return bcbconv_grid_blocks(g);
  if (bdconv_ok(g)) return bdconv_grid_blocks(g);
  BPlan p;
  return bconv_plan(g, p) ? p.gridx : 0;
}

int bconv_stats_finalize(const GatherGeom& g, const double* partial, int total_blocks, int64_t V, float eps, float* mean,
                         float* rstd, hipStream_t s) {
  if (bpw_ok(g) || b3conv_ok(g)) return launch_bn_stats_final(partial, total_blocks, g.Nn, 16, V, eps, mean, rstd, s);
  if (bcbconv_ok(g)) return bcbconv_stats_finalize(g, partial, V, eps, mean, rstd, s);
  if (bdconv_ok(g)) return bdconv_stats_finalize(g, partial, V, eps, mean, rstd, s);
  BPlan p;
  URSN_REQUIRE(bconv_plan(g, p), "bf16 conv: unsupported geometry");
  URSN_TRY(launch_bn_stats_final_blocked(partial, total_blocks, g.Nn, 16 * p.cot, 16 * p.cot, (size_t)total_blocks * 2 * 16 * p.cot, V, eps,
                                         mean, rstd, s));
  return 0;
}

int launch_bconv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                 double* stats_partial, int stats_off, int stats_total, hipStream_t s) {
  if (bpw_ok(g)) return launch_bpw(g, in, w, Kw, Nw, out, stats_partial ? stats_partial + (size_t)stats_off * 32 : nullptr, s);
  if (b3conv_ok(g)) return launch_b3conv(g, in, w, Kw, Nw, wpack, out, stats_partial, stats_off, stats_total, s);
  if (bcbconv_ok(g)) {   // 16 / 32 contraction channels, 3-D k3 s1: the z-marching channel-block kernel (bf16_convcb.hip)
    URSN_REQUIRE(!stats_partial || (stats_off == 0 && (stats_total <= 0 || stats_total == bcbconv_grid_blocks(g))),
                 "bf16 channel-block conv: its statistics partials are one launch's own");
    return launch_bcbconv(g, in, w, Kw, Nw, wpack, out, stats_partial, s);
  }
  if (bdconv_ok(g)) {   // >= 64 contraction channels, 3-D k3 s1: the weight-streaming split-K kernel (bf16_convdeep.hip)
    URSN_REQUIRE(!stats_partial || (stats_off == 0 && (stats_total <= 0 || stats_total == bdconv_grid_blocks(g))),
                 "bf16 deep conv: its statistics partials are one launch's own");
    return launch_bdconv(g, in, w, Kw, Nw, wpack, out, stats_partial, s);
  }
  BPlan p;
  URSN_REQUIRE(bconv_plan(g, p), "bf16 conv: unsupported geometry (channels %d -> %d, strides %d / %d)", g.K, g.Nn, g.in_cs, g.out_cs);
  {
    BPackJob k = bpack_job(BPK_GENERIC);
    k.w = w; k.wp = wpack; k.Kw = Kw > 0 ? Kw : g.K; k.Nw = Nw > 0 ? Nw : g.Nn;
    k.w_tap_stride = g.w_tap_stride; k.w_sk = g.w_sk; k.w_sn = g.w_sn;
    for (int t = 0; t < g.ntaps; ++t) k.tap[t] = g.tap_w[t];
    k.p[0] = g.ntaps; k.p[1] = p.cinc; k.p[2] = p.nchunks; k.p[3] = p.nj; k.p[4] = p.cot; k.p[5] = p.ncob;
    const int64_t total = (int64_t)p.ncob * p.nchunks * p.nj * p.cot * 64 * 8;
    k.blocks = (int)(cdiv64(total, 256) < 2048 ? cdiv64(total, 256) : 2048);
    URSN_TRY(bpack_submit(k, s));
  }
  BConvArgs a;
  a.in = in; a.wp = wpack; a.out = out; a.stats_partial = stats_partial;
  a.zero = wpack + (size_t)p.ncob * p.nchunks * p.nj * p.cot * 64 * 8;
  a.nboxes = p.nboxes; a.per = p.per; a.stage_bytes = p.stage_bytes; a.pp = p.pp; a.nbuf = p.nbuf;
  a.stats_off = stats_off; a.stats_total = stats_total > 0 ? stats_total : p.gridx;
  a.N = g.N;
  for (int j = 0; j < 3; ++j) {
    a.in_d[j] = g.in_d[j]; a.out_d[j] = g.out_d[j]; a.q_d[j] = g.q_d[j]; a.so[j] = g.so[j]; a.po[j] = g.po[j]; a.si[j] = g.si[j];
    a.bq[j] = p.bq[j]; a.nb[j] = p.nb[j]; a.hb[j] = p.hb[j]; a.dmin[j] = p.dmin[j];
  }
  a.Cin = g.K; a.Cout = g.Nn; a.in_cs = g.in_cs; a.out_cs = g.out_cs; a.accumulate = g.accumulate;
  a.cinc = p.cinc; a.nchunks = p.nchunks; a.nj = p.nj; a.cob = 16 * p.cot;
  const int cpb = p.cinc / 8;
  for (int sl = 0; sl < p.nj * 4; ++sl) {
    const int t = sl / cpb, cb = sl - t * cpb;
    a.toff[sl] = 0;
    if (t < g.ntaps)
      a.toff[sl] = ((g.tap_d[t][0] - p.dmin[0]) * p.pp + ((g.tap_d[t][1] - p.dmin[1]) * p.hb[2] + (g.tap_d[t][2] - p.dmin[2])) * cpb + cb) * 16;
  }
  ursn_note_kernel("bconv_bf16");
  int rc = 3;
#define BC(vt_, cot_) if (p.vt == vt_ && p.cot == cot_) { ursn_note_kernel("bconv_bf16<" #vt_ "," #cot_ ">"); rc = bconv_launch<vt_, cot_>(p, a, s); }
  BC(8, 1) BC(8, 2) BC(4, 1) BC(4, 2) BC(2, 1) BC(2, 2) BC(1, 1) BC(1, 2)
#undef BC
  return rc;
}

// =========================================================================================================================
// weight gradient
// =========================================================================================================================
struct BWgradArgs {
  const bf16_t* S;      // gathered tensor (in_d, in_cs), contraction channels K
  const bf16_t* C;      // q-grid tensor (q_d, out_cs), produced channels Nn
  const bf16_t* zero;   // >= 16 zero bytes (padding source of the LDS-DMA staging)
  float* slab;          // [grid.y][grid.x][U*16][16*COT] fp32
  int N;
  int in_d[3], q_d[3], si[3];
  int K, Nn, in_cs, out_cs;
  int bq[3], nb[3], hb[3], dmin[3];
  int cinc, nchunks, ncob, U;     // U: 16-row tiles of (tap, channel) rows per chunk
  int nboxes, per;                // consecutive boxes (z fastest) per workgroup
  int spp, cpp;                   // pieces per staged S halo plane / C box plane, padded to multiples of 256
  int toff[BCONV_MAX_SLOTS];      // LDS byte offset of k slot s = tap * (cinc/8) + channel block (as in bconv)
};

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bfx8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 lo = lds_tr16(p0), hi = lds_tr16(p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bfx8, v);
}

#define BWG_CMAX 4   // C-box staging pieces per thread per plane (plane pieces <= 1024)

// MTW: 16-row tiles per wave (tile u = wave + 4 i), COT: 16-column tiles of produced channels.
// LDS: [S halo box: hb[0] planes of spp pieces][C box: bq[0] planes of cpp pieces]; both filled by LDS-DMA with per-thread
// in-plane geometry computed once (as bconv); one buffer, the co-resident workgroups hide the staging latency.
template <int MTW, int COT>
__global__ __launch_bounds__(256) void bwgrad_kernel(BWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;   // transposed read: lane (row tq, column piece tp) of group G
  const int cpb = a.cinc >> 3;                    // 16-byte pieces per staged S voxel
  constexpr int ccn = 16 * COT;                   // staged C channels per voxel
  constexpr int ccp = ccn >> 3;                   // pieces per staged C voxel
  unsigned char* sbox = smem;
  unsigned char* cbox = smem + (size_t)a.hb[0] * a.spp * 16;
  const int pair = blockIdx.y, ch = pair % a.nchunks, cb_ = pair / a.nchunks;
  const int co0 = cb_ * ccn;

  bf_f32x4 acc[MTW][COT];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};

  __shared__ int toff_s[BCONV_MAX_SLOTS];
  if (tid < BCONV_MAX_SLOTS) toff_s[tid] = a.toff[tid];
  __syncthreads();
  int aoff[MTW];   // per M tile of this wave: LDS byte offset of (k slot of this lane's column piece) relative to the voxel position
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int u = wave + 4 * i, slot = 2 * u + (tp >> 1);
    aoff[i] = ((u < a.U && slot < BCONV_MAX_SLOTS) ? toff_s[slot] : 0) + (tp & 1) * 8;
  }

  // ---- per-thread staging geometry (once) ----
  int s_off[BCONV_PMAX], s_yx[BCONV_PMAX];   // S halo plane: element offset from the plane origin, hy | hx << 16 (or -1: pad)
  int c_off[BWG_CMAX], c_yx[BWG_CMAX];       // C box plane
  {
    const int splane = a.hb[1] * a.hb[2] * cpb, cplane = a.bq[1] * a.bq[2] * ccp;
    const float inv_cpb = 1.0f / (float)cpb, inv_h2 = 1.0f / (float)a.hb[2], inv_b2 = 1.0f / (float)a.bq[2];
#pragma unroll
    for (int k = 0; k < BCONV_PMAX; ++k) {
      const int idx = k * 256 + tid;
      s_off[k] = 0; s_yx[k] = -1;
      if (idx < splane) {
        const int vox = fdiv(idx, cpb, inv_cpb), cb = idx - vox * cpb;
        const int hy = fdiv(vox, a.hb[2], inv_h2), hx = vox - hy * a.hb[2];
        s_off[k] = (hy * a.in_d[2] + hx) * a.in_cs + cb * 8;
        s_yx[k] = hy | (hx << 16);
      }
    }
#pragma unroll
    for (int k = 0; k < BWG_CMAX; ++k) {
      const int idx = k * 256 + tid;
      c_off[k] = 0; c_yx[k] = -1;
      if (idx < cplane) {
        const int vox = idx / ccp, cb = idx - vox * ccp;
        const int qy = fdiv(vox, a.bq[2], inv_b2), qx = vox - qy * a.bq[2];
        if (co0 + cb * 8 < a.Nn) {   // channels beyond the tensor: zero piece
          c_off[k] = (qy * a.q_d[2] + qx) * a.out_cs + cb * 8;
          c_yx[k] = qy | (qx << 16);
        }
      }
    }
  }
  const int nsp = a.spp >> 8, ncp = a.cpp >> 8;
  const int64_t s_plane_stride = (int64_t)a.in_d[1] * a.in_d[2] * a.in_cs, c_plane_stride = (int64_t)a.q_d[1] * a.q_d[2] * a.out_cs;
  const int xruns = a.bq[2] >> 5;
  const float inv_nb0 = 1.0f / (float)a.nb[0], inv_nb2 = 1.0f / (float)a.nb[2], inv_nb1 = 1.0f / (float)a.nb[1];
  const int box_lo = blockIdx.x * a.per;
  int box_hi = box_lo + a.per;
  if (box_hi > a.nboxes) box_hi = a.nboxes;

  for (int box = box_lo; box < box_hi; ++box) {
    int b = box, n, q0[3];
    {
      int t = fdiv(b, a.nb[0], inv_nb0);
      const int bz = b - t * a.nb[0]; b = t;
      t = fdiv(b, a.nb[2], inv_nb2);
      const int bx = b - t * a.nb[2]; b = t;
      n = fdiv(b, a.nb[1], inv_nb1);
      const int by = b - n * a.nb[1];
      q0[0] = bz * a.bq[0]; q0[1] = by * a.bq[1]; q0[2] = bx * a.bq[2];
    }
    const int g0 = q0[0] * a.si[0] + a.dmin[0], g1 = q0[1] * a.si[1] + a.dmin[1], g2 = q0[2] * a.si[2] + a.dmin[2];
    __syncthreads();   // the previous box's reads are done
    {
      const bool inner = g1 >= 0 && g1 + a.hb[1] <= a.in_d[1] && g2 >= 0 && g2 + a.hb[2] <= a.in_d[2];
      const bf16_t* pl0 = a.S + ((((int64_t)n * a.in_d[0] + g0) * a.in_d[1] + g1) * a.in_d[2] + g2) * a.in_cs + ch * a.cinc;
      unsigned char* dst = sbox + (size_t)wave * 1024;
      for (int hz = 0; hz < a.hb[0]; ++hz) {
        const bool zok = (unsigned)(g0 + hz) < (unsigned)a.in_d[0];
        const bf16_t* pl = pl0 + hz * s_plane_stride;
#pragma unroll
        for (int k = 0; k < BCONV_PMAX; ++k) {
          if (k < nsp) {
            const bf16_t* src = a.zero;
            const int yx = s_yx[k];
            bool ok = zok && yx >= 0;
            if (ok && !inner) {
              const int y = g1 + (yx & 0xffff), x = g2 + (yx >> 16);
              ok = (unsigned)y < (unsigned)a.in_d[1] && (unsigned)x < (unsigned)a.in_d[2];
            }
            if (ok) src = pl + s_off[k];
            __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(dst + ((size_t)hz * a.spp + k * 256) * 16), 16, 0, 0);
          }
        }
      }
    }
    {
      const bool full = q0[1] + a.bq[1] <= a.q_d[1] && q0[2] + a.bq[2] <= a.q_d[2];
      const bf16_t* pl0 = a.C + ((((int64_t)n * a.q_d[0] + q0[0]) * a.q_d[1] + q0[1]) * a.q_d[2] + q0[2]) * a.out_cs + co0;
      unsigned char* dst = cbox + (size_t)wave * 1024;
      for (int qz = 0; qz < a.bq[0]; ++qz) {
        const bool zok = q0[0] + qz < a.q_d[0];
        const bf16_t* pl = pl0 + qz * c_plane_stride;
#pragma unroll
        for (int k = 0; k < BWG_CMAX; ++k) {
          if (k < ncp) {
            const bf16_t* src = a.zero;
            const int yx = c_yx[k];
            bool ok = zok && yx >= 0;
            if (ok && !full) ok = q0[1] + (yx & 0xffff) < a.q_d[1] && q0[2] + (yx >> 16) < a.q_d[2];
            if (ok) src = pl + c_off[k];
            __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(dst + ((size_t)qz * a.cpp + k * 256) * 16), 16, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int qz = 0; qz < a.bq[0]; ++qz)
      for (int qy = 0; qy < a.bq[1]; ++qy)
        for (int xrn = 0; xrn < xruns; ++xrn) {
          const int xq = xrn * 32 + 8 * G + tq;   // this lane's voxel row of the first 4 x 16 block
          // B operand: C[k = voxel][n = channel]; lane supplies row tq, columns 4 tp .. 4 tp + 3
          const unsigned char* cp = cbox + ((size_t)qz * a.cpp * 16) + ((size_t)(qy * a.bq[2] + xq) * ccn + tp * 4) * 2;
          bfx8 B[COT];
#pragma unroll
          for (int c = 0; c < COT; ++c) B[c] = tr_pair(cp + c * 32, cp + c * 32 + (size_t)4 * ccn * 2);
          const int sp = (qz * a.si[0]) * a.spp * 16 + ((qy * a.si[1]) * a.hb[2] + xq * a.si[2]) * cpb * 16;   // S position under voxel row tq
          const int sp4 = 4 * a.si[2] * cpb * 16;                                                          // + 4 voxels along x
          bfx8 A[MTW];
#pragma unroll
          for (int i = 0; i < MTW; ++i) A[i] = tr_pair(sbox + sp + aoff[i], sbox + sp + sp4 + aoff[i]);
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[c], acc[i][c], 0, 0, 0);
        }
  }
  // slab: rows 16 u + 4 G + r, columns 16 c + li
  float* sl = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ((size_t)a.U * 16 * ccn);
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int u = wave + 4 * i;
    if (u >= a.U) continue;
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(size_t)(16 * u + 4 * G + r) * ccn + 16 * c + li] = acc[i][c][r];
  }
}

// dw[tap_w[t]][ci][co] += sum over workgroup slabs (fixed order); one thread per (pair, row, column)
struct BWReduceArgs {
  const float* slab; float* dw;
  int nslabs, U, ccn, cinc, nchunks, ncob, ntaps, K, Nn;   // K, Nn: extents of the STORED gradient tensor [t][K][Nn]
  int tap_w[URSN_MAX_TAPS];
};
// block = 64 consecutive elements x 4 slab quarters (threadIdx.y): a single thread walking several hundred slabs per element
// was pure load latency (20 us per launch); the quarters are added in a fixed order
__global__ __launch_bounds__(256) void bwgrad_reduce_kernel(BWReduceArgs a) {
  __shared__ float part[4][64];
  const int cpb = a.cinc >> 3;
  const int64_t per = (int64_t)a.U * 16 * a.ccn, total = per * a.nchunks * a.ncob;
  const int el = threadIdx.x, sl = threadIdx.y;
  const int qn = (a.nslabs + 3) / 4, k0 = sl * qn, k1 = k0 + qn < a.nslabs ? k0 + qn : a.nslabs;
  for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < total; e0 += (int64_t)gridDim.x * 64) {
    const int64_t e = e0 + el;
    float sum = 0.f;
    int t = 0, ci = 0, co = 0;
    bool ok = e < total;
    if (ok) {
      const int pair = (int)(e / per);
      const int64_t rc = e - (int64_t)pair * per;
      const int row = (int)(rc / a.ccn), col = (int)(rc - (int64_t)row * a.ccn);
      const int ch = pair % a.nchunks, cb_ = pair / a.nchunks;
      const int slot = row >> 3;
      t = slot / cpb;
      const int cb = slot - t * cpb;
      ci = ch * a.cinc + cb * 8 + (row & 7);
      co = cb_ * a.ccn + col;
      ok = t < a.ntaps && ci < a.K && co < a.Nn;
      if (ok) {
        const float* p = a.slab + (size_t)pair * a.nslabs * per + rc;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = k0;
        for (; k + 3 < k1; k += 4) {
          s0 += p[(size_t)k * per]; s1 += p[(size_t)(k + 1) * per]; s2 += p[(size_t)(k + 2) * per]; s3 += p[(size_t)(k + 3) * per];
        }
        for (; k < k1; ++k) s0 += p[(size_t)k * per];
        sum = (s0 + s1) + (s2 + s3);
      }
    }
    part[sl][el] = sum;
    __syncthreads();
    if (sl == 0 && ok)
      a.dw[(size_t)a.tap_w[t] * a.K * a.Nn + (size_t)ci * a.Nn + co] += (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    __syncthreads();
  }
}

struct BWPlan {
  int bq[3], nb[3], hb[3], dmin[3];
  int cinc, nchunks, cot, ncob, U, mtw, nboxes, gridx, per, spp, cpp;
  size_t lds;
};

static bool bwgrad_plan(const GatherGeom& g, BWPlan& p) {
  if ((g.K & 7) || (g.Nn & 7) || (g.in_cs & 7) || (g.out_cs & 7) || g.ntaps < 1) return false;
  int dmax[3];
  for (int j = 0; j < 3; ++j) {
    p.dmin[j] = g.tap_d[0][j]; dmax[j] = g.tap_d[0][j];
    for (int t = 1; t < g.ntaps; ++t) {
      if (g.tap_d[t][j] < p.dmin[j]) p.dmin[j] = g.tap_d[t][j];
      if (g.tap_d[t][j] > dmax[j]) dmax[j] = g.tap_d[t][j];
    }
  }
  p.cinc = g.K >= 32 ? 32 : g.K;
  if (g.K > 32 && (g.K % 32)) p.cinc = (g.K % 16) ? 8 : 16;
  p.nchunks = g.K / p.cinc;
  const int slots = g.ntaps * (p.cinc / 8);
  if (slots + 2 > BCONV_MAX_SLOTS) return false;
  p.U = (slots + 1) / 2;
  const int per_wave = (p.U + 3) / 4;
  p.mtw = per_wave <= 1 ? 1 : per_wave <= 2 ? 2 : per_wave <= 4 ? 4 : per_wave <= 7 ? 7 : 14;
  if (per_wave > 14) return false;
  p.cot = g.Nn > 16 ? 2 : 1;
  p.ncob = (g.Nn + 16 * p.cot - 1) / (16 * p.cot);
  static const int cand[][3] = {{2, 8, 32}, {1, 8, 32}, {1, 4, 32}, {1, 2, 32}, {1, 1, 32}};
  const int ncand = (int)(sizeof(cand) / sizeof(cand[0]));
  static size_t first_limit = getenv("URSN_BWGRAD_LDS_KB") ? (size_t)atoi(getenv("URSN_BWGRAD_LDS_KB")) * 1024 : 156 * 1024;   // as bconv: large boxes beat two workgroups per CU (52 KB: 73.2, 156 KB: 74.8 img/s at cfg5)
  const size_t limits[2] = {first_limit, 156 * 1024};
  for (size_t limit : limits)
    for (int ci = 0; ci < ncand; ++ci) {
      int bq[3] = {cand[ci][0], cand[ci][1], cand[ci][2]};
      if (g.q_d[0] == 1) { bq[1] *= bq[0]; bq[0] = 1; }
      for (int j = 0; j < 3; ++j) {
        p.bq[j] = bq[j];
        p.hb[j] = (bq[j] - 1) * g.si[j] + (dmax[j] - p.dmin[j]) + 1;
        p.nb[j] = (g.q_d[j] + bq[j] - 1) / bq[j];
      }
      p.spp = (p.hb[1] * p.hb[2] * (p.cinc / 8) + 255) & ~255;
      p.cpp = (p.bq[1] * p.bq[2] * 2 * p.cot + 255) & ~255;
      if (p.spp > 256 * BCONV_PMAX || p.cpp > 256 * BWG_CMAX) continue;
      p.lds = ((size_t)p.hb[0] * p.spp + (size_t)p.bq[0] * p.cpp) * 16 + 64;
      const int64_t boxes = (int64_t)g.N * p.nb[0] * p.nb[1] * p.nb[2];
      if (p.lds > limit || boxes <= 0 || boxes >= (1ll << 30)) continue;
      // smaller boxes when the problem is small, so that more than a handful of workgroups take part
      static const int64_t wminwg = getenv("URSN_BWGRAD_MINWG") ? atoi(getenv("URSN_BWGRAD_MINWG")) : 256;   // A/B
      if (boxes * p.nchunks * p.ncob < wminwg && ci + 1 < ncand) continue;
      p.nboxes = (int)boxes;
      int64_t occ = (int64_t)(160 * 1024) / (p.lds + 2048);
      if (occ > 4) occ = 4;
      if (occ < 1) occ = 1;
      int64_t gx = (int64_t)ursn_cu_count() * occ / ((int64_t)p.nchunks * p.ncob);
      if (gx < 1) gx = 1;
      if (gx > boxes) gx = boxes;
      p.per = (int)((boxes + gx - 1) / gx);
      p.gridx = (int)((boxes + p.per - 1) / p.per);
      return true;
    }
  return false;
}

size_t bwgrad_scratch_bytes(const GatherGeom& g) {   // 256 bytes (the zero piece) + the slabs
  if (b3wgrad_ok(g)) return b3wgrad_scratch_bytes(g);
  if (bdwgrad_ok(g)) return bdwgrad_scratch_bytes(g);
  if (bs2k8w_ok(g)) return bs2k8w_scratch_bytes(g);
  BWPlan p;
  if (!bwgrad_plan(g, p)) return 0;
  return (size_t)p.nchunks * p.ncob * p.gridx * p.U * 16 * 16 * p.cot * sizeof(float) + 512;
}

template <int MTW, int COT>
static int bwgrad_launch(const BWPlan& p, const BWgradArgs& a, hipStream_t s) {
  auto kern = bwgrad_kernel<MTW, COT>;
  static size_t attr = 48 * 1024;
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.nchunks * p.ncob), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

// 256 zero bytes inside the code object: the source of the padding pieces of the weight gradient's LDS-DMA staging (the launcher
// used to clear the head of the caller's scratch with a hipMemsetAsync per launch: 49 fill kernels of 4.8 us per cfg5 step)
__device__ __attribute__((aligned(256))) unsigned char ursn_zero_piece_dev[256];
static const bf16_t* zero_piece() {
  static const bf16_t* p = nullptr;
  if (!p) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(ursn_zero_piece_dev)) != hipSuccess) return nullptr;
    p = (const bf16_t*)d;
  }
  return p;
}

int launch_bwgrad(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch,
                  size_t scratch_bytes, hipStream_t s) {
  if (b3wgrad_ok(g)) return launch_b3wgrad(g, S, C, dw, Kw, Nw, scratch, scratch_bytes, s);
  if (bdwgrad_ok(g)) return launch_bdwgrad(g, S, C, dw, Kw, Nw, scratch, scratch_bytes, s);   // deep levels (bf16_wgraddeep.hip)
  if (bs2k8w_ok(g)) return launch_bs2k8w(g, S, C, dw, Kw, Nw, scratch, scratch_bytes, nullptr, 0, nullptr, s);   // stride 2, 8 x 16 channels
  BWPlan p;
  URSN_REQUIRE(bwgrad_plan(g, p), "bf16 wgrad: unsupported geometry (channels %d x %d)", g.K, g.Nn);
  URSN_REQUIRE(scratch && scratch_bytes >= bwgrad_scratch_bytes(g), "bf16 wgrad: scratch too small");
  const bf16_t* zero = zero_piece();
  URSN_REQUIRE(zero, "bf16 wgrad: no address for the zero piece");
  scratch = (char*)scratch + 256;   // (the slabs keep their 256-byte offset)
  BWgradArgs a;
  a.S = S; a.C = C; a.slab = (float*)scratch; a.N = g.N;
  for (int j = 0; j < 3; ++j) {
    a.in_d[j] = g.in_d[j]; a.q_d[j] = g.q_d[j]; a.si[j] = g.si[j];
    a.bq[j] = p.bq[j]; a.nb[j] = p.nb[j]; a.hb[j] = p.hb[j]; a.dmin[j] = p.dmin[j];
  }
  a.K = g.K; a.Nn = g.Nn; a.in_cs = g.in_cs; a.out_cs = g.out_cs;
  a.cinc = p.cinc; a.nchunks = p.nchunks; a.ncob = p.ncob; a.U = p.U; a.nboxes = p.nboxes; a.per = p.per;
  a.spp = p.spp; a.cpp = p.cpp;
  a.zero = zero;
  const int cpb = p.cinc / 8, slots = g.ntaps * cpb;
  for (int sl = 0; sl < BCONV_MAX_SLOTS; ++sl) {
    const int t = sl / cpb, cb = sl - t * cpb;
    a.toff[sl] = 0;
    if (sl < slots)
      a.toff[sl] = ((g.tap_d[t][0] - p.dmin[0]) * p.spp + ((g.tap_d[t][1] - p.dmin[1]) * p.hb[2] + (g.tap_d[t][2] - p.dmin[2])) * cpb + cb) * 16;
  }
  ursn_note_kernel("bwgrad_bf16");
  int rc = 3;
#define BW(m_, c_) if (p.mtw == m_ && p.cot == c_) { ursn_note_kernel("bwgrad_bf16<" #m_ "," #c_ ">"); rc = bwgrad_launch<m_, c_>(p, a, s); }
  BW(1, 1) BW(1, 2) BW(2, 1) BW(2, 2) BW(4, 1) BW(4, 2) BW(7, 1) BW(7, 2) BW(14, 1) BW(14, 2)
#undef BW
  if (rc) return rc;
  BWReduceArgs r;
  r.slab = (const float*)scratch; r.dw = dw; r.nslabs = p.gridx; r.U = p.U; r.ccn = 16 * p.cot; r.cinc = p.cinc;
  r.nchunks = p.nchunks; r.ncob = p.ncob; r.ntaps = g.ntaps; r.K = Kw > 0 ? Kw : g.K; r.Nn = Nw > 0 ? Nw : g.Nn;
  for (int t = 0; t < g.ntaps; ++t) r.tap_w[t] = g.tap_w[t];
  const int64_t total = (int64_t)p.U * 16 * 16 * p.cot * p.nchunks * p.ncob;
  int blocks = (int)(cdiv64(total, 64) < 8192 ? cdiv64(total, 64) : 8192);
  hipLaunchKernelGGL(bwgrad_reduce_kernel, dim3(blocks), dim3(64, 4), 0, s, r);
  URSN_HIP(hipGetLastError());
  return 0;
}
