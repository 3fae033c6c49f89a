// Microbenchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 / 16x16x4 with and without LDS operand reads, 1-2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDS_PER, int VALU_PER, int OCC = 2>   // LDS_PER: ds_read_b32 per NACC MFMAs; VALU_PER: extra v_add per NACC MFMAs
__global__ __launch_bounds__(256, OCC) void k4x4(float* out, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
  __syncthreads();
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 2.f;
  int addr = threadIdx.x & 63;
  float extra = 0.f;
  for (int it = 0; it < iters; ++it) {
    float av[LDS_PER > 0 ? LDS_PER : 1];
#pragma unroll
    for (int l = 0; l < LDS_PER; ++l) av[l] = lds[(addr + 64 * l + it) & 4095];
#pragma unroll
    for (int v = 0; v < VALU_PER; ++v) extra = extra * 1.0001f + (float)v;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      float x = LDS_PER > 0 ? av[i % (LDS_PER > 0 ? LDS_PER : 1)] : a;
      acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, b, acc[i], 0, 0, 0);
    }
  }
  float s = extra;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// same operand count through wider reads: NV ds_read_b128 (W = 4) or ds_read_b64 (W = 2) per NACC MFMAs
template <int NACC, int NV, int W>
__global__ __launch_bounds__(256, 2) void k4x4w(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed + i;
  __syncthreads();
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  float b = seed * 2.f;
  int addr = (threadIdx.x & 63) * W;
  for (int it = 0; it < iters; ++it) {
    float av[NV * W];
#pragma unroll
    for (int l = 0; l < NV; ++l) {
      const float* p = &lds[(addr + 64 * W * l + W * it) & (8191 & ~(W - 1))];
      if constexpr (W == 4) { f4 t = *(const f4*)p; av[4 * l] = t[0]; av[4 * l + 1] = t[1]; av[4 * l + 2] = t[2]; av[4 * l + 3] = t[3]; }
      else { typedef float f2 __attribute__((ext_vector_type(2))); f2 t = *(const f2*)p; av[2 * l] = t[0]; av[2 * l + 1] = t[1]; }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i % (NV * W)], b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256, 2) void k16(float* out, int iters, float seed) {
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 2.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, int wg_per_cu, double flop_per_mfma, int nacc, float* out) {
  const int iters = 4000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma = (double)grid * 4 * iters * nacc;
  printf("%-34s wg/cu %d  %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, mfma * flop_per_mfma / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  for (int w = 1; w <= 4; ++w) {
    run("4x4x1 x14 regs only (occ 4)", k4x4<14, 0, 0, 4>, w, 512, 14, out);
    run("4x4x1 x14 + 4 ds_read (occ 4)", k4x4<14, 4, 0, 4>, w, 512, 14, out);
    run("4x4x1 x14 + 8 ds_read (occ 4)", k4x4<14, 8, 0, 4>, w, 512, 14, out);
    run("4x4x1 x14 + 9 ds_read (occ 4)", k4x4<14, 9, 0, 4>, w, 512, 14, out);
  }
  for (int w = 1; w <= 2; ++w) {
    run("4x4x1 x28 regs only", k4x4<28, 0, 0>, w, 512, 28, out);
    run("4x4x1 x28 + 8 ds_read", k4x4<28, 8, 0>, w, 512, 28, out);
    run("4x4x1 x28 + 16 ds_read", k4x4<28, 16, 0>, w, 512, 28, out);
    run("4x4x1 x28 + 16 ds_read + 8 valu", k4x4<28, 16, 8>, w, 512, 28, out);
    run("4x4x1 x28 + 28 ds_read", k4x4<28, 28, 0>, w, 512, 28, out);
    run("4x4x1 x28 + 4 ds_read_b128", k4x4w<28, 4, 4>, w, 512, 28, out);
    run("4x4x1 x28 + 2 ds_read_b128", k4x4w<28, 2, 4>, w, 512, 28, out);
    run("4x4x1 x28 + 8 ds_read_b64", k4x4w<28, 8, 2>, w, 512, 28, out);
    run("4x4x1 x28 + 4 ds_read_b64", k4x4w<28, 4, 2>, w, 512, 28, out);
    run("4x4x1 x8 regs only", k4x4<8, 0, 0>, w, 512, 8, out);
    run("4x4x1 x4 regs only", k4x4<4, 0, 0>, w, 512, 4, out);
    run("4x4x1 x2 regs only", k4x4<2, 0, 0>, w, 512, 2, out);
    run("4x4x1 x1 regs only", k4x4<1, 0, 0>, w, 512, 1, out);
    run("4x4x1 x12 regs only", k4x4<12, 0, 0>, w, 512, 12, out);
    run("4x4x1 x16 regs only", k4x4<16, 0, 0>, w, 512, 16, out);
    run("16x16x4 x8 regs only", k16<8>, w, 2048, 8, out);
  }
  return 0;
}
