// Buffer-instruction staging loads for the z-marching tiled kernels -- gfx950.
#pragma once
#include <hip/hip_runtime.h>

// Staging loads go through buffer instructions: out-of-range elements (image border, planes outside the volume) come back
// as 0 from the hardware bounds check instead of a zero fill + exec-masked branch per element -- on this SIMD every
// instruction issued costs the matrix pipe ~4.6 cycles (tools/micro/mfma_rate.hip), and those were a third of the staging code.
// One resource per z plane: num_records = 0 makes the whole plane read as zeros.
#define URSN_OOB_OFFSET 0x80000000u   // byte offset of an element outside the tile's image footprint: beyond any plane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ursn_plane_rsrc(const float* plane, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane), 0, bytes, 0x00020000);
}
typedef float wgb_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wgb_u32x4 __attribute__((ext_vector_type(4)));
// (the z plane cannot go into the scalar offset with num_records = one plane: measured, planes z > 0 then read as zeros)
__device__ __forceinline__ wgb_f32x4 ursn_buffer_load_f4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(wgb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float ursn_buffer_load_f1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// an out-of-range store is dropped by the hardware (no exec-masked branch around it)
__device__ __forceinline__ void ursn_buffer_store_f4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, wgb_f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wgb_u32x4, v), r, byte_off, 0, 0);
}

// ---- byte-typed forms (bf16 plan) ---------------------------------------------------------------------------------------
typedef unsigned bst_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bst_u32x2 __attribute__((ext_vector_type(2)));
#define URSN_OOB_BYTES 0x40000000u   // offset marker of an element outside the image: past any plane, and so is twice it
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ursn_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bst_u32x4 ursn_bload_b128(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
__device__ __forceinline__ bst_u32x2 ursn_bload_b64(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
__device__ __forceinline__ unsigned ursn_bload_b32(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
__device__ __forceinline__ unsigned ursn_bload_u8(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b8(r, off, 0, 0); }
// an out-of-range store is dropped by the hardware: no exec-masked branch around it
__device__ __forceinline__ void ursn_bstore_b64(bst_u32x2 v, __amdgpu_buffer_rsrc_t r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 0); }
// LDS-DMA through the buffer path: lane l's 16 bytes land at lds + 16 l; an out-of-range lane writes zeros
__device__ __forceinline__ void ursn_bload_lds_b128(__amdgpu_buffer_rsrc_t r, void* lds, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
// ... with a scalar offset on top of the per-lane one (a channel chunk inside the voxel); the bounds check covers the sum
__device__ __forceinline__ void ursn_bload_lds_b128_so(__amdgpu_buffer_rsrc_t r, void* lds, unsigned off, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, soff, 0, 0);
}
