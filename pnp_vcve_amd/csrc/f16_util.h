// Device helpers and A-tile geometry shared by the fp16-operand conv kernels (conv_f16.hip, conv_f16x3.hip).  gfx950 only.
#pragma once
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// 8x16-pixel tile of a 4-wave group; its fp16 A tile in LDS: 10 rows x 18 pixels x (128 B + 16 B pad), row stride 0 mod 256
constexpr int TH = 8, TW = 16, PW = 18, PSB = 144, RSB = 2816;
constexpr int NPIX = (TH + 2) * PW;               // 180 halo pixels
constexpr int A_BYTES = (TH + 2) * RSB;           // 28160 per group
constexpr int UNIT = 1024;                        // one B fragment of the wave: 64 lanes x 8 halfs

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for
// the halo / residual prefetches this kernel deliberately keeps in flight across tiles.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Every per-pixel global access goes through a buffer descriptor: an out-of-image lane gets the offset OOB,
// which the hardware range check turns into "load 0" / "drop the store" -- zero padding, ragged tiles and an
// absent residual (a descriptor of 0 bytes) cost no branch, so a tile's body is one basic block and the
// compiler's vmcnt bookkeeping stays exact across the prefetches in flight.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // readfirstlane: the size must be provably wave-uniform or every access becomes a waterfall loop
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0);
}

__device__ __forceinline__ f32x4 clamp_h(f32x4 v) {       // to fp16's finite range
    return __builtin_elementwise_min(__builtin_elementwise_max(v, (f32x4)(-65504.f)), (f32x4)(65504.f));
}
__device__ __forceinline__ h4 to_h4(f32x4 v) {
    // saturate instead of overflowing to inf (a single out-of-range activation would poison the frame)
    v = __builtin_elementwise_min(__builtin_elementwise_max(v, (f32x4)(-65504.f)), (f32x4)(65504.f));
    return __builtin_convertvector(v, h4);
}

}  // namespace
