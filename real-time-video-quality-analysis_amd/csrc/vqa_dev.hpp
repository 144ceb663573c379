// vqa_dev.hpp — device-side helpers shared by the gfx950 kernels.
// Wave size is 64 on CDNA4; every cross-lane idiom below assumes it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VQA_WAVE 64

namespace vqa {

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v; // valid in lane 0
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned wave_id() { return threadIdx.x >> 6; }

// Block-wide sum of a double: result valid in thread 0.  `red` holds >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (unsigned i = 0; i < (blockDim.x + 63) / 64; i++) t += red[i];
    return t;
}

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *red)
{
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    unsigned long long t = 0;
    if (threadIdx.x == 0)
        for (unsigned i = 0; i < (blockDim.x + 63) / 64; i++) t += red[i];
    return t;
}

// A pointer that is wave-uniform by construction, pinned to scalar registers, in the global address space:
// `uniform_ptr(base)[lane_offset_u32]` compiles to global_load ... v_off, s[base] (no vector ALU address math).
typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;
__device__ __forceinline__ gptr_u8 uniform_ptr(const uint8_t *p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_u8)(((uint64_t)hi << 32) | lo);
}

// cv2.cvtColor BGR2GRAY, uint8: 15-bit fixed point (complexity_metrics.py:358 et al.)
__device__ __forceinline__ uint32_t gray_u8(uint32_t b, uint32_t g, uint32_t r)
{
    return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;
}

} // namespace vqa
