// Shared helpers for libctxnerf.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ctx_nerf.h"

void ctx_set_error(const char *fmt, ...);

#define CTX_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ctx_set_error(__VA_ARGS__);        \
            return CTX_E_ARG;                  \
        }                                      \
    } while (0)

#define CTX_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            ctx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return CTX_E_LAUNCH;                                                   \
        }                                                                          \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// s_barrier plus a compiler-level memory fence.  The builtin alone is IntrNoMem to LLVM: LDS reads, LDS-DMA issues and stores may be
// moved across it at compile time (an LDS read hoisted above the barrier reads a stage other waves' DMA pieces have not landed in
// yet; seen once the steady-state K loops lost the branches that used to pin the order).  Every workgroup barrier that orders LDS
// or global traffic between waves goes through this.
__device__ __forceinline__ void ctx_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Wave-wide sum on the DPP path (no LDS crossbar): 4 DPP adds give every lane its 16-lane row total, 4 readlanes
// combine the rows.  ~40 cycles against ~6 x ds_bpermute for wave_sum; the summation order differs from wave_sum.
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    const int vi = __builtin_bit_cast(int, v);
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
