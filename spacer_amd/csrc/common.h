// Shared device/host helpers for libspacer_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spacer_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SP_WAVE 64

// ---- error plumbing (thread-local message, negative codes; see include/spacer_hip.h) ----
void spacer_set_error(const char* fmt, ...);
#define SP_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            spacer_set_error(__VA_ARGS__);     \
            return (code);                     \
        }                                      \
    } while (0)
#define SP_CHECK_LAUNCH()                                                      \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            spacer_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return SPACER_ELAUNCH;                                             \
        }                                                                      \
    } while (0)

// a caller's launch plan is read only when it declares the struct size this library was compiled with (include/spacer_hip.h)
static inline bool plan_size_ok(const spacer_plan* p) { return !p || p->struct_bytes == (int)sizeof(spacer_plan); }
#define SP_REQUIRE_PLAN(p)                                                                                              \
    SP_REQUIRE(plan_size_ok(p), SPACER_EINVAL, "spacer_plan: struct_bytes = %d but this library's spacer_plan has %d bytes (stale binding?)", \
               (p) ? (p)->struct_bytes : 0, (int)sizeof(spacer_plan))
// ... in a PREDICATE entry point (spacer_gemm_*_fused: "!= 0 means fused"; spacer_gemm_tile: 128 / 256): a rejected plan answers 0 --
// "not fused" / "no tile" -- with the error string set, never a negative code that a caller's truth test would read as "fused"
#define SP_PLAN_OR_ZERO(p)                                                                                             \
    do {                                                                                                               \
        if (!plan_size_ok(p)) {                                                                                        \
            spacer_set_error("spacer_plan: struct_bytes = %d but this library's spacer_plan has %d bytes (stale binding?)", \
                             (p) ? (p)->struct_bytes : 0, (int)sizeof(spacer_plan));                                 \
            return 0;                                                                                                  \
        }                                                                                                              \
    } while (0)

// ---- bf16 <-> f32 ----
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// f32 -> bf16 is the gfx950 hardware conversion (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN), one
// instruction per PAIR of values; a software RNE costs ~7 VALU ops per value and was a third of the attention
// kernels' VALU work.
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// ---- wave / block reductions (wave = 64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block reductions over blockDim.x threads (multiple of 64, <= 1024); `red` = 32 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
