// Does the MFMA shape change the clock the chip sustains on random operands?  Pure register MFMA loops (no memory traffic):
// 16x16x32 bf16 (what the GEMM uses: 2 KB of operand registers read per 16 K flop) vs 32x32x16 (2 KB per 32 K flop), operands
// random bf16 / zeros, 8 waves per CU on all CUs, 8 independent accumulator chains per wave.
// build: hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + i) & 0xfffff]);
        b[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + 4 + i) & 0xfffff]);
    }
    float r = 0.f;
    if (SHAPE == 16) {
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 1) & 3], c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f32x16 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i + 1) & 3], c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) r += c[i][j];
    }
    if (r == 1.2345e-33f) out[0] = r;
}

int main() {
    const long n = 1 << 20;
    uint4* src; float* out;
    hipMalloc(&src, n * 16); hipMalloc(&out, 4);
    unsigned* h = (unsigned*)malloc(n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: random bf16 in [-2, 2) (random mantissa and sign, exponent within a few binades); mode 1: zeros
        srand(1);
        for (long i = 0; i < n * 4; ++i) {
            unsigned lo = (rand() & 0x80ff) | ((0x3c + (rand() & 3)) << 8 & 0x7f00), hi = (rand() & 0x80ff) | ((0x3c + (rand() & 3)) << 8 & 0x7f00);
            h[i] = mode ? 0u : (lo | (hi << 16));
        }
        hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
        for (int shape : {16, 32}) {
            const int iters = 40000;
            const double flop_per_iter = (shape == 16) ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16;
            auto launch = [&]() {
                if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(512), dim3(512), 0, 0, src, out, iters);
                else hipLaunchKernelGGL(k<32>, dim3(512), dim3(512), 0, 0, src, out, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 5.0 * 512 * 8 * iters * flop_per_iter;
            printf("%s operands, mfma %s: %7.1f ms  %7.1f TF/s\n", mode ? "zero  " : "random", shape == 16 ? "16x16x32" : "32x32x16", ms, flops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
