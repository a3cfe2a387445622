// Experiment: what weight-streaming rate does a wave-per-16KB register-double-set pattern reach on MI355X?
// build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe ; run: ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT, int UN>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, unsigned* __restrict__ out, long n_chunks, int chunks_per_wave, int lds_pad) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    // each wave streams chunks_per_wave chunks of UN KiB, contiguous
    const u32x4* p = src + wave * (long)chunks_per_wave * UN * 64 + lane;
    u32x4 a[UN], b[UN];
    auto ld = [&](u32x4 (&w)[UN], int c) {
#pragma unroll
        for (int u = 0; u < UN; ++u) w[u] = NT ? __builtin_nontemporal_load(p + ((long)c * UN + u) * 64) : p[((long)c * UN + u) * 64];
    };
    auto use = [&](const u32x4 (&w)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    };
    ld(a, 0);
    int c = 0;
    while (true) {
        if (c + 1 < chunks_per_wave) ld(b, c + 1);
        use(a);
        if (++c >= chunks_per_wave) break;
        if (c + 1 < chunks_per_wave) ld(a, c + 1);
        use(b);
        if (++c >= chunks_per_wave) break;
    }
    if (lds_pad < 0) smem[0] = (char)acc;
    if (acc == 0x12345678u) out[wave] = acc;
}

int main() {
    const long bytes = 4L << 30;
    u32x4* src; unsigned* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 1 << 24);
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int un, int cpw, int lds) {
        const long per_wave = (long)cpw * un * 1024;
        const long waves = bytes / per_wave;
        const int blocks = (int)(waves / 4);
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, out, 0L, cpw, lds);
        hipEventRecord(e0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, out, 0L, cpw, lds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s un=%d chunks/wave=%3d lds=%6d blocks=%7d : %.2f TB/s\n", name, un, cpw, lds, blocks, 3.0 * bytes / (ms * 1e-3) / 1e12);
    };
    for (int lds : {0, 40 * 1024, 64 * 1024}) {
        run("plain  8x1KiB sets", probe<false, 8>, 8, 4, lds);
        run("nt     8x1KiB sets", probe<true, 8>, 8, 4, lds);
        run("nt     8x1KiB sets, long", probe<true, 8>, 8, 32, lds);
        run("plain  4x1KiB sets", probe<false, 4>, 4, 8, lds);
        run("nt     16x1KiB sets", probe<true, 16>, 16, 4, lds);
    }
    return 0;
}
