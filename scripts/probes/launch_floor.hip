// What does a dependent, (nearly) empty kernel cost inside a captured graph, by launch geometry?  A chain of 200 launches of one
// geometry is captured and replayed; time per launch = replay time / 200.  `touch` = every block reads and writes one cache line of a
// buffer the previous launch wrote (a real producer -> consumer dependence through memory).
// build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(float* p, int touch) {
    extern __shared__ char sm[];
    if (touch) { float v = p[(blockIdx.x * 32 + (threadIdx.x & 31)) % 65536]; if (threadIdx.x < 32) p[(blockIdx.x * 32 + threadIdx.x) % 65536] = v + 1.f; }
}
// the same with 256 VGPRs allocated per lane (two waves per SIMD), like the attention / GEMM kernels
__global__ __launch_bounds__(256, 2) void k_fat(float* p, int touch) {
    extern __shared__ char sm[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (touch) { float v = p[(blockIdx.x * 32 + (threadIdx.x & 31)) % 65536]; if (threadIdx.x < 32) p[(blockIdx.x * 32 + threadIdx.x) % 65536] = v + 1.f; }
}
int main() {
    float* p; hipMalloc(&p, 65536 * 4); hipMemset(p, 0, 65536 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 200;
    for (int touch = 0; touch < 2; ++touch)
        for (int lds : {0, 65536})
            for (int thr : {256, 896})
                for (int grid : {64, 256, 512, 576, 2048}) {
                    hipGraph_t g; hipGraphExec_t ge;
                    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
                    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(thr), lds, s, p, touch);
                    hipStreamEndCapture(s, &g);
                    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
                    hipEventRecord(e0, s);
                    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
                    hipEventRecord(e1, s); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    printf("touch %d  lds %5d  threads %4d  grid %5d : %6.2f us per launch\n", touch, lds, thr, grid, ms * 1e3 / (5 * N));
                    hipGraphExecDestroy(ge); hipGraphDestroy(g);
                }
    hipFuncSetAttribute((const void*)k_fat, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int grid : {64, 256, 512, 2048}) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_fat, dim3(grid), dim3(256), 65536, s, p, 1);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("256 VGPRs, 64 KiB LDS, touch 1, threads 256, grid %5d : %6.2f us per launch\n", grid, ms * 1e3 / (5 * N));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
