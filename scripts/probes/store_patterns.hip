// How fast does a 256x256 bf16 output tile leave the CU, by store pattern?  256 workgroups x 8 waves, every wave writes its
// 128 x 64 sub-tile of successive tiles of a [M, N] bf16 matrix (N = 37888, row stride 75.8 KB), no compute:
//   rows512 : one wave instruction = ONE full tile row, 512 B contiguous (what the LDS-staged epilogue does; 8 waves x 32 rows)
//   seg128  : one instruction = 16 rows x 128 B (lane = 32 contiguous bytes issued as 2 x 16 B: what a permlane-transposed
//             direct epilogue could do -- the wave's 64 columns of 16 rows)
//   seg64x2 : one instruction = 16 rows x 64 B contiguous (lane = 16 B), two instructions complete the 128 B of a row
//   seg32   : one instruction = 16 rows x 32 B (accumulator fragments straight out)
// build: hipcc -O3 --offload-arch=gfx950 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short bf16_t;
template <int MODE>
__global__ __launch_bounds__(512) void k(bf16_t* C, long ldc, int tiles_n, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3, l15 = lane & 15, g = lane >> 4;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x + t * gridDim.x, tm = tile / tiles_n, tn = tile % tiles_n;
        bf16_t* base = C + (long)tm * 256 * ldc + tn * 256;
        const uint4 v = make_uint4(lane, t, wave, 7);
        if (MODE == 0) {
            for (int pass = 0; pass < 2; ++pass)
                for (int it = 0; it < 16; ++it) {
                    const int row = pass * 128 + it * 8 + wave;
                    *(uint2*)(base + (long)row * ldc + lane * 4) = make_uint2(v.x, v.y);
                }
        } else if (MODE == 1) {       // lane: row i*16 + l15, 16 contiguous columns (32 B) at wc*64 + g*16: two 16-byte stores
            for (int i = 0; i < 8; ++i) {
                bf16_t* p = base + (long)(wr * 128 + i * 16 + l15) * ldc + wc * 64 + g * 16;
                *(uint4*)p = v; *(uint4*)(p + 8) = v;
            }
        } else if (MODE == 2) {       // instruction A: lane-row g writes columns wc*64 + g*8 .. +7 (64 B per row), B: +32
            for (int i = 0; i < 8; ++i) {
                bf16_t* p = base + (long)(wr * 128 + i * 16 + l15) * ldc + wc * 64 + g * 8;
                *(uint4*)p = v; *(uint4*)(p + 32) = v;
            }
        } else {                      // fragments straight out: (i, j): 16 rows x 32 B
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 4; ++j)
                    *(uint2*)(base + (long)(wr * 128 + i * 16 + l15) * ldc + wc * 64 + j * 16 + g * 4) = make_uint2(v.x, v.y);
        }
    }
}
int main() {
    const int M = 10996 / 256 * 256, N = 37888, tiles_n = N / 256, tiles = (M / 256) * tiles_n, per = tiles / 256;
    bf16_t* C; hipMalloc(&C, (long)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"rows512", "seg128 (2 x 16 B per lane)", "seg64x2", "seg32"};
    for (int mode = 0; mode < 4; ++mode) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, C, (long)N, tiles_n, per);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, C, (long)N, tiles_n, per);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, C, (long)N, tiles_n, per);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, C, (long)N, tiles_n, per);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)per * 256 * 256 * 256 * 2;
        printf("%-28s %8.1f us per pass over %d tiles/WG  %5.2f TB/s  (%.2f us per tile round)\n", names[mode], ms / 5 * 1e3, per, bytes / (ms / 5 * 1e-3) / 1e12,
               ms / 5 * 1e3 / per);
    }
    return 0;
}
