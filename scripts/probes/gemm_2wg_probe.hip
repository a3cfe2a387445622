// Probe (round 6): the GEMM design DEAD_ENDS.md keeps naming -- a 256 x 128 tile with TWO workgroups per CU, so that
// one workgroup's epilogue / pipeline fill runs under the other's K loop.  This file measures the design's ABORT CRITERION first: is its K
// loop alone within 3 % of the production 256 x 256 kernel's K loop?  (If not, no epilogue overlap can pay for it.)
//
//   tile 256 (M) x 128 (N) x 64 (K), 4 waves as 2 x 2, wave = 128 x 64 = 8 x 4 MFMA 16x16x32 fragments (the production wave tile: same
//   accumulators, same LDS fragment traffic per flop), __launch_bounds__(256, 2): two workgroups = 8 waves per CU as in production.
//   LDS per workgroup 72 KiB = 3 A slots of 16 KiB + 3 B slots of 8 KiB; "A unit h" = the 64-row quadrant h of both wave rows,
//   "B unit h" = the 32-column quadrant h of both wave columns (production's half-tile images, B half as wide).
//   A K tile is 4 phases = the C quadrants (0,0) (0,1) (1,1) (1,0), 16 MFMAs each; reads: A-lo + B-lo | B-hi | A-hi | -.
//   global_load_lds DMA, issued per phase into a slot last read >= 2 phases ago, counted vmcnt:
//       phase 0: A-lo(t+1)   phase 1: B-lo(t+1)   phase 2: B-hi(t+1), A-hi(t+1)   phase 3: -        vmcnt 8 | 6 | - | 6
//   ONE barrier per phase (the production kernel's second barrier staggers its two wave rows so that the two waves of a SIMD alternate
//   LDS and MFMA segments; here the two waves of a SIMD belong to different workgroups and de-phase by themselves).
//   Slots rotate with period 3 units = 1.5 tiles: the tile body is instantiated for the three alignments.
//   L2 -> LDS traffic per flop is 1.5 x production's ((256 + 128) x 64 per 256 x 128 x 64 against (256 + 256) x 64 per 256 x 256 x 64).
//
// build: hipcc -O3 --offload-arch=gfx950 gemm_2wg_probe.hip -o gemm_2wg_probe -L../../spacer_amd -lspacer_hip -Wl,-rpath,'$ORIGIN/../../spacer_amd'
// run:   ./gemm_2wg_probe      (production reference = libspacer_hip.so's spacer_gemm_bf16_nt, alternating with the probe in one process)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BK = 64;
constexpr int AU = 128 * BK * 2;      // A unit, 16 KiB
constexpr int BU = 64 * BK * 2;       // B unit, 8 KiB
constexpr int LDS_BYTES = 3 * AU + 3 * BU;

struct Args {
    const bf16_t* A; const bf16_t* B; float* C;
    int M, N, K, lda, ldb, ldc, tiles_n, store, epi;
    bf16_t* Cb; int ldcb;
    int skew_ticks;      // blocks of the second half of the grid start this many 100 MHz ticks late (de-phases the two residents of a CU)
};

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 hw2; typedef __attribute__((ext_vector_type(2))) float f2;
    const f2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hw2));
}
__device__ __forceinline__ void xpose4_rows(uint32_t (&x)[4]) {
    const auto a = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false);
    const auto c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    x[0] = c[0]; x[2] = c[1]; x[1] = d[0]; x[3] = d[1];
}

template <int TWO_BARRIERS, int PREFETCH>
__global__ __launch_bounds__(256, 2) void gemm_2wg_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 A slots][3 B slots]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nt = g.K / BK;
    const int items = (g.M / 256) * g.tiles_n;
    if (g.skew_ticks && blockIdx.x >= gridDim.x / 2) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)g.skew_ticks) __builtin_amdgcn_s_sleep(32);
    }
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
    // neighbouring blocks (likely the two residents of a CU / the same XCD) take the two N halves of one 256 x 256 super-tile: they share A in L2
    const int tm = (item >> 1) / (g.tiles_n >> 1), tn = ((item >> 1) % (g.tiles_n >> 1)) * 2 + (item & 1);
    const int m0 = tm * 256, n0 = tn * 128;
    unsigned offA[2][4], offB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hr = (wave * 4 + i) * 8 + (lane >> 3);                // A image row 0..127
            const int chunk = (lane & 7) ^ ((hr >> 1) & 7);
            const int ra = m0 + (hr >> 6) * 128 + h * 64 + (hr & 63);
            offA[h][i] = (unsigned)((long)ra * g.lda + chunk * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (wave * 2 + i) * 8 + (lane >> 3);                // B image row 0..63
            const int chunk = (lane & 7) ^ ((hr >> 1) & 7);
            const int rb = n0 + (hr >> 5) * 64 + h * 32 + (hr & 31);
            offB[h][i] = (unsigned)((long)rb * g.ldb + chunk * 8);
        }
    }
    auto stageA = [&](int slot, int h, int t) {
        const int kt = t < nt ? t : nt - 1;
        const bf16_t* base = g.A + kt * BK;
        char* dst = smem + slot * AU + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + offA[h][i]),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    auto stageB = [&](int slot, int h, int t) {
        const int kt = t < nt ? t : nt - 1;
        const bf16_t* base = g.B + kt * BK;
        char* dst = smem + 3 * AU + slot * BU + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + offB[h][i]),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    int fl = lane;
    asm volatile("" : "+v"(fl));
    const int fsw = (fl & 15) >> 1;
    const int fo0 = (fl & 15) * 128 + (((fl >> 4) ^ fsw) << 4);
    const int fo1 = fo0 ^ 64;
    const int aoff = wr * 64 * 128, boff = wc * 32 * 128;
    auto fragA = [&](const char* img, int i, int kk) -> bf16x8 { return *(const bf16x8*)(img + aoff + i * 2048 + (kk ? fo1 : fo0)); };
    auto fragB = [&](const char* img, int j, int kk) -> bf16x8 { return *(const bf16x8*)(img + boff + j * 2048 + (kk ? fo1 : fo0)); };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue = "tile -1": A-lo(0) -> A slot 0, B-lo(0) -> B slot 0, B-hi(0) -> B slot 1, A-hi(0) -> A slot 1
    stageA(0, 0, 0); stageB(0, 0, 0); stageB(1, 1, 0); stageA(1, 1, 0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 a[4][2], blo_e[2][2], blo_o[2][2], bhi[2][2];
    if (PREFETCH) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) blo_e[j][kk] = fragB(smem + 3 * AU, j, kk);
    }
    // S = slot of A-lo(t) and of B-lo(t); A-hi(t) / B-hi(t) in S + 1, A-lo(t+1) / B-lo(t+1) -> S + 2, A-hi(t+1) / B-hi(t+1) -> S   (mod 3)
    auto tile = [&](auto SS, const int t, bf16x8 (&blo)[2][2], bf16x8 (&blo_next)[2][2]) {
        constexpr int S0 = decltype(SS)::value, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;
        const char* imgA = smem;
        const char* imgB = smem + 3 * AU;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p == 0) {
                if (!PREFETCH) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < 2; ++j) blo[j][kk] = fragB(imgB + S0 * BU, j, kk);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i][kk] = fragA(imgA + S0 * AU, i, kk);
            } else if (p == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) bhi[j][kk] = fragB(imgB + S1 * BU, j, kk);
            } else if (p == 2) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i][kk] = fragA(imgA + S1 * AU, i, kk);
            } else if (PREFETCH) {                                   // next tile's B-lo fragments (landed: waited for in phase 2)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) blo_next[j][kk] = fragB(imgB + S2 * BU, j, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p == 0) { stageA(S2, 0, t + 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            if (p == 1) { stageB(S2, 0, t + 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            if (p == 2) { stageB(S0, 1, t + 1); stageA(S0, 1, t + 1); if (PREFETCH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            if (p == 3 && !PREFETCH) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            const int mq = p >> 1;
            const bool hi = (p == 1 || p == 2);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mq * 4 + i][(hi ? 2 : 0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            hi ? bhi[j][kk] : blo[j][kk], a[i][kk], acc[mq * 4 + i][(hi ? 2 : 0) + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (TWO_BARRIERS) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // (with PREFETCH the B-lo register sets alternate per tile: the body is instantiated for 3 slot alignments x 2 parities)
    for (int t0 = 0; t0 < nt; t0 += 6) {
        tile(std::integral_constant<int, 0>{}, t0, blo_e, blo_o);
        if (t0 + 1 < nt) tile(std::integral_constant<int, 2>{}, t0 + 1, blo_o, blo_e);
        if (t0 + 2 < nt) tile(std::integral_constant<int, 1>{}, t0 + 2, blo_e, blo_o);
        if (t0 + 3 < nt) tile(std::integral_constant<int, 0>{}, t0 + 3, blo_o, blo_e);
        if (t0 + 4 < nt) tile(std::integral_constant<int, 2>{}, t0 + 4, blo_e, blo_o);
        if (t0 + 5 < nt) tile(std::integral_constant<int, 1>{}, t0 + 5, blo_o, blo_e);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // (next item's prologue writes slots other waves may still read)

    if (g.store) {                                          // verification: fp32 fragments straight out
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wr * 128 + i * 16 + (lane & 15), n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                *(float4*)(g.C + (long)m * g.ldc + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
    } else if (g.epi) {                                     // the production bf16 direct epilogue (register transpose, 2 x 16-byte stores per lane and row)
        const int ncol = n0 + wc * 64 + (lane >> 4) * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] = pack_bf2(acc[i][j][0], acc[i][j][1]); hi[j] = pack_bf2(acc[i][j][2], acc[i][j][3]); }
            xpose4_rows(lo); xpose4_rows(hi);
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            bf16_t* c = g.Cb + (long)m * g.ldcb + ncol;
            *(uint4*)c = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            *(uint4*)(c + 8) = make_uint4(lo[2], hi[2], lo[3], hi[3]);
        }
    } else {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (r == 1.2345e-33f) g.C[0] = r;
    }
    }
}

static float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

extern "C" {
typedef struct spacer_plan { int struct_bytes, gemm_tile, gemm_no_split, skinny_blocks, skinny_no_balance, cus, skinny_skew; } spacer_plan;
typedef struct spacer_gemm_epilogue { const void* bias; const void* residual; long ldr; int out_f32; int act; float alpha; void* workspace;
                                      long workspace_bytes; const spacer_plan* plan; } spacer_gemm_epilogue;
int spacer_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, const spacer_gemm_epilogue* epi, void* stream);
long spacer_gemm_workspace_bytes(void);
const char* spacer_last_error(void);
}

template <int TB, int PF>
static double run(const Args& g, int grid, int reps) {
    CK(hipFuncSetAttribute((const void*)gemm_2wg_kernel<TB, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gemm_2wg_kernel<TB, PF>), dim3(grid), dim3(256), LDS_BYTES, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_2wg_kernel<TB, PF>), dim3(grid), dim3(256), LDS_BYTES, 0, g);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e-3 / reps;
}
static double run_variant(int v, const Args& g, int grid, int reps) {
    switch (v) {
        case 0: return run<0, 0>(g, grid, reps);
        case 1: return run<1, 0>(g, grid, reps);
        case 2: return run<0, 1>(g, grid, reps);
        default: return run<1, 1>(g, grid, reps);
    }
}
static const char* VN[4] = {"1 barrier/phase          ", "2 barriers/phase         ", "1 barrier + B-lo prefetch", "2 barriers + B-lo prefetch"};

int main(int argc, char** argv) {
    // ---- correctness on a small problem (K = 3, 4, 5, 7, 13 tiles: every slot alignment / register parity and the redundant tail stages)
    for (int K : {192, 256, 320, 448, 832}) {
        const int M = 512, N = 256;
        std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
        srand(K);
        for (auto& v : hA) v = f2bf((rand() % 2001 - 1000) / 1000.f);
        for (auto& v : hB) v = f2bf((rand() % 2001 - 1000) / 1000.f);
        bf16_t *dA, *dB; float* dC;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        Args g = {dA, dB, dC, M, N, K, K, K, N, N / 128, 1, 0, nullptr, 0, 0};
        for (int v = 0; v < 4; ++v) {
            CK(hipMemset(dC, 0, (size_t)M * N * 4));
            run_variant(v, g, 3, 1);                                  // 3 blocks walk the 4 tiles: the persistent loop too
            std::vector<float> hC((size_t)M * N);
            CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int m = 0; m < M; m += 7)
                for (int n = 0; n < N; n += 3) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
                    worst = fmax(worst, fabs(s - hC[(size_t)m * N + n]));
                }
            if (worst >= 1e-3 || K == 832) printf("check K=%d %s: max |C - ref| = %.3e %s\n", K, VN[v], worst, worst < 1e-3 ? "ok" : "WRONG");
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    // ---- exact rounds: 8192 x 6144 = 1536 tiles of 256 x 128 = 3 rounds of 512 workgroups = 768 tiles of 256 x 256 = 3 rounds of 256 (production);
    //      4096 x 4096 = one round of either.  Production (libspacer_hip.so, bf16 output = persistent, direct epilogue) and the probe ALTERNATE.
    void* ws; CK(hipMalloc(&ws, spacer_gemm_workspace_bytes()));
    spacer_plan plan = {(int)sizeof(spacer_plan), 0, 0, 0, 0, 0, 0};
    spacer_gemm_epilogue epi = {nullptr, nullptr, 0, 0, 0, 1.f, ws, spacer_gemm_workspace_bytes(), &plan};
    struct Shape { int M, N, K; } shapes[] = {{4096, 4096, 3584}, {8192, 6144, 3584}, {8192, 6144, 18944}, {4096, 4096, 18944}};
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        bf16_t *dA, *dB; float* dC; bf16_t* dCb;
        std::vector<bf16_t> h((size_t)(M > N ? M : N) * K);
        srand(1);
        for (auto& v : h) v = f2bf((rand() % 2001 - 1000) / 1000.f);
        CK(hipMalloc(&dA, (size_t)M * K * 2)); CK(hipMalloc(&dB, (size_t)N * K * 2)); CK(hipMalloc(&dC, 64)); CK(hipMalloc(&dCb, (size_t)M * N * 2));
        CK(hipMemcpy(dA, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
        for (auto& v : h) v = f2bf((rand() % 2001 - 1000) / 1000.f);
        CK(hipMemcpy(dB, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
        const double fl = 2.0 * M * N * K;
        const int tile_ticks = (int)(K / 64 * 1.6 * 100 / 2);          // half a tile of the probe in 100 MHz ticks
        double best[16]; for (double& b : best) b = 1e9;
        const int REPS = 5, ROUNDS = 4;
        for (int r = 0; r < ROUNDS; ++r) {
            {   // production
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                if (spacer_gemm_bf16_nt(dA, K, dB, K, dCb, N, M, N, K, &epi, nullptr)) { printf("production: %s\n", spacer_last_error()); return 1; }
                CK(hipEventRecord(e0));
                for (int i = 0; i < REPS; ++i) spacer_gemm_bf16_nt(dA, K, dB, K, dCb, N, M, N, K, &epi, nullptr);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best[0] = fmin(best[0], ms * 1e-3 / REPS);
            }
            for (int v = 0; v < 4; ++v)
                for (int mode = 0; mode < 3; ++mode) {               // 0: K loop only, 1: bf16 direct epilogue, 2: epilogue + second half of the grid starts half a tile late
                    Args g = {dA, dB, dC, M, N, K, K, K, N, N / 128, 0, mode > 0, dCb, N, mode == 2 ? tile_ticks : 0};
                    best[1 + v * 3 + mode] = fmin(best[1 + v * 3 + mode], run_variant(v, g, 512, REPS));
                }
        }
        printf("\n%d x %d x %d (best of %d x %d launches, alternating)\n  production 256 x 256, 1 workgroup / CU, bf16 direct epilogue: %8.1f us = %7.1f TF/s\n", M, N, K, ROUNDS, REPS,
               best[0] * 1e6, fl / best[0] / 1e12);
        for (int v = 0; v < 4; ++v)
            printf("  2wg %s: K loop only %8.1f us = %7.1f TF/s (%.3f us / K tile / round) | + epilogue %8.1f us = %7.1f TF/s | + skewed start %8.1f us = %7.1f TF/s\n", VN[v],
                   best[1 + v * 3] * 1e6, fl / best[1 + v * 3] / 1e12, best[1 + v * 3] * 1e6 / (K / 64) / ((double)M * N / (256 * 128) / 512),
                   best[2 + v * 3] * 1e6, fl / best[2 + v * 3] / 1e12, best[3 + v * 3] * 1e6, fl / best[3 + v * 3] / 1e12);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dCb));
    }
    return 0;
}
