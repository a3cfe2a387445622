// Can a weight-streaming decode GEMM hide its launch + first-load latency behind the small dependent kernel in front of it?
//
// Decode layer chain (rollout.py): ... o-proj GEMM -> norm (64 rows, ~7 us) -> gate|up GEMM (272 MB of weights) -> ...  Today the GEMM
// is launched after the norm (graph edge).  "early" form: the GEMM sits in a PARALLEL graph branch that starts together with the
// norm; its workgroups become resident, request their first weight slices (which do not depend on the norm) and then wait on a
// device flag that the norm's blocks raise (release) when their rows are written; bounded spin (never hangs: after ~20 ms a block
// gives up and poisons the result).  28 (norm, GEMM) pairs per graph = one token step's worth; time per pair, both forms, same
// kernels.  The stand-in GEMM streams its share of the weights with non-temporal 16-byte loads and dots them with the norm's output
// (HBM-bound like gemm_skinny_kernel, 512 workgroups x 256 threads, two per CU).
// build: hipcc -O3 --offload-arch=gfx950 early_launch.hip -o early_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int ROWS = 64, H = 3584;
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ uint4 nt_load(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// producer: RMS-normalises row b of x into h (bf16), then raises the flag (one arrival per block)
__global__ __launch_bounds__(256) void k_norm(const float* __restrict__ x, bf16_t* __restrict__ h, unsigned* flag) {
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    float ss = 0.f, v[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) { v[i] = x[b * H + i * 256 + t]; ss += v[i] * v[i]; }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((t & 63) == 0) red[t >> 6] = ss;
    __syncthreads();
    const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) / H + 1e-6f);
#pragma unroll
    for (int i = 0; i < 14; ++i) h[b * H + i * 256 + t] = (bf16_t)(__float_as_uint(v[i] * r) >> 16);
    if (flag) {
        __threadfence();                                   // release: this block's rows are visible device-wide
        __syncthreads();
        if (t == 0) atomicAdd(flag, 1u);
    }
}

// consumer: block j streams weight rows [j*rows_per, ..) x H (bf16) and writes y[row] = sum_k w[row][k] * h[row & 63][k]
__global__ __launch_bounds__(256, 2) void k_gemm(const bf16_t* __restrict__ w, const bf16_t* __restrict__ h, float* __restrict__ y,
                                                 int rows_per, const unsigned* flag, unsigned target, int* poisoned) {
    const int t = threadIdx.x;
    const long row0 = (long)blockIdx.x * rows_per;
    // first weight slice: independent of the producer -- issued BEFORE the wait in the early form
    const uint4* wp = (const uint4*)(w + row0 * H);
    uint4 pre[2];
    pre[0] = nt_load(wp + t);
    pre[1] = nt_load(wp + 256 + t);
    if (flag) {
        if (t == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > 2000000) { *poisoned = 1; break; }      // 100 MHz counter: 20 ms
            }
            __threadfence();                               // acquire
        }
        __syncthreads();
    }
    // the dependent read: one activation value per thread (what the real GEMM stages into LDS), then the weight stream:
    // 8 x 16-byte non-temporal loads in flight per thread, folded with integer adds (the stand-in is HBM-bound by construction)
    const float hv = bf2f(h[(blockIdx.x & 63) * H + t]);
    const uint4* wr = (const uint4*)(w + row0 * H);
    const int n16 = rows_per * H / 8;                       // 16-byte chunks of this block's weight share
    unsigned fold = pre[0].x + pre[1].y;
    for (int c = 512 + t; c < n16; c += 256 * 8) {
        uint4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = (c + u * 256 < n16) ? nt_load(wr + c + u * 256) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) fold += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    }
    float acc = hv * (float)(fold & 0xff);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((t & 63) == 0) atomicAdd(&y[blockIdx.x], acc);
}

int main(int argc, char** argv) {
    const int N = 37888, PAIRS = 28, GRID = 512;            // gate|up: 37888 weight rows x 3584
    const int rows_per = N / GRID;                          // 74
    float *x, *y; bf16_t *h, *w; unsigned* flag; int* poisoned;
    hipMalloc(&x, ROWS * H * 4); hipMalloc(&h, ROWS * H * 2); hipMalloc(&y, GRID * 4);
    const size_t wbytes = (size_t)N * H * 2;
    const int NW = 4;                                       // rotate weight copies: no Infinity-Cache residency between pairs
    hipMalloc(&w, wbytes * NW); hipMalloc(&flag, 4); hipMalloc(&poisoned, 4);
    std::vector<float> hx(ROWS * H);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemset(w, 0x3c, wbytes * NW); hipMemset(y, 0, GRID * 4); hipMemset(flag, 0, 4); hipMemset(poisoned, 0, 4);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<hipEvent_t> ev(4 * PAIRS);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);

    for (int form = 0; form < 3; ++form) {                  // 0: chain; 1: early-launched consumer with flag; 2: GEMM only (no norm)
        hipGraph_t g; hipGraphExec_t ge;
        hipMemset(flag, 0, 4);
        hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < PAIRS; ++i) {
            const bf16_t* wi = w + (size_t)(i % NW) * N * H;
            if (form == 0) {
                hipLaunchKernelGGL(k_norm, dim3(ROWS), dim3(256), 0, s0, x, h, (unsigned*)nullptr);
                hipLaunchKernelGGL(k_gemm, dim3(GRID), dim3(256), 0, s0, wi, h, y, rows_per, (const unsigned*)nullptr, 0u, poisoned);
            } else if (form == 1) {
                // (the flag counts arrivals monotonically within a replay: pair i waits for 64 (i + 1); reset before every replay)
                // fork: the consumer starts together with the producer (both after the previous pair), joins back into s0
                hipEventRecord(ev[4 * i], s0); hipStreamWaitEvent(s1, ev[4 * i], 0);
                hipLaunchKernelGGL(k_gemm, dim3(GRID), dim3(256), 0, s1, wi, h, y, rows_per, (const unsigned*)flag, 64u * (i + 1), poisoned);
                hipLaunchKernelGGL(k_norm, dim3(ROWS), dim3(256), 0, s0, x, h, flag);
                hipEventRecord(ev[4 * i + 1], s1); hipStreamWaitEvent(s0, ev[4 * i + 1], 0);
            } else {
                hipLaunchKernelGGL(k_gemm, dim3(GRID), dim3(256), 0, s0, wi, h, y, rows_per, (const unsigned*)nullptr, 0u, poisoned);
            }
        }
        hipStreamEndCapture(s0, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const int REPS = 6;
        float best = 1e9f;
        for (int r = 0; r < REPS; ++r) {
            hipMemsetAsync(flag, 0, 4, s0);
            hipEventRecord(e0, s0);
            hipGraphLaunch(ge, s0);
            hipEventRecord(e1, s0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        int hp = 0; hipMemcpy(&hp, poisoned, 4, hipMemcpyDeviceToHost);
        printf("form %d (%s): %7.2f us per pair (%d pairs, best of %d replays)%s\n", form,
               form == 0 ? "norm -> GEMM chain" : form == 1 ? "GEMM launched with the norm, waits on its flag" : "GEMM alone", best * 1e3 / PAIRS,
               PAIRS, REPS - 1, hp ? "  [POISONED: a spin timed out]" : "");
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
