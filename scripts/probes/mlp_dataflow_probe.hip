// Probe (round 6): the decode MLP half at 8 rows as ONE persistent dataflow launch -- gate|up -> (flag hand-off of the
// 8 x 18944 activations, weights streaming across the edge) -> down -- against the two launches the decode graph runs today.
//
// Qwen2-VL-7B decode layer, 8 rows (cfg4 = the reference script's launch shape): gate|up streams 271.6 MB of packed weights and leaves
// act bf16 [8, 18944] (303 KB); down streams 135.8 MB and needs, per workgroup, the act columns of its K range.  Production: two launches
// (46.1 + ~28 us in the decode graph, profiles/r05_cfg4_step_kernel_stats.md).  Dataflow form: 512 co-resident workgroups; workgroup j
//   (1) streams its 1/512 of the gate|up weights, publishes its 37 act columns (8 rows, padded to 40: 640 B) with write-through (sc1)
//       16-byte stores, drains them (vmcnt(0)) and arrives on the counter of its K range (64 producers per range, agent-scope atomic);
//   (2) requests the first slices of its down-projection weights (they do not depend on anything), THEN waits for its range's counter
//       (one lane polls with s_sleep, bounded: a stuck launch poisons the result instead of hanging the GPU), gathers the 40 KB of
//       activations of its range with sc1 loads, streams the rest of its weights, flushes 8 x 56 fp32 sums with atomics.
// Stand-in arithmetic (the weight bytes are folded with integer adds: both forms are HBM-bound by construction, as gemm_skinny_kernel is);
// what is measured is the EDGE: per pair, dataflow launch vs graph chain of the same two bodies, vs the same launch with the waits
// compiled out (upper bound: no dependency at all).
// build: hipcc -O3 --offload-arch=gfx950 mlp_dataflow_probe.hip -o mlp_dataflow_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int H = 3584, I = 18944, ROWS = 8, GRID = 512, PAIRS = 28;
constexpr int GU_ROWS_PER = 2 * I / GRID;          // 74 weight rows of gate|up per workgroup
constexpr int ACT_COLS = I / GRID;                 // 37 act columns per workgroup, stored padded to 40
constexpr int KR = 8, CG = GRID / KR;              // down: 8 K ranges x 64 column groups
constexpr int KR_COLS = I / KR;                    // 2368 act columns per K range = 64 producers x 37
constexpr int DN_COLS = H / CG;                    // 56 output columns per column group
constexpr int NBUF = 3;                            // act buffers (layer L uses L % 3)
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ uint4 nt_load(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_sc1(uint4* p, uint4 v) {      // write-through: visible to other XCDs once vmcnt drains
    const u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
// agent-scope relaxed atomic loads = loads that bypass the non-coherent L1 / L2 lines (sc1); the compiler tracks their completion itself
// (a first version issued them as inline asm: hipcc reused a destination register as the next load's address while the data was in flight
// -- a memory fault on the box; asynchronous results must not be asm outputs unless the wait is inside the same asm statement)
__device__ __forceinline__ unsigned long long load_sc1_u64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stream n16 16-byte chunks starting at wp (this block's share), 8 loads in flight per thread; returns a fold of the bytes
__device__ __forceinline__ unsigned stream(const uint4* wp, int n16, int t, int first) {
    unsigned fold = 0;
    for (int c = first + t; c < n16; c += 256 * 8) {
        uint4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = (c + u * 256 < n16) ? nt_load(wp + c + u * 256) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) fold += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    }
    return fold;
}

// ---- body A: gate|up share of block j -> its 640-byte act record
__device__ __forceinline__ void body_gateup(const bf16_t* __restrict__ wgu, const float* __restrict__ x, uint4* __restrict__ act_rec, int j, int t, bool write_through) {
    const float xv = x[(t & 7) * H + (t >> 3)];                        // (the real kernel stages the 8 x 3584 fp32 rows: 115 KB, L2-resident)
    const unsigned fold = stream((const uint4*)(wgu + (long)j * GU_ROWS_PER * H), GU_ROWS_PER * H / 8, t, 0);
    const float v = xv * (float)(fold & 0xff);
    if (t < 40) {
        const uint4 rec = make_uint4(__float_as_uint(v), fold, t, j);
        if (write_through) store_sc1(act_rec + t, rec); else act_rec[t] = rec;
    }
}
// ---- body B: down share of block j = (column group j % 64, K range j / 64): gather the range's 64 records, stream 56 x 2368 weights, flush
__device__ __forceinline__ void body_down(const bf16_t* __restrict__ wdn, const uint4* __restrict__ act_base, float* __restrict__ out, int j, int t,
                                          uint4 pre0, uint4 pre1, bool coherent_loads) {
    const int kr = j / CG, cg = j % CG;
    const uint4* rec = act_base + (long)kr * CG * 40;                  // 64 producers x 40 chunks = 40 KB
    unsigned af = 0;
    if (coherent_loads) {
        unsigned long long a[20];
        const unsigned long long* r8 = (const unsigned long long*)rec;
#pragma unroll
        for (int u = 0; u < 20; ++u) a[u] = load_sc1_u64(r8 + u * 256 + t);        // 40 KB per block, 20 x 8 bytes per thread in flight
#pragma unroll
        for (int u = 0; u < 20; ++u) af += (unsigned)a[u] ^ (unsigned)(a[u] >> 32);
    } else {
#pragma unroll
        for (int u = 0; u < 10; ++u) { const uint4 q = rec[u * 256 + t]; af += q.x ^ q.y; }
    }
    const uint4* wp = (const uint4*)(wdn + ((long)cg * DN_COLS * I + (long)kr * KR_COLS * DN_COLS));   // (block-contiguous stand-in layout: 265 KB)
    unsigned fold = pre0.x + pre1.y + stream(wp, DN_COLS * KR_COLS / 8, t, 512);
    float acc = (float)((fold ^ af) & 0xff);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    for (int idx = t; idx < ROWS * DN_COLS; idx += 256)                 // 8 x 56 sums per block, as the real K-split flush: 448 fp32 atomics
        atomicAdd(&out[(idx / DN_COLS) * H + cg * DN_COLS + idx % DN_COLS], acc);
}

__global__ __launch_bounds__(256, 2) void k_gateup(const bf16_t* wgu, const float* x, uint4* act) {
    body_gateup(wgu, x, act + (long)blockIdx.x * 40, blockIdx.x, threadIdx.x, false);
}
__global__ __launch_bounds__(256, 2) void k_down(const bf16_t* wdn, const uint4* act, float* out) {
    const int j = blockIdx.x, t = threadIdx.x, kr = j / CG, cg = j % CG;
    const uint4* wp = (const uint4*)(wdn + ((long)cg * DN_COLS * I + (long)kr * KR_COLS * DN_COLS));
    body_down(wdn, act, out, j, t, nt_load(wp + t), nt_load(wp + 256 + t), false);
}
// the persistent dataflow launch: `pairs` (gate|up, down) pairs; WAIT = 0: the hand-off waits compiled out (no dependency: upper bound)
template <int WAIT>
__global__ __launch_bounds__(256, 2) void k_dataflow(const bf16_t* wgu, const bf16_t* wdn, const float* x, uint4* act, float* out, unsigned* cnt,
                                                     int pairs, int nw, int* poisoned) {
    const int j = blockIdx.x, t = threadIdx.x, kr = j / CG, cg = j % CG;
    __shared__ int bad;
    for (int L = 0; L < pairs; ++L) {
        const bf16_t* wg = wgu + (size_t)(L % nw) * 2 * I * H;
        const bf16_t* wd = wdn + (size_t)(L % nw) * I * H;
        uint4* abuf = act + (size_t)(L % NBUF) * GRID * 40;
        body_gateup(wg, x, abuf + (long)j * 40, j, t, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the record has left this CU (write-through)
        __syncthreads();
        if (t == 0) __hip_atomic_fetch_add(cnt + (j * ACT_COLS) / KR_COLS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // my columns' K range
        // the down share's first weight slices do not depend on the hand-off: requested before the wait
        const uint4* wp = (const uint4*)(wd + ((long)cg * DN_COLS * I + (long)kr * KR_COLS * DN_COLS));
        const uint4 pre0 = nt_load(wp + t), pre1 = nt_load(wp + 256 + t);
        if (WAIT) {
            if (t == 0) {
                bad = 0;
                const unsigned target = (unsigned)CG * (L + 1);
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(cnt + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > 2000000) { bad = 1; *poisoned = 1; break; }     // 20 ms: never hang the box
                }
            }
            __syncthreads();
            if (bad) return;
        }
        body_down(wd, abuf, out, j, t, pre0, pre1, true);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    const int NW = 4;                                                     // rotating weight copies: nothing stays in the 256 MiB Infinity Cache
    bf16_t *wgu, *wdn; float *x, *out; uint4* act; unsigned* cnt; int* poisoned;
    CK(hipMalloc(&wgu, (size_t)NW * 2 * I * H * 2)); CK(hipMalloc(&wdn, (size_t)NW * I * H * 2));
    CK(hipMalloc(&x, ROWS * H * 4)); CK(hipMalloc(&out, ROWS * H * 4)); CK(hipMalloc(&act, (size_t)NBUF * GRID * 40 * 16));
    CK(hipMalloc(&cnt, KR * 4)); CK(hipMalloc(&poisoned, 4));
    CK(hipMemset(wgu, 0x3c, (size_t)NW * 2 * I * H * 2)); CK(hipMemset(wdn, 0x3c, (size_t)NW * I * H * 2));
    CK(hipMemset(x, 0, ROWS * H * 4)); CK(hipMemset(out, 0, ROWS * H * 4)); CK(hipMemset(act, 0, (size_t)NBUF * GRID * 40 * 16)); CK(hipMemset(poisoned, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double mb_pair = (2.0 * I * H * 2 + (double)I * H * 2) / 1e6;
    auto time_graph = [&](int form) {                                     // 0: gate|up -> down chain, 1: gate|up only, 2: down only
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < PAIRS; ++i) {
            if (form != 2) hipLaunchKernelGGL(k_gateup, dim3(GRID), dim3(256), 0, s, wgu + (size_t)(i % NW) * 2 * I * H, x, act + (size_t)(i % NBUF) * GRID * 40);
            if (form != 1) hipLaunchKernelGGL(k_down, dim3(GRID), dim3(256), 0, s, wdn + (size_t)(i % NW) * I * H, act + (size_t)(i % NBUF) * GRID * 40, out);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return best * 1e3 / PAIRS;
    };
    auto time_flow = [&](int wait) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            CK(hipMemsetAsync(cnt, 0, KR * 4, s));
            CK(hipEventRecord(e0, s));
            if (wait) hipLaunchKernelGGL(k_dataflow<1>, dim3(GRID), dim3(256), 0, s, wgu, wdn, x, act, out, cnt, PAIRS, NW, poisoned);
            else hipLaunchKernelGGL(k_dataflow<0>, dim3(GRID), dim3(256), 0, s, wgu, wdn, x, act, out, cnt, PAIRS, NW, poisoned);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        return best * 1e3 / PAIRS;
    };
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_dataflow<1>, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("co-residency: %d blocks per CU x %d CUs = %d slots for %d blocks\n", occ, prop.multiProcessorCount, occ * prop.multiProcessorCount, GRID);
    if (occ * prop.multiProcessorCount < GRID) { printf("grid would not be co-resident: not running the dataflow form\n"); return 1; }
    for (int rep = 0; rep < 2; ++rep) {
        const double chain = time_graph(0), gu = time_graph(1), dn = time_graph(2), flow = time_flow(1), free_ = time_flow(0);
        int hp = 0; CK(hipMemcpy(&hp, poisoned, 4, hipMemcpyDeviceToHost));
        printf("per (gate|up, down) pair, %.1f MB of weights, 8 rows, %d pairs, best of 5:\n"
               "  graph chain of two launches            %7.2f us  (%.2f TB/s)   [gate|up alone %.2f, down alone %.2f, sum %.2f]\n"
               "  ONE persistent dataflow launch         %7.2f us  (%.2f TB/s)%s\n"
               "  the same launch, hand-off waits off    %7.2f us  (%.2f TB/s)   (no dependency: what any hand-off could at best reach)\n",
               mb_pair, PAIRS, chain, mb_pair / chain, gu, dn, gu + dn, flow, mb_pair / flow, hp ? "  [POISONED: a spin timed out]" : "", free_, mb_pair / free_);
    }
    return 0;
}
