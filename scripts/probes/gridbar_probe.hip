// Grid-barrier probe for a persistent decode kernel on MI355X: how long does one all-blocks barrier take when every
// workgroup of a co-resident grid (2 per CU x 256 threads) arrives on a monotonically increasing counter and spins, with
// agent-scope release / acquire fences around it (the L2s of the 8 XCDs are not coherent with each other without them)?
// Also checks visibility: before barrier i every block writes a word, after it every block reads its neighbour's word
// (a block that lives on another XCD) and counts mismatches.  Spins carry a clock timeout so a bug cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000ll) { *err = 1; break; }      // 1 s at 100 MHz
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return true;
}

// mode 1: no fences (cost of the counter alone); mode 2: two-level tree -- groups of 16 blocks on their own 128-byte lines, the last
// arriver of a group (known from the fetch_add result) bumps the root; everybody polls the root.  mode 3: tree, pollers watch a
// per-group release word that the root's last arriver... (kept simple: pollers watch the root).
__device__ __forceinline__ void grid_barrier_tree(unsigned* base, unsigned gen, int* err, bool fences) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned grp = blockIdx.x >> 4, ngrp = (gridDim.x + 15) >> 4;
        const unsigned in_grp = min(16u, gridDim.x - grp * 16);
        unsigned* gc = base + 64 + grp * 32;                                  // 128-byte line per group
        const unsigned old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * in_grp) __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * ngrp) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000ll) { *err = 1; break; }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void probe_tree(unsigned* base, unsigned* words, int* err, int* bad, int iters, int fences) {
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x == 0) words[blockIdx.x] = (unsigned)(i * 1000003u + blockIdx.x);
        grid_barrier_tree(base, (unsigned)(2 * i + 1), err, fences);
        if (threadIdx.x == 0) {
            const unsigned o = (blockIdx.x + 1) % nb;
            if (words[o] != (unsigned)(i * 1000003u + o)) atomicAdd(bad, 1);
        }
        grid_barrier_tree(base, (unsigned)(2 * i + 2), err, fences);
    }
}

__global__ __launch_bounds__(256, 2) void probe(unsigned* counter, unsigned* words, int* err, int* bad, int iters, int check) {
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        if (check && threadIdx.x == 0) words[blockIdx.x] = (unsigned)(i * 1000003u + blockIdx.x);
        grid_barrier(counter, (unsigned)(i + 1) * nb, err);
        if (check && threadIdx.x == 0) {
            const unsigned o = (blockIdx.x + 1) % nb;                        // round-robin dispatch: the next block is on the next XCD
            if (words[o] != (unsigned)(i * 1000003u + o)) atomicAdd(bad, 1);
        }
        if (check) grid_barrier(counter + 32, (unsigned)(i + 1) * nb, err);  // nobody overwrites before everyone has read
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int blocks = p.multiProcessorCount * per_cu;
    printf("CUs %d, grid %d x 256 threads\n", p.multiProcessorCount, blocks);
    unsigned *counter, *words; int *err, *bad;
    hipMalloc(&counter, 256); hipMalloc(&words, blocks * 4); hipMalloc(&err, 4); hipMalloc(&bad, 4);
    for (int check = 0; check < 2; ++check)
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 2000;
            hipMemset(counter, 0, 256); hipMemset(err, 0, 4); hipMemset(bad, 0, 4);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, counter, words, err, bad, iters, check);
            hipEventRecord(b, 0);
            hipError_t e = hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            int herr, hbad; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            printf("check %d: %s, %d barriers%s in %.3f ms = %.2f us each, timeout flag %d, stale reads %d\n", check, hipGetErrorString(e),
                   iters * (check ? 2 : 1), check ? " (+ write/read)" : "", ms, ms * 1e3 / (iters * (check ? 2 : 1)), herr, hbad);
        }
    unsigned* base; hipMalloc(&base, 64 * 1024);
    for (int fences = 1; fences >= 0; --fences) {
        const int iters = 2000;
        hipMemset(base, 0, 64 * 1024); hipMemset(err, 0, 4); hipMemset(bad, 0, 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe_tree, dim3(blocks), dim3(256), 0, 0, base, words, err, bad, iters, fences);
        hipEventRecord(b, 0);
        hipError_t e = hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        int herr, hbad; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        printf("tree, fences %d: %s, %d barriers in %.3f ms = %.2f us each, timeout flag %d, stale reads %d\n", fences, hipGetErrorString(e),
               iters * 2, ms, ms * 1e3 / (iters * 2), herr, hbad);
    }
    return 0;
}
