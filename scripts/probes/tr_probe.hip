// Semantics probe for gfx950 ds_read_b64_tr_b16: LDS holds v[i] = i (u16) for i < 4096; every lane passes its own byte
// address; we dump the 4 x u16 each lane receives.  Two address patterns: (a) lane-linear 8 B per lane, (b) a [4 rows][16 cols]
// block per 16-lane group inside a row-major image with 64-element rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                             // lane-linear
    else {                                                                   // group g reads rows 4g..4g+3 (64 elems per row), cols 0..15
        const int g = l >> 4, i = l & 15;
        addr = ((4 * g + (i >> 2)) * 64 + 4 * (i & 3)) * 2;
    }
    uint64_t v;
    if (mode == 2) v = *(const uint64_t*)((const char*)lds + l * 8);          // plain read, sanity
    else asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 2; mode >= 0; --mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipError_t e1 = hipDeviceSynchronize();
        uint16_t h[256]; hipError_t e2 = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("launch/sync: %s, copy: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { if (l < 20 || l % 16 == 0) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    }
    return 0;
}
