// Anatomy of the decode (M = 64) weight-streaming GEMM: the production loop of spacer_amd/csrc/decode.hip (packed weights,
// 64 columns x a K range per workgroup, A slice global -> regs -> LDS per 256-wide K slice) with parts switched off, next to a
// pure read of the same bytes.  dbg bits: 1 = no A global loads, 2 = no LDS staging / barriers, 4 = no MFMA, 8 = no epilogue.
// build: hipcc -O3 --offload-arch=gfx950 skinny_anatomy.hip -o skinny_anatomy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

// NW waves of 16 columns each per workgroup (64 / 128 / 256 columns share one staged A slice).  dbg bit 16: the grid is 1-D with
// id = range * groups8 + group (groups8 = groups rounded up to 8), so that every K range of a column group lands on the same
// XCD (workgroups go to XCDs round-robin by id), and the flush uses WORKGROUP-scope atomics = performed in that XCD's L2.
// KS = K-slice width (256: 64 KiB of LDS per workgroup, 2 per CU; 128: 32 KiB, OCC = 3 or 4 workgroups per CU -- every one of
// the 592 gate|up column groups is then resident at once and the launch has no second round)
template <int NW, int KS = 256, int OCC = 2>
__global__ __launch_bounds__(NW * 64, (NW == 16) ? 1 : OCC) void skinny(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                 float* __restrict__ C, long ldc, int M, int N, int K, int spr, int dbg, int groups8,
                                                 float* __restrict__ scratch, int* __restrict__ tickets) {
    constexpr int ROWB = KS * 2, NU = KS / 32, MF = 4, CH = KS / 8, CHS = (KS == 256) ? 5 : 4, NA = (64 * CH) / (NW * 64), RS = (NW * 64) / CH;
    __shared__ __attribute__((aligned(16))) char smem[2][64 * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bx = (dbg & 16) ? (int)blockIdx.x % groups8 : (int)blockIdx.x, by = (dbg & 16) ? (int)blockIdx.x / groups8 : (int)blockIdx.y;
    const int total = K / KS, s_begin = by * spr, s_end = min(total, s_begin + spr);
    const int n0 = bx * (NW * 16) + wave * 16;
    if (s_begin >= s_end || bx * (NW * 16) >= N) return;
    const int ar0 = tid >> CHS, ach = tid & (CH - 1);
    uint4 areg[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) areg[j] = make_uint4(0, 0, 0, 0);
    auto load_a = [&](int slice) {
        if (dbg & 1) return;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = ar0 + RS * j;
            if (row < M) areg[j] = *(const uint4*)(A + (long)row * lda + slice * KS + ach * 8);
        }
    };
    auto store_a = [&](char* buf) {
        if (dbg & 2) return;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = ar0 + RS * j;
            *(uint4*)(buf + row * ROWB + ((ach ^ (row & 15)) * 16)) = areg[j];
        }
        __syncthreads();
    };
    const bf16_t* bbase = B + ((long)(n0 >> 4) * (K >> 5)) * 512 + lane * 8;
    auto load_w = [&](u32x4 (&w)[NU], int slice) {
#pragma unroll
        for (int u = 0; u < NU; ++u) w[u] = __builtin_nontemporal_load((const u32x4*)(bbase + ((long)slice * NU + u) * 512));
    };
    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u32x4 (&w)[NU], const char* buf) {
        if (dbg & 4) {
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u & 3][0] += __uint_as_float(w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3]);
            return;
        }
        bf16x8 af[2][MF];
        auto read_a = [&](bf16x8 (&dst)[MF], int u) {
            const int ch = u * 4 + g;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int row = mf * 16 + l15;
                dst[mf] = *(const bf16x8*)(buf + row * ROWB + ((ch ^ (row & 15)) * 16));
            }
        };
        read_a(af[0], 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) read_a(af[(u + 1) & 1], u + 1);
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][mf], wf, acc[mf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    u32x4 wa[NU], wb[NU];
    load_a(s_begin);
    load_w(wa, s_begin);
    int s = s_begin;
    while (true) {
        store_a(smem[0]);
        if (s + 1 < s_end) { load_a(s + 1); load_w(wb, s + 1); }
        compute(wa, smem[0]);
        if (++s >= s_end) break;
        store_a(smem[1]);
        if (s + 1 < s_end) { load_a(s + 1); load_w(wa, s + 1); }
        compute(wb, smem[1]);
        if (++s >= s_end) break;
    }
    const int n = n0 + l15;
    const bool whole_k = (s_begin == 0 && s_end == total);
    if (dbg & 8) {
        float t = 0.f;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) t += acc[mf][0] + acc[mf][1] + acc[mf][2] + acc[mf][3];
        if (t == 1.2345e-30f) C[0] = t;
        return;
    }
    // (the scratch + ticket flush variants -- agent-scope fences, sc1 dword / 16-byte stores and loads -- were measured with this
    // probe and are recorded in DESIGN.md 7b; their code is gone so that it does not distort the register count of the rest)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mf * 16 + g * 4 + r;
            if (m < M) {
                float* c = C + (long)m * ldc + n;
                if (whole_k) *c = acc[mf][r];
                else if (dbg & 16) __hip_atomic_fetch_add(c, acc[mf][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else atomicAdd(c, acc[mf][r]);
            }
        }
}


// 8-wave variant: waves 0-3 and 4-7 take alternate K slices of the workgroup's range for the same 64 columns (two A slices staged
// per step), the two partial sums meet in LDS, then ONE flush: half the K ranges (= half the atomics) for the same number of
// weight-streaming waves.  One workgroup per CU (128 KiB of LDS).
__global__ __launch_bounds__(512, 2) void skinny_k2(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                    float* __restrict__ C, long ldc, int M, int N, int K, int spr, int dbg) {
    constexpr int KS = 256, ROWB = 512, NU = 8, MF = 4, CH = 32, CHS = 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];        // [2 parities][2 slices][64 * ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int kh = wave >> 2, cw = wave & 3;
    const int total = K / KS, s_begin = blockIdx.y * spr, s_end = min(total, s_begin + spr);
    const int n0 = blockIdx.x * 64 + cw * 16;
    if (s_begin >= s_end) return;
    const int t2 = tid & 255, ar0 = t2 >> CHS, ach = t2 & (CH - 1);    // threads 0-255 stage slice s, 256-511 slice s+1
    uint4 areg[8];
    auto load_a = [&](int s0) {
        const int slice = s0 + kh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = ar0 + 8 * j;
            areg[j] = (row < M && slice < s_end) ? *(const uint4*)(A + (long)row * lda + slice * KS + ach * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_a = [&](char* buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = ar0 + 8 * j;
            *(uint4*)(buf + kh * 64 * ROWB + row * ROWB + ((ach ^ (row & 15)) * 16)) = areg[j];
        }
        __syncthreads();
    };
    const bf16_t* bbase = B + ((long)(n0 >> 4) * (K >> 5)) * 512 + lane * 8;
    auto load_w = [&](u32x4 (&w)[NU], int s0) {
        const int slice = min(s0 + kh, total - 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) w[u] = __builtin_nontemporal_load((const u32x4*)(bbase + ((long)slice * NU + u) * 512));
    };
    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u32x4 (&w)[NU], const char* buf0, int s0) {
        if (s0 + kh >= s_end) return;                                   // odd slice count: the second half idles on the last step
        const char* buf = buf0 + kh * 64 * ROWB;
        bf16x8 af[2][MF];
        auto read_a = [&](bf16x8 (&dst)[MF], int u) {
            const int ch = u * 4 + g;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int row = mf * 16 + l15;
                dst[mf] = *(const bf16x8*)(buf + row * ROWB + ((ch ^ (row & 15)) * 16));
            }
        };
        read_a(af[0], 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) read_a(af[(u + 1) & 1], u + 1);
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][mf], wf, acc[mf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    u32x4 wa[NU], wb[NU];
    char* b0 = smem; char* b1 = smem + 2 * 64 * ROWB;
    load_a(s_begin);
    load_w(wa, s_begin);
    int s = s_begin;
    while (true) {
        store_a(b0);
        if (s + 2 < s_end) { load_a(s + 2); load_w(wb, s + 2); }
        compute(wa, b0, s);
        if ((s += 2) >= s_end) break;
        store_a(b1);
        if (s + 2 < s_end) { load_a(s + 2); load_w(wa, s + 2); }
        compute(wb, b1, s);
        if ((s += 2) >= s_end) break;
    }
    __syncthreads();                                                    // everyone is done with the A images
    float* red = (float*)smem;                                          // [4 column waves][MF][4][64 lanes]
    if (kh == 1) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((cw * MF + mf) * 4 + r) * 64 + lane] = acc[mf][r];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mf][r] += red[((cw * MF + mf) * 4 + r) * 64 + lane];
    const int n = n0 + l15;
    const bool whole_k = (s_begin == 0 && s_end == total);
    if (dbg & 8) {
        float t = 0.f;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) t += acc[mf][0] + acc[mf][1] + acc[mf][2] + acc[mf][3];
        if (t == 1.2345e-30f) C[0] = t;
        return;
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mf * 16 + g * 4 + r;
            if (m < M) {
                float* c = C + (long)m * ldc + n;
                if (whole_k) *c = acc[mf][r];
                else atomicAdd(c, acc[mf][r]);
            }
        }
}


// Loader-wave variant: waves 0-3 stream weights and run the MFMAs; wave 4 does nothing but bring the A tiles in, by LDS-DMA into a
// ring of NSLOT 128-wide slots, running ahead of the consumers.  Its memory queue holds only A requests (L2 hits), so an A tile
// never waits behind the weight stream's in-order returns, and the consumers never wait for A.  One barrier per slice.
template <int NSLOT>
__global__ __launch_bounds__(320, 2) void skinny_ld(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                    float* __restrict__ C, long ldc, int M, int N, int K, int spr, int dbg) {
    constexpr int KS = 128, ROWB = 256, NU = 4, MF = 4, SLOT = 64 * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];                 // [NSLOT][64 rows][256 B], chunk ^= row & 15
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = K / KS, s_begin = blockIdx.y * spr, s_end = min(total, s_begin + spr), ns = s_end - s_begin;
    if (ns <= 0) return;
    if (wave == 4) {
        // ---------------------------------------------------------------- loader: 16 DMA instructions (1 KiB each) per slice
        // instruction i covers rows 4i .. 4i+3: lane -> row 4i + (lane >> 4), LDS position lane & 15 holds chunk (lane & 15) ^ (row & 15)
        auto issue = [&](int s) {
            char* dst = smem + (s % NSLOT) * SLOT;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 4 * i + g, ch = l15 ^ (row & 15);
                const bf16_t* src = A + (long)min(row, M - 1) * lda + (long)(s_begin + s) * KS + ch * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        };
        for (int s = 0; s < NSLOT - 1 && s < ns; ++s) issue(s);
        for (int s = 0; s < ns; ++s) {
            // slices s+1 .. s+NSLOT-2 may stay in flight; slice s must have landed
            const int later = min(ns - 1, s + NSLOT - 2) - s;
            if (later >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                          // slot s ready; consumers are done with slot s-1
            if (s + NSLOT - 1 < ns) issue(s + NSLOT - 1);
        }
        return;
    }
    // -------------------------------------------------------------------- consumers
    const int n0 = blockIdx.x * 64 + wave * 16;
    const bf16_t* bbase = B + ((long)(n0 >> 4) * (K >> 5)) * 512 + lane * 8;
    auto load_w = [&](u32x4 (&w)[NU], int s) {
#pragma unroll
        for (int u = 0; u < NU; ++u) w[u] = __builtin_nontemporal_load((const u32x4*)(bbase + ((long)(s_begin + s) * NU + u) * 512));
    };
    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u32x4 (&w)[NU], int s) {
        const char* buf = smem + (s % NSLOT) * SLOT;
        bf16x8 af[2][MF];
        auto read_a = [&](bf16x8 (&dst)[MF], int u) {
            const int ch = u * 4 + g;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int row = mf * 16 + l15;
                dst[mf] = *(const bf16x8*)(buf + row * ROWB + ((ch ^ (row & 15)) * 16));
            }
        };
        read_a(af[0], 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) read_a(af[(u + 1) & 1], u + 1);
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][mf], wf, acc[mf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // four weight register sets = three slices of prefetch (as many bytes in flight per wave as two 256-wide sets)
    u32x4 w0[NU], w1[NU], w2[NU], w3[NU];
    load_w(w0, 0);
    if (1 < ns) load_w(w1, 1);
    if (2 < ns) load_w(w2, 2);
    for (int s = 0; s < ns; s += 4) {
        __builtin_amdgcn_s_barrier(); if (s + 3 < ns) load_w(w3, s + 3); compute(w0, s);
        if (s + 1 >= ns) break;
        __builtin_amdgcn_s_barrier(); if (s + 4 < ns) load_w(w0, s + 4); compute(w1, s + 1);
        if (s + 2 >= ns) break;
        __builtin_amdgcn_s_barrier(); if (s + 5 < ns) load_w(w1, s + 5); compute(w2, s + 2);
        if (s + 3 >= ns) break;
        __builtin_amdgcn_s_barrier(); if (s + 6 < ns) load_w(w2, s + 6); compute(w3, s + 3);
    }
    const int n = n0 + l15;
    const bool whole_k = (s_begin == 0 && s_end == total);
    if (dbg & 8) {
        float t = 0.f;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) t += acc[mf][0] + acc[mf][1] + acc[mf][2] + acc[mf][3];
        if (t == 1.2345e-30f) C[0] = t;
        return;
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mf * 16 + g * 4 + r;
            if (m < M) {
                float* c = C + (long)m * ldc + n;
                if (whole_k) *c = acc[mf][r];
                else atomicAdd(c, acc[mf][r]);
            }
        }
}


// Row-split variant: no K split and no atomics.  The 64 rows are cut into four groups of 16; a workgroup = 64 columns x 16 rows x ALL
// of K (one MFMA row fragment per wave), so a column group's weights are streamed by four workgroups -- placed on ONE XCD (block ids
// 8 apart, dispatched within 32 ids of each other) so that three of the four reads hit that XCD's L2 -- and every output element has
// one owner: plain stores, any epilogue.  HBM traffic stays 1x, L2 -> CU traffic is 4x.
__global__ __launch_bounds__(256, 4) void skinny_rs(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                    float* __restrict__ C, long ldc, int M, int N, int K, int dbg) {
    constexpr int KS = 256, ROWB = 512, NU = 8;
    __shared__ __attribute__((aligned(16))) char smem[2][16 * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int id = blockIdx.x, cg = (id >> 5) * 8 + (id & 7), rg = (id >> 3) & 3;       // column group, row group
    if (cg * 64 >= N) return;
    const int total = K / KS, n0 = cg * 64 + wave * 16, r0 = rg * 16;
    const int arow = tid >> 5, ach = tid & 31;                 // 8 rows x 32 chunks per pass, 2 passes
    uint4 areg[2];
    auto load_a = [&](int slice) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = r0 + arow + 8 * j;
            areg[j] = row < M ? *(const uint4*)(A + (long)row * lda + slice * KS + ach * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_a = [&](char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = arow + 8 * j;
            *(uint4*)(buf + row * ROWB + ((ach ^ (row & 15)) * 16)) = areg[j];
        }
        __syncthreads();
    };
    const bf16_t* bbase = B + ((long)(n0 >> 4) * (K >> 5)) * 512 + lane * 8;
    auto load_w = [&](u32x4 (&w)[NU], int slice) {
#pragma unroll
        for (int u = 0; u < NU; ++u) w[u] = *(const u32x4*)(bbase + ((long)slice * NU + u) * 512);   // plain (L2-allocating): the other three row groups of this column group hit
    };
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u32x4 (&w)[NU], const char* buf) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bf16x8 af = *(const bf16x8*)(buf + l15 * ROWB + (((u * 4 + g) ^ l15) * 16));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, w[u]), acc, 0, 0, 0);
        }
    };
    u32x4 wa[NU], wb[NU], wc[NU];                               // three weight sets: two slices of prefetch
    load_a(0); load_w(wa, 0);
    if (1 < total) load_w(wb, 1);
    for (int s = 0; s < total; s += 3) {
        store_a(smem[0]); if (s + 1 < total) load_a(s + 1); if (s + 2 < total) load_w(wc, s + 2); compute(wa, smem[0]);
        if (s + 1 >= total) break;
        store_a(smem[1]); if (s + 2 < total) load_a(s + 2); if (s + 3 < total) load_w(wa, s + 3); compute(wb, smem[1]);
        if (s + 2 >= total) break;
        store_a(smem[0]); if (s + 3 < total) load_a(s + 3); if (s + 4 < total) load_w(wb, s + 4); compute(wc, smem[0]);
    }
    if (dbg & 8) { if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) C[0] = acc[0]; return; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = r0 + g * 4 + r;
        if (m < M) C[(long)m * ldc + n0 + l15] = acc[r];
    }
}

// pure read with the GEMM's own access pattern (each wave: its 16-column fragment stream, 8 KiB per slice)
__global__ __launch_bounds__(256, 2) void stream_only(const bf16_t* __restrict__ B, float* __restrict__ C, int K, int spr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = K / 256, s_begin = blockIdx.y * spr, s_end = min(total, s_begin + spr);
    const int n0 = blockIdx.x * 64 + wave * 16;
    const bf16_t* bbase = B + ((long)(n0 >> 4) * (K >> 5)) * 512 + lane * 8;
    u32x4 x = {0, 0, 0, 0};
    for (int s = s_begin; s < s_end; ++s) {
        u32x4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load((const u32x4*)(bbase + ((long)s * 8 + u) * 512));
#pragma unroll
        for (int u = 0; u < 8; ++u) x ^= w[u];
    }
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678u) C[0] = 1.f;
}

int main() {
    const int M = 64;
    const long wbytes_max = 37888L * 3584 * 2;
    const int COPIES = 6;
    bf16_t *A, *W; float *C, *scratch; int* tickets;
    hipMalloc(&scratch, 1024L * 64 * 64 * 4 * 2); hipMalloc(&tickets, 4096); hipMemset(tickets, 0, 4096);
    hipMalloc(&A, (long)M * 18944 * 2); hipMalloc(&W, wbytes_max * COPIES); hipMalloc(&C, (long)M * 152064 * 4);
    hipMemset(A, 0, (long)M * 18944 * 2); hipMemset(W, 0, wbytes_max * COPIES); hipMemset(C, 0, (long)M * 152064 * 4);
    {   // bf16 1.0 everywhere: every element of A.W^T is K
        const long na = (long)M * 18944, nw = wbytes_max * COPIES / 2;
        unsigned short* h = (unsigned short*)malloc(nw * 2);
        for (long i = 0; i < nw; ++i) h[i] = 0x3F80;
        hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice); hipMemcpy(W, h, nw * 2, hipMemcpyHostToDevice);
        free(h);
    }
    float* hc = (float*)malloc((long)M * 152064 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, double bytes) {
        for (int w = 0; w < 3; ++w) launch(w);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 24;
        for (int r = 0; r < reps; ++r) launch(r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %-34s %7.1f us  %5.2f TB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    };
    struct Shape { const char* name; int N, K, ranges; };
    const Shape shapes[] = {{"qkv 4608x3584 (72 groups x 7)", 4608, 3584, 7}, {"o 3584x3584 (56 x 9 -> 7)", 3584, 3584, 9},
                            {"down 3584x18944 (56 x 9)", 3584, 18944, 9}, {"gate|up 32768x3584 (512 x 1)", 32768, 3584, 1},
                            {"gate|up 37888x3584 (592 x 1)", 37888, 3584, 1}};
    for (const Shape& sh : shapes) {
        const int groups = sh.N / 64, slices = sh.K / 256;
        int spr = (slices + sh.ranges - 1) / sh.ranges;
        const int ranges = (slices + spr - 1) / spr;
        const double bytes = (double)sh.N * sh.K * 2;
        const long wstride = (long)sh.N * sh.K;
        printf("%s: %d blocks, %d slices each, %.0f MB\n", sh.name, groups * ranges, spr, bytes / 1e6);
        run("stream only", [&](int r) { hipLaunchKernelGGL(stream_only, dim3(groups, ranges), dim3(256), 0, 0, W + (r % COPIES) * wstride, C, sh.K, spr); }, bytes);
        const struct { const char* n; int d; } cfgs[] = {{"full kernel", 0}, {"no epilogue", 8}, {"no A loads", 1}, {"no A loads, no epilogue", 9}};
        auto launch = [&](int nw, int d, int r) {
            const int g = sh.N / (nw * 16), g8 = (g + 7) / 8 * 8;
            const dim3 grid = (d & 16) ? dim3(g8 * ranges, 1) : dim3(g, ranges);
            const bf16_t* w = W + (r % COPIES) * wstride;
            if (nw == 4) hipLaunchKernelGGL(skinny<4>, grid, dim3(256), 0, 0, A, (long)sh.K, w, C, (long)sh.N, M, sh.N, sh.K, spr, d, g8, scratch, tickets);
            else if (nw == 8) hipLaunchKernelGGL(skinny<8>, grid, dim3(512), 0, 0, A, (long)sh.K, w, C, (long)sh.N, M, sh.N, sh.K, spr, d, g8, scratch, tickets);
            else hipLaunchKernelGGL(skinny<16>, grid, dim3(1024), 0, 0, A, (long)sh.K, w, C, (long)sh.N, M, sh.N, sh.K, spr, d, g8, scratch, tickets);
        };
        hipFuncSetAttribute((const void*)skinny_k2, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 512);
        for (int r2 : {ranges, (ranges + 1) / 2, (ranges + 2) / 3}) {
            const int spr2 = (slices + r2 - 1) / r2, rr = (slices + spr2 - 1) / spr2;
            for (int d : {0, 8}) {
                char nm[96]; snprintf(nm, 96, "8 waves (2 K halves), %d ranges x %d slices%s", rr, spr2, d ? ", no epilogue" : "");
                run(nm, [&](int r) { hipLaunchKernelGGL(skinny_k2, dim3(groups, rr), dim3(512), 4 * 64 * 512, 0, A, (long)sh.K, W + (r % COPIES) * wstride, C, (long)sh.N, M, sh.N, sh.K, spr2, d); }, bytes);
                if (d == 0) {
                    hipMemset(C, 0, (long)M * sh.N * 4);
                    hipLaunchKernelGGL(skinny_k2, dim3(groups, rr), dim3(512), 4 * 64 * 512, 0, A, (long)sh.K, W, C, (long)sh.N, M, sh.N, sh.K, spr2, 0);
                    hipDeviceSynchronize();
                    hipMemcpy(hc, C, (long)M * sh.N * 4, hipMemcpyDeviceToHost);
                    long bad = 0; for (long i = 0; i < (long)M * sh.N; ++i) bad += (hc[i] != (float)sh.K);
                    if (bad) printf("      WRONG: %ld of %ld elements != K\n", bad, (long)M * sh.N);
                }
            }
        }
        {
            const int g8 = (groups + 7) / 8 * 8;
            for (int d : {0, 8}) {
                char nm[96]; snprintf(nm, 96, "row split (4 x 16 rows, whole K, no atomics)%s", d == 8 ? ", no epilogue" : "");
                run(nm, [&](int r) { hipLaunchKernelGGL(skinny_rs, dim3(g8 * 4), dim3(256), 0, 0, A, (long)sh.K, W + (r % COPIES) * wstride, C, (long)sh.N, M, sh.N, sh.K, d); }, bytes);
                if (d == 0) {
                    hipMemset(C, 0, (long)M * sh.N * 4);
                    hipLaunchKernelGGL(skinny_rs, dim3(g8 * 4), dim3(256), 0, 0, A, (long)sh.K, W, C, (long)sh.N, M, sh.N, sh.K, 0);
                    hipDeviceSynchronize();
                    hipMemcpy(hc, C, (long)M * sh.N * 4, hipMemcpyDeviceToHost);
                    long bad = 0; for (long i = 0; i < (long)M * sh.N; ++i) bad += (hc[i] != (float)sh.K);
                    if (bad) printf("      WRONG: %ld of %ld elements != K\n", bad, (long)M * sh.N);
                }
            }
        }
        {
            hipFuncSetAttribute((const void*)skinny_ld<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 256);
            const int spr_l = 2 * spr;                                     // same K ranges, in 128-wide slices
            for (int d : {0, 8}) {
                char nm[96]; snprintf(nm, 96, "loader wave + 4 consumers, 4-slot A ring%s", d == 8 ? ", no epilogue" : "");
                run(nm, [&](int r) { hipLaunchKernelGGL(skinny_ld<4>, dim3(groups, ranges), dim3(320), 4 * 64 * 256, 0, A, (long)sh.K, W + (r % COPIES) * wstride, C, (long)sh.N, M, sh.N, sh.K, spr_l, d); }, bytes);
                if (d == 0) {
                    hipMemset(C, 0, (long)M * sh.N * 4);
                    hipLaunchKernelGGL(skinny_ld<4>, dim3(groups, ranges), dim3(320), 4 * 64 * 256, 0, A, (long)sh.K, W, C, (long)sh.N, M, sh.N, sh.K, spr_l, 0);
                    hipDeviceSynchronize();
                    hipMemcpy(hc, C, (long)M * sh.N * 4, hipMemcpyDeviceToHost);
                    long bad = 0; for (long i = 0; i < (long)M * sh.N; ++i) bad += (hc[i] != (float)sh.K);
                    if (bad) printf("      WRONG: %ld of %ld elements != K\n", bad, (long)M * sh.N);
                }
            }
        }
        if (false) {
            for (int occ : {3, 4}) {
                for (int d : {0, 8, 1}) {
                    char nm[96]; snprintf(nm, 96, "128-wide slices, %d workgroups/CU%s", occ, d == 8 ? ", no epilogue" : d == 1 ? ", no A loads" : "");
                    run(nm, [&](int r) {
                        const bf16_t* w = W + (r % COPIES) * wstride;
                        if (occ == 3) hipLaunchKernelGGL((skinny<4, 128, 3>), dim3(groups, 1), dim3(256), 0, 0, A, (long)sh.K, w, C, (long)sh.N, M, sh.N, sh.K, sh.K / 128, d, 0, scratch, tickets);
                        else hipLaunchKernelGGL((skinny<4, 128, 4>), dim3(groups, 1), dim3(256), 0, 0, A, (long)sh.K, w, C, (long)sh.N, M, sh.N, sh.K, sh.K / 128, d, 0, scratch, tickets);
                    }, bytes);
                    if (d == 0) {
                        hipMemset(C, 0, (long)M * sh.N * 4);
                        hipLaunchKernelGGL((skinny<4, 128, 4>), dim3(groups, 1), dim3(256), 0, 0, A, (long)sh.K, W, C, (long)sh.N, M, sh.N, sh.K, sh.K / 128, 0, 0, scratch, tickets);
                        hipDeviceSynchronize();
                        hipMemcpy(hc, C, (long)M * sh.N * 4, hipMemcpyDeviceToHost);
                        long bad = 0; for (long i = 0; i < (long)M * sh.N; ++i) bad += (hc[i] != (float)sh.K);
                        if (bad) printf("      WRONG: %ld of %ld elements != K\n", bad, (long)M * sh.N);
                    }
                }
            }
        }
        if (false) {
            for (int d : {0, 8, 1}) {
                char nm[96]; snprintf(nm, 96, "8 waves x 128 columns, 128-wide slices, 2 workgroups/CU%s", d == 8 ? ", no epilogue" : d == 1 ? ", no A loads" : "");
                run(nm, [&](int r) {
                    hipLaunchKernelGGL((skinny<8, 128, 4>), dim3(sh.N / 128, 1), dim3(512), 0, 0, A, (long)sh.K, W + (r % COPIES) * wstride, C, (long)sh.N, M, sh.N, sh.K, sh.K / 128, d, 0, scratch, tickets);
                }, bytes);
                if (d == 0) {
                    hipMemset(C, 0, (long)M * sh.N * 4);
                    hipLaunchKernelGGL((skinny<8, 128, 4>), dim3(sh.N / 128, 1), dim3(512), 0, 0, A, (long)sh.K, W, C, (long)sh.N, M, sh.N, sh.K, sh.K / 128, 0, 0, scratch, tickets);
                    hipDeviceSynchronize();
                    hipMemcpy(hc, C, (long)M * sh.N * 4, hipMemcpyDeviceToHost);
                    long bad = 0; for (long i = 0; i < (long)M * sh.N; ++i) bad += (hc[i] != (float)sh.K);
                    if (bad) printf("      WRONG: %ld of %ld elements != K\n", bad, (long)M * sh.N);
                }
            }
        }
        for (int nw : {4}) {
            if (sh.N % (nw * 16)) continue;
            for (auto& c : cfgs) {
                if ((c.d & 112) && ranges == 1) continue;
                char nm[96]; snprintf(nm, 96, "%2d waves: %s", nw, c.n);
                run(nm, [&](int r) { launch(nw, c.d, r); }, bytes);
                if (c.d == 0 || c.d == 16 || c.d == 32 || c.d == 64) {          // one launch from zero: every element must equal K
                    hipMemset(C, 0, (long)M * sh.N * 4); launch(nw, c.d, 0); hipDeviceSynchronize();
                    hipMemcpy(hc, C, (long)M * sh.N * 4, hipMemcpyDeviceToHost);
                    long bad = 0; for (long i = 0; i < (long)M * sh.N; ++i) bad += (hc[i] != (float)sh.K);
                    if (bad) printf("      WRONG: %ld of %ld elements != K\n", bad, (long)M * sh.N);
                }
            }
        }
    }
    return 0;
}
