// Decode-loop kernels (the HBM-bound half of HF generate, SG_RLVR_trainer.py:463): weight-streaming skinny
// GEMM with fp32 split-K accumulation, the q/k/v finishing step (bias + rotary + KV-cache append), SwiGLU
// from fp32 accumulators, and single-query attention over [shared prompt KV | per-rollout tail KV].
// All step-dependent scalars (tail length, step index) are read from device memory so one decode step can
// be captured once in a hipGraph and replayed.
#include <stdlib.h>

#include "attn_common.h"

namespace {

// =============================================================================== skinny GEMM (M <= 64)
// C32[M,N] += A[M,K] . B[N,K]^T, weights streamed exactly once from HBM.
// Workgroup = 64 columns (wave = 16) x a contiguous RANGE of 256-wide K slices.  Per slice the A slice (<= 64 x 256
// bf16) goes global -> registers -> LDS (double-buffered, one barrier per slice, next slice's loads in flight
// under the MFMAs); B is streamed HBM -> VGPR with non-temporal 16-byte fragment loads through two 8-deep register
// sets that stay in flight across the barriers.  The sums live in registers over the whole K range and are flushed
// ONCE: a plain read-add-write when the workgroup covers all of K, fp32 atomics when K is split (small N only --
// the L2 atomic units, not HBM, bound the kernel when every 64x64 tile is flushed per slice).
// PACKED = B stored fragment-major (spacer_pack_weight_frag): [N/16][K/32][64 lanes][8 bf16], so every wave load is
// 1 KiB of contiguous memory; row-major B gives 16 x 64-byte segments per instruction, ~30 % slower.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// SWIGLU (packed weights only, whole K per block): the 16 columns of a wave are [8 gate | 8 up] columns of the same 8
// outputs (spacer_pack_weight_frag_swiglu), so lanes l and l^8 hold gate and up of one element; the epilogue writes
// y = silu(gate) * up as bf16 [M, N/2] and C is that bf16 buffer -- no fp32 round trip, no separate SwiGLU launch.
// MT = 64-row blocks of A handled per weight pass: 1 (M <= 64, 256-wide K slices) or 2 (M <= 128, 128-wide K slices -- the
// same 64 KiB of LDS and two workgroups per CU; every weight fragment feeds 8 instead of 4 MFMAs, so 128 rows stream the
// weights once instead of twice).
// SWIGLU tail balance (split_groups > 0): N / 64 column groups on 512 resident slots leave a short last round (592 groups at
// 7B: 80 workgroups alone on the chip stream at the per-workgroup rate, 33 GB/s each: +15 us).  The LAST `split_groups`
// column groups are therefore cut into `split_ranges` K ranges each, dispatched FIRST (lowest block ids, so they run beside the
// whole-K blocks instead of after them); their partial sums meet in an fp32 scratch tile through agent-scope atomics, an
// agent-scope ticket per column group counts arrivals, and the last arriver takes the totals back with atomic exchanges (read +
// re-zero in one read-modify-write: both sides atomics on the same words, so no fence is needed) and applies the SwiGLU epilogue.
// SWIGLU one-round form (wide_groups > 0; round 4): when the N / 16 column fragments number between 4 and 5 per resident
// workgroup slot (7B gate|up: 2368 fragments on 512 slots), the launch is EXACTLY one resident round of 512 workgroups: the first
// `wide_groups` (= fragments - 4 * slots) own FIVE fragments, the others four.  Wave w of a wide workgroup streams its own fragment
// over all of K as before plus one quarter of the k-steps of every slice of the fifth fragment (MT = 1: k-steps 2w, 2w + 1; MT = 2: k-step w); the four partial
// sums of the fifth fragment meet in LDS after the K loop and wave w finishes rows 16w .. 16w + 15 of it.  No atomics, no ticket,
// no second round: time = 9.4 us + bytes / 6.75 TB/s like any one-round launch of this kernel (scripts/probes/swiglu_equal_work.py)
// instead of the 1.16 rounds of the 64-column decomposition (58.7 us with the tail balance, 49.6 us by that law).
// NORMA (packed weights, MT = 1, K-split form): A is the fp32 residual stream x [M, K] itself and the RMSNorm in front of the
// projection is folded in:  norm(x) W^T = rstd[m] * (bf16(x) (W diag(w))^T)  -- exact algebra; W diag(w) is folded into the packed
// weights by the caller, the staging threads round x to bf16 on the way into LDS, and the workgroups of column group 0 also sum x^2 per
// row into `rowss` (one fp32 atomic per row per K range) from which the finishing kernel forms rstd.  One ~6 us launch (the norm) less
// per layer; the A slice costs twice the L2 -> CU bytes, which the q|k|v projection (33 MB of weights, latency-bound) does not
// feel; on gate|up (592 column groups re-reading 64 x 3584 fp32) the same fold measured slower and is not used.
// SMALL (round 5; MT = 1): the batch has <= 16 rows (cfg4's one prompt group per GPU = 8 rows): only row fragment 0 is staged,
// read and multiplied (MF = 1: a quarter of the LDS traffic, 12 accumulator registers instead of 48).  With NORMA + SWIGLU + SMALL the
// post-attention RMSNorm is folded into the gate|up projection: A is the fp32 residual stream (at <= 16 rows its re-staging by every
// workgroup costs 16 x 256 x 4 B per slice beside 40 KB of weights -- at 64 rows the same fold measured slower), every
// workgroup sums x^2 per row over its whole K (the SWIGLU forms never split K), rstd meets the epilogue through LDS, and the packed
// weights carry W diag(w_ln2): y = silu(rstd g) * (rstd u).  One ~5 us launch (the norm) less per layer of an 8-row decode step.
template <bool PACKED, bool SWIGLU = false, int MT = 1, bool NORMA = false, bool SMALL = false>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const bf16_t* __restrict__ A, long lda,
                                                             const bf16_t* __restrict__ B, long ldb,
                                                             float* __restrict__ C, long ldc, int M, int N, int K,
                                                             int slices_per_range, int mflush, int overwrite,
                                                             int split_groups = 0, int split_ranges = 1,
                                                             float* __restrict__ scratch = nullptr, int* __restrict__ tickets = nullptr,
                                                             float* __restrict__ rowss = nullptr, int wide_groups = 0, float eps = 0.f) {
    static_assert(!SMALL || MT == 1, "SMALL is a 64-row-block form");
    constexpr int KS = 256 / MT, ROWB = KS * 2;            // LDS row bytes (512 / 256); 16-byte chunk index ^= row & 15
    constexpr int NU = KS / 32, MF = SMALL ? 1 : 4 * MT;   // MFMA k-steps per slice, 16-row A fragments
    constexpr int NJ = SMALL ? 2 : 8;                      // staging instructions per thread (8 rows each)
    constexpr int CH = KS / 8, CHS = (MT == 1) ? 5 : 4;    // 16-byte chunks per LDS row and log2
    __shared__ __attribute__((aligned(16))) char smem[2][64 * MT * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int total_slices = K / KS;
    int cgroup = blockIdx.x, s_begin = blockIdx.y * slices_per_range, s_end = min(total_slices, s_begin + slices_per_range);
    if (slices_per_range < 0) {
        // skewed K ranges (spacer_plan::skinny_skew = k + 1, passed as -(k + 1)): range r of R gets a share proportional to
        // 1 + alpha (2 r / (R - 1) - 1), alpha = k / 16 -- the ranges of a column group finish at different times, so their atomic
        // flushes overlap the other ranges' weight streaming instead of all landing when the stream ends; k = 0: an even split
        const int R = (int)gridDim.y, r = (int)blockIdx.y;
        const float al = (float)(-slices_per_range - 1) * (1.f / 16.f);
        auto st = [&](int q) { const float c = (float)q + (R > 1 ? al * (float)q * (float)(q - R) / (float)(R - 1) : 0.f);
                               return (int)((float)total_slices * c / (float)R + 0.5f); };
        s_begin = max(0, min(total_slices, st(r))); s_end = max(0, min(total_slices, st(r + 1)));     // (alpha <= 1 is enforced by the launchers)
    }
    bool split_block = false;
    if (SWIGLU && split_groups > 0) {
        const int n_split_blocks = split_groups * split_ranges, whole = (N >> 6) - split_groups;
        const int bx = (int)blockIdx.x;        // split blocks first: measured 55.4 us vs 56.0 us with them last (592 groups, 7B)
        if (bx < n_split_blocks) {
            split_block = true;
            cgroup = whole + bx / split_ranges;
            const int r = bx % split_ranges, spr = (total_slices + split_ranges - 1) / split_ranges;
            s_begin = r * spr; s_end = min(total_slices, s_begin + spr);
        } else {
            cgroup = bx - n_split_blocks;
        }
    }
    int n0 = cgroup * 64 + wave * 16;
    // one-round form: workgroup bx owns fragments [f0, f0 + 5) (bx < wide_groups) or [f0, f0 + 4)
    const bool wide = SWIGLU && wide_groups > 0 && (int)blockIdx.x < wide_groups;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // scalar copy: the fifth-fragment branch below is wave-uniform
    int n5 = 0;
    if (SWIGLU && wide_groups > 0) {
        const int bx = (int)blockIdx.x;
        const int f0 = bx < wide_groups ? 5 * bx : 5 * wide_groups + 4 * (bx - wide_groups);
        n0 = (f0 + wave) * 16;
        n5 = (f0 + 4) * 16;
    }
    if (s_begin >= s_end) return;

    // ---- staging map: instruction j of a thread covers row (tid >> CHS) + (256 / CH) j, 16-byte chunk tid & (CH - 1)
    // (a wave reads whole rows of the slice: 2 x 512 or 4 x 256 contiguous bytes per load instruction); 8 per thread
    const int ar0 = tid >> CHS, ach = tid & (CH - 1);
    uint4 areg[NJ];
    float4 areg32[NORMA ? NJ : 1][2];                  // NORMA: the same 8 elements per row as fp32
    float ss[NORMA ? NJ : 1];                          // NORMA: running sum of x^2 of this thread's chunk of 8 rows
    // who sums x^2: column group 0 of a K-split launch (atomics into rowss), EVERY workgroup of the SwiGLU fold (its own rstd)
    const bool sums = NORMA && (SWIGLU || blockIdx.x == 0);
    if (NORMA) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) ss[NORMA ? j : 0] = 0.f;
    }
    auto load_a = [&](int slice) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = ar0 + (256 / CH) * j;
            if (NORMA) {
                const float* p = (const float*)A + (long)row * lda + slice * KS + ach * 8;
                areg32[NORMA ? j : 0][0] = row < M ? *(const float4*)p : make_float4(0.f, 0.f, 0.f, 0.f);
                areg32[NORMA ? j : 0][1] = row < M ? *(const float4*)(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < M) v = *(const uint4*)(A + (long)row * lda + slice * KS + ach * 8);
                areg[j] = v;
            }
        }
    };
    auto store_a = [&](char* buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = ar0 + (256 / CH) * j;
            if (NORMA) {
                const float4 lo = areg32[NORMA ? j : 0][0], hi = areg32[NORMA ? j : 0][1];
                if (sums) ss[NORMA ? j : 0] += lo.x * lo.x + lo.y * lo.y + lo.z * lo.z + lo.w * lo.w + hi.x * hi.x + hi.y * hi.y + hi.z * hi.z + hi.w * hi.w;
                areg[j] = make_uint4(pack_bf2(lo.x, lo.y), pack_bf2(lo.z, lo.w), pack_bf2(hi.x, hi.y), pack_bf2(hi.z, hi.w));
            }
            *(uint4*)(buf + row * ROWB + ((ach ^ (row & 15)) * 16)) = areg[j];
        }
    };
    const bf16_t* bbase;
    if (PACKED) bbase = B + ((long)min(n0 >> 4, (N >> 4) - 1) * (K >> 5)) * 512 + lane * 8;
    else bbase = B + (long)min(n0 + l15, N - 1) * ldb + g * 8;
    auto load_w = [&](u32x4 (&w)[NU], int slice) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bf16_t* p = PACKED ? bbase + ((long)slice * NU + u) * 512 : bbase + slice * KS + u * 32;
            w[u] = __builtin_nontemporal_load((const u32x4*)p);
        }
    };
    // fifth fragment of a wide workgroup: this wave's quarter of the k-steps (N5 * wave .. + N5 - 1) of every slice
    constexpr int N5 = NU / 4;
    const bf16_t* bbase5 = B + ((long)min(n5 >> 4, (N >> 4) - 1) * (K >> 5)) * 512 + lane * 8;
    auto load_w5 = [&](u32x4 (&w)[N5], int slice) {
#pragma unroll
        for (int q = 0; q < N5; ++q)
            w[q] = __builtin_nontemporal_load((const u32x4*)(bbase5 + ((long)slice * NU + N5 * wave_u + q) * 512));
    };
    f32x4 acc[MF], acc5[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc5[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto compute = [&](const u32x4 (&w)[NU], const u32x4 (&w5)[N5], const char* buf) {
        // A fragments double-buffered by hand and the scheduler fenced per k-step (left alone hipcc hoists all 32
        // fragment reads of the slice and spills)
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
            for (int mf = 0; mf < MF; ++mf)
                acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][mf], wf, acc[mf], 0, 0, 0);   // D[m][n]
            if (SWIGLU && wide && (u / N5) == wave_u) {                // wave-uniform: this wave's share of the fifth fragment
                const bf16x8 wf5 = __builtin_bit_cast(bf16x8, w5[u % N5]);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
                    acc5[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][mf], wf5, acc5[mf], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    u32x4 wa[NU], wb[NU], wa5[N5], wb5[N5];
#pragma unroll
    for (int q = 0; q < N5; ++q) { wa5[q] = (u32x4){0u, 0u, 0u, 0u}; wb5[q] = (u32x4){0u, 0u, 0u, 0u}; }
    load_a(s_begin);
    load_w(wa, s_begin);
    if (wide) load_w5(wa5, s_begin);
    int s = s_begin;
    while (true) {
        store_a(smem[0]);
        __syncthreads();
        if (s + 1 < s_end) { load_a(s + 1); load_w(wb, s + 1); if (wide) load_w5(wb5, s + 1); }
        compute(wa, wa5, smem[0]);
        if (++s >= s_end) break;
        store_a(smem[1]);
        __syncthreads();
        if (s + 1 < s_end) { load_a(s + 1); load_w(wa, s + 1); if (wide) load_w5(wa5, s + 1); }
        compute(wb, wb5, smem[1]);
        if (++s >= s_end) break;
    }
    __shared__ float rstd_lds[SWIGLU && NORMA ? 64 : 1];
    if (NORMA && sums) {
        // the CH = 32 threads that share a row are one half wave: reduce, one atomic per row per K range
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float v = ss[NORMA ? j : 0];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int row = ar0 + (256 / CH) * j;
            if (SWIGLU) { if (ach == 0) rstd_lds[SWIGLU && NORMA ? row : 0] = rsqrtf(v / (float)K + eps); }   // whole K in this workgroup
            else if (ach == 0 && row < M) atomicAdd(rowss + row, v);
        }
    }
    if (SWIGLU && NORMA) __syncthreads();
    // lane holds C[m = mf*16 + g*4 + r][n = n0 + l15]
    const int n = n0 + l15;
    if (SWIGLU && split_block) {
        // partial sums -> scratch tile of this column group ([64 rows][64 cols] fp32), then the arrival ticket
        float* tile = scratch + (long)(cgroup - ((N >> 6) - split_groups)) * 64 * 64 * MT;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_fetch_add(tile + (mf * 16 + g * 4 + r) * 64 + wave * 16 + l15, acc[mf][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's atomics are performed before it reaches the barrier
        __syncthreads();
        __shared__ int last_flag;
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(tickets + (cgroup - ((N >> 6) - split_groups)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (t == split_ranges - 1);
        }
        __syncthreads();
        if (!last_flag) return;
        // last arriver: every other block's atomics preceded its ticket; read the totals (agent-scope loads), re-zero
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // exchange = read the total AND re-zero in one read-modify-write at the same place the adds were performed
                // (a plain or sc1 load could be served by this XCD's L2)
                float* p = tile + (mf * 16 + g * 4 + r) * 64 + wave * 16 + l15;
                acc[mf][r] = __hip_atomic_exchange(p, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        if (tid == 0) __hip_atomic_exchange(tickets + (cgroup - ((N >> 6) - split_groups)), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (SWIGLU) {
        bf16_t* Y = (bf16_t*)C;
        const int col = (n0 >> 1) + (l15 & 7);                               // output column of this gate/up pair
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mf * 16 + g * 4 + r;
                const float rs = (SWIGLU && NORMA) ? rstd_lds[SWIGLU && NORMA ? m : 0] : 1.f;
                const float other = __shfl_xor(acc[mf][r], 8) * rs;
                if (l15 < 8 && m < mflush && n < N) {
                    const float gv = acc[mf][r] * rs;
                    Y[(long)m * ldc + col] = f2bf(gv / (1.f + __expf(-gv)) * other);
                }
            }
        if (wide) {
            // fifth fragment: the four waves' K quarters meet in LDS ([wave][row block][lane] float4, 16 KiB x MT of the A buffers,
            // which every wave has left: barrier first), wave w sums row blocks w, w + 4, .. in wave order and runs the same epilogue
            __syncthreads();
            float4* part = (float4*)&smem[0][0];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
                part[(wave * MF + mf) * 64 + lane] = make_float4(acc5[mf][0], acc5[mf][1], acc5[mf][2], acc5[mf][3]);
            __syncthreads();
            const int col5 = (n5 >> 1) + (l15 & 7);
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int mf = wave + 4 * b;                                 // MF = 4 MT row blocks, MT per wave
                if (SMALL && mf >= MF) break;                                // <= 16 rows: row block 0 only (wave 0 finishes it)
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float4 t = part[(w * MF + mf) * 64 + lane];
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mf * 16 + g * 4 + r;
                    const float rs = (SWIGLU && NORMA) ? rstd_lds[SWIGLU && NORMA ? m : 0] : 1.f;
                    const float other = __shfl_xor(v[r], 8) * rs, gv = v[r] * rs;
                    if (l15 < 8 && m < mflush && n5 + l15 < N)
                        Y[(long)m * ldc + col5] = f2bf(gv / (1.f + __expf(-gv)) * other);
                }
            }
        }
        return;
    }
    const bool whole_k = (s_begin == 0 && s_end == total_slices);
    if (n < N) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mf * 16 + g * 4 + r;
                if (m < mflush) {
                    float* c = C + (long)m * ldc + n;
                    if (whole_k) *c = overwrite ? acc[mf][r] : *c + acc[mf][r];      // overwrite: C = A.B^T, no zero fill needed
                    else atomicAdd(c, acc[mf][r]);
                }
            }
    }
}

// W [N, K] row-major -> fragment-major copy for the decode loop (rebuilt once per optimizer step)
// half > 0 (N = 2*half, rows [0,half) gate, [half,N) up): column group nb takes gate rows 8nb..8nb+7 then up rows
// half+8nb..half+8nb+7, the layout the SWIGLU epilogue of gemm_skinny_kernel expects.
__global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t* __restrict__ W, long ld, bf16_t* __restrict__ out, int N, int K,
                                                        int half) {
    const long total = (long)(N >> 4) * (K >> 5) * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long frag = i >> 6;
        const int kb = (int)(frag % (K >> 5));
        const long nb = frag / (K >> 5);
        const int p = lane & 15;
        const long row = half > 0 ? (p < 8 ? nb * 8 + p : half + nb * 8 + (p - 8)) : nb * 16 + p;
        const uint4 v = *(const uint4*)(W + row * ld + kb * 32 + (lane >> 4) * 8);
        *(uint4*)(out + i * 8) = v;
    }
}

// =============================================================================== rotary table for the step
// cos/sin [B, D] for text position pos_base[b] + step (all three M-RoPE rows equal for generated tokens)
__global__ void decode_rope_table_kernel(const int* __restrict__ pos_base, const int* __restrict__ step, float theta,
                                         float* __restrict__ cs, float* __restrict__ sn, int B, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (D / 2)) return;
    const int b = i / (D / 2), j = i % (D / 2);
    const float inv = powf(theta, -(float)(2 * j) / (float)D);
    const float ang = (float)(pos_base[b] + *step) * inv;
    float s, c;
    sincosf(ang, &s, &c);
    cs[b * D + j] = c; cs[b * D + j + D / 2] = c;
    sn[b * D + j] = s; sn[b * D + j + D / 2] = s;
}

// =============================================================================== q/k/v finishing
// acc32 [B, (Hq+2Hkv)*D] (fp32 split-K sums, re-zeroed here) + bias -> rotary(q), rotary(k) ->
// q_out bf16 [B, Hq*D]; k, v appended to the tail cache at position *tail_len.
// rowss != NULL (norm-folded projection): acc holds bf16(x) (W diag(w))^T; every value is scaled by rstd[b] =
// rsqrt(rowss[b] / norm_cols + eps) first.  rowss_zero (the row sums the NEXT norm-folded launch will add into) is cleared here.
__global__ __launch_bounds__(256) void decode_qkv_finish_kernel(float* __restrict__ acc, const bf16_t* __restrict__ bias,
                                                                const float* __restrict__ cs, const float* __restrict__ sn,
                                                                bf16_t* __restrict__ q_out, bf16_t* __restrict__ tail_k,
                                                                bf16_t* __restrict__ tail_v, const int* __restrict__ tail_len,
                                                                int B, int Hq, int Hkv, int D, int Cmax,
                                                                const float* __restrict__ rowss, float* __restrict__ rowss_zero,
                                                                int norm_cols, float eps) {
    const int half = D / 2, heads = Hq + 2 * Hkv;
    const int total = B * heads * half;
    const int pos = *tail_len;
    if (rowss_zero && blockIdx.x == 0 && threadIdx.x < B) rowss_zero[threadIdx.x] = 0.f;
    // every load of an element is issued before the first use (round 6): with the uses next to the loads (row sum -> rsqrt, sums + bias,
    // the rotary tables behind `if (hh < ...)`) the kernel was four dependent round trips; absent operands (no bias / no folded norm) read a
    // valid dummy address and are masked, so there is no branch between the loads
    const bf16_t* bias_p = bias ? bias : (const bf16_t*)cs;
    const float* rowss_p = rowss ? rowss : cs;
    const float bias_on = bias ? 1.f : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i % half, hh = (i / half) % heads, b = i / (half * heads);
        float* a = acc + ((long)b * heads + hh) * D;
        const float rsum = rowss_p[rowss ? b : 0];
        const float a1 = a[j], a2 = a[j + half];
        const bf16_t b1 = bias_p[bias ? hh * D + j : 0], b2 = bias_p[bias ? hh * D + j + half : 0];
        const float c1 = cs[b * D + j], s1 = sn[b * D + j], c2 = cs[b * D + j + half], s2 = sn[b * D + j + half];
        const float rs = rowss ? rsqrtf(rsum / (float)norm_cols + eps) : 1.f;
        float x1 = a1 * rs + bias_on * bf2f(b1);
        float x2 = a2 * rs + bias_on * bf2f(b2);
        a[j] = 0.f; a[j + half] = 0.f;
        if (hh < Hq + Hkv) {   // rotary on q and k heads
            const float r1 = x1 * c1 - x2 * s1, r2 = x2 * c2 + x1 * s2;
            x1 = r1; x2 = r2;
        }
        bf16_t* dst;
        if (hh < Hq) dst = q_out + ((long)b * Hq + hh) * D;
        else if (hh < Hq + Hkv) dst = tail_k + (((long)b * Cmax + pos) * Hkv + (hh - Hq)) * D;
        else dst = tail_v + (((long)b * Cmax + pos) * Hkv + (hh - Hq - Hkv)) * D;
        dst[j] = f2bf(x1); dst[j + half] = f2bf(x2);
    }
}

// y bf16 [B, I] = silu(gate) * up from fp32 acc [B, 2I] (re-zeroed)
__global__ __launch_bounds__(256) void swiglu_f32_kernel(float* __restrict__ acc, bf16_t* __restrict__ y, int B, int I) {
    const long total = (long)B * I;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / I; const int c = (int)(i % I);
        float* gp = acc + b * 2 * I + c;
        const float gv = gp[0], uv = gp[I];
        gp[0] = 0.f; gp[I] = 0.f;
        y[i] = f2bf(gv / (1.f + __expf(-gv)) * uv);
    }
}

// x32 += 0 helper not needed; residual adds land in the fp32 stream through the skinny GEMM atomics.

// =============================================================================== decode attention: shared prompt keys
// The K rollouts of one prompt all attend the same prompt keys.  Workgroup = (prompt, kv head, key split): the query
// COLUMNS are the (rollout, q-head) pairs of the prompt (Kn * REP <= 64, one 16-column block per wave), the prompt's
// K / V tiles are staged once in LDS (row-major K image + transposed V image, as in the prefill kernel) and scored
// against all columns by MFMA.  Per-column partial (O, m, l) go to a small fp32 workspace; the tail kernel merges them.
// Each prompt key is read once per kv head instead of once per rollout (per-CU load bandwidth is what bounds decode
// attention: ~24 GB/s per CU, so re-reading the prompt in all 8 rollouts' workgroups costs 8x the time).
constexpr int PRE_SPLITS = 8;


// =============================================================================== decode attention (MFMA)
// Workgroup = (sequence b, kv head); the REP q-heads of the GQA group are the MFMA's 16-wide N dimension (zero
// padded), so every K/V byte is read once per workgroup and scored against all heads by the matrix cores:
//   S^T = K . Q^T   (A = K fragment loaded straight from global memory, row-per-lane 16 B; B = Q in registers)
//   O^T += V^T . P^T (V tile transposed through a wave-private LDS region, same slot permutation as the forward kernel)
// The 4 waves stride over 64-key tiles of [shared prompt KV | per-rollout tail KV] and are merged through LDS.
template <int REP, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4) ? 2 : 1) void attn_decode_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ pk,
                                                                  const bf16_t* __restrict__ pv, const int* __restrict__ plen,
                                                                  const int* __restrict__ prompt_of, const bf16_t* __restrict__ tk,
                                                                  const bf16_t* __restrict__ tv, const int* __restrict__ tail_len,
                                                                  bf16_t* __restrict__ o, int Pmax, int Cmax, int Hq, int Hkv,
                                                                  float scale, const float* __restrict__ pre, int Kn) {
    // pre != nullptr: the prompt keys were already scored for all rollouts of the prompt by attn_decode_prefix_kernel;
    // this launch covers the tail keys only and folds the PRE_SPLITS prefix partials into its merge.
    constexpr int D = 128, DC = 4, DF = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];               // NW x 16 KiB: V^T per wave, then the merge
    const int b = blockIdx.x, hk = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int pr = prompt_of[b], P = pre ? 0 : plen[pr], total = P + *tail_len + 1;
    char* vt = smem + wave * AT_T_BYTES(D);

    bf16x8 qf[DC];
    {
        const bf16_t* qp = q + ((long)b * Hq + hk * REP + min(l15, REP - 1)) * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (l15 < REP) t = *(const uint4*)(qp + dc * 32 + g * 8);
            qf[dc] = __builtin_bit_cast(bf16x8, t);
        }
    }
    auto key_ptr = [&](const bf16_t* pre, const bf16_t* tail, int key) -> const bf16_t* {
        key = min(key, total - 1);
        return key < P ? pre + (((long)pr * Pmax + key) * Hkv + hk) * D : tail + (((long)b * Cmax + (key - P)) * Hkv + hk) * D;
    };
    f32x4 oacc[DF];
#pragma unroll
    for (int d = 0; d < DF; ++d) oacc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = wave * 64; k0 < total; k0 += NW * 64) {
        // ---- V tile -> registers (4 passes: d-chunk c = lane & 15, row quad rq = (lane >> 4) + 4 i), issued first
        uint4 vreg[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = k0 + (g + 4 * i) * 4 + j;
                vreg[i][j] = *(const uint4*)(key_ptr(pv, tv, key) + l15 * 8);
            }
        // ---- S^T = K . Q^T : st[kf] = keys k0 + kf*16 + g*4 + r for head q = l15
        f32x4 st[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            st[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16_t* kp = key_ptr(pk, tk, k0 + kf * 16 + l15) + g * 8;
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const bf16x8 kfr = __builtin_bit_cast(bf16x8, *(const uint4*)(kp + dc * 32));
                st[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[dc], st[kf], 0, 0, 0);
            }
        }
        float p[4][4], mx = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sv = (k0 + kf * 16 + g * 4 + r < total) ? st[kf][r] * scale : -INFINITY;
                p[kf][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);           // finite: every tile a wave visits has >= 1 valid key
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[kf][r] = __expf(p[kf][r] - m_new); psum += p[kf][r]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[d] *= alpha;
        const bf16x8 pf0 = pack_slots(p[0], p[1]), pf1 = pack_slots(p[2], p[3]);
        // ---- V^T through the wave-private LDS image
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rq = g + 4 * i;
            const uint32_t w[4][4] = {{vreg[i][0].x, vreg[i][0].y, vreg[i][0].z, vreg[i][0].w}, {vreg[i][1].x, vreg[i][1].y, vreg[i][1].z, vreg[i][1].w},
                                      {vreg[i][2].x, vreg[i][2].y, vreg[i][2].z, vreg[i][2].w}, {vreg[i][3].x, vreg[i][3].y, vreg[i][3].z, vreg[i][3].w}};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = l15 * 8 + e, wi = e >> 1, hi = e & 1;
                uint32_t x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = hi ? (w[j][wi] >> 16) : (w[j][wi] & 0xffffu);
                *(uint2*)(vt + d * AT_T_ROW_BYTES + ((rq ^ vswz(d)) * 8)) = make_uint2(x[0] | (x[1] << 16), x[2] | (x[3] << 16));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image complete before any lane reads fragments from it
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            oacc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_t(vt, df, 0, lane), pf0, oacc[df], 0, 0, 0);
            oacc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_t(vt, df, 1, lane), pf1, oacc[df], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments read before the next tile overwrites the image
        __builtin_amdgcn_wave_barrier();
    }
    // ---- merge the 4 waves: per wave m[q], l[q], O^T[d][q] (q = l15 < REP)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __syncthreads();                                   // every wave is done with its V^T image
    float* mo = (float*)smem;                          // [NW][D][16] O^T, then [NW][16] m, [NW][16] l
    float* mm = mo + NW * D * 16;
    float* ml = mm + NW * 16;
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int r = 0; r < 4; ++r) mo[(wave * D + df * 16 + g * 4 + r) * 16 + l15] = oacc[df][r];
    if (g == 0) { mm[wave * 16 + l15] = m_run; ml[wave * 16 + l15] = l_run; }
    __syncthreads();
    for (int i = tid; i < REP * D; i += NW * 64) {
        const int qh = i / D, d = i % D;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, mm[w * 16 + qh]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = mm[w * 16 + qh];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L += ml[w * 16 + qh] * f;
            O += mo[(w * D + d) * 16 + qh] * f;
        }
        if (pre) {      // prefix partials of column (rollout within prompt, head): [prompt][hk][split][64 cols][D + 2]
            const int col = (b - pr * Kn) * REP + qh;
            const float* pp = pre + ((long)(pr * Hkv + hk) * PRE_SPLITS * 64 + col) * (D + 2);
            float M2 = M;
#pragma unroll
            for (int sp = 0; sp < PRE_SPLITS; ++sp) M2 = fmaxf(M2, pp[(long)sp * 64 * (D + 2) + D]);
            const float f0 = (M == -INFINITY) ? 0.f : __expf(M - M2);
            L *= f0; O *= f0;
#pragma unroll
            for (int sp = 0; sp < PRE_SPLITS; ++sp) {
                const float* q1 = pp + (long)sp * 64 * (D + 2);
                const float ms = q1[D];
                const float f = (ms == -INFINITY) ? 0.f : __expf(ms - M2);
                L += q1[D + 1] * f;
                O += q1[d] * f;
            }
        }
        o[((long)b * Hq + hk * REP + qh) * D + d] = f2bf(O / L);
    }
}

// =============================================================================== decode attention, shared prompt: one launch + merge
// The prompt-key blocks (role A: prompt, kv head, key split -- 64 (rollout, head) columns) and the tail-key blocks (role B:
// sequence, kv head -- REP columns) are independent until the final softmax merge, so they run in ONE launch (512 co-resident
// workgroups at the cfg3 shape) and a small merge kernel combines the PRE_SPLITS prompt partials with the tail partial.
// Before: prefix kernel -> tail kernel (which also merged) = two latency-bound launches in series, 21.5 us per layer at an
// empty tail.  V tiles are staged row-major and read through the transposing LDS read (frag_tr) in both roles; role A keeps
// two K/V tiles in flight in registers.
template <int REP>
__global__ __launch_bounds__(256, 2) void attn_decode_split_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ pk,
                                                                   const bf16_t* __restrict__ pv, const int* __restrict__ plen,
                                                                   const bf16_t* __restrict__ tk, const bf16_t* __restrict__ tv,
                                                                   const int* __restrict__ tail_len, float* __restrict__ pre,
                                                                   float* __restrict__ tailp, int n_prefix_blocks, int Kn, int Pmax,
                                                                   int Cmax, int Hq, int Hkv, float scale, const int* __restrict__ row0) {
    // row0 != nullptr (round 6): prompt pr owns the rows [row0[pr], row0[pr + 1]) -- per-prompt rollout counts (the T-GRPO twins take
    // G / 2 rollouts, TR:473); nullptr: the uniform layout, prompt pr owns rows [pr Kn, (pr + 1) Kn)
    constexpr int D = 128, DC = 4, DF = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];               // 64 KiB: A: K + V images; B: 4 wave-private V images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    f32x4 oacc[DF];
#pragma unroll
    for (int d = 0; d < DF; ++d) oacc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // one 64-key tile of scores st (K-major C layout) -> online softmax -> O^T += V^T . P^T with V^T fragments from `v_img`
    auto softmax_pv = [&](f32x4 (&st)[4], int key0, int n_valid, const char* v_img) {
        float p[4][4], mx = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sv = (key0 + kf * 16 + g * 4 + r < n_valid) ? st[kf][r] * scale : -INFINITY;
                p[kf][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);           // finite: a visited tile has >= 1 valid key
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[kf][r] = __expf(p[kf][r] - m_new); psum += p[kf][r]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[d] *= alpha;
        const bf16x8 pf0 = pack_slots(p[0], p[1]), pf1 = pack_slots(p[2], p[3]);
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            oacc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(v_img, df, 0, lane), pf0, oacc[df], 0, 0, 0);
            oacc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(v_img, df, 1, lane), pf1, oacc[df], 0, 0, 0);
        }
    };

    if ((int)blockIdx.x < n_prefix_blocks) {
        // ------------------------------------------------------------------ role A: prompt keys x all columns of the prompt
        const int sp = blockIdx.x % PRE_SPLITS, hk = (blockIdx.x / PRE_SPLITS) % Hkv, pr = blockIdx.x / (PRE_SPLITS * Hkv);
        char* k_lds = smem;
        char* v_lds = smem + AT_RM_BYTES;
        const int P = plen[pr];
        const int tiles = (P + 63) >> 6, tps = (tiles + PRE_SPLITS - 1) / PRE_SPLITS;
        const int t0 = sp * tps, t1 = min(tiles, t0 + tps);
        const int col = wave * 16 + l15;                       // (rollout, head) column of this lane
        const int r0 = row0 ? row0[pr] : pr * Kn, kcnt = row0 ? row0[pr + 1] - r0 : Kn;
        const bool col_ok = col < kcnt * REP;
        const long row_stride = (long)Hkv * D;
        bf16x8 qf[DC];
        uint4 ka[4], va[4], kb[4], vb[4];
        auto fetch = [&](uint4 (&kr_)[4], uint4 (&vr_)[4], int t) {
            const long off = (((long)pr * Pmax + t * 64) * Hkv + hk) * D;
            tile_load<D>(kr_, pk + off, row_stride, P - t * 64, tid);
            tile_load<D>(vr_, pv + off, row_stride, P - t * 64, tid);
        };
        auto step = [&](uint4 (&kr_)[4], uint4 (&vr_)[4], int t) {
            __syncthreads();
            tile_store<D, true, false>(kr_, k_lds, nullptr, tid);
            tile_store<D, true, false>(vr_, v_lds, nullptr, tid);
            __syncthreads();
            if (t + 2 < t1) fetch(kr_, vr_, t + 2);              // this register set is free again: two tiles stay in flight
            f32x4 st[4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                st[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dc = 0; dc < DC; ++dc)
                    st[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(k_lds, kf, dc, lane), qf[dc], st[kf], 0, 0, 0);
            }
            softmax_pv(st, t * 64, P, v_lds);
        };
        {   // q fragments first, then the tile requests (in-order vmcnt: the first tile does not wait for the later ones)
            const int kr = col_ok ? col / REP : 0, hr = col_ok ? col % REP : 0;
            const bf16_t* qp = q + ((long)(r0 + kr) * Hq + hk * REP + hr) * D;
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                uint4 t = make_uint4(0, 0, 0, 0);
                if (col_ok) t = *(const uint4*)(qp + dc * 32 + g * 8);
                qf[dc] = __builtin_bit_cast(bf16x8, t);
            }
        }
        // (a third register set with all of a split's 3 tiles in flight at once: 16.3 vs 17.4 us per layer in the standalone
        // probe, 18.0 vs 16.7 us inside the decode step -- measured with rocprof in one process; not kept)
        if (t0 < t1) fetch(ka, va, t0);
        if (t0 + 1 < t1) fetch(kb, vb, t0 + 1);
        for (int t = t0; t < t1; t += 2) {
            step(ka, va, t);
            if (t + 1 < t1) step(kb, vb, t + 1);
        }
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        // lane holds O^T[d = df*16 + g*4 + r][col]; partial record = [D floats O | m | l]
        float* rec = pre + (((long)(pr * Hkv + hk) * PRE_SPLITS + sp) * 64 + col) * (D + 2);
#pragma unroll
        for (int df = 0; df < DF; ++df)
            *(float4*)(rec + df * 16 + g * 4) = make_float4(oacc[df][0], oacc[df][1], oacc[df][2], oacc[df][3]);
        if (g == 0) { rec[D] = m_run; rec[D + 1] = l_run; }
        return;
    }
    // ---------------------------------------------------------------------- role B: the sequence's own (generated) keys
    const int bi = blockIdx.x - n_prefix_blocks;
    const int b = bi / Hkv, hk = bi % Hkv;
    const int total = *tail_len + 1;
    char* v_img = smem + wave * AT_RM_BYTES;
    bf16x8 qf[DC];
    {
        const bf16_t* qp = q + ((long)b * Hq + hk * REP + min(l15, REP - 1)) * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (l15 < REP) t = *(const uint4*)(qp + dc * 32 + g * 8);
            qf[dc] = __builtin_bit_cast(bf16x8, t);
        }
    }
    auto key_ptr = [&](const bf16_t* base, int key) -> const bf16_t* {
        return base + (((long)b * Cmax + min(key, total - 1)) * Hkv + hk) * D;
    };
    for (int k0 = wave * 64; k0 < total; k0 += 4 * 64) {
        // V tile -> registers first (lane: 16-byte chunk l15 of rows (g + 4 i) * 4 + j), then K fragments straight from memory
        uint4 vreg[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) vreg[i][j] = *(const uint4*)(key_ptr(tv, k0 + (g + 4 * i) * 4 + j) + l15 * 8);
        f32x4 st[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            st[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16_t* kp = key_ptr(tk, k0 + kf * 16 + l15) + g * 8;
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const bf16x8 kfr = __builtin_bit_cast(bf16x8, *(const uint4*)(kp + dc * 32));
                st[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[dc], st[kf], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (g + 4 * i) * 4 + j;
                *(uint4*)(v_img + row * AT_RM_ROW_BYTES + ((l15 ^ (row & 15)) * 16)) = vreg[i][j];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private image complete before any lane reads fragments
        __builtin_amdgcn_wave_barrier();
        softmax_pv(st, k0, total, v_img);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments read before the next tile overwrites the image
        __builtin_amdgcn_wave_barrier();
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __syncthreads();                                   // every wave is done with its V image
    float* mo = (float*)smem;                          // [4][D][16] O^T, then [4][16] m, [4][16] l
    float* mm = mo + 4 * D * 16;
    float* ml = mm + 4 * 16;
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int r = 0; r < 4; ++r) mo[(wave * D + df * 16 + g * 4 + r) * 16 + l15] = oacc[df][r];
    if (g == 0) { mm[wave * 16 + l15] = m_run; ml[wave * 16 + l15] = l_run; }
    __syncthreads();
    float* rec0 = tailp + (long)bi * REP * (D + 2);
    for (int i = tid; i < REP * (D + 2); i += 256) {
        const int qh = i / (D + 2), d = i % (D + 2);
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, mm[w * 16 + qh]);        // finite: wave 0 always has key 0
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = mm[w * 16 + qh];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            acc += (d < D ? mo[(w * D + d) * 16 + qh] : ml[w * 16 + qh]) * f;
        }
        rec0[i] = d == D ? M : acc;                    // [D floats O | m | l] like the prefix records (slot D+1 = l)
    }
}

// o[b, head, :] = softmax-merge of the PRE_SPLITS prompt partials (column (b - pr*Kn)*REP + qh) and the tail partial.
// One thread per output element (REP*128 threads), all 3 x (PRE_SPLITS + 1) loads independent of each other: ONE memory round
// trip (the records were just written from other XCDs); the 256-thread form looped 3.5 outputs per thread, a round trip each.
template <int REP>
__global__ __launch_bounds__(REP * 128) void attn_decode_merge_kernel(const float* __restrict__ pre, const float* __restrict__ tailp,
                                                                      bf16_t* __restrict__ o, int Kn, int Hq, int Hkv,
                                                                      const int* __restrict__ prompt_of, const int* __restrict__ row0) {
    constexpr int D = 128;
    const int b = blockIdx.x, hk = blockIdx.y, pr = row0 ? prompt_of[b] : b / Kn;
    const int qh = threadIdx.x >> 7, d = threadIdx.x & 127;
    const float* tp = tailp + ((long)b * Hkv + hk) * REP * (D + 2) + qh * (D + 2);
    const int col = (b - (row0 ? row0[pr] : pr * Kn)) * REP + qh;
    const float* pp = pre + ((long)(pr * Hkv + hk) * PRE_SPLITS * 64 + col) * (D + 2);
    float ms[PRE_SPLITS + 1], ls[PRE_SPLITS + 1], os[PRE_SPLITS + 1];
#pragma unroll
    for (int sp = 0; sp < PRE_SPLITS; ++sp) {
        const float* q1 = pp + (long)sp * 64 * (D + 2);
        ms[sp] = q1[D]; ls[sp] = q1[D + 1]; os[sp] = q1[d];
    }
    ms[PRE_SPLITS] = tp[D]; ls[PRE_SPLITS] = tp[D + 1]; os[PRE_SPLITS] = tp[d];
    float M = ms[PRE_SPLITS];                                   // finite: the tail always holds the current token's key
#pragma unroll
    for (int sp = 0; sp < PRE_SPLITS; ++sp) M = fmaxf(M, ms[sp]);
    // same order of additions as before: tail first, then the prompt partials
    const float ft = __expf(ms[PRE_SPLITS] - M);
    float L = ls[PRE_SPLITS] * ft, O = os[PRE_SPLITS] * ft;
#pragma unroll
    for (int sp = 0; sp < PRE_SPLITS; ++sp) {
        const float f = (ms[sp] == -INFINITY) ? 0.f : __expf(ms[sp] - M);
        L += ls[sp] * f;
        O += os[sp] * f;
    }
    o[((long)b * Hq + hk * REP + qh) * D + d] = f2bf(O / L);
}


}  // namespace

// launch-plan helpers: the resident workgroup slots of a decode GEMM launch are 2 per CU of the caller's CU budget
static inline int plan_cus(const spacer_plan* plan) { return plan && plan->cus > 0 ? plan->cus : 256; }
static inline int skinny_target_blocks(const spacer_plan* plan) {
    return plan && plan->skinny_blocks > 0 ? plan->skinny_blocks : 2 * plan_cus(plan);
}

// K-range shape of a K-split launch: 0 = equal ranges of ceil(slices / ranges); k + 1 = shares 1 + (k / 16)(2 r / (R - 1) - 1) (kernel).
// Default rule (plan->skinny_skew == 0): a launch that fills the resident slots with SHORT ranges (<= 2 slices each: the 7B q|k|v
// projection, 72 column groups x 7 ranges of 2) ends with every workgroup flushing its atomics at the same moment; skewing the ranges
// (alpha = 0.375: 1,2,2,2,2,2,3 slices; 0.5 and more measured worse) staggers the flushes under the other ranges' streaming: 16.8 -> 15.0 us
// (norm-folded), 14.7 -> 13.3 us (bf16 A)
// (scripts/probes/decode_gemm_times.py, A/B in one process).  Launches that do not fill the slots (o: 392 workgroups) or have long
// ranges (down: 8-9 slices) measured slower with a skew and keep equal ranges.  plan->skinny_skew < 0 forces equal ranges.
static inline int skinny_skew(const spacer_plan* plan, int col_groups, int ranges, int slices, int target_blocks) {
    if (ranges <= 1) return 0;
    if (plan && plan->skinny_skew != 0) return plan->skinny_skew > 0 ? plan->skinny_skew : 0;       // range-checked by the launchers
    return (col_groups * ranges >= target_blocks - target_blocks / 16 && cdiv(slices, ranges) <= 2 && slices >= 2 * ranges) ? 7 : 0;
}

static int launch_skinny(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                         const spacer_gemm_epilogue* epi, bool packed, const spacer_plan* plan, hipStream_t s, bool overwrite = false) {
    SP_REQUIRE(A && B && C, SPACER_EINVAL, "gemm_skinny: null operand");
    SP_REQUIRE_PLAN(plan);
    SP_REQUIRE(!plan || plan->skinny_skew <= 17, SPACER_EINVAL, "gemm_skinny: plan.skinny_skew = %d out of range (<= 17: alpha = (skew - 1) / 16 <= 1 keeps the K ranges monotone)", plan->skinny_skew);
    SP_REQUIRE(M > 0 && M <= (packed ? 128 : 64), SPACER_EINVAL, "gemm_skinny: M=%d must be in 1..%d", M, packed ? 128 : 64);
    SP_REQUIRE(K % 256 == 0, SPACER_EINVAL, "gemm_skinny: K=%d must be a multiple of 256", K);
    const int MTv = M > 64 ? 2 : 1, KSv = 256 / MTv;          // 65..128 rows: two 64-row blocks per weight pass, 128-wide K slices
    SP_REQUIRE(lda % 8 == 0 && (packed || ldb % 8 == 0), SPACER_EINVAL, "gemm_skinny: lda/ldb must be multiples of 8");
    SP_REQUIRE(!packed || N % 16 == 0, SPACER_EINVAL, "gemm_skinny: packed weights need N %% 16 == 0");
    SP_REQUIRE(!epi || (epi->out_f32 && !epi->bias && epi->act == 0 && (!epi->residual || epi->residual == C)),
               SPACER_EINVAL, "gemm_skinny: only fp32 accumulate-into-C is supported (C32 += A.B^T)");
    // K ranges: 1 (no atomics) when the column groups alone fill the chip, else just enough ranges for ~2 workgroups / CU
    const int col_groups = cdiv(N, 64), slices = K / KSv;
    int ranges = 1;
    // spacer_plan::skinny_blocks = 1 forces one K range per column group (no atomics: bit-reproducible sums)
    const int target_blocks = skinny_target_blocks(plan);
    // as many K ranges as keep the whole launch in ONE resident round (2 workgroups x 256 CUs): down-proj at 7B (56 column
    // groups x 74 slices) ran as 560 blocks = a full round + a 48-block tail before; now 9 ranges = 504 blocks
    if (col_groups < target_blocks - target_blocks / 8) ranges = max(1, min(slices, target_blocks / col_groups));
    int spr = cdiv(slices, ranges);
    const int skew = skinny_skew(plan, col_groups, ranges, slices, target_blocks);
    if (skew > 0) spr = -skew;                 // skewed / even ranges: the kernel derives them from gridDim.y
    else ranges = cdiv(slices, spr);
    SP_REQUIRE(!overwrite || ranges == 1, SPACER_EINVAL, "gemm_skinny: C = A.B^T (store form) needs whole-K workgroups; N=%d splits K %d ways", N, ranges);
    const int mflush = M;
    if (packed && MTv == 2)
        hipLaunchKernelGGL((gemm_skinny_kernel<true, false, 2>), dim3(col_groups, ranges), dim3(256), 0, s, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K, spr, mflush, overwrite ? 1 : 0);
    else if (packed)
        hipLaunchKernelGGL(gemm_skinny_kernel<true>, dim3(col_groups, ranges), dim3(256), 0, s, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K, spr, mflush, overwrite ? 1 : 0);
    else
        hipLaunchKernelGGL(gemm_skinny_kernel<false>, dim3(col_groups, ranges), dim3(256), 0, s, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K, spr, mflush, overwrite ? 1 : 0);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_gemm_skinny_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                                       int K, const spacer_gemm_epilogue* epi, spacer_stream_t stream) {
    return launch_skinny(A, lda, B, ldb, C, ldc, M, N, K, epi, false, epi ? epi->plan : nullptr, (hipStream_t)stream);
}

extern "C" int spacer_gemm_skinny_packed_bf16(const void* A, long lda, const void* Bpacked, void* C, long ldc, int M, int N,
                                              int K, const spacer_plan* plan, spacer_stream_t stream) {
    return launch_skinny(A, lda, Bpacked, 0, C, ldc, M, N, K, nullptr, true, plan, (hipStream_t)stream);
}

// C32[M,N] += bf16(X32[M,K]) . Wp[N,K]^T and rowss[m] += sum_k X32[m,k]^2: the K-split decode projection with the RMSNorm in
// front of it folded in (see gemm_skinny_kernel NORMA); Wp = packed W diag(w_norm).  M <= 64.
extern "C" int spacer_gemm_skinny_packed_normed(const float* X32, long ldx, const void* Bpacked, float* C, long ldc, float* rowss,
                                                int M, int N, int K, const spacer_plan* plan, spacer_stream_t stream) {
    SP_REQUIRE(X32 && Bpacked && C && rowss, SPACER_EINVAL, "gemm_skinny_normed: null operand");
    SP_REQUIRE_PLAN(plan);
    SP_REQUIRE(!plan || plan->skinny_skew <= 17, SPACER_EINVAL, "gemm_skinny_normed: plan.skinny_skew = %d out of range (<= 17)", plan->skinny_skew);
    SP_REQUIRE(M > 0 && M <= 64 && K % 256 == 0 && N % 16 == 0 && ldx % 4 == 0, SPACER_EINVAL,
               "gemm_skinny_normed: need 0 < M <= 64, K %% 256 == 0, N %% 16 == 0, ldx %% 4 == 0 (M=%d N=%d K=%d)", M, N, K);
    const int col_groups = cdiv(N, 64), slices = K / 256;
    const int target_blocks = skinny_target_blocks(plan);
    int ranges = col_groups < target_blocks - target_blocks / 8 ? max(1, min(slices, target_blocks / col_groups)) : 1;
    int spr = cdiv(slices, ranges);
    const int skew = skinny_skew(plan, col_groups, ranges, slices, target_blocks);
    if (skew > 0) spr = -skew;
    else ranges = cdiv(slices, spr);
    hipLaunchKernelGGL((gemm_skinny_kernel<true, false, 1, true>), dim3(col_groups, ranges), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)X32, ldx, (const bf16_t*)Bpacked, 0L, C, ldc, M, N, K, spr, M, 0, 0, 1, (float*)nullptr,
                       (int*)nullptr, rowss);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_gemm_skinny_packed_store_bf16(const void* A, long lda, const void* Bpacked, void* C, long ldc, int M, int N,
                                                    int K, const spacer_plan* plan, spacer_stream_t stream) {
    return launch_skinny(A, lda, Bpacked, 0, C, ldc, M, N, K, nullptr, true, plan, (hipStream_t)stream, true);
}

extern "C" int spacer_pack_weight_frag(const void* W, long ld, void* out, int N, int K, spacer_stream_t stream) {
    SP_REQUIRE(N % 16 == 0 && K % 32 == 0 && ld % 8 == 0, SPACER_EINVAL, "pack_weight_frag: need N %% 16 == 0, K %% 32 == 0");
    const long total = (long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)W, ld, (bf16_t*)out, N, K, 0);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_pack_weight_frag_swiglu(const void* W, long ld, void* out, int inter, int K, spacer_stream_t stream) {
    SP_REQUIRE(inter % 32 == 0 && K % 32 == 0 && ld % 8 == 0, SPACER_EINVAL,
               "pack_weight_frag_swiglu: need inter %% 32 == 0, K %% 32 == 0");
    const long total = (long)(2 * inter / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)W, ld, (bf16_t*)out, 2 * inter, K, inter);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

// workspace of the SwiGLU decode GEMM's tail balance: scratch tiles + tickets, ZERO-initialised once by the caller (the kernel
// leaves it zeroed); launches that share it must be ordered on one stream
constexpr int SWIGLU_MAX_SPLIT_GROUPS = 256;
extern "C" long spacer_gemm_skinny_swiglu_workspace_bytes(void) { return (long)SWIGLU_MAX_SPLIT_GROUPS * (64 * 64 * 4 + 4); }

// normed: A is the fp32 residual stream and the RMSNorm in front of the projection is folded in (<= 16 rows; gemm_skinny_kernel SMALL)
static int launch_skinny_swiglu(const void* A, long lda, const void* Bpacked, void* Y, long ldy, int M, int inter, int K, void* ws,
                                long ws_bytes, const spacer_plan* plan, hipStream_t stream, bool normed = false, float eps = 0.f) {
    SP_REQUIRE(A && Bpacked && Y, SPACER_EINVAL, "gemm_skinny_swiglu: null operand");
    SP_REQUIRE_PLAN(plan);
    SP_REQUIRE(M > 0 && M <= 128, SPACER_EINVAL, "gemm_skinny_swiglu: M=%d must be in 1..128", M);
    SP_REQUIRE(K % 256 == 0 && inter % 32 == 0 && lda % 8 == 0, SPACER_EINVAL,
               "gemm_skinny_swiglu: need K %% 256 == 0, inter %% 32 == 0, lda %% 8 == 0");
    SP_REQUIRE(!normed || M <= 16, SPACER_EINVAL, "gemm_skinny_swiglu_normed: the norm-folded form takes <= 16 rows (M=%d)", M);
    const bool small = M <= 16;
    const int N = 2 * inter, col_groups = cdiv(N, 64);
    // one-round form (round 4): between 4 and 5 column fragments per resident slot -> exactly `slots` workgroups, the first
    // `wide` of them five fragments wide (gemm_skinny_kernel: wide_groups); 65..128 rows: the same with two row blocks per pass
    const int slots1 = 2 * plan_cus(plan), frags = N / 16;
    const bool one_round = N % 16 == 0 && frags > 4 * slots1 && frags < 5 * slots1 && !(plan && (plan->skinny_no_balance || plan->skinny_blocks));
    const int wide = one_round ? frags - 4 * slots1 : 0;
    if (M > 64) {
        hipLaunchKernelGGL((gemm_skinny_kernel<true, true, 2>), dim3(one_round ? slots1 : col_groups, 1), dim3(256), 0, stream,
                           (const bf16_t*)A, lda, (const bf16_t*)Bpacked, 0L, (float*)Y, ldy, M, N, K, K / 128, M, 0, 0, 1, (float*)nullptr,
                           (int*)nullptr, (float*)nullptr, wide);
        SP_CHECK_LAUNCH();
        return SPACER_OK;
    }
    if (one_round || normed) {
        // (the norm-folded form outside the one-round window: one whole-K workgroup per 64 columns, no tail balance -- the small models'
        // shapes, measured at 16 rows of 2B in profiles/r05_decode_small_rows.md; every workgroup must cover all of K to sum x^2 itself, so
        // plan->skinny_blocks / skinny_no_balance do not apply to it.  ADVICE r5: the NON-normed <= 16-row launches outside the window
        // keep the tail-balanced path below, which honours those switches, as before round 5)
        const dim3 grid(one_round ? slots1 : col_groups, 1);
#define SKINNY_SWIGLU_LAUNCH(NRM, SML)                                                                                              \
        hipLaunchKernelGGL((gemm_skinny_kernel<true, true, 1, NRM, SML>), grid, dim3(256), 0, stream, (const bf16_t*)A, lda,       \
                           (const bf16_t*)Bpacked, 0L, (float*)Y, ldy, M, N, K, K / 256, M, 0, 0, 1, (float*)nullptr, (int*)nullptr, \
                           (float*)nullptr, wide, eps)
        if (normed) SKINNY_SWIGLU_LAUNCH(true, true);
        else if (small) SKINNY_SWIGLU_LAUNCH(false, true);
        else SKINNY_SWIGLU_LAUNCH(false, false);
#undef SKINNY_SWIGLU_LAUNCH
        SP_CHECK_LAUNCH();
        return SPACER_OK;
    }
    // tail balance: the column groups beyond the last full round of 512 resident workgroups, when that tail is short
    int split_groups = 0, split_ranges = 1;
    const int slots = 2 * plan_cus(plan), rem = col_groups % slots, slices = K / 256;
    const bool enabled = ws && ws_bytes >= spacer_gemm_skinny_swiglu_workspace_bytes() && !(plan && plan->skinny_no_balance)
                         && !(plan && plan->skinny_blocks) && N % 64 == 0;
    if (enabled && col_groups > slots && rem > 0 && rem <= 192 && rem <= SWIGLU_MAX_SPLIT_GROUPS && slices >= 4) {
        split_groups = rem;
        const int want = min(slices / 2, max(2, slots / rem));                // >= 2 K slices per range, about one round of small blocks
        const int spr = cdiv(slices, want);
        split_ranges = cdiv(slices, spr);                                     // every range non-empty (the kernel derives the same spr)
    }
    const int blocks = split_groups * split_ranges + (col_groups - split_groups);
    float* scratch = (float*)ws;
    int* tickets = ws ? (int*)((char*)ws + (long)SWIGLU_MAX_SPLIT_GROUPS * 64 * 64 * 4) : nullptr;
    hipLaunchKernelGGL((gemm_skinny_kernel<true, true, 1>), dim3(blocks, 1), dim3(256), 0, stream, (const bf16_t*)A, lda,
                       (const bf16_t*)Bpacked, 0L, (float*)Y, ldy, M, N, K, K / 256, M, 0, split_groups,
                       split_ranges, scratch, tickets);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_gemm_skinny_swiglu_bf16(const void* A, long lda, const void* Bpacked, void* Y, long ldy, int M, int inter,
                                              int K, spacer_stream_t stream) {
    return launch_skinny_swiglu(A, lda, Bpacked, Y, ldy, M, inter, K, nullptr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int spacer_gemm_skinny_swiglu_bf16_ws(const void* A, long lda, const void* Bpacked, void* Y, long ldy, int M, int inter,
                                                 int K, void* workspace, long workspace_bytes, const spacer_plan* plan,
                                                 spacer_stream_t stream) {
    return launch_skinny_swiglu(A, lda, Bpacked, Y, ldy, M, inter, K, workspace, workspace_bytes, plan, (hipStream_t)stream);
}

// y[M, inter] = silu(rstd g) * (rstd u), [g | u] = bf16(x32) . Wp^T, rstd = rsqrt(mean(x32^2) + eps): the decode gate|up projection of
// <= 16 rows with the post-attention RMSNorm folded in (Wp = spacer_pack_weight_frag_swiglu of W diag(w_norm)); one launch for HF's
// post_attention_layernorm + gate_proj / up_proj + act_fn of one generate step (TR:463).
extern "C" int spacer_gemm_skinny_swiglu_normed(const float* X32, long ldx, const void* Bpacked, void* Y, long ldy, int M, int inter,
                                                int K, float eps, void* workspace, long workspace_bytes, const spacer_plan* plan,
                                                spacer_stream_t stream) {
    SP_REQUIRE(ldx % 4 == 0 && ((uintptr_t)X32 % 16) == 0, SPACER_EINVAL, "gemm_skinny_swiglu_normed: x32 rows must be 16-byte aligned");
    return launch_skinny_swiglu(X32, ldx, Bpacked, Y, ldy, M, inter, K, workspace, workspace_bytes, plan, (hipStream_t)stream, true, eps);
}

extern "C" int spacer_decode_rope_table(const int* pos_base, const int* step_dev, float theta, float* cos_t, float* sin_t,
                                        int B, int D, spacer_stream_t stream) {
    if (B <= 0) return SPACER_OK;
    hipLaunchKernelGGL(decode_rope_table_kernel, dim3(cdiv(B * (D / 2), 256)), dim3(256), 0, (hipStream_t)stream, pos_base,
                       step_dev, theta, cos_t, sin_t, B, D);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_decode_qkv_finish(float* acc32, const void* bias, const float* cos_t, const float* sin_t, void* q_out,
                                        void* tail_k, void* tail_v, const int* tail_len_dev, int B, int Hq, int Hkv, int D,
                                        int Cmax, spacer_stream_t stream) {
    if (B <= 0) return SPACER_OK;
    const int total = B * (Hq + 2 * Hkv) * (D / 2);
    hipLaunchKernelGGL(decode_qkv_finish_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, acc32,
                       (const bf16_t*)bias, cos_t, sin_t, (bf16_t*)q_out, (bf16_t*)tail_k, (bf16_t*)tail_v, tail_len_dev, B, Hq,
                       Hkv, D, Cmax, (const float*)nullptr, (float*)nullptr, 0, 0.f);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_decode_qkv_finish_normed(float* acc32, const void* bias, const float* cos_t, const float* sin_t, void* q_out,
                                               void* tail_k, void* tail_v, const int* tail_len_dev, const float* rowss,
                                               float* rowss_zero, int norm_cols, float eps, int B, int Hq, int Hkv, int D, int Cmax,
                                               spacer_stream_t stream) {
    SP_REQUIRE(rowss && norm_cols > 0 && B <= 256, SPACER_EINVAL, "decode_qkv_finish_normed: row sums missing or B=%d > 256", B);
    if (B <= 0) return SPACER_OK;
    const int total = B * (Hq + 2 * Hkv) * (D / 2);
    hipLaunchKernelGGL(decode_qkv_finish_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, acc32,
                       (const bf16_t*)bias, cos_t, sin_t, (bf16_t*)q_out, (bf16_t*)tail_k, (bf16_t*)tail_v, tail_len_dev, B, Hq,
                       Hkv, D, Cmax, rowss, rowss_zero, norm_cols, eps);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_swiglu_f32_fwd(float* acc32, void* y, int B, int inter, spacer_stream_t stream) {
    if (B <= 0) return SPACER_OK;
    hipLaunchKernelGGL(swiglu_f32_kernel, dim3(min(cdiv((long)B * inter, 256), 4096)), dim3(256), 0, (hipStream_t)stream, acc32,
                       (bf16_t*)y, B, inter);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

static int launch_attn_decode(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                              const int* prompt_of, const void* tail_k, const void* tail_v, const int* tail_len_dev,
                              void* o, float* pre_ws, int Kn, int B, int Pmax, int Cmax, int Hq, int Hkv, int D, float scale,
                              spacer_stream_t stream, const int* row0 = nullptr, int n_prompts = 0) {
    SP_REQUIRE(D == 128, SPACER_EINVAL, "attn_decode: head_dim %d unsupported (128)", D);
    SP_REQUIRE(Hkv > 0 && Hq % Hkv == 0, SPACER_EINVAL, "attn_decode: bad head counts");
    if (B <= 0) return SPACER_OK;
    const int rep = Hq / Hkv;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(R)                                                                                                      \
    {                                                                                                                  \
        static const int once = hipFuncSetAttribute((const void*)attn_decode_mfma_kernel<R, 8>,                        \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 8 * AT_T_BYTES(128));  \
        (void)once;                                                                                                    \
        hipLaunchKernelGGL((attn_decode_mfma_kernel<R, 8>), dim3(B, Hkv), dim3(512), 8 * AT_T_BYTES(128), s, (const bf16_t*)q, \
                           (const bf16_t*)prefix_k, (const bf16_t*)prefix_v, prefix_len, prompt_of, (const bf16_t*)tail_k, \
                           (const bf16_t*)tail_v, tail_len_dev, (bf16_t*)o, Pmax, Cmax, Hq, Hkv, scale, (float*)nullptr, Kn); \
    }
    if (pre_ws) {
        SP_REQUIRE(Kn > 0 && Kn * rep <= 64 && (row0 || B % Kn == 0), SPACER_EINVAL, "attn_decode_shared: Kn*rep=%d must be <= 64", Kn * rep);
        SP_REQUIRE(!row0 || (n_prompts > 0 && prompt_of), SPACER_EINVAL, "attn_decode_shared_rows: n_prompts and prompt_of are required");
        const int np = row0 ? n_prompts : B / Kn;
        const int nA = np * Hkv * PRE_SPLITS;
        float* tailp = pre_ws + (long)np * Hkv * PRE_SPLITS * 64 * (128 + 2);
#define LAUNCH_SPLIT(R)                                                                                                 \
        {                                                                                                               \
            static const int once = hipFuncSetAttribute((const void*)attn_decode_split_kernel<R>,                       \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 4 * AT_RM_BYTES);   \
            (void)once;                                                                                                 \
            hipLaunchKernelGGL((attn_decode_split_kernel<R>), dim3(nA + B * Hkv), dim3(256), 4 * AT_RM_BYTES, s, (const bf16_t*)q, \
                               (const bf16_t*)prefix_k, (const bf16_t*)prefix_v, prefix_len, (const bf16_t*)tail_k,       \
                               (const bf16_t*)tail_v, tail_len_dev, pre_ws, tailp, nA, Kn, Pmax, Cmax, Hq, Hkv, scale, row0); \
            hipLaunchKernelGGL((attn_decode_merge_kernel<R>), dim3(B, Hkv), dim3(R * 128), 0, s, (const float*)pre_ws,   \
                               (const float*)tailp, (bf16_t*)o, Kn, Hq, Hkv, prompt_of, row0);                          \
        }
        switch (rep) {
            case 1: LAUNCH_SPLIT(1); break; case 2: LAUNCH_SPLIT(2); break; case 3: LAUNCH_SPLIT(3); break; case 4: LAUNCH_SPLIT(4); break;
            case 5: LAUNCH_SPLIT(5); break; case 6: LAUNCH_SPLIT(6); break; case 7: LAUNCH_SPLIT(7); break; case 8: LAUNCH_SPLIT(8); break;
            default: SP_REQUIRE(false, SPACER_EINVAL, "attn_decode: GQA ratio %d not instantiated", rep);
        }
#undef LAUNCH_SPLIT
        SP_CHECK_LAUNCH();
        return SPACER_OK;
    }
    switch (rep) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        case 5: LAUNCH(5); break;
        case 6: LAUNCH(6); break;
        case 7: LAUNCH(7); break;
        case 8: LAUNCH(8); break;
        default: SP_REQUIRE(false, SPACER_EINVAL, "attn_decode: GQA ratio %d not instantiated", rep);
    }
#undef LAUNCH
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_attn_decode(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                                  const int* prompt_of, const void* tail_k, const void* tail_v, const int* tail_len_dev,
                                  void* o, int B, int Pmax, int Cmax, int Hq, int Hkv, int D, float scale,
                                  spacer_stream_t stream) {
    return launch_attn_decode(q, prefix_k, prefix_v, prefix_len, prompt_of, tail_k, tail_v, tail_len_dev, o, nullptr, 0, B, Pmax,
                              Cmax, Hq, Hkv, D, scale, stream);
}

extern "C" long spacer_attn_decode_workspace_bytes(int n_prompts, int Hkv) {
    // PRE_SPLITS prompt partials of 64 columns + one tail partial per (sequence, head) (Kn * rep <= 64 of them per prompt)
    return (long)n_prompts * Hkv * (PRE_SPLITS + 1) * 64 * (128 + 2) * (long)sizeof(float);
}

extern "C" int spacer_attn_decode_shared(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                                         const int* prompt_of, const void* tail_k, const void* tail_v,
                                         const int* tail_len_dev, void* o, void* workspace, int B, int Kn, int Pmax, int Cmax,
                                         int Hq, int Hkv, int D, float scale, spacer_stream_t stream) {
    SP_REQUIRE(workspace != nullptr, SPACER_EINVAL, "attn_decode_shared: workspace required");
    return launch_attn_decode(q, prefix_k, prefix_v, prefix_len, prompt_of, tail_k, tail_v, tail_len_dev, o, (float*)workspace, Kn,
                              B, Pmax, Cmax, Hq, Hkv, D, scale, stream);
}

// Per-prompt rollout counts (round 6): prompt p owns the decode rows [row0[p], row0[p + 1]) (row0 int32 [n_prompts + 1], device), at most
// Kmax of them (Kmax * Hq / Hkv <= 64).  The T-GRPO twin of a sample generates G / 2 rollouts (TR:473): a step's main + twin rollouts decode
// as 8 x 8 + 8 x 4 = 96 rows instead of 128.  workspace: spacer_attn_decode_workspace_bytes(n_prompts, Hkv).
extern "C" int spacer_attn_decode_shared_rows(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                                              const int* prompt_of, const int* row0, const void* tail_k, const void* tail_v,
                                              const int* tail_len_dev, void* o, void* workspace, int B, int n_prompts, int Kmax, int Pmax,
                                              int Cmax, int Hq, int Hkv, int D, float scale, spacer_stream_t stream) {
    SP_REQUIRE(workspace != nullptr && row0 != nullptr, SPACER_EINVAL, "attn_decode_shared_rows: workspace and row0 required");
    return launch_attn_decode(q, prefix_k, prefix_v, prefix_len, prompt_of, tail_k, tail_v, tail_len_dev, o, (float*)workspace, Kmax,
                              B, Pmax, Cmax, Hq, Hkv, D, scale, stream, row0, n_prompts);
}

