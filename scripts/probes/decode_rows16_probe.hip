// ARCHIVED PROBE (round 5) -- not part of libspacer_hip.so.  This file was spacer_amd/csrc/decode_rows16.hip in commits 1760bd4 .. f5c47b4^
// (C-ABI entries spacer_gemm_rows16_packed_bf16 / spacer_decode_qkv_rows16 / spacer_pack_weight_frag_rope, wired into
// RolloutEngine._decode_step behind SPACER_DECODE_SMALL=qkv,o,down) and was measured there: slower than the K-split kernels in every
// position (profiles/r05_decode_small_rows.md).  To rebuild it, copy it back into spacer_amd/csrc/ (it includes "common.h").  Known
// hazard left unfixed: the ring's trailing inline-asm loads land in registers the compiler considers dead (the q|k|v form was not
// bit-reproducible between runs).
//
// Decode GEMMs for SMALL row counts (M <= 16: one prompt group of K = 8 rollouts per GPU is the reference script's own launch shape,
// run_SpaceR_SG_RLVR.sh:21,39 = BASELINE configs[3]; cfg2's 4 groups x K = 4 are 16 rows).  Round 5.
//
// The 64-row kernels of decode.hip are tuned for the 64-row batch: column groups of 64 x K RANGES, so q|k|v / o / down split K
// across workgroups and meet through fp32 atomics (2 M atomics per launch, 3-4.6 us of flush), q|k|v needs a zero-filled fp32
// accumulator and a separate finishing kernel that reads the atomically produced sums back (bias, rotary, KV append), and every
// workgroup re-stages A through LDS.  At <= 16 rows none of that is needed:
//   * a workgroup owns ONE 16-column weight fragment over ALL of K: its four waves take the k-steps u = 4 j + w, so every A
//     element is used by exactly one wave of the workgroup exactly once -- A fragments go global -> registers (row l15, 16 bytes per
//     lane), no LDS staging, no barrier in the K loop;
//   * the whole weight share of a wave is a handful of 1 KiB non-temporal loads (K = 3584: 28 per wave); a ring of DEPTH loads stays
//     in flight from the first instruction on, so the launch is a pure stream after one latency;
//   * the four K-interleaved partial sums meet in LDS, wave 0 runs the epilogue -- no atomics, no zero fill, bit-reproducible;
//   * epilogues: ACC (C32 += sums: o / down projections into the fp32 residual stream), STORE, and QKV: the layer's input RMSNorm
//     folded in (A = the fp32 stream, row sums of x^2 from the SAME loads, W diag(w_ln) in the packed weights), + bias, rotary,
//     bf16 q rows and the KV-cache append in one go -- the packed q|k|v fragments hold dims [8 j .. 8 j + 7] and [64 + 8 j .. 64 + 8 j + 7]
//     of ONE head (spacer_pack_weight_frag_rope), so a rotary pair sits in lanes l and l ^ 8 of the same wave.
// One launch replaces {norm-folded K-split GEMM + finishing kernel} (q|k|v) and the atomic K-split launches of o / down.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int r16_u32x4;

enum { R16_ACC = 0, R16_STORE = 1, R16_QKV = 2 };

struct Rows16Args {
    const void* A; long lda;               // bf16 [M, K], or fp32 [M, K] (NORMA)
    const bf16_t* Bp;                      // fragment-major weights [N/16][K/32][64 lanes][8]
    float* C; long ldc;                    // ACC / STORE: fp32 [M, N]
    int M, N, K;
    // QKV epilogue
    const bf16_t* bias;                    // [N] in the ORIGINAL column order (head-major), or null
    const float* cos_t; const float* sin_t;   // [M, D]
    bf16_t* q_out; bf16_t* tail_k; bf16_t* tail_v;
    const int* tail_len;
    int Hq, Hkv, D, Cmax;
    float eps;
};

template <int EPI, bool NORMA, int DEPTH, int NW>
__global__ __launch_bounds__(NW * 64, 2) void gemm_rows16_kernel(Rows16Args a) {
    __shared__ float4 part[NW][64];
    __shared__ float rowss[NW][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nb = blockIdx.x;                                  // fragment column
    const int nsteps = a.K >> 5;
    const int S = (nsteps - wave + NW - 1) / NW;                // k-steps of this wave: u = wave + NW j, j < S (wave-uniform)
    // rows >= M of the MFMA tile compute garbage that is never stored (an output row depends on its own A row only): the row index is
    // clamped for address validity, nothing is zeroed
    const int arow = l15 < a.M ? l15 : a.M - 1;
    // addresses = wave-uniform 64-bit base (SGPR pair, advanced per k-step by scalar adds) + a fixed 32-bit lane offset (one VGPR each
    // for the weights and for A): the ring's loads cost no address registers
    const bf16_t* wbase = a.Bp + ((long)nb * nsteps) * 512;                         // + u * 512 elements per k-step
    const unsigned woff = lane * 16;
    const char* abase = (const char*)a.A;                                            // + u * 32 elements per k-step
    const unsigned aoff = (unsigned)(((long)arow * a.lda + g * 8) * (NORMA ? 4 : 2));
    // The K loop is BRANCH-FREE: a ring of DEPTH (weight fragment, A fragment) loads per wave stays in flight from the first
    // instruction on; past the wave's last k-step the ring slots are refilled from the first KiB of the weights / the first k-step of A
    // (always the same cached lines) and their products are masked out, so the compiler's vmcnt bookkeeping never meets a control-flow merge
    // (with `if (u < nsteps)` around the loads it drained the queue -- vmcnt(0) -- every ring round: 40 us for the 7B down projection
    // against 25 us for the K-split kernel it was meant to beat).
    // Loads and waits are inline asm with COUNTED vmcnt: through the builtins hipcc's waitcnt pass gives up at the loop back-edge and
    // puts one vmcnt(0) at the top of every ring round (the queue drains DEPTH steps at a time).  VMEM returns are in order, so after
    // vmcnt(LPS (DEPTH - 1)) the oldest slot's LPS loads have landed; the wait takes the slot's registers as in/out operands so
    // that the MFMA cannot be scheduled above it.
    constexpr int LPS = NORMA ? 3 : 2;                          // loads per k-step
    r16_u32x4 w[DEPTH];
    r16_u32x4 ab[NORMA ? 1 : DEPTH];
    r16_u32x4 af[NORMA ? DEPTH : 1][2];
    auto issue = [&](int i, int j) {                            // k-step j of this wave into ring slot i
        const bool ok = j < S;
        const long u = wave + NW * (long)j;
        const bf16_t* wp = ok ? wbase + u * 512 : a.Bp;           // scalar selects
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(w[i]) : "v"(woff), "s"(wp) : "memory");
        if (NORMA) {
            const char* p = ok ? abase + u * 32 * 4 : abase;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(af[NORMA ? i : 0][0]) : "v"(aoff), "s"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(af[NORMA ? i : 0][1]) : "v"(aoff), "s"(p) : "memory");
        } else {
            const char* p = ok ? abase + u * 32 * 2 : abase;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ab[NORMA ? 0 : i]) : "v"(aoff), "s"(p) : "memory");
        }
    };
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i, i);
    for (int j0 = 0; j0 < S; j0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const bool ok = j0 + i < S;                          // wave-uniform: a scalar select, not a branch
            if (NORMA) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[i]), "+v"(af[NORMA ? i : 0][0]), "+v"(af[NORMA ? i : 0][1]) : "n"(LPS * (DEPTH - 1)));
            else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[i]), "+v"(ab[NORMA ? 0 : i]) : "n"(LPS * (DEPTH - 1)));
            bf16x8 av;
            if (NORMA) {
                const float4 lo = __builtin_bit_cast(float4, af[NORMA ? i : 0][0]), hi = __builtin_bit_cast(float4, af[NORMA ? i : 0][1]);
                const float q = lo.x * lo.x + lo.y * lo.y + lo.z * lo.z + lo.w * lo.w + hi.x * hi.x + hi.y * hi.y + hi.z * hi.z + hi.w * hi.w;
                ss += ok ? q : 0.f;
                av = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(lo.x, lo.y), pack_bf2(lo.z, lo.w), pack_bf2(hi.x, hi.y), pack_bf2(hi.z, hi.w)));
            } else {
                av = __builtin_bit_cast(bf16x8, ab[NORMA ? 0 : i]);
            }
            const r16_u32x4 wz = ok ? w[i] : (r16_u32x4){0u, 0u, 0u, 0u};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wz), acc, 0, 0, 0);   // D[m = g*4 + r][n = l15]
            __builtin_amdgcn_sched_barrier(0);                   // the slot's registers are read before its refill is issued
            issue(i, j0 + i + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the NW K-interleaved partial sums (and row sums of x^2) meet in LDS; wave 0 finishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the ring's trailing dummy loads)
    part[wave][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (NORMA) {
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) rowss[wave][l15] = ss;
    }
    __syncthreads();
    if (wave != 0) return;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) {
        const float4 p = part[w_][lane];
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    }
    const int n = nb * 16 + l15;
    if (EPI == R16_ACC || EPI == R16_STORE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = g * 4 + r;
            if (m < a.M) {
                float* c = a.C + (long)m * a.ldc + n;
                *c = EPI == R16_ACC ? *c + v[r] : v[r];
            }
        }
        return;
    }
    // ---- q|k|v: rstd (norm fold) -> + bias -> rotary on q / k heads -> bf16 q rows, KV-cache append at *tail_len
    const int D = a.D, half = D >> 1, fph = D >> 4;                  // fragments per head
    const int head = nb / fph, j = nb % fph;
    const int d = (l15 < 8 ? 0 : half) + j * 8 + (l15 & 7);           // this lane's dim inside the head
    const float bv = a.bias ? bf2f(a.bias[head * D + d]) : 0.f;
    const int pos = *a.tail_len;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = g * 4 + r;
        float x = v[r];
        if (NORMA) {
            float s2 = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < NW; ++w_) s2 += rowss[w_][m];
            x *= rsqrtf(s2 / (float)a.K + a.eps);
        }
        x += bv;
        const float other = __shfl_xor(x, 8, 64);                    // the rotary partner (dim d +- D/2), every lane takes part
        if (m >= a.M) continue;
        if (head < a.Hq + a.Hkv) {
            const float c = a.cos_t[m * D + d], s = a.sin_t[m * D + d];
            x = l15 < 8 ? x * c - other * s : x * c + other * s;
        }
        bf16_t* dst;
        if (head < a.Hq) dst = a.q_out + ((long)m * a.Hq + head) * D;
        else if (head < a.Hq + a.Hkv) dst = a.tail_k + (((long)m * a.Cmax + pos) * a.Hkv + (head - a.Hq)) * D;
        else dst = a.tail_v + (((long)m * a.Cmax + pos) * a.Hkv + (head - a.Hq - a.Hkv)) * D;
        dst[d] = f2bf(x);
    }
}

// W [N, K] row-major (N = heads * D) -> fragment-major copy whose 16-column fragments pair the rotary halves of one head:
// fragment nb = (head, j): columns p < 8 -> row head*D + 8 j + p, p >= 8 -> row head*D + D/2 + 8 j + (p - 8).
// scale (bf16 [K], may be null): W diag(scale) is folded in (the decode path's input RMSNorm weight), rounded once to bf16.
__global__ __launch_bounds__(256) void pack_frag_rope_kernel(const bf16_t* __restrict__ W, long ld, const bf16_t* __restrict__ scale,
                                                             bf16_t* __restrict__ out, int N, int K, int D) {
    const long total = (long)(N >> 4) * (K >> 5) * 64;
    const int fph = D >> 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long frag = i >> 6;
        const int kb = (int)(frag % (K >> 5));
        const long nb = frag / (K >> 5);
        const int p = lane & 15;
        const long row = (nb / fph) * D + (p < 8 ? 0 : D / 2) + (nb % fph) * 8 + (p & 7);
        const int k0 = kb * 32 + (lane >> 4) * 8;
        uint4 v = *(const uint4*)(W + row * ld + k0);
        if (scale) {
            const uint4 sc = *(const uint4*)(scale + k0);
            v.x = pack_bf2(bf_lo(v.x) * bf_lo(sc.x), bf_hi(v.x) * bf_hi(sc.x));
            v.y = pack_bf2(bf_lo(v.y) * bf_lo(sc.y), bf_hi(v.y) * bf_hi(sc.y));
            v.z = pack_bf2(bf_lo(v.z) * bf_lo(sc.z), bf_hi(v.z) * bf_hi(sc.z));
            v.w = pack_bf2(bf_lo(v.w) * bf_lo(sc.w), bf_hi(v.w) * bf_hi(sc.w));
        }
        *(uint4*)(out + i * 8) = v;
    }
}

constexpr int R16_DEPTH = 12, R16_NW = 8;        // bf16 A: 8 waves x 12 KiB of weights in flight per workgroup (120 VGPRs)
constexpr int R16_DEPTH_N = 8, R16_NW_N = 4;    // fp32 A (norm fold): 12 registers per ring slot -> 4 waves x 14

}  // namespace

extern "C" int spacer_pack_weight_frag_rope(const void* W, long ld, const void* scale, void* out, int N, int K, int head_dim,
                                            spacer_stream_t stream) {
    SP_REQUIRE(W && out, SPACER_EINVAL, "pack_weight_frag_rope: null operand");
    SP_REQUIRE(head_dim >= 16 && head_dim % 16 == 0 && N % head_dim == 0 && K % 32 == 0 && ld % 8 == 0, SPACER_EINVAL,
               "pack_weight_frag_rope: need head_dim %% 16 == 0, N %% head_dim == 0, K %% 32 == 0 (N=%d K=%d D=%d)", N, K, head_dim);
    const long total = (long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_frag_rope_kernel, dim3((int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)W, ld, (const bf16_t*)scale, (bf16_t*)out, N, K, head_dim);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

// C32[M, N] (+)= A[M, K] . Wp^T for M <= 16 rows, one whole-K workgroup per 16-column fragment (no atomics).  store != 0: C = A . Wp^T.
extern "C" int spacer_gemm_rows16_packed_bf16(const void* A, long lda, const void* Bpacked, float* C, long ldc, int M, int N, int K,
                                              int store, spacer_stream_t stream) {
    SP_REQUIRE(A && Bpacked && C, SPACER_EINVAL, "gemm_rows16: null operand");
    SP_REQUIRE(M > 0 && M <= 16 && N % 16 == 0 && N > 0 && K % 32 == 0 && K > 0 && lda % 8 == 0 && ((uintptr_t)A % 16) == 0, SPACER_EINVAL,
               "gemm_rows16: need 0 < M <= 16, N %% 16 == 0, K %% 32 == 0, lda %% 8 == 0 (M=%d N=%d K=%d)", M, N, K);
    Rows16Args a = {};
    a.A = A; a.lda = lda; a.Bp = (const bf16_t*)Bpacked; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    if (store) hipLaunchKernelGGL((gemm_rows16_kernel<R16_STORE, false, R16_DEPTH, R16_NW>), dim3(N / 16), dim3(R16_NW * 64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_rows16_kernel<R16_ACC, false, R16_DEPTH, R16_NW>), dim3(N / 16), dim3(R16_NW * 64), 0, (hipStream_t)stream, a);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

// The decode q|k|v projection of one layer for M <= 16 rows in ONE launch (HF: input_layernorm + q/k/v_proj + rotary + cache update of
// one generate step, TR:463): x32 [M, K] = the fp32 residual stream; Wp = spacer_pack_weight_frag_rope(W, scale = the norm weight).
extern "C" int spacer_decode_qkv_rows16(const float* x32, long ldx, const void* Wp_rope, const void* bias, const float* cos_t,
                                        const float* sin_t, void* q_out, void* tail_k, void* tail_v, const int* tail_len_dev, int M,
                                        int K, float eps, int Hq, int Hkv, int D, int Cmax, spacer_stream_t stream) {
    SP_REQUIRE(x32 && Wp_rope && cos_t && sin_t && q_out && tail_k && tail_v && tail_len_dev, SPACER_EINVAL, "decode_qkv_rows16: null operand");
    SP_REQUIRE(M > 0 && M <= 16 && K % 32 == 0 && ldx % 4 == 0 && ((uintptr_t)x32 % 16) == 0 && D % 16 == 0 && D >= 16 && Hq > 0 && Hkv > 0, SPACER_EINVAL,
               "decode_qkv_rows16: need 0 < M <= 16, K %% 32 == 0, head_dim %% 16 == 0 (M=%d K=%d D=%d)", M, K, D);
    Rows16Args a = {};
    a.A = x32; a.lda = ldx; a.Bp = (const bf16_t*)Wp_rope; a.M = M; a.N = (Hq + 2 * Hkv) * D; a.K = K;
    a.bias = (const bf16_t*)bias; a.cos_t = cos_t; a.sin_t = sin_t; a.q_out = (bf16_t*)q_out; a.tail_k = (bf16_t*)tail_k;
    a.tail_v = (bf16_t*)tail_v; a.tail_len = tail_len_dev; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.Cmax = Cmax; a.eps = eps;
    hipLaunchKernelGGL((gemm_rows16_kernel<R16_QKV, true, R16_DEPTH_N, R16_NW_N>), dim3(a.N / 16), dim3(R16_NW_N * 64), 0, (hipStream_t)stream, a);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
