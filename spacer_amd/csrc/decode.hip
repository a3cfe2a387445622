// Decode-loop kernels (the HBM-bound half of HF generate, SG_RLVR_trainer.py:463): weight-streaming skinny
// GEMM with fp32 split-K accumulation, the q/k/v finishing step (bias + rotary + KV-cache append), SwiGLU
// from fp32 accumulators, and single-query attention over [shared prompt KV | per-rollout tail KV].
// All step-dependent scalars (tail length, step index) are read from device memory so one decode step can
// be captured once in a hipGraph and replayed.
#include "common.h"

namespace {

// =============================================================================== skinny GEMM (M <= 64)
// C32[M,N] += A[M,K] . B[N,K]^T.  Workgroup = 64 columns (wave = 16) x one K slice of KS; the A slice
// (<= 64 x KS bf16) is staged once in LDS (swizzled) and shared by the 4 waves; B (the weights) is streamed
// HBM -> VGPR with 16-byte fragment loads, 8 k-steps in flight per lane; fp32 atomics reduce across K slices.
template <int KS>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const bf16_t* __restrict__ A, long lda,
                                                             const bf16_t* __restrict__ B, long ldb,
                                                             float* __restrict__ C, long ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [64][KS] bf16, chunk ^= row & 15
    constexpr int ROWB = KS * 2, CHUNKS = KS / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 64 + wave * 16, k0 = blockIdx.y * KS;
    const int l15 = lane & 15, g = lane >> 4;

    // stage A slice (zero rows >= M)
    for (int idx = tid; idx < 64 * CHUNKS; idx += 256) {
        const int row = idx / CHUNKS, ch = idx % CHUNKS;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < M) v = *(const uint4*)(A + (long)row * lda + k0 + ch * 8);
        *(uint4*)(smem + row * ROWB + ((ch ^ (row & 15)) * 16)) = v;
    }
    const int nrow = min(n0 + l15, N - 1);
    const bf16_t* bp = B + (long)nrow * ldb + k0 + g * 8;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    constexpr int STEPS = KS / 32, UN = 8;
#pragma unroll 1
    for (int s0 = 0; s0 < STEPS; s0 += UN) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        u32x4 w[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) w[u] = __builtin_nontemporal_load((const u32x4*)(bp + (s0 + u) * 32));
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int ch = (s0 + u) * 4 + g;
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int row = mf * 16 + l15;
                const bf16x8 af = *(const bf16x8*)(smem + row * ROWB + ((ch ^ (row & 15)) * 16));
                acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf, acc[mf], 0, 0, 0);   // D[m][n]
            }
        }
    }
    // lane holds C[m = mf*16 + g*4 + r][n = n0 + l15]
    if (n0 + l15 < N) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mf * 16 + g * 4 + r;
                if (m < M) atomicAdd(C + (long)m * ldc + n0 + l15, acc[mf][r]);
            }
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
__global__ __launch_bounds__(256) void decode_qkv_finish_kernel(float* __restrict__ acc, const bf16_t* __restrict__ bias,
                                                                const float* __restrict__ cs, const float* __restrict__ sn,
                                                                bf16_t* __restrict__ q_out, bf16_t* __restrict__ tail_k,
                                                                bf16_t* __restrict__ tail_v, const int* __restrict__ tail_len,
                                                                int B, int Hq, int Hkv, int D, int Cmax) {
    const int half = D / 2, heads = Hq + 2 * Hkv;
    const int total = B * heads * half;
    const int pos = *tail_len;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i % half, hh = (i / half) % heads, b = i / (half * heads);
        float* a = acc + ((long)b * heads + hh) * D;
        float x1 = a[j] + (bias ? bf2f(bias[hh * D + j]) : 0.f);
        float x2 = a[j + half] + (bias ? bf2f(bias[hh * D + j + half]) : 0.f);
        a[j] = 0.f; a[j + half] = 0.f;
        if (hh < Hq + Hkv) {   // rotary on q and k heads
            const float c1 = cs[b * D + j], s1 = sn[b * D + j], c2 = cs[b * D + j + half], s2 = sn[b * D + j + half];
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

// =============================================================================== decode attention
// Workgroup = (sequence b, kv head).  16 lanes x 8 dims cover one key; a wave scores 4 keys per step, the
// 4 waves stride over keys; all `REP` q heads of the GQA group are scored against each loaded key/value.
template <int D, int REP>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ pk,
                                                          const bf16_t* __restrict__ pv, const int* __restrict__ plen,
                                                          const int* __restrict__ prompt_of, const bf16_t* __restrict__ tk,
                                                          const bf16_t* __restrict__ tv, const int* __restrict__ tail_len,
                                                          bf16_t* __restrict__ o, int Pmax, int Cmax, int Hq, int Hkv,
                                                          float scale) {
    static_assert(D == 128, "decode attention is written for head_dim 128");
    __shared__ float red_m[4][REP], red_l[4][REP];
    __shared__ float red_o[4][REP][D];
    const int b = blockIdx.x, hk = blockIdx.y;
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;   // 16 key groups
    const int pr = prompt_of[b], P = plen[pr], Tl = *tail_len + 1;    // tail includes the token appended this step
    float qv[REP][8];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        const uint4 t = *(const uint4*)(q + ((long)b * Hq + hk * REP + r) * D + sub * 8);
        qv[r][0] = bf_lo(t.x) * scale; qv[r][1] = bf_hi(t.x) * scale; qv[r][2] = bf_lo(t.y) * scale; qv[r][3] = bf_hi(t.y) * scale;
        qv[r][4] = bf_lo(t.z) * scale; qv[r][5] = bf_hi(t.z) * scale; qv[r][6] = bf_lo(t.w) * scale; qv[r][7] = bf_hi(t.w) * scale;
    }
    float m[REP], l[REP], acc[REP][8];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        m[r] = -INFINITY; l[r] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
    }
    const int total = P + Tl;
    for (int key = grp; key < total; key += 16) {
        const bf16_t *kp, *vp;
        if (key < P) {
            const long off = (((long)pr * Pmax + key) * Hkv + hk) * D + sub * 8;
            kp = pk + off; vp = pv + off;
        } else {
            const long off = (((long)b * Cmax + (key - P)) * Hkv + hk) * D + sub * 8;
            kp = tk + off; vp = tv + off;
        }
        const uint4 kk = *(const uint4*)kp, vv = *(const uint4*)vp;
        const float kf[8] = {bf_lo(kk.x), bf_hi(kk.x), bf_lo(kk.y), bf_hi(kk.y), bf_lo(kk.z), bf_hi(kk.z), bf_lo(kk.w), bf_hi(kk.w)};
        const float vf[8] = {bf_lo(vv.x), bf_hi(vv.x), bf_lo(vv.y), bf_hi(vv.y), bf_lo(vv.z), bf_hi(vv.z), bf_lo(vv.w), bf_hi(vv.w)};
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qv[r][e] * kf[e];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            const float mn = fmaxf(m[r], s);
            const float al = __expf(m[r] - mn), p = __expf(s - mn);
            l[r] = l[r] * al + p; m[r] = mn;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = acc[r][e] * al + p * vf[e];
        }
    }
    // combine: first the 4 key groups inside each wave (lanes sub + 16*j) with shuffles, then the 4 waves via LDS
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        float M = fmaxf(m[r], __shfl_xor(m[r], 16, 64));
        M = fmaxf(M, __shfl_xor(M, 32, 64));
        const float w = (m[r] == -INFINITY) ? 0.f : __expf(m[r] - M);
        float L = l[r] * w;
        L += __shfl_xor(L, 16, 64); L += __shfl_xor(L, 32, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = acc[r][e] * w;
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            if (lane < 16) red_o[wave][r][sub * 8 + e] = a;
        }
        if (lane == 0) { red_m[wave][r] = M; red_l[wave][r] = L; }
    }
    __syncthreads();
    for (int i = tid; i < REP * D; i += 256) {
        const int r = i / D, d = i % D;
        float M = -INFINITY;
        for (int gq = 0; gq < 4; ++gq) M = fmaxf(M, red_m[gq][r]);
        float L = 0.f, O = 0.f;
        for (int gq = 0; gq < 4; ++gq) {
            const float w = (red_m[gq][r] == -INFINITY) ? 0.f : __expf(red_m[gq][r] - M);
            L += red_l[gq][r] * w; O += red_o[gq][r][d] * w;
        }
        o[((long)b * Hq + hk * REP + r) * D + d] = f2bf(O / L);
    }
}

}  // namespace

extern "C" int spacer_gemm_skinny_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                                       int K, const spacer_gemm_epilogue* epi, spacer_stream_t stream) {
    SP_REQUIRE(A && B && C, SPACER_EINVAL, "gemm_skinny: null operand");
    SP_REQUIRE(M > 0 && M <= 64, SPACER_EINVAL, "gemm_skinny: M=%d must be in 1..64", M);
    SP_REQUIRE(K % 256 == 0, SPACER_EINVAL, "gemm_skinny: K=%d must be a multiple of 256", K);
    SP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, SPACER_EINVAL, "gemm_skinny: lda/ldb must be multiples of 8");
    SP_REQUIRE(!epi || (epi->out_f32 && !epi->bias && epi->act == 0 && (!epi->residual || epi->residual == C)),
               SPACER_EINVAL, "gemm_skinny: only fp32 accumulate-into-C is supported (C32 += A.B^T)");
    hipStream_t s = (hipStream_t)stream;
    if (K % 512 == 0 && (long)cdiv(N, 64) * (K / 512) >= 384) {
        static const int once = hipFuncSetAttribute((const void*)gemm_skinny_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 512 * 2);
        (void)once;
        hipLaunchKernelGGL(gemm_skinny_kernel<512>, dim3(cdiv(N, 64), K / 512), dim3(256), 64 * 512 * 2, s, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K);
    } else {
        hipLaunchKernelGGL(gemm_skinny_kernel<256>, dim3(cdiv(N, 64), K / 256), dim3(256), 64 * 256 * 2, s, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K);
    }
    SP_CHECK_LAUNCH();
    return SPACER_OK;
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
                       Hkv, D, Cmax);
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

extern "C" int spacer_attn_decode(const void* q, const void* prefix_k, const void* prefix_v, const int* prefix_len,
                                  const int* prompt_of, const void* tail_k, const void* tail_v, const int* tail_len_dev,
                                  void* o, int B, int Pmax, int Cmax, int Hq, int Hkv, int D, float scale,
                                  spacer_stream_t stream) {
    SP_REQUIRE(D == 128, SPACER_EINVAL, "attn_decode: head_dim %d unsupported (128)", D);
    SP_REQUIRE(Hkv > 0 && Hq % Hkv == 0, SPACER_EINVAL, "attn_decode: bad head counts");
    if (B <= 0) return SPACER_OK;
    const int rep = Hq / Hkv;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(R)                                                                                                      \
    hipLaunchKernelGGL((attn_decode_kernel<128, R>), dim3(B, Hkv), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)prefix_k, \
                       (const bf16_t*)prefix_v, prefix_len, prompt_of, (const bf16_t*)tail_k, (const bf16_t*)tail_v,      \
                       tail_len_dev, (bf16_t*)o, Pmax, Cmax, Hq, Hkv, scale)
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
