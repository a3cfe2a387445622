// Precise scoring mode: the kernels that let `Qwen2VLEngine.score_groups(precise=True)` meet the north-star's 1e-3 log-prob
// tolerance at full depth (SG_RLVR_trainer.py:353-366 is the pinned quantity; DESIGN.md section 4 has the error budget).
//
// Every tensor that the fast path rounds to bf16 between operators is carried here as a PAIR of bf16 arrays (hi, lo) with
// hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits instead of 8, i.e. a relative error of 2^-17 per operand.  Weights are
// exactly bf16 on both sides, so a linear layer is TWO accumulate passes of the production GEMM (A_hi . W^T, then += A_lo . W^T)
// into an fp32 output; attention needs both operands of both products split: S = Qh Kh + Qh Kl + Ql Kh, O = Ph Vh + Ph Vl + Pl Vh
// (the lo x lo terms are 2^-16 of the result and dropped).  Everything between the matmuls (norms, rotary, activations, softmax)
// runs in fp32 and writes a pair.  The forward can also EMIT THE TAPE of the fast path's backward (round 4: norm statistics, the
// attention log-sum-exp, bf16 pre-activations; the hi halves of the pairs are the bf16 activations): the training step then takes its
// log-probs, KL and loss from this mode and its gradient from the production backward kernels on those activations.
#include "attn_common.h"

namespace {

constexpr int NT = 256;
inline int grid_for(long work, int cap = 256 * 16) { long b = (work + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

// (a, b) fp32 -> packed bf16 pairs hi and lo; a - hi is exact in fp32 (hi shares a's leading bits)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf2(a, b);
    lo = pack_bf2(a - bf_lo(hi), b - bf_hi(hi));
}
__device__ __forceinline__ void store_pair4(bf16_t* __restrict__ yh, bf16_t* __restrict__ yl, long idx, const float v[4]) {
    uint32_t h0, l0, h1, l1;
    split2(v[0], v[1], h0, l0); split2(v[2], v[3], h1, l1);
    *(uint2*)(yh + idx) = make_uint2(h0, h1);
    *(uint2*)(yl + idx) = make_uint2(l0, l1);
}
__device__ __forceinline__ void loadbf4(const bf16_t* x, long idx, float v[4]) {
    const uint2 t = *(const uint2*)(x + idx);
    v[0] = bf_lo(t.x); v[1] = bf_hi(t.x); v[2] = bf_lo(t.y); v[3] = bf_hi(t.y);
}

// ------------------------------------------------------------------ norms: fp32 rows -> (hi, lo)
constexpr int MAXIT = 8;      // cols <= NT*4*MAXIT = 8192
template <bool LAYER>
__global__ __launch_bounds__(NT) void norm_pair_kernel(const float* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const bf16_t* __restrict__ b, bf16_t* __restrict__ yh,
                                                       bf16_t* __restrict__ yl, int rows, int cols, float eps,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ float red[32];
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const long base = (long)row * cols;
        float xv[MAXIT][4];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
                const float4 t = *(const float4*)(x + base + c);
                xv[it][0] = t.x; xv[it][1] = t.y; xv[it][2] = t.z; xv[it][3] = t.w;
                s += (t.x + t.y) + (t.z + t.w);
            }
        }
        float mu = 0.f;
        if (LAYER) mu = block_sum(s, red) / cols;
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xv[it][e] - mu; ss += d * d; }
            }
        }
        const float var = block_sum(ss, red) / cols;
        const float rstd = 1.f / sqrtf(var + eps);
        if (threadIdx.x == 0) {                       // the statistics the fast path's backward kernels take (taped precise forward)
            if (LAYER && mean_out) mean_out[row] = mu;
            if (rstd_out) rstd_out[row] = rstd;
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
                float wv[4], bv[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
                loadbf4(w, c, wv);
                if (LAYER) loadbf4(b, c, bv);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (xv[it][e] - mu) * rstd * wv[e] + bv[e];
                store_pair4(yh, yl, base + c, o);
            }
        }
    }
}

// ------------------------------------------------------------------ rotary on fp32 q|k|v rows -> (hi, lo)
// x fp32 [tokens, heads*D] (ldx floats between tokens); the first rot_heads heads are rotated (HF rotate_half convention, fp32
// tables [tokens, D]), the others (v) are split as they are.  One work item = 4 dims of the first half + the matching 4 of the second.
__global__ __launch_bounds__(NT) void rope_pair_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ cs,
                                                       const float* __restrict__ sn, bf16_t* __restrict__ yh,
                                                       bf16_t* __restrict__ yl, long ldy, int tokens, int rot_heads, int heads, int D) {
    const int half = D >> 1, per_head = half >> 2;
    const long total = (long)tokens * heads * per_head;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % per_head);
        const long th = i / per_head;
        const int h = (int)(th % heads);
        const long t = th / heads;
        const float* p = x + t * ldx + (long)h * D + c * 4;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + half);
        float x1[4] = {a.x, a.y, a.z, a.w}, x2[4] = {b.x, b.y, b.z, b.w}, r1[4], r2[4];
        if (h < rot_heads) {
            const float* c1 = cs + t * D + c * 4; const float* s1 = sn + t * D + c * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r1[e] = x1[e] * c1[e] - x2[e] * s1[e];
                r2[e] = x2[e] * c1[e + half] + x1[e] * s1[e + half];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { r1[e] = x1[e]; r2[e] = x2[e]; }
        }
        const long o = t * ldy + (long)h * D + c * 4;
        store_pair4(yh, yl, o, r1);
        store_pair4(yh, yl, o + half, r2);
    }
}

// ------------------------------------------------------------------ activations on fp32 -> (hi, lo)
__device__ __forceinline__ float sigm_p(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float act_p(float v, int act) {
    if (act == SPACER_ACT_QUICK_GELU) return v * sigm_p(1.702f * v);
    if (act == SPACER_ACT_GELU_ERF) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (act == SPACER_ACT_SILU) return v * sigm_p(v);
    return v;
}
__global__ __launch_bounds__(NT) void act_pair_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ yh,
                                                      bf16_t* __restrict__ yl, long ldy, int rows, int cols, int act,
                                                      bf16_t* __restrict__ pre) {
    const int per_row = cols >> 2;
    const long total = (long)rows * per_row;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per_row; const int c = (int)(i % per_row) * 4;
        const float4 t = *(const float4*)(x + r * ldx + c);
        const float o[4] = {act_p(t.x, act), act_p(t.y, act), act_p(t.z, act), act_p(t.w, act)};
        store_pair4(yh, yl, r * ldy + c, o);
        if (pre) *(uint2*)(pre + r * ldy + c) = make_uint2(pack_bf2(t.x, t.y), pack_bf2(t.z, t.w));   // bf16(x): what act_bwd differentiates at
    }
}
// gu fp32 [rows, 2*inter] = [gate | up] -> silu(gate) * up as a pair [rows, inter]
__global__ __launch_bounds__(NT) void swiglu_pair_kernel(const float* __restrict__ gu, bf16_t* __restrict__ yh,
                                                         bf16_t* __restrict__ yl, int rows, int inter, bf16_t* __restrict__ gu16) {
    const int per_row = inter >> 2;
    const long total = (long)rows * per_row;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per_row; const int c = (int)(i % per_row) * 4;
        const float4 g = *(const float4*)(gu + r * 2 * inter + c), u = *(const float4*)(gu + r * 2 * inter + inter + c);
        const float o[4] = {g.x * sigm_p(g.x) * u.x, g.y * sigm_p(g.y) * u.y, g.z * sigm_p(g.z) * u.z, g.w * sigm_p(g.w) * u.w};
        store_pair4(yh, yl, r * inter + c, o);
        if (gu16) {                                   // bf16(gate | up): what swiglu_bwd differentiates at (taped precise forward)
            *(uint2*)(gu16 + r * 2 * inter + c) = make_uint2(pack_bf2(g.x, g.y), pack_bf2(g.z, g.w));
            *(uint2*)(gu16 + r * 2 * inter + inter + c) = make_uint2(pack_bf2(u.x, u.y), pack_bf2(u.z, u.w));
        }
    }
}

// ------------------------------------------------------------------ embedding gather with fp32 vision rows
__global__ __launch_bounds__(NT) void embed_f32video_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                            const float* __restrict__ video, const int* __restrict__ vrow,
                                                            float* __restrict__ out, int T, int H) {
    const int per = H >> 2;
    const long total = (long)T * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long t = i / per; const int c = (int)(i % per) * 4;
        const int vr = vrow ? vrow[t] : -1;
        float4 v;
        if (vr >= 0) v = *(const float4*)(video + (long)vr * H + c);
        else {
            const uint2 w = *(const uint2*)(table + ids[t] * (long)H + c);
            v = make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y));
        }
        *(float4*)(out + t * H + c) = v;
    }
}

// ------------------------------------------------------------------ attention forward on (hi, lo) operands
// The register-staged flash decomposition of attention.hip (S^T = K Q^T per 64-key tile, online softmax per lane column,
// O^T += V^T P^T straight from the score registers) with every MFMA operand as a pair.  Workgroup = 4 waves x 16*F query rows.
struct AttnPairArgs {
    const bf16_t *q_hi, *q_lo, *k_hi, *k_lo, *v_hi, *v_lo;
    bf16_t *o_hi, *o_lo;
    float* lse;
    long q_stride, kv_stride, o_stride;
    const spacer_attn_segment* segs;
    int num_segs, nqb, T, Hq, Hkv, causal;
    float scale;
};
constexpr int PBKV = 64;

template <int D, int F>
__global__ __launch_bounds__(256, 2) void attn_fwd_pair_kernel(AttnPairArgs a) {
    constexpr int DC = (D + 31) / 32;
    constexpr int DF = D / 16;
    constexpr int BQP = 64 * F;
    extern __shared__ __attribute__((aligned(16))) char smem[];    // K_hi | K_lo | V_hi | V_lo, row-major images
    char* kh_lds = smem;
    char* kl_lds = smem + AT_RM_BYTES;
    char* vh_lds = smem + 2 * AT_RM_BYTES;
    char* vl_lds = smem + 3 * AT_RM_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg_id = a.num_segs - 1 - blockIdx.x / a.nqb;        // longest first (attention.hip)
    const int qb = a.nqb - 1 - blockIdx.x % a.nqb;
    const int h = blockIdx.y, hk = h / (a.Hq / a.Hkv);
    const spacer_attn_segment seg = a.segs[seg_id];
    const int qb0 = qb * BQP;
    if (qb0 >= seg.q_len) return;
    const int l15 = lane & 15, g = lane >> 4;
    const int wq0 = qb0 + wave * 16 * F;

    bf16x8 qh[F][DC], ql[F][DC];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        int qi = wq0 + f * 16 + l15;
        qi = qi < seg.q_len ? qi : seg.q_len - 1;
        const long off = (long)(seg.q_start + qi) * a.q_stride + (long)h * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
            qh[f][dc] = frag_global<D>(a.q_hi + off, dc, lane);
            ql[f][dc] = frag_global<D>(a.q_lo + off, dc, lane);
        }
    }
    f32x4 oacc[F][DF];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[f][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[F], l_run[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { m_run[f] = -INFINITY; l_run[f] = 0.f; }
    const float c2 = a.scale * 1.4426950408889634f;

    const int own_len = a.causal ? min(seg.q_len, qb0 + BQP) : seg.q_len;
    const int n_pre = (seg.pre_len + PBKV - 1) / PBKV, n_tiles = n_pre + (own_len + PBKV - 1) / PBKV;

    for (int t = 0; t < n_tiles; ++t) {
        int start_abs, len, rel0; bool own;
        if (t < n_pre) { own = false; rel0 = t * PBKV; start_abs = seg.pre_start; len = seg.pre_len; }
        else { own = true; rel0 = (t - n_pre) * PBKV; start_abs = seg.q_start; len = own_len; }
        {
            const long off = (long)(start_abs + rel0) * a.kv_stride + (long)hk * D;
            uint4 r0[4], r1[4];
            tile_load<D>(r0, a.k_hi + off, a.kv_stride, len - rel0, tid);
            tile_load<D>(r1, a.k_lo + off, a.kv_stride, len - rel0, tid);
            __syncthreads();                                       // everyone is done with the previous tile's images
            tile_store<D, true, false>(r0, kh_lds, nullptr, tid);
            tile_store<D, true, false>(r1, kl_lds, nullptr, tid);
            tile_load<D>(r0, a.v_hi + off, a.kv_stride, len - rel0, tid);
            tile_load<D>(r1, a.v_lo + off, a.kv_stride, len - rel0, tid);
            tile_store<D, true, false>(r0, vh_lds, nullptr, tid);
            tile_store<D, true, false>(r1, vl_lds, nullptr, tid);
            __syncthreads();
        }
        if (own && a.causal && rel0 > wq0 + 16 * F - 1) continue;   // wave-uniform: tile entirely above this wave's last row

        // ---- S^T = K . Q^T with (hi, lo) operands; the small products first
        f32x4 st[4][F];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
            for (int f = 0; f < F; ++f) st[kf][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const bf16x8 kh = frag_rm(kh_lds, kf, dc, lane), kl = frag_rm(kl_lds, kf, dc, lane);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    st[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[f][dc], st[kf][f], 0, 0, 0);
                    st[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[f][dc], st[kf][f], 0, 0, 0);
                    st[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[f][dc], st[kf][f], 0, 0, 0);
                }
            }
        }

        // ---- online softmax (exp2 domain), P as a pair
        const bool need_mask = (rel0 + PBKV > len) || (own && a.causal && rel0 + PBKV - 1 > wq0);
        bf16x8 ph[F][2], pl[F][2];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int qi = wq0 + f * 16 + l15;
            float mx = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kr = rel0 + kf * 16 + g * 4 + r;
                        const bool ok = kr < len && !(own && a.causal && kr > qi);
                        st[kf][f][r] = ok ? st[kf][f][r] : -INFINITY;
                    }
            }
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kf][f][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[f], mx * c2);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run[f] - m_use);
            float psum = 0.f;
            float p[4][4], q[4][4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kf][r] = exp2f(__builtin_fmaf(st[kf][f][r], c2, -m_use));
                    psum += p[kf][r];
                }
            l_run[f] = l_run[f] * alpha + psum;
            m_run[f] = m_new;
#pragma unroll
            for (int d = 0; d < DF; ++d) oacc[f][d] *= alpha;
            ph[f][0] = pack_slots(p[0], p[1]);
            ph[f][1] = pack_slots(p[2], p[3]);
            // residuals p - bf16(p): exact in fp32
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint4 w = __builtin_bit_cast(uint4, ph[f][c]);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {            // word j holds slots 2j, 2j+1 = (kf = 2c + (j >> 1), r = 2 (j & 1) ..)
                    const int kf = 2 * c + (j >> 1), r = 2 * (j & 1);
                    q[kf][r] = p[kf][r] - bf_lo(ww[j]);
                    q[kf][r + 1] = p[kf][r + 1] - bf_hi(ww[j]);
                }
            }
            pl[f][0] = pack_slots(q[0], q[1]);
            pl[f][1] = pack_slots(q[2], q[3]);
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 vh = frag_tr(vh_lds, df, c, lane), vl = frag_tr(vl_lds, df, c, lane);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    oacc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph[f][c], oacc[f][df], 0, 0, 0);
                    oacc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl[f][c], oacc[f][df], 0, 0, 0);
                    oacc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph[f][c], oacc[f][df], 0, 0, 0);
                }
            }
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l as a pair; lane owns q = l15, d = df*16 + g*4 + r
#pragma unroll
    for (int f = 0; f < F; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qi = wq0 + f * 16 + l15;
        if (qi >= seg.q_len) continue;
        const float inv = 1.f / l;
        const long tok = seg.q_start + qi;
        const long o = tok * a.o_stride + (long)h * D;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const f32x4 v = oacc[f][df];
            const float ov[4] = {v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
            store_pair4(a.o_hi, a.o_lo, o + df * 16 + g * 4, ov);
        }
        if (a.lse && g == 0) a.lse[(long)h * a.T + tok] = m_run[f] * 0.6931471805599453f + logf(l);
    }
}


// ------------------------------------------------------------------ attention forward on (hi, lo) operands, DMA-staged (round 5)
// The same arithmetic as attn_fwd_pair_kernel (same MFMA order per wave, so the same bits), restructured around what the profile of
// the precise step showed: 899 us per launch on the two-group cfg3 layout against ~315 us of pure MFMA time -- the register-staged
// kernel issues a tile's global loads at the top of that tile (no prefetch: every tile pays its latency), and its two 128-row
// workgroups per CU each pull their own 64 KiB K_hi | K_lo | V_hi | V_lo tile.  Here a workgroup is 8 waves = 256 query rows sharing
// one tile (half the L2 -> CU bytes per query row), the four images reach LDS by global_load_lds DMA (no registers, no ds_write
// pass) into TWO buffer sets (128 KiB, one workgroup per CU, still 2 waves per SIMD), tile t+1 is requested right after the barrier
// that publishes tile t, one barrier per tile.
struct PairKeyTile { int start_abs, len, rel0; bool own; };
__device__ __forceinline__ PairKeyTile pair_key_tile(const spacer_attn_segment& seg, int n_pre, int own_len, int t) {
    PairKeyTile kt;
    if (t < n_pre) { kt.own = false; kt.rel0 = t * PBKV; kt.start_abs = seg.pre_start; kt.len = seg.pre_len; }
    else { kt.own = true; kt.rel0 = (t - n_pre) * PBKV; kt.start_abs = seg.q_start; kt.len = own_len; }
    return kt;
}

template <int D>
__global__ __launch_bounds__(512, 2) void attn_fwd_pair_dma_kernel(AttnPairArgs a) {
    constexpr int F = 2;
    constexpr int DC = (D + 31) / 32;
    constexpr int DF = D / 16;
    constexpr int BQP = 8 * 16 * F;                                 // 256 query rows per workgroup
    constexpr int BUF = 4 * AT_RM_BYTES;                            // K_hi | K_lo | V_hi | V_lo
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][BUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seg_id = a.num_segs - 1 - blockIdx.x / a.nqb;        // longest first
    const int qb = a.nqb - 1 - blockIdx.x % a.nqb;
    const int h = blockIdx.y, hk = h / (a.Hq / a.Hkv);
    const spacer_attn_segment seg = a.segs[seg_id];
    const int qb0 = qb * BQP;
    if (qb0 >= seg.q_len) return;
    const int l15 = lane & 15, g = lane >> 4;
    const int wq0 = qb0 + wave * 16 * F;

    bf16x8 qh[F][DC], ql[F][DC];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        int qi = wq0 + f * 16 + l15;
        qi = qi < seg.q_len ? qi : seg.q_len - 1;
        const long off = (long)(seg.q_start + qi) * a.q_stride + (long)h * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
            qh[f][dc] = frag_global<D>(a.q_hi + off, dc, lane);
            ql[f][dc] = frag_global<D>(a.q_lo + off, dc, lane);
        }
    }
    f32x4 oacc[F][DF];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[f][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[F], l_run[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { m_run[f] = -INFINITY; l_run[f] = 0.f; }
    const float c2 = a.scale * 1.4426950408889634f;

    const int own_len = a.causal ? min(seg.q_len, qb0 + BQP) : seg.q_len;
    const int n_pre = (seg.pre_len + PBKV - 1) / PBKV, n_tiles = n_pre + (own_len + PBKV - 1) / PBKV;

    // ---- DMA map: waves 2 i, 2 i + 1 fill image i (0 K_hi, 1 K_lo, 2 V_hi, 3 V_lo); a wave's piece j = rows (wave & 1) * 32 + 4 j + g,
    // j = 0..7; lane (g, l15) sits at physical chunk l15 of its row and fetches the logical chunk l15 ^ (row & 15) (frag_rm / frag_tr)
    const int img = wave >> 1;
    const bf16_t* gsrc = (img == 0 ? a.k_hi : img == 1 ? a.k_lo : img == 2 ? a.v_hi : a.v_lo) + (long)hk * D;
    const int img_off = img * AT_RM_BYTES + (wave & 1) * 8 * 1024;
    const int prow = (wave & 1) * 32 + g;
    const unsigned stride_b = (unsigned)a.kv_stride * 2u;
    unsigned doff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = prow + 4 * j;
        int c = l15 ^ (row & 15);
        if (c >= D / 8) c = 0;                                       // D = 80: padding chunks take finite filler (they meet Q's zeros / are never read)
        doff[j] = (unsigned)row * stride_b + (unsigned)c * 16u;
    }
    auto issue_tile = [&](int t, int buf) {
        const PairKeyTile kt = pair_key_tile(seg, n_pre, own_len, t);
        const int valid = kt.len - kt.rel0;                          // >= 1
        const char* base = (const char*)(gsrc + (long)(kt.start_abs + kt.rel0) * a.kv_stride);
        char* dst = smem + buf * BUF + img_off;
        if (valid >= PBKV) {
            const char* base16 = base + 16l * stride_b;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((i < 4 ? base : base16) + doff[i & 3]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int row = prow + 4 * i;
                row = row < valid ? row : valid - 1;                 // finite filler; those scores are masked, P is 0 there
                const unsigned dch = doff[i & 3] - (unsigned)(prow + 4 * (i & 3)) * stride_b;          // (ragged tiles only: recomputed, not kept)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)row * stride_b + dch),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
    };

    issue_tile(0, 0);
    for (int t = 0; t < n_tiles; ++t) {
        const PairKeyTile kt = pair_key_tile(seg, n_pre, own_len, t);
        const int rel0 = kt.rel0, len = kt.len;
        const bool own = kt.own;
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // everyone's pieces of tile t landed; everyone is done with tile t - 1
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n_tiles) issue_tile(t + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (own && a.causal && rel0 > wq0 + 16 * F - 1) continue;    // wave-uniform: tile entirely above this wave's last row
        const char* kh_lds = smem + buf * BUF;
        const char* kl_lds = kh_lds + AT_RM_BYTES;
        const char* vh_lds = kh_lds + 2 * AT_RM_BYTES;
        const char* vl_lds = kh_lds + 3 * AT_RM_BYTES;

        // ---- S^T = K . Q^T with (hi, lo) operands; the small products first
        f32x4 st[4][F];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
            for (int f = 0; f < F; ++f) st[kf][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const bf16x8 kh = frag_rm(kh_lds, kf, dc, lane), kl = frag_rm(kl_lds, kf, dc, lane);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    st[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[f][dc], st[kf][f], 0, 0, 0);
                    st[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[f][dc], st[kf][f], 0, 0, 0);
                    st[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[f][dc], st[kf][f], 0, 0, 0);
                }
            }
        }

        // ---- online softmax (exp2 domain), P as a pair
        const bool need_mask = (rel0 + PBKV > len) || (own && a.causal && rel0 + PBKV - 1 > wq0);
        bf16x8 ph[F][2], pl[F][2];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int qi = wq0 + f * 16 + l15;
            float mx = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kr = rel0 + kf * 16 + g * 4 + r;
                        const bool ok = kr < len && !(own && a.causal && kr > qi);
                        st[kf][f][r] = ok ? st[kf][f][r] : -INFINITY;
                    }
            }
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kf][f][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[f], mx * c2);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run[f] - m_use);
            float psum = 0.f;
            float p[4][4], q[4][4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kf][r] = exp2f(__builtin_fmaf(st[kf][f][r], c2, -m_use));
                    psum += p[kf][r];
                }
            l_run[f] = l_run[f] * alpha + psum;
            m_run[f] = m_new;
#pragma unroll
            for (int d = 0; d < DF; ++d) oacc[f][d] *= alpha;
            ph[f][0] = pack_slots(p[0], p[1]);
            ph[f][1] = pack_slots(p[2], p[3]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint4 w = __builtin_bit_cast(uint4, ph[f][c]);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kf = 2 * c + (j >> 1), r = 2 * (j & 1);
                    q[kf][r] = p[kf][r] - bf_lo(ww[j]);
                    q[kf][r + 1] = p[kf][r + 1] - bf_hi(ww[j]);
                }
            }
            pl[f][0] = pack_slots(q[0], q[1]);
            pl[f][1] = pack_slots(q[2], q[3]);
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 vh = frag_tr(vh_lds, df, c, lane), vl = frag_tr(vl_lds, df, c, lane);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    oacc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph[f][c], oacc[f][df], 0, 0, 0);
                    oacc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl[f][c], oacc[f][df], 0, 0, 0);
                    oacc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph[f][c], oacc[f][df], 0, 0, 0);
                }
            }
    }

#pragma unroll
    for (int f = 0; f < F; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qi = wq0 + f * 16 + l15;
        if (qi >= seg.q_len) continue;
        const float inv = 1.f / l;
        const long tok = seg.q_start + qi;
        const long o = tok * a.o_stride + (long)h * D;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const f32x4 v = oacc[f][df];
            const float ov[4] = {v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
            store_pair4(a.o_hi, a.o_lo, o + df * 16 + g * 4, ov);
        }
        if (a.lse && g == 0) a.lse[(long)h * a.T + tok] = m_run[f] * 0.6931471805599453f + logf(l);
    }
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" int spacer_split_f32_pair(const float* x, long ldx, void* y_hi, void* y_lo, long ldy, int rows, int cols,
                                     spacer_stream_t stream) {
    SP_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, SPACER_EINVAL, "split_f32_pair: cols / strides must be multiples of 4");
    if (rows <= 0 || cols <= 0) return SPACER_OK;
    hipLaunchKernelGGL(act_pair_kernel, dim3(grid_for((long)rows * cols / 4)), dim3(NT), 0, (hipStream_t)stream, x, ldx,
                       (bf16_t*)y_hi, (bf16_t*)y_lo, ldy, rows, cols, SPACER_ACT_NONE, (bf16_t*)nullptr);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_act_f32_pair(const float* x, long ldx, void* y_hi, void* y_lo, long ldy, int rows, int cols, int act,
                                   void* pre_bf16, spacer_stream_t stream) {
    SP_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, SPACER_EINVAL, "act_f32_pair: cols / strides must be multiples of 4");
    SP_REQUIRE(act >= SPACER_ACT_NONE && act <= SPACER_ACT_SILU, SPACER_EINVAL, "act_f32_pair: unknown activation %d", act);
    if (rows <= 0 || cols <= 0) return SPACER_OK;
    hipLaunchKernelGGL(act_pair_kernel, dim3(grid_for((long)rows * cols / 4)), dim3(NT), 0, (hipStream_t)stream, x, ldx,
                       (bf16_t*)y_hi, (bf16_t*)y_lo, ldy, rows, cols, act, (bf16_t*)pre_bf16);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_swiglu_f32_pair(const float* gu, void* y_hi, void* y_lo, int rows, int inter, void* gu_bf16,
                                      spacer_stream_t stream) {
    SP_REQUIRE(inter % 4 == 0, SPACER_EINVAL, "swiglu_f32_pair: inter=%d must be a multiple of 4", inter);
    if (rows <= 0 || inter <= 0) return SPACER_OK;
    hipLaunchKernelGGL(swiglu_pair_kernel, dim3(grid_for((long)rows * inter / 4)), dim3(NT), 0, (hipStream_t)stream, gu,
                       (bf16_t*)y_hi, (bf16_t*)y_lo, rows, inter, (bf16_t*)gu_bf16);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_norm_f32_pair(const float* x, const void* w, const void* b, void* y_hi, void* y_lo, int rows, int cols,
                                    float eps, int layer, float* mean_out, float* rstd_out, spacer_stream_t stream) {
    SP_REQUIRE(cols % 4 == 0 && cols <= NT * 4 * MAXIT, SPACER_EINVAL, "norm_f32_pair: cols=%d must be a multiple of 4, <= %d", cols,
               NT * 4 * MAXIT);
    SP_REQUIRE(!layer || b, SPACER_EINVAL, "norm_f32_pair: LayerNorm needs a bias");
    if (rows <= 0) return SPACER_OK;
    const int grid = rows < 256 * 8 ? rows : 256 * 8;
    if (layer)
        hipLaunchKernelGGL(norm_pair_kernel<true>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, (const bf16_t*)w, (const bf16_t*)b,
                           (bf16_t*)y_hi, (bf16_t*)y_lo, rows, cols, eps, mean_out, rstd_out);
    else
        hipLaunchKernelGGL(norm_pair_kernel<false>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, (const bf16_t*)w, (const bf16_t*)nullptr,
                           (bf16_t*)y_hi, (bf16_t*)y_lo, rows, cols, eps, mean_out, rstd_out);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_rope_f32_pair(const float* x, long ldx, const float* cos_t, const float* sin_t, void* y_hi, void* y_lo, long ldy,
                                    int tokens, int rot_heads, int heads, int head_dim, spacer_stream_t stream) {
    SP_REQUIRE(head_dim % 8 == 0 && ldx % 4 == 0 && ldy % 4 == 0, SPACER_EINVAL, "rope_f32_pair: head_dim=%d must be a multiple of 8", head_dim);
    SP_REQUIRE(rot_heads >= 0 && rot_heads <= heads, SPACER_EINVAL, "rope_f32_pair: rot_heads=%d of %d", rot_heads, heads);
    if (tokens <= 0 || heads <= 0) return SPACER_OK;
    hipLaunchKernelGGL(rope_pair_kernel, dim3(grid_for((long)tokens * heads * head_dim / 8)), dim3(NT), 0, (hipStream_t)stream, x, ldx,
                       cos_t, sin_t, (bf16_t*)y_hi, (bf16_t*)y_lo, ldy, tokens, rot_heads, heads, head_dim);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_embed_fwd_f32video(const int64_t* ids, const void* table, const float* video, const int* video_row_of_token,
                                         float* out, int T, int H, spacer_stream_t stream) {
    SP_REQUIRE(H % 4 == 0, SPACER_EINVAL, "embed_fwd_f32video: H=%d must be a multiple of 4", H);
    if (T <= 0) return SPACER_OK;
    hipLaunchKernelGGL(embed_f32video_kernel, dim3(grid_for((long)T * H / 4)), dim3(NT), 0, (hipStream_t)stream, ids,
                       (const bf16_t*)table, video, video_row_of_token, out, T, H);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_attn_fwd_pair(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* v_hi,
                                    const void* v_lo, void* o_hi, void* o_lo, float* lse, long q_stride, long kv_stride, long o_stride,
                                    const spacer_attn_segment* segs_dev, int num_segs, int max_q_len, int T, int Hq, int Hkv, int D,
                                    int causal, float scale, int variant, spacer_stream_t stream) {
    SP_REQUIRE(D == 80 || D == 128, SPACER_EINVAL, "attn_fwd_pair: head_dim=%d (80 or 128)", D);
    SP_REQUIRE(variant == 0 || variant == 1, SPACER_EINVAL, "attn_fwd_pair: variant %d (0 = DMA-staged, 1 = register-staged)", variant);
    SP_REQUIRE(Hkv > 0 && Hq % Hkv == 0, SPACER_EINVAL, "attn_fwd_pair: Hq=%d must be a multiple of Hkv=%d", Hq, Hkv);
    SP_REQUIRE(q_stride % 8 == 0 && kv_stride % 8 == 0 && o_stride % 4 == 0, SPACER_EINVAL, "attn_fwd_pair: strides must keep 16-byte rows");
    if (num_segs <= 0 || max_q_len <= 0) return SPACER_OK;
    constexpr int F = 2;
    const bool dma = variant == 0;                                  // 0 = DMA-staged 256-row workgroups (round 5), 1 = the register-staged round-3 kernel
    AttnPairArgs a = {};
    a.q_hi = (const bf16_t*)q_hi; a.q_lo = (const bf16_t*)q_lo; a.k_hi = (const bf16_t*)k_hi; a.k_lo = (const bf16_t*)k_lo;
    a.v_hi = (const bf16_t*)v_hi; a.v_lo = (const bf16_t*)v_lo; a.o_hi = (bf16_t*)o_hi; a.o_lo = (bf16_t*)o_lo; a.lse = lse;
    a.q_stride = q_stride; a.kv_stride = kv_stride; a.o_stride = o_stride; a.segs = segs_dev; a.num_segs = num_segs;
    a.nqb = cdiv(max_q_len, dma ? 256 : 64 * F); a.T = T; a.Hq = Hq; a.Hkv = Hkv; a.causal = causal; a.scale = scale;
    if (dma) {
        constexpr int LDS2 = 8 * AT_RM_BYTES;
        static const int once2 = hipFuncSetAttribute((const void*)attn_fwd_pair_dma_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2)
                               + hipFuncSetAttribute((const void*)attn_fwd_pair_dma_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        SP_REQUIRE(once2 == 0, SPACER_ELAUNCH, "attn_fwd_pair: cannot raise the dynamic LDS limit to %d bytes", LDS2);
        const dim3 grid2(num_segs * a.nqb, Hq);
        if (D == 128) hipLaunchKernelGGL(attn_fwd_pair_dma_kernel<128>, grid2, dim3(512), LDS2, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(attn_fwd_pair_dma_kernel<80>, grid2, dim3(512), LDS2, (hipStream_t)stream, a);
        SP_CHECK_LAUNCH();
        return SPACER_OK;
    }
    constexpr int LDS = 4 * AT_RM_BYTES;
    static const int once = hipFuncSetAttribute((const void*)attn_fwd_pair_kernel<128, F>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                          + hipFuncSetAttribute((const void*)attn_fwd_pair_kernel<80, F>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SP_REQUIRE(once == 0, SPACER_ELAUNCH, "attn_fwd_pair: cannot raise the dynamic LDS limit to %d bytes", LDS);
    const dim3 grid(num_segs * a.nqb, Hq);
    if (D == 128) hipLaunchKernelGGL((attn_fwd_pair_kernel<128, F>), grid, dim3(256), LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_pair_kernel<80, F>), grid, dim3(256), LDS, (hipStream_t)stream, a);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
