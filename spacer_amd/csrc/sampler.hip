// Token sampling for the rollout loop: temperature -> top-k -> top-p -> multinomial, the warper chain HF
// generate applies for the reference's GenerationConfig (SG_RLVR_trainer.py:277-284: do_sample, top_p 0.95,
// temperature 1, and the era-default top_k = 50).  One 1024-thread workgroup per sequence:
//   1. exact k-th largest logit by 4-pass radix select on order-preserving keys (wave-aggregated LDS atomics)
//   2. candidates (logit >= k-th, ties kept as HF does) gathered to LDS, bitonic-sorted (value desc, index asc)
//   3. nucleus: keep rank i while the descending exclusive cumulative probability < top_p  (== HF's
//      "remove ascending cumsum <= 1 - top_p", min_tokens_to_keep = 1)
//   4. inverse-CDF draw with a Philox4x32-10 uniform keyed by (seed, step, row)
#include "common.h"

namespace {

constexpr int SNT = 1024;
constexpr int CAP = 1024;    // candidate capacity (top_k <= CAP; ties beyond CAP are dropped)

__device__ __forceinline__ uint32_t okey(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t step, uint32_t row) {
    uint32_t c[4] = {step, row, 0x5bd1e995u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
}

__device__ __forceinline__ float block_excl_scan(float v, float* wsum, float& total) {
    // exclusive prefix over SNT threads; wsum = 16 floats of LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    float base = 0.f, tot = 0.f;
    for (int i = 0; i < SNT / 64; ++i) { if (i < w) base += wsum[i]; tot += wsum[i]; }
    total = tot;
    return base + inc - v;
}

// exact k-th largest key of `count` values produced by load(i): 4-pass radix select, wave-aggregated LDS atomics (hist: 256 ints of LDS)
template <class Load>
__device__ __forceinline__ uint32_t kth_key_fn(Load&& load, int count, int k, int* hist, int& sel_bin, int& sel_k) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += SNT) hist[i] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < count; i0 += SNT) {
            const int i = i0 + tid;
            bool live = i < count;
            const uint32_t key = live ? okey(load(i)) : 0u;
            if (pass > 0) live = live && ((key >> (shift + 8)) == prefix);
            const int bin = (key >> shift) & 255;
            unsigned long long act = __ballot(live);
            while (act) {
                const int leader = __ffsll((long long)act) - 1;
                const int lb = __shfl(bin, leader, 64);
                const unsigned long long same = __ballot(live && bin == lb);
                if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
                act &= ~same;
            }
        }
        __syncthreads();
        if (tid < 64) {
            // first bin, walking down from 255, where the running count reaches k -- by one wave (4 bins per lane, shuffle
            // prefix) instead of one thread walking 256 dependent LDS reads per pass (that walk was half of the kernel's time)
            int hb[4], own = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { hb[q] = hist[255 - (lane * 4 + q)]; own += hb[q]; }
            int incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            int cum = incl - own, pos = 256, cum_at = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (pos == 256 && cum + hb[q] >= k) { pos = lane * 4 + q; cum_at = cum; }
                cum += hb[q];
            }
            int first = pos;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
            if (first == 256) { if (lane == 63) { sel_bin = 0; sel_k = k - incl; } }  // k exceeds the population (cannot happen for count >= k): what the serial walk left
            else if (pos == first) { sel_bin = 255 - pos; sel_k = k - cum_at; }
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint32_t)sel_bin;
        k = sel_k;
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(SNT) void sample_kernel(const float* __restrict__ logits, long ld, int vocab, int top_k,
                                                     float top_p, float inv_temp, uint64_t seed,
                                                     const int* __restrict__ step_dev, int eos, int pad, int suppress_eos,
                                                     int* __restrict__ finished, int64_t* __restrict__ out_ids,
                                                     float* __restrict__ out_logp, int step_bias, int64_t* __restrict__ out_mat,
                                                     long out_ld, const float* __restrict__ pre_val, const int* __restrict__ pre_idx,
                                                     const int* __restrict__ pre_n) {
    // pre_n != nullptr (wide form): the candidates of row b were gathered by sample_gather_kernel into pre_val / pre_idx [b][CAP]; a row whose
    // candidates overflowed CAP falls back to this kernel's own two-pass / radix-select path
    __shared__ int hist[256];
    __shared__ int sel_bin, sel_k, ncand;
    __shared__ float cval[CAP];
    __shared__ int cidx[CAP];
    __shared__ float wsum[SNT / 64];
    __shared__ float red[32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (finished && finished[b]) {
        if (tid == 0) {
            out_ids[b] = pad;
            if (out_logp) out_logp[b] = 0.f;
            if (out_mat) out_mat[(long)b * out_ld + (*step_dev + step_bias)] = pad;
        }
        return;
    }
    const float* row = logits + (long)b * ld;
    auto fix = [&](float raw, int i) -> float { return (suppress_eos && i == eos) ? -INFINITY : raw * inv_temp; };
    auto val = [&](int i) -> float { return fix(row[i], i); };
    const bool vec_ok = (ld % 4 == 0) && (((uintptr_t)logits & 15) == 0);

    auto kth_key = [&](auto&& load, int count, int k) -> uint32_t { return kth_key_fn(load, count, k, hist, sel_bin, sel_k); };
    auto gather = [&](uint32_t thr) {
        if (tid == 0) ncand = 0;
        __syncthreads();
        auto consider = [&](float x, int i) {
            if (okey(x) >= thr) {
                const int slot = atomicAdd(&ncand, 1);
                if (slot < CAP) { cval[slot] = x; cidx[slot] = i; }
            }
        };
        if (vec_ok) {
            for (int i4 = tid; i4 < (vocab >> 2); i4 += SNT) {
                const float4 v = *(const float4*)(row + i4 * 4);
                consider(fix(v.x, i4 * 4), i4 * 4); consider(fix(v.y, i4 * 4 + 1), i4 * 4 + 1);
                consider(fix(v.z, i4 * 4 + 2), i4 * 4 + 2); consider(fix(v.w, i4 * 4 + 3), i4 * 4 + 3);
            }
            for (int i = (vocab & ~3) + tid; i < vocab; i += SNT) consider(val(i), i);
        } else {
            for (int i = tid; i < vocab; i += SNT) consider(val(i), i);
        }
        __syncthreads();
    };

    // ---- 1. candidates.  Fast path (2 passes over the row): the top_k-th largest of the 1024 per-thread maxima is a
    // lower bound of the top_k-th largest logit, so "logit >= bound" keeps a superset of the top-k (typically ~2k
    // values) that is then trimmed exactly after the sort.  Pathological rows (mass ties) overflow CAP and take the
    // exact 4-pass radix select over the whole row.
    bool trimmed_exactly = false;
    const bool pre = pre_n && pre_n[b] <= CAP;
    if (pre) {
        if (tid == 0) ncand = pre_n[b];
        if (tid < pre_n[b]) { cval[tid] = pre_val[(long)b * CAP + tid]; cidx[tid] = pre_idx[(long)b * CAP + tid]; }
        __syncthreads();
    } else {
    float lmax = -INFINITY;
    if (vec_ok) {
        for (int i4 = tid; i4 < (vocab >> 2); i4 += SNT) {
            const float4 v = *(const float4*)(row + i4 * 4);
            lmax = fmaxf(fmaxf(lmax, fmaxf(fix(v.x, i4 * 4), fix(v.y, i4 * 4 + 1))), fmaxf(fix(v.z, i4 * 4 + 2), fix(v.w, i4 * 4 + 3)));
        }
        for (int i = (vocab & ~3) + tid; i < vocab; i += SNT) lmax = fmaxf(lmax, val(i));
    } else {
        for (int i = tid; i < vocab; i += SNT) lmax = fmaxf(lmax, val(i));
    }
    cval[tid] = lmax;
    __syncthreads();
    const uint32_t bound = kth_key([&](int i) { return cval[i]; }, SNT, top_k);
    __syncthreads();
    gather(bound);
    if (ncand > CAP) {
        const uint32_t kth = kth_key(val, vocab, top_k);
        gather(kth);
        trimmed_exactly = true;
    }
    }
    const int n = min(ncand, CAP);   // (ties beyond CAP are dropped)
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += SNT) { cval[i] = -INFINITY; cidx[i] = 0x7fffffff; }
    __syncthreads();
    // bitonic sort: descending value, ascending index on ties
    for (int sz = 2; sz <= np2; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int i = tid; i < np2; i += SNT) {
                const int j = i ^ st;
                if (j > i) {
                    const bool desc = ((i & sz) == 0);
                    const float vi = cval[i], vj = cval[j];
                    const int ii = cidx[i], ij = cidx[j];
                    const bool i_first = (vi > vj) || (vi == vj && ii < ij);   // i should precede j in final order
                    if (i_first != desc) { cval[i] = vj; cval[j] = vi; cidx[i] = ij; cidx[j] = ii; }
                }
            }
            __syncthreads();
        }

    // exact top-k with ties (HF keeps every logit >= the k-th largest): candidates are sorted descending
    int n_keep = n;
    if (!trimmed_exactly) {
        const float kval = cval[min(top_k, n) - 1];
        __shared__ int cut;
        if (tid == 0) cut = 0;
        __syncthreads();
        if (tid < n && cval[tid] >= kval) atomicMax(&cut, tid + 1);
        __syncthreads();
        n_keep = cut;
    }
    // ---- 3. nucleus over the sorted candidates (n <= CAP <= SNT: one candidate per thread)
    const float vmax = cval[0];
    const float p = (tid < n_keep) ? __expf(cval[tid] - vmax) : 0.f;
    float Z;
    const float excl = block_excl_scan(p, wsum, Z);
    const bool keep = (tid < n_keep) && (tid == 0 || excl < top_p * Z);
    const float pk = keep ? p : 0.f;
    float Zk;
    const float exk = block_excl_scan(pk, wsum, Zk);

    // ---- 4. draw
    const int step_now = *step_dev + step_bias;
    const float u = philox_uniform(seed, (uint32_t)step_now, (uint32_t)b) * Zk;
    __shared__ int chosen;
    if (tid == 0) chosen = -1;
    __syncthreads();
    if (keep && exk <= u && u < exk + pk) chosen = tid;     // at most one thread satisfies this
    __syncthreads();
    if (chosen < 0) {   // numeric edge (u == Zk after rounding): take the last kept candidate
        const int cand = keep ? tid : -1;
        int best = cand;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
        if (lane == 0) atomicMax(&chosen, best);
        __syncthreads();
    }
    const int tok = cidx[chosen];
    if (out_logp) {     // log-prob of the drawn token under the full (untruncated) softmax
        float m = -INFINITY;
        for (int i = tid; i < vocab; i += SNT) m = fmaxf(m, val(i));
        const float M = block_max(m, red);
        float s = 0.f;
        for (int i = tid; i < vocab; i += SNT) s += __expf(val(i) - M);
        const float S = block_sum(s, red);
        if (tid == 0) out_logp[b] = cval[chosen] - M - logf(S);
    }
    if (tid == 0) {
        out_ids[b] = tok;
        if (out_mat) out_mat[(long)b * out_ld + step_now] = tok;      // column = decode step: no per-step copy kernel on the host side
        if (finished && tok == eos) finished[b] = 1;
    }
}

// ---- wide form (round 6): the two passes over a row's logits by S workgroups per row instead of one.  At decode batch sizes (8 ... 128 rows of
// 152 064 logits) one workgroup per row leaves most CUs idle and each reads 1.2 MB alone.  Measured (scripts/probes/sampler_time.py):
// 32.6 / 39.4 / 46.9 / 58.3 us per call at 8 / 12 / 64 / 128 rows in that form, 30.4 / 30.7 / 34.3 / 47.2 us in this one -- the passes shrink, the
// select / sort / draw chain (now three dependent launches) stays.  Here
//   sample_partmax_kernel   (B x S workgroups)  per-thread maxima of a slice of the row -> pmax[b][S x 1024]
//   sample_gather_kernel    (B x S workgroups)  bound = top_k-th largest of the row's S x 1024 maxima (each is a logit of the row, so
//                                               the bound is <= the top_k-th largest logit: the candidates are a superset of the top-k,
//                                               as in the single-workgroup form, with a tighter bound); slice candidates >= bound are
//                                               collected in LDS and appended to the row's list with ONE global atomic per workgroup
//   sample_kernel (pre_*)   (B workgroups)      sort / exact top-k trim / nucleus / draw on the gathered list -- the same code, hence the
//                                               same token for the same logits and counter (the sort orders by (value, index))
__global__ __launch_bounds__(SNT) void sample_partmax_kernel(const float* __restrict__ logits, long ld, int vocab, int S, float inv_temp,
                                                             int eos, int suppress_eos, const int* __restrict__ finished,
                                                             float* __restrict__ pmax, int* __restrict__ pre_n) {
    const int b = blockIdx.x / S, sl = blockIdx.x % S, tid = threadIdx.x;
    if (sl == 0 && tid == 0) pre_n[b] = 0;
    if (finished && finished[b]) return;
    const float* row = logits + (long)b * ld;
    auto fix = [&](float raw, int i) -> float { return (suppress_eos && i == eos) ? -INFINITY : raw * inv_temp; };
    const int n4 = vocab >> 2, per = (n4 + S - 1) / S, q0 = sl * per, q1 = min(n4, q0 + per);
    float lmax = -INFINITY;
    for (int i4 = q0 + tid; i4 < q1; i4 += SNT) {
        const float4 v = *(const float4*)(row + i4 * 4);
        lmax = fmaxf(fmaxf(lmax, fmaxf(fix(v.x, i4 * 4), fix(v.y, i4 * 4 + 1))), fmaxf(fix(v.z, i4 * 4 + 2), fix(v.w, i4 * 4 + 3)));
    }
    if (sl == S - 1)
        for (int i = (vocab & ~3) + tid; i < vocab; i += SNT) lmax = fmaxf(lmax, fix(row[i], i));
    pmax[((long)b * S + sl) * SNT + tid] = lmax;
}

constexpr int WIDE_MAX_S = 8;
__global__ __launch_bounds__(SNT) void sample_gather_kernel(const float* __restrict__ logits, long ld, int vocab, int S, int top_k, float inv_temp,
                                                            int eos, int suppress_eos, const int* __restrict__ finished,
                                                            const float* __restrict__ pmax, float* __restrict__ pre_val,
                                                            int* __restrict__ pre_idx, int* __restrict__ pre_n) {
    __shared__ float mx[WIDE_MAX_S * SNT];
    __shared__ int hist[256];
    __shared__ int sel_bin, sel_k, nloc, base;
    __shared__ float lval[CAP];
    __shared__ int lidx[CAP];
    const int b = blockIdx.x / S, sl = blockIdx.x % S, tid = threadIdx.x;
    if (finished && finished[b]) return;
    for (int i = tid; i < S * SNT; i += SNT) mx[i] = pmax[(long)b * S * SNT + i];
    if (tid == 0) nloc = 0;
    __syncthreads();
    const uint32_t thr = kth_key_fn([&](int i) { return mx[i]; }, S * SNT, top_k, hist, sel_bin, sel_k);
    const float* row = logits + (long)b * ld;
    auto fix = [&](float raw, int i) -> float { return (suppress_eos && i == eos) ? -INFINITY : raw * inv_temp; };
    auto consider = [&](float x, int i) {
        if (okey(x) >= thr) {
            const int slot = atomicAdd(&nloc, 1);
            if (slot < CAP) { lval[slot] = x; lidx[slot] = i; }
        }
    };
    const int n4 = vocab >> 2, per = (n4 + S - 1) / S, q0 = sl * per, q1 = min(n4, q0 + per);
    for (int i4 = q0 + tid; i4 < q1; i4 += SNT) {
        const float4 v = *(const float4*)(row + i4 * 4);
        consider(fix(v.x, i4 * 4), i4 * 4); consider(fix(v.y, i4 * 4 + 1), i4 * 4 + 1);
        consider(fix(v.z, i4 * 4 + 2), i4 * 4 + 2); consider(fix(v.w, i4 * 4 + 3), i4 * 4 + 3);
    }
    if (sl == S - 1)
        for (int i = (vocab & ~3) + tid; i < vocab; i += SNT) consider(fix(row[i], i), i);
    __syncthreads();
    if (tid == 0) base = atomicAdd(&pre_n[b], nloc);          // (a slice that overflowed CAP pushes the row's count past CAP: fallback)
    __syncthreads();
    const int n = min(nloc, CAP);
    for (int i = tid; i < n; i += SNT)
        if (base + i < CAP) { pre_val[(long)b * CAP + base + i] = lval[i]; pre_idx[(long)b * CAP + base + i] = lidx[i]; }
}

// Synthetic completion lengths (bench / tests): row b may emit EOS at token index eos_at[b] and nowhere else.  Runs on the step's
// logits right before the sampler: the EOS logit becomes -inf (never drawn) or, at the scheduled index, so large that every other
// candidate's probability underflows to 0 (the nucleus keeps the top candidate unconditionally).
__global__ void eos_schedule_kernel(float* __restrict__ logits, long ld, int B, const int* __restrict__ step_dev, int step_bias,
                                    const int* __restrict__ eos_at, int eos) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    logits[(long)b * ld + eos] = (*step_dev + step_bias == eos_at[b]) ? 1e30f : -INFINITY;
}

}  // namespace

// workspace of the wide form: per row WIDE_MAX_S x 1024 partial maxima + CAP candidate (value, index) pairs + a counter
extern "C" long spacer_sample_workspace_bytes(int B, int vocab) {
    (void)vocab;
    return (long)B * ((long)WIDE_MAX_S * SNT * 4 + (long)CAP * 8 + 4);
}

static int sample_launch(const float* logits, long ld, int B, int vocab, int top_k, float top_p, float temperature, uint64_t seed,
                         const int* step_dev, int step_bias, int eos_id, int pad_id, int suppress_eos, int* finished, int64_t* out_ids,
                         float* out_logp, int64_t* out_mat, long out_ld, spacer_stream_t stream, void* ws = nullptr, long ws_bytes = 0) {
    SP_REQUIRE(top_k >= 1 && top_k <= CAP, SPACER_EINVAL,
               "sample_top_p: top_k=%d must be in 1..%d (full-vocabulary nucleus, top_k=0, is not implemented yet)", top_k, CAP);
    SP_REQUIRE(top_p > 0.f && top_p <= 1.f && temperature > 0.f, SPACER_EINVAL, "sample_top_p: bad top_p/temperature");
    SP_REQUIRE(vocab >= top_k, SPACER_EINVAL, "sample_top_p: vocab < top_k");
    if (B <= 0) return SPACER_OK;
    // wide form: S workgroups per row for the two passes over the logits when the batch alone cannot fill the chip (a caller's
    // workspace selects it; rows of >= 64 K logits, 16-byte aligned)
    int S = B >= 256 ? 1 : (256 + B - 1) / B;
    S = S > WIDE_MAX_S ? WIDE_MAX_S : S;
    const bool wide = ws && ws_bytes >= spacer_sample_workspace_bytes(B, vocab) && S > 1 && vocab >= 65536 && ld % 4 == 0 &&
                      ((uintptr_t)logits & 15) == 0 && ((uintptr_t)ws & 15) == 0 && top_k <= WIDE_MAX_S * SNT;
    if (wide) {
        float* pmax = (float*)ws;
        float* pre_val = pmax + (long)B * WIDE_MAX_S * SNT;
        int* pre_idx = (int*)(pre_val + (long)B * CAP);
        int* pre_n = pre_idx + (long)B * CAP;
        hipLaunchKernelGGL(sample_partmax_kernel, dim3(B * S), dim3(SNT), 0, (hipStream_t)stream, logits, ld, vocab, S, 1.f / temperature,
                           eos_id, suppress_eos, (const int*)finished, pmax, pre_n);
        hipLaunchKernelGGL(sample_gather_kernel, dim3(B * S), dim3(SNT), 0, (hipStream_t)stream, logits, ld, vocab, S, top_k,
                           1.f / temperature, eos_id, suppress_eos, (const int*)finished, (const float*)pmax, pre_val, pre_idx, pre_n);
        hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(SNT), 0, (hipStream_t)stream, logits, ld, vocab, top_k, top_p,
                           1.f / temperature, seed, step_dev, eos_id, pad_id, suppress_eos, finished, out_ids, out_logp, step_bias, out_mat,
                           out_ld, (const float*)pre_val, (const int*)pre_idx, (const int*)pre_n);
        SP_CHECK_LAUNCH();
        return SPACER_OK;
    }
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(SNT), 0, (hipStream_t)stream, logits, ld, vocab, top_k, top_p,
                       1.f / temperature, seed, step_dev, eos_id, pad_id, suppress_eos, finished, out_ids, out_logp, step_bias, out_mat,
                       out_ld, (const float*)nullptr, (const int*)nullptr, (const int*)nullptr);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_sample_top_p(const float* logits, long ld, int B, int vocab, int top_k, float top_p,
                                   float temperature, uint64_t seed, const int* step_dev, int eos_id, int pad_id,
                                   int suppress_eos, int* finished, int64_t* out_ids, float* out_logp, void* workspace,
                                   long workspace_bytes, spacer_stream_t stream) {
    return sample_launch(logits, ld, B, vocab, top_k, top_p, temperature, seed, step_dev, 0, eos_id, pad_id, suppress_eos, finished,
                         out_ids, out_logp, nullptr, 0, stream, workspace, workspace_bytes);
}

extern "C" int spacer_sample_top_p_step(const float* logits, long ld, int B, int vocab, int top_k, float top_p, float temperature,
                                        uint64_t seed, const int* step_dev, int step_bias, int eos_id, int pad_id, int suppress_eos,
                                        int* finished, int64_t* out_ids, int64_t* out_matrix, long out_ld, spacer_stream_t stream) {
    return sample_launch(logits, ld, B, vocab, top_k, top_p, temperature, seed, step_dev, step_bias, eos_id, pad_id, suppress_eos,
                         finished, out_ids, nullptr, out_matrix, out_ld, stream);
}

extern "C" int spacer_sample_top_p_step_ws(const float* logits, long ld, int B, int vocab, int top_k, float top_p, float temperature,
                                           uint64_t seed, const int* step_dev, int step_bias, int eos_id, int pad_id, int suppress_eos,
                                           int* finished, int64_t* out_ids, int64_t* out_matrix, long out_ld, void* workspace,
                                           long workspace_bytes, spacer_stream_t stream) {
    return sample_launch(logits, ld, B, vocab, top_k, top_p, temperature, seed, step_dev, step_bias, eos_id, pad_id, suppress_eos,
                         finished, out_ids, nullptr, out_matrix, out_ld, stream, workspace, workspace_bytes);
}

extern "C" int spacer_eos_schedule(float* logits, long ld, int B, int vocab, const int* step_dev, int step_bias, const int* eos_at,
                                   int eos_id, spacer_stream_t stream) {
    if (B <= 0) return SPACER_OK;
    SP_REQUIRE(logits && step_dev && eos_at, SPACER_EINVAL, "eos_schedule: null operand");
    SP_REQUIRE(eos_id >= 0 && eos_id < vocab, SPACER_EINVAL, "eos_schedule: eos_id %d outside the vocabulary (%d)", eos_id, vocab);
    hipLaunchKernelGGL(eos_schedule_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, logits, ld, B, step_dev, step_bias,
                       eos_at, eos_id);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
