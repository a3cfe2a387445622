// Per-token log-prob (log-softmax + gather), its backward, first-EOS mask, GRPO loss + analytic gradient,
// grad-norm and fused AdamW.  Reference arithmetic: SG_RLVR_trainer.py:353-366 (logps), :493-498 (mask),
// :551-552 (k3 KL), :640-643 (loss), :682 (kl metric); optimizer = AdamW behind HF Trainer.training_step.
#include "common.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ logp = logit[tgt] - logsumexp(row)
// one workgroup per row; online (max, sum) per thread over float4 loads, then a block combine.
__global__ __launch_bounds__(NT) void logprob_fwd_kernel(const float* __restrict__ logits, long ld,
                                                         const int64_t* __restrict__ tgt, float* __restrict__ logp,
                                                         float* __restrict__ lse_out, int rows, int vocab) {
    __shared__ float red[32];
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* p = logits + (long)row * ld;
        float m = -INFINITY, s = 0.f;
        const int v4 = vocab >> 2;
        for (int i = threadIdx.x; i < v4; i += NT) {
            const float4 x = *(const float4*)(p + i * 4);
            const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
            if (mx > m) { s *= __expf(m - mx); m = mx; }
            s += __expf(x.x - m) + __expf(x.y - m) + __expf(x.z - m) + __expf(x.w - m);
        }
        for (int i = v4 * 4 + threadIdx.x; i < vocab; i += NT) {
            const float x = p[i];
            if (x > m) { s *= __expf(m - x); m = x; }
            s += __expf(x - m);
        }
        const float M = block_max(m, red);
        const float S = block_sum(m == -INFINITY ? 0.f : s * __expf(m - M), red);
        if (threadIdx.x == 0) {
            const float lse = M + logf(S);
            if (lse_out) lse_out[row] = lse;
            logp[row] = p[tgt[row]] - lse;
        }
    }
}

// dlogits[r, v] = d logp[r] / d logits[r, v] * g[r] = ([v == tgt[r]] - softmax(r)[v]) * g[r]   (bf16 out)
__global__ __launch_bounds__(NT) void logprob_bwd_kernel(const float* __restrict__ logits, long ld,
                                                         const int64_t* __restrict__ tgt, const float* __restrict__ lse,
                                                         const float* __restrict__ g, bf16_t* __restrict__ dl, long ldd,
                                                         int rows, int vocab) {
    const int v4 = vocab >> 2;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* p = logits + (long)row * ld;
        bf16_t* o = dl + (long)row * ldd;
        const float L = lse[row], gr = -g[row];   // kernel computes (softmax - onehot) * (-g)
        const int t = (int)tgt[row];
        for (int i = blockIdx.x * NT + threadIdx.x; i < v4; i += gridDim.x * NT) {
            const float4 x = *(const float4*)(p + i * 4);
            float d[4] = {__expf(x.x - L), __expf(x.y - L), __expf(x.z - L), __expf(x.w - L)};
            if ((t >> 2) == i) d[t & 3] -= 1.f;
            *(uint2*)(o + i * 4) = make_uint2(pack_bf2(d[0] * gr, d[1] * gr), pack_bf2(d[2] * gr, d[3] * gr));
        }
        if (blockIdx.x == 0)
            for (int i = v4 * 4 + threadIdx.x; i < vocab; i += NT) o[i] = f2bf((__expf(p[i] - L) - (i == t ? 1.f : 0.f)) * gr);
    }
}

// ------------------------------------------------------------------ the same over VOCABULARY CHUNKS
// The lm_head GEMM runs chunk by chunk over the vocabulary (rows of lm_head); a chunk's fp32 logits [rows, cols] update the running
// (max, sum-exp, target logit) of every row and are then dead -- the [rows, vocab] logits tensor (2.5 GB per 7B prompt group, the
// reference's is 4.6 GB bf16 over all positions, TR:357-366) never exists.  One workgroup per row, as logprob_fwd_kernel.
__global__ __launch_bounds__(NT) void lse_chunk_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ tgt,
                                                       int col0, int cols, float* __restrict__ m_run, float* __restrict__ s_run,
                                                       float* __restrict__ t_run, int rows, int first) {
    __shared__ float red[32];
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* p = logits + (long)row * ld;
        float m = -INFINITY, s = 0.f;
        const int v4 = cols >> 2;
        for (int i = threadIdx.x; i < v4; i += NT) {
            const float4 x = *(const float4*)(p + i * 4);
            const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
            if (mx > m) { s *= __expf(m - mx); m = mx; }
            s += __expf(x.x - m) + __expf(x.y - m) + __expf(x.z - m) + __expf(x.w - m);
        }
        for (int i = v4 * 4 + threadIdx.x; i < cols; i += NT) {
            const float x = p[i];
            if (x > m) { s *= __expf(m - x); m = x; }
            s += __expf(x - m);
        }
        const float M = block_max(m, red);
        const float S = block_sum(m == -INFINITY ? 0.f : s * __expf(m - M), red);
        if (threadIdx.x == 0) {
            const float m0 = first ? -INFINITY : m_run[row], s0 = first ? 0.f : s_run[row];
            const float mn = fmaxf(m0, M);
            s_run[row] = (m0 == -INFINITY ? 0.f : s0 * __expf(m0 - mn)) + S * __expf(M - mn);
            m_run[row] = mn;
            const long t = tgt[row] - col0;
            if (t >= 0 && t < cols) t_run[row] = p[t];
        }
    }
}
__global__ __launch_bounds__(NT) void lse_finish_kernel(const float* __restrict__ m_run, const float* __restrict__ s_run,
                                                        const float* __restrict__ t_run, float* __restrict__ logp,
                                                        float* __restrict__ lse_out, int rows) {
    const int r = blockIdx.x * NT + threadIdx.x;
    if (r < rows) {
        const float lse = m_run[r] + logf(s_run[r]);
        if (lse_out) lse_out[r] = lse;
        logp[r] = t_run[r] - lse;
    }
}
// logprob_bwd_kernel on the column chunk [col0, col0 + cols) of the vocabulary: dl [rows, cols] bf16
__global__ __launch_bounds__(NT) void logprob_bwd_chunk_kernel(const float* __restrict__ logits, long ld,
                                                               const int64_t* __restrict__ tgt, int col0, const float* __restrict__ lse,
                                                               const float* __restrict__ g, bf16_t* __restrict__ dl, long ldd,
                                                               int rows, int cols) {
    const int v4 = cols >> 2;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* p = logits + (long)row * ld;
        bf16_t* o = dl + (long)row * ldd;
        const float L = lse[row], gr = -g[row];
        const long tl = tgt[row] - col0;
        const int t = (tl >= 0 && tl < cols) ? (int)tl : -1;
        for (int i = blockIdx.x * NT + threadIdx.x; i < v4; i += gridDim.x * NT) {
            const float4 x = *(const float4*)(p + i * 4);
            float d[4] = {__expf(x.x - L), __expf(x.y - L), __expf(x.z - L), __expf(x.w - L)};
            if (t >= 0 && (t >> 2) == i) d[t & 3] -= 1.f;
            *(uint2*)(o + i * 4) = make_uint2(pack_bf2(d[0] * gr, d[1] * gr), pack_bf2(d[2] * gr, d[3] * gr));
        }
        if (blockIdx.x == 0)
            for (int i = v4 * 4 + threadIdx.x; i < cols; i += NT) o[i] = f2bf((__expf(p[i] - L) - (i == t ? 1.f : 0.f)) * gr);
    }
}

// ------------------------------------------------------------------ packed (EOS-trimmed) positions <-> the [G, C] rectangle
__global__ __launch_bounds__(NT) void gather_f32_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                        float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(NT) void scatter_f32_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                         float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) dst[idx[i]] = src[i];
}

// ------------------------------------------------------------------ first-EOS mask
__global__ __launch_bounds__(NT) void completion_mask_kernel(const int64_t* __restrict__ ids, int eos, int* __restrict__ mask,
                                                             int* __restrict__ lengths, int G, int C) {
    __shared__ int first;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) first = C;
    __syncthreads();
    for (int t = threadIdx.x; t < C; t += NT)
        if (ids[(long)g * C + t] == eos) atomicMin(&first, t);
    __syncthreads();
    const int f = first;
    for (int t = threadIdx.x; t < C; t += NT) mask[(long)g * C + t] = (t <= f) ? 1 : 0;
    if (threadIdx.x == 0 && lengths) lengths[g] = min(C, f + 1);
}

// ------------------------------------------------------------------ GRPO loss + d loss / d logp
// grid = G blocks; each computes its row's masked sums, last step combined with atomics (loss/kl pre-zeroed
// by this launcher through a tiny memset kernel).
__global__ __launch_bounds__(NT) void grpo_loss_kernel(const float* __restrict__ lp, const float* __restrict__ ref,
                                                       const float* __restrict__ adv, const int* __restrict__ mask,
                                                       float beta, float* __restrict__ loss, float* __restrict__ mkl,
                                                       float* __restrict__ dlp, int G, int C) {
    __shared__ float red[32];
    const int g = blockIdx.x;
    const float A = adv[g];
    float sl = 0.f, sk = 0.f, sm = 0.f;
    for (int t = threadIdx.x; t < C; t += NT) {
        const long i = (long)g * C + t;
        const float x = fminf(fmaxf(ref[i] - lp[i], -10.f), 10.f);
        const float kl = __expf(x) - x - 1.f;
        const float mk = (float)mask[i];
        sl += -(A - beta * kl) * mk;     // exp(lp - sg(lp)) == 1 in value
        sk += kl * mk;
        sm += mk;
    }
    const float SL = block_sum(sl, red), SK = block_sum(sk, red), SM = block_sum(sm, red);
    if (threadIdx.x == 0) {
        atomicAdd(loss, SL / SM / G);
        atomicAdd(mkl, SK / SM / G);
    }
    if (dlp) {
        const float inv = 1.f / (SM * G);
        for (int t = threadIdx.x; t < C; t += NT) {
            const long i = (long)g * C + t;
            const float raw = ref[i] - lp[i];
            const float x = fminf(fmaxf(raw, -10.f), 10.f);
            const float inside = (raw >= -10.f && raw <= 10.f) ? 1.f : 0.f;
            // d/dlp [-(ratio*A - beta*kl)] = -A + beta * dkl/dlp ,  dkl/dlp = (e^x - 1) * dx/dlp = (1 - e^x) inside the clamp
            dlp[i] = (-A + beta * (1.f - __expf(x)) * inside) * (float)mask[i] * inv;
        }
    }
}
__global__ void zero2_kernel(float* a, float* b) { if (threadIdx.x == 0) { *a = 0.f; *b = 0.f; } }

// ------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(NT) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ acc) {
    __shared__ float red[32];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        const float4 v = *(const float4*)(g + i * 4);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
    const float S = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, S);
}

// torch.optim.AdamW semantics (decoupled weight decay), fp32 master + bf16 shadow refresh, grad pre-scaled
// by grad_scale (e.g. 1/accumulation or 1/world) and by the global-norm clip coefficient.
__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, bf16_t* __restrict__ sh, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ g, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                   const float* __restrict__ sumsq, float max_norm, float gscale) {
    float clip = gscale;
    if (sumsq && max_norm > 0.f) {
        const float norm = sqrtf(*sumsq) * gscale;
        clip = gscale * fminf(1.f, max_norm / (norm + 1e-6f));
    }
    const float step = lr / bc1, rbc2 = rsqrtf(bc2);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float gi = g[i] * clip;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) * rbc2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (sh) sh[i] = f2bf(pi);
    }
}

}  // namespace

extern "C" int spacer_logprob_fwd(const float* logits, long ld, const int64_t* targets, float* logp, float* lse,
                                  int rows, int vocab, spacer_stream_t stream) {
    SP_REQUIRE(ld % 4 == 0, SPACER_EINVAL, "logprob: ld must be a multiple of 4");
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(logprob_fwd_kernel, dim3(min(rows, 4096)), dim3(NT), 0, (hipStream_t)stream, logits, ld, targets,
                       logp, lse, rows, vocab);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_logprob_bwd(const float* logits, long ld, const int64_t* targets, const float* lse, const float* g,
                                  void* dlogits, long ldd, int rows, int vocab, spacer_stream_t stream) {
    SP_REQUIRE(ld % 4 == 0 && ldd % 4 == 0, SPACER_EINVAL, "logprob: ld must be a multiple of 4");
    if (rows <= 0) return SPACER_OK;
    const int gx = max(1, min(16, cdiv(vocab / 4, NT)));
    hipLaunchKernelGGL(logprob_bwd_kernel, dim3(gx, min(rows, 8192)), dim3(NT), 0, (hipStream_t)stream, logits, ld,
                       targets, lse, g, (bf16_t*)dlogits, ldd, rows, vocab);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_lse_chunk(const float* logits, long ld, const int64_t* targets, int col0, int cols, float* m_run,
                                float* s_run, float* t_run, int rows, int first, spacer_stream_t stream) {
    SP_REQUIRE(ld % 4 == 0 && cols > 0, SPACER_EINVAL, "lse_chunk: ld must be a multiple of 4, cols > 0");
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(lse_chunk_kernel, dim3(min(rows, 4096)), dim3(NT), 0, (hipStream_t)stream, logits, ld, targets, col0, cols,
                       m_run, s_run, t_run, rows, first);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_lse_finish(const float* m_run, const float* s_run, const float* t_run, float* logp, float* lse, int rows,
                                 spacer_stream_t stream) {
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(lse_finish_kernel, dim3(cdiv(rows, NT)), dim3(NT), 0, (hipStream_t)stream, m_run, s_run, t_run, logp, lse, rows);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_logprob_bwd_chunk(const float* logits, long ld, const int64_t* targets, int col0, const float* lse,
                                        const float* g, void* dlogits, long ldd, int rows, int cols, spacer_stream_t stream) {
    SP_REQUIRE(ld % 4 == 0 && ldd % 4 == 0, SPACER_EINVAL, "logprob_bwd_chunk: ld must be a multiple of 4");
    if (rows <= 0 || cols <= 0) return SPACER_OK;
    const int gx = max(1, min(16, cdiv(cols / 4, NT)));
    hipLaunchKernelGGL(logprob_bwd_chunk_kernel, dim3(gx, min(rows, 8192)), dim3(NT), 0, (hipStream_t)stream, logits, ld, targets,
                       col0, lse, g, (bf16_t*)dlogits, ldd, rows, cols);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_completion_mask(const int64_t* ids, int eos_id, int* mask, int* lengths, int G, int C,
                                      spacer_stream_t stream) {
    if (G <= 0 || C <= 0) return SPACER_OK;
    hipLaunchKernelGGL(completion_mask_kernel, dim3(G), dim3(NT), 0, (hipStream_t)stream, ids, eos_id, mask, lengths, G, C);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_gather_f32(const float* src, const int64_t* idx, float* dst, long n, spacer_stream_t stream) {
    if (n <= 0) return SPACER_OK;
    SP_REQUIRE(src && idx && dst, SPACER_EINVAL, "gather_f32: null operand");
    const int grid = (int)((n + NT - 1) / NT < 1024 ? (n + NT - 1) / NT : 1024);
    hipLaunchKernelGGL(gather_f32_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, src, idx, dst, n);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_scatter_f32(const float* src, const int64_t* idx, float* dst, long n, spacer_stream_t stream) {
    if (n <= 0) return SPACER_OK;
    SP_REQUIRE(src && idx && dst, SPACER_EINVAL, "scatter_f32: null operand");
    const int grid = (int)((n + NT - 1) / NT < 1024 ? (n + NT - 1) / NT : 1024);
    hipLaunchKernelGGL(scatter_f32_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, src, idx, dst, n);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_grpo_loss(const float* logp, const float* ref_logp, const float* adv, const int* mask, float beta,
                                float* loss, float* mean_kl, float* dlogp, int G, int C, spacer_stream_t stream) {
    SP_REQUIRE(G > 0 && C > 0, SPACER_EINVAL, "grpo_loss: empty");
    hipLaunchKernelGGL(zero2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, mean_kl);
    hipLaunchKernelGGL(grpo_loss_kernel, dim3(G), dim3(NT), 0, (hipStream_t)stream, logp, ref_logp, adv, mask, beta, loss,
                       mean_kl, dlogp, G, C);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_sumsq_f32(const float* g, long n, float* acc, spacer_stream_t stream) {
    if (n <= 0) return SPACER_OK;
    const int grid = (int)((n / 4 + NT - 1) / NT < 2048 ? (n / 4 + NT - 1) / NT + 1 : 2048);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, g, n, acc);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_adamw_step(float* master, void* shadow_bf16, float* m, float* v, const float* grad, long n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, float bias_c1,
                                 float bias_c2, const float* sumsq_dev, float max_norm, float grad_scale,
                                 spacer_stream_t stream) {
    if (n <= 0) return SPACER_OK;
    const int grid = (int)((n + NT - 1) / NT < 4096 ? (n + NT - 1) / NT : 4096);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, master, (bf16_t*)shadow_bf16, m, v,
                       grad, n, lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2, sumsq_dev, max_norm, grad_scale);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
