// HBM-bound element-wise / gather kernels of the hot path: rotary, SwiGLU, activations, bias-grad,
// casts, embedding gather/scatter, frame patchify, transpose.  All use 8..16-byte accesses per lane.
#include "common.h"

namespace {

constexpr int NT = 256;
inline int grid_for(long work, int cap = 256 * 16) { long b = (work + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

// ------------------------------------------------------------------ rotary (M-RoPE / ViT 2-D RoPE share this)
// x [tokens, heads, D] bf16 (token_stride elems between tokens, heads contiguous), cos/sin fp32 [tokens, D].
// One work item = 8 dims of the first half + the matching 8 of the second half.
__global__ __launch_bounds__(NT) void rope_kernel(bf16_t* __restrict__ x, long token_stride,
                                                  const float* __restrict__ cs, const float* __restrict__ sn,
                                                  int tokens, int heads, int D, int inverse) {
    const int half = D >> 1, per_head = half >> 3;   // 8 (D=128) or 5 (D=80)
    const long total = (long)tokens * heads * per_head;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % per_head);
        const long th = i / per_head;
        const int h = (int)(th % heads);
        const long t = th / heads;
        bf16_t* p = x + t * token_stride + (long)h * D + c * 8;
        const uint4 a = *(const uint4*)p, b = *(const uint4*)(p + half);
        const float* c1 = cs + t * D + c * 8; const float* s1 = sn + t * D + c * 8;
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t oa[4], ob[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x1[2] = {bf_lo(aw[e]), bf_hi(aw[e])}, x2[2] = {bf_lo(bw[e]), bf_hi(bw[e])}, r1[2], r2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int d = e * 2 + u;
                const float ca = c1[d], sa = s1[d], cb = c1[d + half], sb = s1[d + half];
                if (!inverse) {
                    r1[u] = x1[u] * ca - x2[u] * sa;      // out[d]      = x[d] cos[d]   - x[d+h] sin[d]
                    r2[u] = x2[u] * cb + x1[u] * sb;      // out[d+h]    = x[d+h] cos[d+h] + x[d] sin[d+h]
                } else {
                    r1[u] = x1[u] * ca + x2[u] * sb;      // dx[d]       = g[d] cos[d]   + g[d+h] sin[d+h]
                    r2[u] = x2[u] * cb - x1[u] * sa;      // dx[d+h]     = g[d+h] cos[d+h] - g[d] sin[d]
                }
            }
            oa[e] = pack_bf2(r1[0], r1[1]); ob[e] = pack_bf2(r2[0], r2[1]);
        }
        *(uint4*)p = make_uint4(oa[0], oa[1], oa[2], oa[3]);
        *(uint4*)(p + half) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
    }
}

// ------------------------------------------------------------------ SwiGLU
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(NT) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ y,
                                                        int rows, int inter) {
    const int per_row = inter >> 3;
    const long total = (long)rows * per_row;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per_row; const int c = (int)(i % per_row) * 8;
        const uint4 g = *(const uint4*)(gu + r * 2 * inter + c), u = *(const uint4*)(gu + r * 2 * inter + inter + c);
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = bf_lo(gw[e]), g1 = bf_hi(gw[e]);
            o[e] = pack_bf2(g0 * sigm(g0) * bf_lo(uw[e]), g1 * sigm(g1) * bf_hi(uw[e]));
        }
        *(uint4*)(y + r * inter + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(NT) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dy,
                                                        bf16_t* __restrict__ dgu, int rows, int inter) {
    const int per_row = inter >> 3;
    const long total = (long)rows * per_row;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per_row; const int c = (int)(i % per_row) * 8;
        const uint4 g = *(const uint4*)(gu + r * 2 * inter + c), u = *(const uint4*)(gu + r * 2 * inter + inter + c);
        const uint4 d = *(const uint4*)(dy + r * inter + c);
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w}, dw[4] = {d.x, d.y, d.z, d.w};
        uint32_t og[4], ou[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float dg[2], du[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float gv = h ? bf_hi(gw[e]) : bf_lo(gw[e]), uv = h ? bf_hi(uw[e]) : bf_lo(uw[e]);
                const float dv = h ? bf_hi(dw[e]) : bf_lo(dw[e]);
                const float s = sigm(gv);
                dg[h] = dv * uv * s * (1.f + gv * (1.f - s));
                du[h] = dv * gv * s;
            }
            og[e] = pack_bf2(dg[0], dg[1]); ou[e] = pack_bf2(du[0], du[1]);
        }
        *(uint4*)(dgu + r * 2 * inter + c) = make_uint4(og[0], og[1], og[2], og[3]);
        *(uint4*)(dgu + r * 2 * inter + inter + c) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
    }
}

// ------------------------------------------------------------------ activations
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == SPACER_ACT_QUICK_GELU) return v * sigm(1.702f * v);
    if (act == SPACER_ACT_GELU_ERF) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (act == SPACER_ACT_SILU) return v * sigm(v);
    return v;
}
__device__ __forceinline__ float act_df(float v, int act) {
    if (act == SPACER_ACT_QUICK_GELU) { const float s = sigm(1.702f * v); return s * (1.f + 1.702f * v * (1.f - s)); }
    if (act == SPACER_ACT_GELU_ERF)
        return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
    if (act == SPACER_ACT_SILU) { const float s = sigm(v); return s * (1.f + v * (1.f - s)); }
    return 1.f;
}

__global__ __launch_bounds__(NT) void act_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                 bf16_t* __restrict__ out, long n8, int act, int bwd) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n8; i += (long)gridDim.x * NT) {
        const uint4 a = *(const uint4*)(x + i * 8);
        uint4 d = make_uint4(0, 0, 0, 0);
        if (bwd) d = *(const uint4*)(dy + i * 8);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {d.x, d.y, d.z, d.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v0 = bf_lo(aw[e]), v1 = bf_hi(aw[e]);
            o[e] = bwd ? pack_bf2(bf_lo(dw[e]) * act_df(v0, act), bf_hi(dw[e]) * act_df(v1, act))
                       : pack_bf2(act_f(v0, act), act_f(v1, act));
        }
        *(uint4*)(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------ bias grad: db[c] += sum_r dy[r, c]
// Block = 32 column groups (8 bf16 = 16 bytes each: 256 columns) x 8 row lanes over BG_ROWS rows: every thread keeps 16
// independent 16-byte loads in flight (the first form read 4 bytes per thread per row: 23 us for 4160 x 1280), the 8 row
// lanes meet in LDS and the block flushes one atomic per column.
constexpr int BG_ROWS = 128;
__global__ __launch_bounds__(NT) void bias_grad_kernel(const bf16_t* __restrict__ dy, long ld, float* __restrict__ db,
                                                       int rows, int cols) {
    __shared__ float part[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * BG_ROWS, r1 = min(rows, r0 + BG_ROWS);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < cols) {            // cols % 8 == 0 (checked by the launcher)
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 8) {
            const uint4 v = *(const uint4*)(dy + (long)r * ld + c);
            s[0] += bf_lo(v.x); s[1] += bf_hi(v.x); s[2] += bf_lo(v.y); s[3] += bf_hi(v.y);
            s[4] += bf_lo(v.z); s[5] += bf_hi(v.z); s[6] += bf_lo(v.w); s[7] += bf_hi(v.w);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[rl][cg * 8 + e] = s[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
        atomicAdd(db + cc, t);
    }
}

// ------------------------------------------------------------------ casts
__global__ __launch_bounds__(NT) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        const float4 v = *(const float4*)(in + i * 4);
        *(uint2*)(out + i * 4) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = f2bf(in[n4 * 4 + threadIdx.x]);
}
__global__ __launch_bounds__(NT) void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        const uint2 v = *(const uint2*)(in + i * 4);
        *(float4*)(out + i * 4) = make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = bf2f(in[n4 * 4 + threadIdx.x]);
}

// rows x cols fp32 (ld_in) -> bf16 (ld_out); cols % 4 == 0
__global__ __launch_bounds__(NT) void f32_to_bf16_strided_kernel(const float* __restrict__ in, long ld_in,
                                                                 bf16_t* __restrict__ out, long ld_out, int rows, int cols) {
    const int per = cols >> 2;
    const long total = (long)rows * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per; const int c = (int)(i % per) * 4;
        const float4 v = *(const float4*)(in + r * ld_in + c);
        *(uint2*)(out + r * ld_out + c) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
}

// ------------------------------------------------------------------ embedding gather / scatter
__global__ __launch_bounds__(NT) void embed_fwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                       const bf16_t* __restrict__ video, const int* __restrict__ vrow,
                                                       float* __restrict__ out, int T, int H, int* __restrict__ inc0 = nullptr,
                                                       int* __restrict__ inc1 = nullptr) {
    // decode loop: the step's first kernel advances the device-side step counters (nobody reads them during this kernel)
    if (inc0 && blockIdx.x == 0 && threadIdx.x == 0) { *inc0 += 1; if (inc1) *inc1 += 1; }
    const int per = H >> 3;
    const long total = (long)T * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long t = i / per; const int c = (int)(i % per) * 8;
        const int vr = vrow ? vrow[t] : -1;
        const bf16_t* src = vr >= 0 ? video + (long)vr * H + c : table + ids[t] * (long)H + c;
        const uint4 v = *(const uint4*)src;
        float* o = out + t * H + c;
        *(float4*)o = make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
        *(float4*)(o + 4) = make_float4(bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w));
    }
}
__global__ __launch_bounds__(NT) void embed_bwd_kernel(const int64_t* __restrict__ ids, const int* __restrict__ vrow,
                                                       const float* __restrict__ d_out, float* __restrict__ d_table,
                                                       float* __restrict__ d_video, int T, int H) {
    const int per = H >> 2;
    const long total = (long)T * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long t = i / per; const int c = (int)(i % per) * 4;
        const float4 g = *(const float4*)(d_out + t * H + c);
        const int vr = vrow ? vrow[t] : -1;
        if (vr >= 0) {
            if (d_video) *(float4*)(d_video + (long)vr * H + c) = g;
        } else if (d_table) {
            float* p = d_table + ids[t] * (long)H + c;
            atomicAdd(p, g.x); atomicAdd(p + 1, g.y); atomicAdd(p + 2, g.z); atomicAdd(p + 3, g.w);
        }
    }
}

// ------------------------------------------------------------------ frames -> normalised patch rows
__global__ __launch_bounds__(NT) void patchify_kernel(const uint8_t* __restrict__ fr, bf16_t* __restrict__ out, int F,
                                                      int Hpx, int Wpx, int ps, int tp, int mg, int Kpad, int gt,
                                                      int gh, int gw) {
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float istd[3] = {1.f / 0.26862954f, 1.f / 0.26130258f, 1.f / 0.27577711f};
    const int K = 3 * tp * ps * ps;
    const long total = (long)gt * gh * gw * Kpad;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int f = (int)(i % Kpad);
        const long tok = i / Kpad;
        float v = 0.f;
        if (f < K) {
            const int px = f % ps, py = (f / ps) % ps, t2 = (f / (ps * ps)) % tp, c = f / (ps * ps * tp);
            const int in = (int)(tok % (mg * mg));
            const long blk = tok / (mg * mg);
            const int bw = (int)(blk % (gw / mg)), bh = (int)((blk / (gw / mg)) % (gh / mg));
            const int t = (int)(blk / ((long)(gw / mg) * (gh / mg)));
            const int ih = in / mg, iw = in % mg;
            int frame = t * tp + t2; frame = frame < F ? frame : F - 1;
            const int y = (bh * mg + ih) * ps + py, xx = (bw * mg + iw) * ps + px;
            const float p = (float)fr[(((long)frame * 3 + c) * Hpx + y) * Wpx + xx];
            v = (p * (1.f / 255.f) - mean[c]) * istd[c];
        }
        out[i] = f2bf(v);
    }
}

// ------------------------------------------------------------------ row gather / scatter-add (logit positions)
__global__ __launch_bounds__(NT) void gather_rows_kernel(const bf16_t* __restrict__ src, long ld, const int* __restrict__ idx,
                                                         bf16_t* __restrict__ out, int n, int cols) {
    const int per = cols >> 3;
    const long total = (long)n * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per; const int c = (int)(i % per) * 8;
        *(uint4*)(out + r * cols + c) = *(const uint4*)(src + (long)idx[r] * ld + c);
    }
}
__global__ __launch_bounds__(NT) void scatter_add_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx,
                                                              float* __restrict__ dst, long ld, int n, int cols) {
    const int per = cols >> 2;
    const long total = (long)n * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / per; const int c = (int)(i % per) * 4;
        const uint2 v = *(const uint2*)(src + r * cols + c);
        float* p = dst + (long)idx[r] * ld + c;
        atomicAdd(p, bf_lo(v.x)); atomicAdd(p + 1, bf_hi(v.x)); atomicAdd(p + 2, bf_lo(v.y)); atomicAdd(p + 3, bf_hi(v.y));
    }
}

// ------------------------------------------------------------------ transpose with zero padding
// out[c, r] = in[r, c] for r < R, 0 for R <= r < Rpad.  64x64 tiles.  A thread loads the same 8 columns of TWO adjacent
// rows (2 x 16 B), so every transposed LDS store is a full dword {in[r][c], in[r+1][c]} (no sub-dword writes); the LDS
// image is tile[c][32 dwords] with the 16-byte group index XOR-swizzled by c/8, which makes both the dword stores
// (2-way at most) and the 16-byte row reads conflict-free; global stores are 16 B, 128 B contiguous per 8 lanes.
__global__ __launch_bounds__(NT) void transpose_kernel(const bf16_t* __restrict__ in, long ld_in,
                                                       bf16_t* __restrict__ out, long ld_out, int R, int C, int Rpad) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[64 * 32];   // tile[c][row pair], swizzled
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    const bool fast = (ld_in % 8 == 0) && (ld_out % 8 == 0) && (((uintptr_t)in | (uintptr_t)out) % 16 == 0);
    {
        const int p = t >> 3, cc = (t & 7) * 8;                       // row pair (rows 2p, 2p+1), first of 8 columns
        uint32_t w[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + 2 * p + h;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < R) {
                if (fast && c0 + cc + 8 <= C) v = *(const uint4*)(in + (long)r * ld_in + c0 + cc);
                else {
                    bf16_t tmp[8];
                    for (int e = 0; e < 8; ++e) tmp[e] = (c0 + cc + e < C) ? in[(long)r * ld_in + c0 + cc + e] : (bf16_t)0;
                    v = make_uint4(tmp[0] | (uint32_t)tmp[1] << 16, tmp[2] | (uint32_t)tmp[3] << 16, tmp[4] | (uint32_t)tmp[5] << 16,
                                   tmp[6] | (uint32_t)tmp[7] << 16);
                }
            }
            w[h][0] = v.x; w[h][1] = v.y; w[h][2] = v.z; w[h][3] = v.w;
        }
        const int grp = ((p >> 2) ^ (t & 7)) * 4 + (p & 3);            // swizzle key (c >> 3) & 7 == t & 7 for all 8 columns
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t lo = (e & 1) ? (w[0][e >> 1] >> 16) : (w[0][e >> 1] & 0xffffu);
            const uint32_t hi = (e & 1) ? (w[1][e >> 1] & 0xffff0000u) : (w[1][e >> 1] << 16);
            tile[(cc + e) * 32 + grp] = lo | hi;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (t >> 3) + 32 * j, g4 = t & 7, rr = g4 * 8;     // 8 lanes write 128 contiguous bytes of out row c
        if (c0 + c < C && r0 + rr < Rpad) {
            const uint4 v = *(const uint4*)(tile + c * 32 + ((g4 ^ ((c >> 3) & 7)) * 4));
            bf16_t* dst = out + (long)(c0 + c) * ld_out + r0 + rr;
            if (fast && r0 + rr + 8 <= Rpad) *(uint4*)dst = v;
            else {
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                for (int e = 0; e < 8 && r0 + rr + e < Rpad; ++e) dst[e] = (bf16_t)((e & 1) ? (wv[e >> 1] >> 16) : (wv[e >> 1] & 0xffffu));
            }
        }
    }
}

}  // namespace

extern "C" int spacer_rope_inplace(void* x, long token_stride, const float* cos_t, const float* sin_t, int tokens,
                                   int heads, int head_dim, int inverse, spacer_stream_t stream) {
    SP_REQUIRE(head_dim % 16 == 0, SPACER_EINVAL, "rope: head_dim=%d must be a multiple of 16", head_dim);
    SP_REQUIRE(token_stride % 8 == 0, SPACER_EINVAL, "rope: token_stride must be a multiple of 8");
    if (tokens <= 0) return SPACER_OK;
    const long work = (long)tokens * heads * (head_dim / 16);
    hipLaunchKernelGGL(rope_kernel, dim3(grid_for(work)), dim3(NT), 0, (hipStream_t)stream, (bf16_t*)x, token_stride,
                       cos_t, sin_t, tokens, heads, head_dim, inverse);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_swiglu_fwd(const void* gu, void* y, int rows, int inter, spacer_stream_t stream) {
    SP_REQUIRE(inter % 8 == 0, SPACER_EINVAL, "swiglu: inter must be a multiple of 8");
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((long)rows * inter / 8)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)gu, (bf16_t*)y, rows, inter);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_swiglu_bwd(const void* gu, const void* dy, void* dgu, int rows, int inter, spacer_stream_t stream) {
    SP_REQUIRE(inter % 8 == 0, SPACER_EINVAL, "swiglu: inter must be a multiple of 8");
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long)rows * inter / 8)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)gu, (const bf16_t*)dy, (bf16_t*)dgu, rows, inter);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_act_fwd(const void* x, void* y, long n, int act, spacer_stream_t stream) {
    SP_REQUIRE(n % 8 == 0, SPACER_EINVAL, "act: n must be a multiple of 8");
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(n / 8)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, nullptr,
                       (bf16_t*)y, n / 8, act, 0);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_act_bwd(const void* x, const void* dy, void* dx, long n, int act, spacer_stream_t stream) {
    SP_REQUIRE(n % 8 == 0, SPACER_EINVAL, "act: n must be a multiple of 8");
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(n / 8)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, n / 8, act, 1);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_bias_grad(const void* dy, long ld, float* db, int rows, int cols, spacer_stream_t stream) {
    SP_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ((uintptr_t)dy % 16) == 0, SPACER_EINVAL, "bias_grad: cols/ld must be multiples of 8, dy 16-byte aligned");
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(cdiv(cols, 256), cdiv(rows, BG_ROWS)), dim3(NT), 0,
                       (hipStream_t)stream, (const bf16_t*)dy, ld, db, rows, cols);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_cast_f32_to_bf16(const float* in, void* out, long n, spacer_stream_t stream) {
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, in,
                       (bf16_t*)out, n);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_cast_bf16_to_f32(const void* in, float* out, long n, spacer_stream_t stream) {
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)in, out, n);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_cast_f32_to_bf16_strided(const float* in, long ld_in, void* out, long ld_out, int rows, int cols,
                                               spacer_stream_t stream) {
    SP_REQUIRE(cols % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0, SPACER_EINVAL, "cast_strided: cols/ld must be multiples of 4");
    if (rows <= 0) return SPACER_OK;
    hipLaunchKernelGGL(f32_to_bf16_strided_kernel, dim3(grid_for((long)rows * cols / 4)), dim3(NT), 0, (hipStream_t)stream, in,
                       ld_in, (bf16_t*)out, ld_out, rows, cols);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_embed_fwd(const int64_t* ids, const void* table, const void* video, const int* video_row_of_token,
                                float* out, int T, int H, spacer_stream_t stream) {
    SP_REQUIRE(H % 8 == 0, SPACER_EINVAL, "embed: H must be a multiple of 8");
    if (T <= 0) return SPACER_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((long)T * H / 8)), dim3(NT), 0, (hipStream_t)stream, ids,
                       (const bf16_t*)table, (const bf16_t*)video, video_row_of_token, out, T, H, (int*)nullptr, (int*)nullptr);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
// zero fill of a device buffer on the launch stream (dK / dV accumulators of the attention backward, scatter targets, the flat
// gradient): the runtime's fill kernel, so that a training step consists of this library's launches only
extern "C" int spacer_zero(void* p, long bytes, spacer_stream_t stream) {
    if (bytes <= 0) return SPACER_OK;
    const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    SP_REQUIRE(e == hipSuccess, SPACER_ELAUNCH, "zero: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return SPACER_OK;
}
extern "C" int spacer_decode_embed(const int64_t* ids, const void* table, float* out, int B, int H, int* counter0, int* counter1,
                                   spacer_stream_t stream) {
    SP_REQUIRE(H % 8 == 0 && counter0, SPACER_EINVAL, "decode_embed: H must be a multiple of 8, counter0 non-null");
    if (B <= 0) return SPACER_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((long)B * H / 8)), dim3(NT), 0, (hipStream_t)stream, ids,
                       (const bf16_t*)table, (const bf16_t*)nullptr, (const int*)nullptr, out, B, H, counter0, counter1);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_embed_bwd(const int64_t* ids, const int* video_row_of_token, const float* d_out, float* d_table,
                                float* d_video, int T, int H, spacer_stream_t stream) {
    SP_REQUIRE(H % 4 == 0, SPACER_EINVAL, "embed: H must be a multiple of 4");
    if (T <= 0) return SPACER_OK;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((long)T * H / 4)), dim3(NT), 0, (hipStream_t)stream, ids,
                       video_row_of_token, d_out, d_table, d_video, T, H);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_patchify(const uint8_t* frames, void* out, int F, int Hpx, int Wpx, int patch, int tpatch,
                               int merge, int Kpad, spacer_stream_t stream) {
    SP_REQUIRE(F > 0 && Hpx % (patch * merge) == 0 && Wpx % (patch * merge) == 0, SPACER_EINVAL,
               "patchify: frame %dx%d not a multiple of patch*merge", Hpx, Wpx);
    SP_REQUIRE(Kpad >= 3 * tpatch * patch * patch, SPACER_EINVAL, "patchify: Kpad too small");
    const int gt = (F + tpatch - 1) / tpatch, gh = Hpx / patch, gw = Wpx / patch;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((long)gt * gh * gw * Kpad)), dim3(NT), 0, (hipStream_t)stream,
                       frames, (bf16_t*)out, F, Hpx, Wpx, patch, tpatch, merge, Kpad, gt, gh, gw);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_gather_rows_bf16(const void* src, long ld, const int* idx, void* out, int n, int cols,
                                       spacer_stream_t stream) {
    SP_REQUIRE(cols % 8 == 0 && ld % 8 == 0, SPACER_EINVAL, "gather_rows: cols/ld must be multiples of 8");
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * cols / 8)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)src, ld, idx, (bf16_t*)out, n, cols);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_scatter_add_rows_f32(const void* src, const int* idx, float* dst, long ld, int n, int cols,
                                           spacer_stream_t stream) {
    SP_REQUIRE(cols % 4 == 0, SPACER_EINVAL, "scatter_add_rows: cols must be a multiple of 4");
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((long)n * cols / 4)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)src, idx, dst, ld, n, cols);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
extern "C" int spacer_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int R, int C, int Rpad,
                                     spacer_stream_t stream) {
    SP_REQUIRE(R > 0 && C > 0 && Rpad >= R && ld_out >= Rpad && ld_in >= C, SPACER_EINVAL, "transpose: bad shape");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 64), cdiv(Rpad, 64)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C, Rpad);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
