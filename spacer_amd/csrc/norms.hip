// RMSNorm / LayerNorm forward + backward (HBM-bound; one workgroup walks whole rows with 16-byte accesses).
// Semantics: HF Qwen2VLRMSNorm (fp32 statistics) and nn.LayerNorm(eps=1e-6), the norms the reference's
// model forward (SG_RLVR_trainer.py:357) executes.  x is the (fp32 or bf16) residual stream, y is bf16.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int NT = 256;       // threads per block
constexpr int MAXIT = 8;      // cols <= NT*4*MAXIT = 8192

template <bool XF32>
__device__ __forceinline__ void load4(const void* x, long idx, float v[4]) {
    if (XF32) {
        const float4 t = *(const float4*)((const float*)x + idx);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const uint2 t = *(const uint2*)((const bf16_t*)x + idx);
        v[0] = bf_lo(t.x); v[1] = bf_hi(t.x); v[2] = bf_lo(t.y); v[3] = bf_hi(t.y);
    }
}
__device__ __forceinline__ void loadbf4(const void* x, long idx, float v[4]) { load4<false>(x, idx, v); }

template <bool XF32>
__device__ __forceinline__ void store4(void* x, long idx, const float v[4]) {
    if (XF32) {
        *(float4*)((float*)x + idx) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *(uint2*)((bf16_t*)x + idx) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    }
}

// ------------------------------------------------------------------ forward
template <bool XF32, bool LAYER>
__global__ __launch_bounds__(NT) void norm_fwd_kernel(const void* __restrict__ x, const bf16_t* __restrict__ w,
                                                      const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      int rows, int cols, float eps) {
    __shared__ float red[32];
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const long base = (long)row * cols;
        float xv[MAXIT][4];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
                load4<XF32>(x, base + c, xv[it]);
                s += xv[it][0] + xv[it][1] + xv[it][2] + xv[it][3];
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
        const float rstd = rsqrtf(var + eps);
        if (threadIdx.x == 0) {
            if (rstd_out) rstd_out[row] = rstd;
            if (LAYER && mean_out) mean_out[row] = mu;
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
                store4<false>(y, base + c, o);
            }
        }
    }
}

// ------------------------------------------------------------------ backward
// Each block walks ROWS_PER_BLOCK consecutive rows; per-thread dw/db partials stay in registers and are
// flushed with one fp32 atomicAdd per column per block.
// 16 rows per block: the dw / db flush is cols fp32 atomics PER BLOCK and the L2 atomic units retire only ~37 G of them
// per second (8 rows: +33 us at 5498 x 3584, 4 rows: +84 us), while more rows per block leave CUs without a block.
constexpr int ROWS_PER_BLOCK = 16;
static int rows_per_block() { return ROWS_PER_BLOCK; }

template <bool XF32, bool LAYER>
__global__ __launch_bounds__(NT) void norm_bwd_kernel(const void* __restrict__ x, const bf16_t* __restrict__ w,
                                                      const bf16_t* __restrict__ dy, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, void* __restrict__ dx,
                                                      int dx_acc, float* __restrict__ dw, float* __restrict__ db,
                                                      int rows, int cols, int rpb) {
    __shared__ float red[32];
    float dwp[MAXIT][4], dbp[MAXIT][4], wv[MAXIT][4];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * NT + threadIdx.x) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dwp[it][e] = 0.f; dbp[it][e] = 0.f; wv[it][e] = 0.f; }
        if (c < cols) loadbf4(w, c, wv[it]);
    }
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    // the loads of row r+1 are issued before the block reductions of row r: a row is one dependent chain
    // (load -> reduce -> store), and with one row in flight per block the kernel ran at 1.8 TB/s
    float xn[MAXIT][4], dn[MAXIT][4];
    auto fetch = [&](int row) {
        const long base = (long)row * cols;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
                load4<XF32>(x, base + c, xn[it]);
                loadbf4(dy, base + c, dn[it]);
            }
        }
    };
    if (r0 < r1) fetch(r0);
    for (int row = r0; row < r1; ++row) {
        const long base = (long)row * cols;
        const float rs = rstd[row], mu = LAYER ? mean[row] : 0.f;
        float xh[MAXIT][4], g[MAXIT][4];
        float s1 = 0.f, s2 = 0.f;   // sum(g), sum(g * xhat) with g = dy * w
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xv = xn[it][e], dv = dn[it][e];
                    xh[it][e] = (xv - mu) * rs;
                    g[it][e] = dv * wv[it][e];
                    s1 += g[it][e];
                    s2 += g[it][e] * xh[it][e];
                    dwp[it][e] += dv * xh[it][e];
                    dbp[it][e] += dv;
                }
            }
        }
        if (row + 1 < r1) fetch(row + 1);
        const float m2 = block_sum(s2, red) / cols;
        float m1 = 0.f;
        if (LAYER) m1 = block_sum(s1, red) / cols;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (c < cols) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - m1 - xh[it][e] * m2);
                if (dx_acc) {
                    float old[4];
                    load4<XF32>(dx, base + c, old);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += old[e];
                }
                store4<XF32>(dx, base + c, o);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * NT + threadIdx.x) * 4;
        if (c < cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dw) atomicAdd(dw + c + e, dwp[it][e]);
                if (LAYER && db) atomicAdd(db + c + e, dbp[it][e]);
            }
        }
    }
}

int check_shape(const char* who, int rows, int cols) {
    SP_REQUIRE(rows > 0 && cols > 0, SPACER_EINVAL, "%s: empty shape", who);
    SP_REQUIRE(cols % 4 == 0 && cols <= NT * 4 * MAXIT, SPACER_EINVAL, "%s: cols=%d must be a multiple of 4 and <= %d",
               who, cols, NT * 4 * MAXIT);
    return SPACER_OK;
}

}  // namespace

extern "C" int spacer_rmsnorm_fwd(const void* x, int x_f32, const void* w, void* y, float* rstd, int rows, int cols,
                                  float eps, spacer_stream_t stream) {
    if (int rc = check_shape("rmsnorm_fwd", rows, cols)) return rc;
    const int grid = min(rows, 256 * 8);
    if (x_f32)
        hipLaunchKernelGGL((norm_fwd_kernel<true, false>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, nullptr, (bf16_t*)y, nullptr, rstd, rows, cols, eps);
    else
        hipLaunchKernelGGL((norm_fwd_kernel<false, false>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, nullptr, (bf16_t*)y, nullptr, rstd, rows, cols, eps);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_layernorm_fwd(const void* x, int x_f32, const void* w, const void* b, void* y, float* mean,
                                    float* rstd, int rows, int cols, float eps, spacer_stream_t stream) {
    if (int rc = check_shape("layernorm_fwd", rows, cols)) return rc;
    const int grid = min(rows, 256 * 8);
    if (x_f32)
        hipLaunchKernelGGL((norm_fwd_kernel<true, true>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    else
        hipLaunchKernelGGL((norm_fwd_kernel<false, true>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_rmsnorm_bwd(const void* x, int x_f32, const void* w, const void* dy, const float* rstd, void* dx,
                                  int dx_accumulate, float* dw, int rows, int cols, spacer_stream_t stream) {
    if (int rc = check_shape("rmsnorm_bwd", rows, cols)) return rc;
    const int rpb = rows_per_block(), grid = cdiv(rows, rpb);
    if (x_f32)
        hipLaunchKernelGGL((norm_bwd_kernel<true, false>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)dy, nullptr, rstd, dx, dx_accumulate, dw, nullptr, rows,
                           cols, rpb);
    else
        hipLaunchKernelGGL((norm_bwd_kernel<false, false>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)dy, nullptr, rstd, dx, dx_accumulate, dw, nullptr, rows,
                           cols, rpb);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_layernorm_bwd(const void* x, int x_f32, const void* w, const void* dy, const float* mean,
                                    const float* rstd, void* dx, int dx_accumulate, float* dw, float* db, int rows,
                                    int cols, spacer_stream_t stream) {
    if (int rc = check_shape("layernorm_bwd", rows, cols)) return rc;
    const int rpb = rows_per_block(), grid = cdiv(rows, rpb);
    if (x_f32)
        hipLaunchKernelGGL((norm_bwd_kernel<true, true>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)dy, mean, rstd, dx, dx_accumulate, dw, db, rows, cols, rpb);
    else
        hipLaunchKernelGGL((norm_bwd_kernel<false, true>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)dy, mean, rstd, dx, dx_accumulate, dw, db, rows, cols, rpb);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
