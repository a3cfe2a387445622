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
        // all of the row's loads are issued before the first value is used (round 6): with the use inside each `if (c < cols)` block hipcc
        // emitted load / s_waitcnt vmcnt(0) / add per block -- up to 8 dependent round trips per row; in the decode loop (64 x 3584 rows,
        // one row per workgroup) the kernel is nothing but that latency.  Columns past the row read column 0 again and are zeroed.
        const int nit = (cols + NT * 4 - 1) / (NT * 4);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (it < nit) load4<XF32>(x, base + (c < cols ? c : 0), xv[it]);
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (it < nit && c < cols) s += xv[it][0] + xv[it][1] + xv[it][2] + xv[it][3];
        }
        // (the weight / bias values do not depend on the statistics: requested before the reductions, all at once, for the same reason)
        // (raw bf16 bits: the conversion would be a use, and a use inside the block puts the wait right behind the load again)
        uint2 wraw[MAXIT], braw[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            wraw[it] = make_uint2(0, 0); braw[it] = make_uint2(0, 0);
            if (it < nit) {
                wraw[it] = *(const uint2*)(w + (c < cols ? c : 0));
                if (LAYER) braw[it] = *(const uint2*)(b + (c < cols ? c : 0));
            }
        }
        float mu = 0.f;
        if (LAYER) mu = block_sum(s, red) / cols;
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = (it * NT + threadIdx.x) * 4;
            if (it < nit && c < cols) {
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
            if (it < nit && c < cols) {
                const float wv[4] = {bf_lo(wraw[it].x), bf_hi(wraw[it].x), bf_lo(wraw[it].y), bf_hi(wraw[it].y)};
                const float bv[4] = {bf_lo(braw[it].x), bf_hi(braw[it].x), bf_lo(braw[it].y), bf_hi(braw[it].y)};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (xv[it][e] - mu) * rstd * wv[e] + (LAYER ? bv[e] : 0.f);
                store4<false>(y, base + c, o);
            }
        }
    }
}

// Narrow rows (cols <= 1536: the vision tower's 1280): one wave per row, shuffle reductions, no workgroup barrier -- the
// block-per-row form spends a 5 KiB row mostly in its two block-wide reductions (19.8 us for 4160 x 1280).  Same arithmetic.
template <bool XF32, bool LAYER, int WG>
__global__ __launch_bounds__(NT) void norm_fwd_wave_kernel(const void* __restrict__ x, const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wv[WG][4], bv[WG][4];
#pragma unroll
    for (int it = 0; it < WG; ++it) {
        const int c = (it * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { wv[it][e] = 0.f; bv[it][e] = 0.f; }
        if (c < cols) { loadbf4(w, c, wv[it]); if (LAYER) loadbf4(b, c, bv[it]); }
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const long base = (long)row * cols;
        float xv[WG][4];
        float s = 0.f;
        // (all loads of the row before the first use, as in norm_fwd_kernel: columns past the row re-read column 0 and are not summed)
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
            load4<XF32>(x, base + (c < cols ? c : 0), xv[it]);
        }
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) s += xv[it][0] + xv[it][1] + xv[it][2] + xv[it][3];
        }
        const float mu = LAYER ? wave_sum(s) / cols : 0.f;
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xv[it][e] - mu; ss += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / cols + eps);
        if (lane == 0) {
            if (rstd_out) rstd_out[row] = rstd;
            if (LAYER && mean_out) mean_out[row] = mu;
        }
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (xv[it][e] - mu) * rstd * wv[it][e] + bv[it][e];
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

// IT = 16-byte column groups per thread (cols <= NT * 4 * IT): sized to the row so that the per-thread arrays of a 1280- or
// 3584-wide row do not carry the registers of an 8192-wide one (254 VGPRs, 1-2 waves per SIMD with IT = 8 for every shape).
template <bool XF32, bool LAYER, int IT>
__global__ __launch_bounds__(NT) void norm_bwd_kernel(const void* __restrict__ x, const bf16_t* __restrict__ w,
                                                      const bf16_t* __restrict__ dy, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, void* __restrict__ dx,
                                                      int dx_acc, float* __restrict__ dw, float* __restrict__ db,
                                                      int rows, int cols, int rpb, float* __restrict__ part) {
    __shared__ float red[32];
    float dwp[IT][4], dbp[IT][4], wv[IT][4];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * NT + threadIdx.x) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dwp[it][e] = 0.f; dbp[it][e] = 0.f; wv[it][e] = 0.f; }
        if (c < cols) loadbf4(w, c, wv[it]);
    }
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    // the loads of row r+1 are issued before the block reductions of row r: a row is one dependent chain
    // (load -> reduce -> store), and with one row in flight per block the kernel ran at 1.8 TB/s
    float xn[IT][4], dn[IT][4];
    auto fetch = [&](int row) {
        const long base = (long)row * cols;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
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
        float xh[IT][4], g[IT][4];
        float s1 = 0.f, s2 = 0.f;   // sum(g), sum(g * xhat) with g = dy * w
#pragma unroll
        for (int it = 0; it < IT; ++it) {
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
        for (int it = 0; it < IT; ++it) {
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
    for (int it = 0; it < IT; ++it) {
        const int c = (it * NT + threadIdx.x) * 4;
        if (c < cols) {
            if (part) {          // two-stage form: this block's partial row (dw | db) in the workspace, summed by norm_dw_reduce_kernel
                *(float4*)(part + (long)blockIdx.x * cols + c) = make_float4(dwp[it][0], dwp[it][1], dwp[it][2], dwp[it][3]);
                if (LAYER) *(float4*)(part + ((long)gridDim.x + blockIdx.x) * cols + c) = make_float4(dbp[it][0], dbp[it][1], dbp[it][2], dbp[it][3]);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dw) atomicAdd(dw + c + e, dwp[it][e]);
                if (LAYER && db) atomicAdd(db + c + e, dbp[it][e]);
            }
        }
    }
}

// Narrow rows (cols <= 64 * 4 * WG = 1536: the vision tower's 1280): one WAVE per row instead of one workgroup.  A 1280-wide row
// is 5 KiB -- a 256-thread block spends its time in the two block-wide reductions (4 barriers per row, rows strictly one after
// the other: 80-128 us for 4160 x 1280); here the reductions are wave shuffles, the 4 waves of a block walk 4 rows at once
// (next row's loads in flight), and the per-wave dw / db partials meet in LDS before the one atomic per column per block.
constexpr int WROWS = 16;     // rows per block (4 per wave); with the two-stage dw reduction more, smaller blocks win (32 is better under atomics)
template <bool XF32, bool LAYER, int WG>
__global__ __launch_bounds__(NT) void norm_bwd_wave_kernel(const void* __restrict__ x, const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ dy, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, void* __restrict__ dx, int dx_acc,
                                                           float* __restrict__ dw, float* __restrict__ db, int rows, int cols, float* __restrict__ gpart) {
    __shared__ float part[2][3][64 * 4 * WG];           // waves 1..3 park their dw (and db) partials here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dwp[WG][4], dbp[WG][4], wv[WG][4];
#pragma unroll
    for (int it = 0; it < WG; ++it) {
        const int c = (it * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dwp[it][e] = 0.f; dbp[it][e] = 0.f; wv[it][e] = 0.f; }
        if (c < cols) loadbf4(w, c, wv[it]);
    }
    const int r0 = blockIdx.x * WROWS, r1 = min(rows, r0 + WROWS);
    float xn[WG][4], dn[WG][4];
    auto fetch = [&](int row) {
        const long base = (long)row * cols;
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) { load4<XF32>(x, base + c, xn[it]); loadbf4(dy, base + c, dn[it]); }
        }
    };
    if (r0 + wave < r1) fetch(r0 + wave);
    for (int row = r0 + wave; row < r1; row += 4) {
        const long base = (long)row * cols;
        const float rs = rstd[row], mu = LAYER ? mean[row] : 0.f;
        float xh[WG][4], g[WG][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
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
        if (row + 4 < r1) fetch(row + 4);
        const float m2 = wave_sum(s2) / cols;
        const float m1 = LAYER ? wave_sum(s1) / cols : 0.f;
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
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
    // ---- the block's 4 partials -> wave 0 -> one atomic per column
    if (wave > 0) {
#pragma unroll
        for (int it = 0; it < WG; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part[0][wave - 1][(it * 64 + lane) * 4 + e] = dwp[it][e];
                if (LAYER) part[1][wave - 1][(it * 64 + lane) * 4 + e] = dbp[it][e];
            }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int it = 0; it < WG; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
                float a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = dwp[it][e]; b[e] = dbp[it][e];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { a[e] += part[0][k][c + e]; if (LAYER) b[e] += part[1][k][c + e]; }
                }
                if (gpart) {
                    *(float4*)(gpart + (long)blockIdx.x * cols + c) = make_float4(a[0], a[1], a[2], a[3]);
                    if (LAYER) *(float4*)(gpart + ((long)gridDim.x + blockIdx.x) * cols + c) = make_float4(b[0], b[1], b[2], b[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (dw) atomicAdd(dw + c + e, a[e]);
                        if (LAYER && db) atomicAdd(db + c + e, b[e]);
                    }
                }
            }
        }
    }
}

// second stage of the two-stage dw / db reduction: dw[c] += sum over the blocks' partial rows (coalesced across c)
__global__ __launch_bounds__(NT) void norm_dw_reduce_kernel(const float* __restrict__ part, int nblocks, int cols, float* __restrict__ dw,
                                                            float* __restrict__ db) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= cols) return;
    const float* p = part + (blockIdx.y ? (long)nblocks * cols : 0) + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 4 <= nblocks; b += 4) {
        s0 += p[(long)b * cols]; s1 += p[(long)(b + 1) * cols]; s2 += p[(long)(b + 2) * cols]; s3 += p[(long)(b + 3) * cols];
    }
    for (; b < nblocks; ++b) s0 += p[(long)b * cols];
    float* out = blockIdx.y ? db : dw;
    if (out) out[c] += (s0 + s1) + (s2 + s3);
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
    if (cols <= 64 * 4 * 6 && rows >= 256) {         // narrow rows, enough of them to fill the chip with one wave per row
        const int wgrid = min(cdiv(rows, 4), 256 * 8);
        if (x_f32) hipLaunchKernelGGL((norm_fwd_wave_kernel<true, false, 6>), dim3(wgrid), dim3(NT), 0, (hipStream_t)stream, x,
                                      (const bf16_t*)w, nullptr, (bf16_t*)y, nullptr, rstd, rows, cols, eps);
        else hipLaunchKernelGGL((norm_fwd_wave_kernel<false, false, 6>), dim3(wgrid), dim3(NT), 0, (hipStream_t)stream, x,
                                (const bf16_t*)w, nullptr, (bf16_t*)y, nullptr, rstd, rows, cols, eps);
    } else if (x_f32)
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
    if (cols <= 64 * 4 * 6 && rows >= 256) {
        const int wgrid = min(cdiv(rows, 4), 256 * 8);
        if (x_f32) hipLaunchKernelGGL((norm_fwd_wave_kernel<true, true, 6>), dim3(wgrid), dim3(NT), 0, (hipStream_t)stream, x,
                                      (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
        else hipLaunchKernelGGL((norm_fwd_wave_kernel<false, true, 6>), dim3(wgrid), dim3(NT), 0, (hipStream_t)stream, x,
                                (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    } else if (x_f32)
        hipLaunchKernelGGL((norm_fwd_kernel<true, true>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    else
        hipLaunchKernelGGL((norm_fwd_kernel<false, true>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x,
                           (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

// Shared launcher of the backward kernels.  workspace (optional, fp32 scratch of >= blocks * cols * (LAYER ? 2 : 1) floats): the
// blocks leave their dw / db partial rows there and a second small kernel sums them into dw / db -- no atomics (their flush
// costs ~33 us of a 5498 x 3584 launch: 1.2 M fp32 atomics at the ~37 G/s the L2 atomic units retire).  NULL: atomics.
template <bool LAYER>
static int launch_norm_bwd(const char* who, const void* x, int x_f32, const void* w, const void* dy, const float* mean, const float* rstd,
                           void* dx, int dx_accumulate, float* dw, float* db, int rows, int cols, void* workspace, long workspace_bytes,
                           spacer_stream_t stream) {
    if (int rc = check_shape(who, rows, cols)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool narrow = cols <= 64 * 4 * 6;
    const int rpb = rows_per_block(), grid = narrow ? cdiv(rows, WROWS) : cdiv(rows, rpb);
    const long need = (long)grid * cols * (LAYER ? 2 : 1) * (long)sizeof(float);
    float* part = (workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16) == 0 && (dw || db)) ? (float*)workspace : nullptr;
#define BWD(F32, IT)                                                                                                     \
    hipLaunchKernelGGL((norm_bwd_kernel<F32, LAYER, IT>), dim3(grid), dim3(NT), 0, s, x, (const bf16_t*)w, (const bf16_t*)dy, mean, \
                       rstd, dx, dx_accumulate, dw, db, rows, cols, rpb, part)
    const int it = cols <= NT * 4 * 2 ? 2 : cols <= NT * 4 * 4 ? 4 : MAXIT;
    if (narrow) {
        if (x_f32) hipLaunchKernelGGL((norm_bwd_wave_kernel<true, LAYER, 6>), dim3(grid), dim3(NT), 0, s, x, (const bf16_t*)w,
                                      (const bf16_t*)dy, mean, rstd, dx, dx_accumulate, dw, db, rows, cols, part);
        else hipLaunchKernelGGL((norm_bwd_wave_kernel<false, LAYER, 6>), dim3(grid), dim3(NT), 0, s, x, (const bf16_t*)w,
                                (const bf16_t*)dy, mean, rstd, dx, dx_accumulate, dw, db, rows, cols, part);
    } else if (x_f32) { if (it == 2) BWD(true, 2); else if (it == 4) BWD(true, 4); else BWD(true, MAXIT); }
    else { if (it == 2) BWD(false, 2); else if (it == 4) BWD(false, 4); else BWD(false, MAXIT); }
#undef BWD
    if (part) hipLaunchKernelGGL(norm_dw_reduce_kernel, dim3(cdiv(cols, NT), LAYER ? 2 : 1), dim3(NT), 0, s, part, grid, cols, dw, db);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_rmsnorm_bwd(const void* x, int x_f32, const void* w, const void* dy, const float* rstd, void* dx,
                                  int dx_accumulate, float* dw, int rows, int cols, spacer_stream_t stream) {
    return launch_norm_bwd<false>("rmsnorm_bwd", x, x_f32, w, dy, nullptr, rstd, dx, dx_accumulate, dw, nullptr, rows, cols, nullptr, 0, stream);
}
extern "C" int spacer_rmsnorm_bwd_ws(const void* x, int x_f32, const void* w, const void* dy, const float* rstd, void* dx,
                                     int dx_accumulate, float* dw, int rows, int cols, void* workspace, long workspace_bytes,
                                     spacer_stream_t stream) {
    return launch_norm_bwd<false>("rmsnorm_bwd", x, x_f32, w, dy, nullptr, rstd, dx, dx_accumulate, dw, nullptr, rows, cols, workspace,
                                  workspace_bytes, stream);
}

extern "C" int spacer_layernorm_bwd(const void* x, int x_f32, const void* w, const void* dy, const float* mean,
                                    const float* rstd, void* dx, int dx_accumulate, float* dw, float* db, int rows,
                                    int cols, spacer_stream_t stream) {
    return launch_norm_bwd<true>("layernorm_bwd", x, x_f32, w, dy, mean, rstd, dx, dx_accumulate, dw, db, rows, cols, nullptr, 0, stream);
}
extern "C" int spacer_layernorm_bwd_ws(const void* x, int x_f32, const void* w, const void* dy, const float* mean,
                                       const float* rstd, void* dx, int dx_accumulate, float* dw, float* db, int rows,
                                       int cols, void* workspace, long workspace_bytes, spacer_stream_t stream) {
    return launch_norm_bwd<true>("layernorm_bwd", x, x_f32, w, dy, mean, rstd, dx, dx_accumulate, dw, db, rows, cols, workspace,
                                 workspace_bytes, stream);
}
