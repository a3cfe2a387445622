// bf16 GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . B[N,K]^T )   (both operands K-contiguous)
//
// This is the one matmul shape the whole hot path is expressed in (every nn.Linear of the ViT / LLM the
// reference trainer drives is y = x W^T with W stored [out,in]); backward GEMMs reuse it through the
// transpose kernel (dX = dY . (W^T)^T, dW = dY^T . (X^T)^T).
//
// Design (MI355X_MICROARCH / cdna_hip_programming "step-3" structure + T1/T2/T3):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA 16x16x32 frags)
//   * BK = 64; A/B tiles go HBM -> LDS directly with global_load_lds_dwordx4 (1 KiB per wave-instruction),
//     double-buffered: loads of tile t+1 are issued before the MFMAs of tile t, one barrier per tile
//   * LDS image is lane-linear (DMA constraint); the bank-conflict XOR swizzle is applied on the per-lane
//     SOURCE address and again on the ds_read_b128 address (16-byte chunk ^= (row>>1)&7)
//   * MFMA is issued with swapped operands (D^T = B.A^T) so each lane owns 4 consecutive N of one row:
//     8-byte bf16 / 16-byte fp32 epilogue stores
//   * XCD-aware bijective block remap + grouped tile order so neighbouring tiles share L2
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    long lda, ldb, ldc;
    int M, N, K;
    const bf16_t* bias;     // [N] or null
    const void* resid;      // same dtype as C, or null
    long ldr;
    int out_f32;            // 0: C/resid bf16, 1: fp32
    int act;                // spacer_act
    float alpha;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case SPACER_ACT_QUICK_GELU: return v / (1.f + __expf(-1.702f * v));
        case SPACER_ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case SPACER_ACT_SILU: return v / (1.f + __expf(-v));
        default: return v;
    }
}

// Issue the global->LDS DMA of one 128x64 bf16 tile (rows r0.., cols k0..k0+63) into `lds` (16 KiB).
// 16 wave-instructions of 1 KiB cover the tile; wave w issues instructions w*4 .. w*4+3.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, long ld, int r0, int rmax, int k0,
                                           char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        const int row = inst * 8 + (lane >> 3);          // tile row 0..127
        const int pchunk = lane & 7;                     // physical 16-B chunk in the LDS row
        const int chunk = pchunk ^ ((row >> 1) & 7);     // logical chunk this lane must fetch
        int gr = r0 + row;
        gr = gr < rmax ? gr : rmax - 1;
        const bf16_t* src = G + (long)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + inst * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* lds, int row, int chunk) {
    const int p = chunk ^ ((row >> 1) & 7);
    return *(const bf16x8*)(lds + row * 128 + p * 16);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> tile mapping: bijective XCD remap, then grouped (GM tile-rows) order ----
    const int nwg = g.tiles_m * g.tiles_n;
    int pid;
    {
        const int b = blockIdx.x, x = b & 7, q = nwg >> 3, r = nwg & 7;
        pid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;  // wave's 64x64 sub-tile
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = g.K / BK;
    stage_tile(g.A, g.lda, m0, g.M, 0, smem, wave, lane);
    stage_tile(g.B, g.ldb, n0, g.N, 0, smem + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const char* cur = smem + (t & 1) * (2 * TILE_BYTES);
        char* nxt = smem + ((t + 1) & 1) * (2 * TILE_BYTES);
        if (t + 1 < nt) {
            stage_tile(g.A, g.lda, m0, g.M, (t + 1) * BK, nxt, wave, lane);
            stage_tile(g.B, g.ldb, n0, g.N, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[4], b[4];
            const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds_frag(cur, wm + i * 16 + (lane & 15), chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = lds_frag(cur + TILE_BYTES, wn + j * 16 + (lane & 15), chunk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // swapped operands: D'[n][m] -> lane holds n = (lane>>4)*4 + r, m = lane&15
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = m0+wm+i*16+(lane&15), n = n0+wn+j*16+(lane>>4)*4 ----
    const bool n_vec_ok = (g.N % 4) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + i * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + j * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const int nv = min(4, g.N - n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] *= g.alpha;
                if (g.bias && e < nv) v[e] += bf2f(g.bias[n + e]);
                v[e] = apply_act(v[e], g.act);
            }
            if (g.out_f32) {
                float* c = (float*)g.C + (long)m * g.ldc + n;
                const float* r = g.resid ? (const float*)g.resid + (long)m * g.ldr + n : nullptr;
                if (nv == 4 && n_vec_ok && (g.ldc % 4) == 0 && (!r || (g.ldr % 4) == 0)) {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (r) { const float4 rr = *(const float4*)r; o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
                    *(float4*)c = o;
                } else {
                    for (int e = 0; e < nv; ++e) c[e] = v[e] + (r ? r[e] : 0.f);
                }
            } else {
                bf16_t* c = (bf16_t*)g.C + (long)m * g.ldc + n;
                const bf16_t* r = g.resid ? (const bf16_t*)g.resid + (long)m * g.ldr + n : nullptr;
                if (nv == 4 && n_vec_ok && (g.ldc % 4) == 0 && (!r || (g.ldr % 4) == 0)) {
                    if (r) {
                        const uint2 rr = *(const uint2*)r;
                        v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
                    }
                    *(uint2*)c = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                } else {
                    for (int e = 0; e < nv; ++e) c[e] = f2bf(v[e] + (r ? bf2f(r[e]) : 0.f));
                }
            }
        }
    }
}

}  // namespace

extern "C" int spacer_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M,
                                   int N, int K, const spacer_gemm_epilogue* epi, spacer_stream_t stream) {
    SP_REQUIRE(A && B && C, SPACER_EINVAL, "gemm: null operand");
    SP_REQUIRE(M > 0 && N > 0 && K > 0, SPACER_EINVAL, "gemm: empty shape M=%d N=%d K=%d", M, N, K);
    SP_REQUIRE(K % BK == 0, SPACER_EINVAL, "gemm: K=%d must be a multiple of %d (pad the contraction dim)", K, BK);
    SP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, SPACER_EINVAL, "gemm: lda/ldb must be multiples of 8 elements");
    SP_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, SPACER_EINVAL, "gemm: A/B must be 16-byte aligned");
    GemmArgs g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = epi ? (const bf16_t*)epi->bias : nullptr;
    g.resid = epi ? epi->residual : nullptr;
    g.ldr = epi ? epi->ldr : 0;
    g.out_f32 = epi ? epi->out_f32 : 0;
    g.act = epi ? epi->act : SPACER_ACT_NONE;
    g.alpha = epi ? epi->alpha : 1.f;
    if (epi && epi->alpha == 0.f) g.alpha = 1.f;
    const int esz = g.out_f32 ? 4 : 2;
    SP_REQUIRE(((uintptr_t)C % (4 * esz)) == 0, SPACER_EINVAL, "gemm: C misaligned");
    g.tiles_m = cdiv(M, BM); g.tiles_n = cdiv(N, BN);
    const int grid = g.tiles_m * g.tiles_n;
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(grid), dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, g);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
