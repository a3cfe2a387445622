// bf16 GEMM for gfx950:  C[M,N] = epilogue( sum_k A(m,k) B(n,k) )
//
// Every nn.Linear of the ViT / LLM the reference trainer drives is y = x W^T with W stored [out,in]: both operands
// contraction-contiguous ("NT", spacer_gemm_bf16_nt).  The backward GEMMs contract over the OTHER dim of the stored arrays
// (dX = dY . W over `out`, dW = dY^T . X over tokens); the 256 tile reads such operands in place (trans_a / trans_b of
// spacer_gemm_bf16: contraction-major LDS images + transposing LDS reads, gemm_halftile.h) -- no transpose pass, no W^T copies.
//
// Design (MI355X_MICROARCH / cdna_hip_programming T1/T2/T3):
//   * two kernels picked by problem size (spacer_gemm_tile):
//       gemm_bf16_nt_kernel       128x128x64, 4 waves as 2x2, wave = 64x64 (4x4 MFMA 16x16x32 frags), 64 KiB LDS,
//                                 2 workgroups / CU; double-buffered K tiles, one barrier per tile
//       gemm_bf16_nt_256h_kernel  256x256x64, 8 waves as 2x4, wave = 128x64 (8x4 frags), 128 KiB LDS, 1 workgroup / CU;
//                                 half-tile DMA pipeline that never drains (gemm_halftile.h)
//   * A/B tiles go HBM -> LDS directly with global_load_lds_dwordx4 (1 KiB per wave-instruction)
//   * LDS image is lane-linear (DMA constraint); the bank-conflict XOR swizzle is applied on the per-lane
//     SOURCE address and again on the ds_read_b128 address (16-byte chunk ^= (row>>1)&7)
//   * MFMA is issued with swapped operands (D^T = B.A^T) so each lane owns 4 consecutive N of one row:
//     8-byte bf16 / 16-byte fp32 epilogue stores
//   * XCD-aware bijective block remap + grouped tile order so neighbouring tiles share L2
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BK = 64;

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
    int full_tiles, splits;     // 256h kernel: tiles owned whole / K splits of each remaining (tail) tile
    float* slabs;               // split-K workspace (caller's): [tail tile][split][256x256] fp32 partial tiles
    // fused SwiGLU (256h kernel only, swiglu_inter = I > 0): B = [gate rows 0..I) | up rows I..2I); tile column tn takes gate
    // rows tn*128.. as its columns 0..127 and up rows I + tn*128.. as its columns 128..255, and the epilogue writes
    // act[m, tn*128 + c] = silu(gate) * up into C (bf16 [M, I]); gate|up themselves go to C2 (bf16 [M, 2I]) when it is given.
    int swiglu_inter;
    void* C2; long ldc2;
    // contraction-major operands with a ragged K: global K tile `k_tail_tile` (the last one) is staged from zero-padded
    // 64-row copies of the operands' last K % 64 rows (same row strides lda / ldb); -1 = none
    int k_tail_tile;
    const bf16_t* A_tail; const bf16_t* B_tail;
    // 256h kernel, persistent form: work items (whole tiles + K ranges of tail tiles) in the launch, walked by min(items, CUs)
    // workgroups; stage_bf16 = bf16 output without a residual (and the SwiGLU form): the epilogue stages bf16 through half of the
    // LDS and the next item's first K tile is requested under it (gemm_halftile.h)
    int total_blocks, stage_bf16;
};

// The pair forms' own arguments (a SECOND kernel argument of gemm_bf16_pair_256h_kernel only: the production kernels' argument block
// is byte for byte round 3's -- growing GemmArgs changed their register allocation, see profiles/r05_gemm_ab.md).
struct PairArgs {
    // K-concatenated pair form (precise mode, NT operands only): C = [A | A2] . [B | B]^T in ONE launch -- global K tiles
    // kt < kt_wrap come from A, the others from A2 at tile kt - kt_wrap, and B's K tile index wraps at kt_wrap; GemmArgs::K = 2 x the
    // contraction length of one pass
    const bf16_t* A2; int kt_wrap;
    // pair forms with a fused producer epilogue (gemm_halftile.h: PAIR_SWIGLU / PAIR_ROPE / PAIR_ACT): C = hi, C2 = lo (bf16, same ld);
    // C3 = bf16 tape output (gate | up, or the pre-activation) or null; rotary tables fp32 [M, 128], heads below rope_heads are rotated
    int pair_mode;
    void* C3; long ldc3;
    const float* rope_cos; const float* rope_sin; int rope_heads;
};

// workgroups of a persistent 256-tile launch: one per CU
static unsigned persistent_grid(long items) {
    static const int cus = [] {
        hipDeviceProp_t p;
        int d = 0;
        return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256;
    }();
    return (unsigned)(items < cus ? items : cus);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case SPACER_ACT_QUICK_GELU: return v / (1.f + __expf(-1.702f * v));
        case SPACER_ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case SPACER_ACT_SILU: return v / (1.f + __expf(-v));
        default: return v;
    }
}

// four values at once, NOT inlined: the 256 tile's staged epilogue calls it from 32 unrolled fragment bodies
__device__ __noinline__ f32x4 apply_act4(f32x4 v, int act) {
    return (f32x4){apply_act(v[0], act), apply_act(v[1], act), apply_act(v[2], act), apply_act(v[3], act)};
}

// Issue the global->LDS DMA of one ROWS x 64 bf16 tile (rows r0.., cols k0..k0+63) into `lds` (ROWS*128 B).
// ROWS/8 wave-instructions of 1 KiB cover the tile; wave w issues instructions w*PER .. w*PER+PER-1.
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, long ld, int r0, int rmax, int k0,
                                           char* lds, int wave, int lane) {
    constexpr int PER = ROWS / 8 / NWAVES;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int inst = wave * PER + i;
        const int row = inst * 8 + (lane >> 3);          // tile row
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

// One accumulator fragment -> memory with the fused epilogue.  Deliberately NOT inlined: the 256x256 tile has 32
// fragments per lane and a fully inlined epilogue exceeds hipcc's unroll budget, which leaves the fragment index
// dynamic and pushes the whole accumulator array to scratch (5x slower).
struct EpiArgs {
    void* C; const bf16_t* bias; const void* resid; long ldc, ldr; int M, N, out_f32, act; float alpha;
};
__device__ __noinline__ void store_frag(EpiArgs g, int m, int n, f32x4 a) {
    if (m >= g.M || n >= g.N) return;
    float v[4] = {a[0], a[1], a[2], a[3]};
    const int nv = min(4, g.N - n);
    const bool n_vec_ok = (g.N % 4) == 0;
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

// Back half of the epilogue for a row-contiguous group of 4 outputs whose bias / activation were already applied:
// + residual, convert, one 16-byte (fp32) or 8-byte (bf16) store.  Used by the LDS-staged epilogue of the 256 tile.
__device__ __forceinline__ void store_row4(const EpiArgs& g, int m, int n, float4 x) {
    float v[4] = {x.x, x.y, x.z, x.w};
    const int nv = min(4, g.N - n);
    if (g.out_f32) {
        float* c = (float*)g.C + (long)m * g.ldc + n;
        const float* r = g.resid ? (const float*)g.resid + (long)m * g.ldr + n : nullptr;
        if (nv == 4 && (g.ldc % 4) == 0 && (!r || (g.ldr % 4) == 0)) {
            float4 o = x;
            if (r) { const float4 rr = *(const float4*)r; o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
            *(float4*)c = o;
        } else {
            for (int e = 0; e < nv; ++e) c[e] = v[e] + (r ? r[e] : 0.f);
        }
    } else {
        bf16_t* c = (bf16_t*)g.C + (long)m * g.ldc + n;
        const bf16_t* r = g.resid ? (const bf16_t*)g.resid + (long)m * g.ldr + n : nullptr;
        if (nv == 4 && (g.ldc % 4) == 0 && (!r || (g.ldr % 4) == 0)) {
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

template <int WAVES_M, int WAVES_N, int FM, int FN>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (WAVES_M * WAVES_N == 4) ? 2 : 1) void gemm_bf16_nt_kernel(GemmArgs g) {
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * FM * 16, BN = WAVES_N * FN * 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> tile mapping: bijective XCD remap, then grouped (GM tile-rows) order ----
    const int nwg = g.tiles_m * g.tiles_n;
    int pid;
    {
        const int b = blockIdx.x, x = b & 7, q = nwg >> 3, r = nwg & 7;
        pid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    constexpr int GM = (BM == 128) ? 8 : 4;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int wm = (wave / WAVES_N) * (FM * 16), wn = (wave % WAVES_N) * (FN * 16);  // wave's sub-tile origin
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = g.K / BK;
    stage_tile<BM, NWAVES>(g.A, g.lda, m0, g.M, 0, smem, wave, lane);
    stage_tile<BN, NWAVES>(g.B, g.ldb, n0, g.N, 0, smem + A_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const char* cur = smem + (t & 1) * STAGE;
        char* nxt = smem + ((t + 1) & 1) * STAGE;
        if (t + 1 < nt) {
            stage_tile<BM, NWAVES>(g.A, g.lda, m0, g.M, (t + 1) * BK, nxt, wave, lane);
            stage_tile<BN, NWAVES>(g.B, g.ldb, n0, g.N, (t + 1) * BK, nxt + A_BYTES, wave, lane);
        }
        // The wave tile is walked in (k-step, 4-row-fragment group) passes of 4 x FN MFMAs; the scheduler is fenced
        // between passes of the big tile, otherwise hipcc hoists every fragment read of the K tile and spills.
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 b[FN];
            const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = lds_frag(cur + A_BYTES, wn + j * 16 + (lane & 15), chunk);
#pragma unroll
            for (int ih = 0; ih < FM; ih += 4) {
                bf16x8 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = lds_frag(cur, wm + (ih + i) * 16 + (lane & 15), chunk);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        // swapped operands: D'[n][m] -> lane holds n = (lane>>4)*4 + r, m = lane&15
                        acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[ih + i][j], 0, 0, 0);
                if (NWAVES == 8) __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue, staged through LDS ([BM rows][BN cols] fp32 = the whole 64 KiB of the 128x128 tile) so that global
    // stores cover whole tile rows (512 B fp32 / 256 B bf16) instead of 32-byte fragment segments.  alpha / bias /
    // activation on the way in, residual + conversion on the way out (one rounding).  16-byte chunk index ^= row & 7.
    const EpiArgs e = {g.C, g.bias, g.resid, g.ldc, g.ldr, g.M, g.N, g.out_f32, g.act, g.alpha};
    static_assert(BM * BN * 4 <= 2 * STAGE, "output tile must fit the operand buffers");
    constexpr int ROWB = BN * 4, CPR = BN / 4;                               // row bytes, 16-byte chunks per row
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int nl = wn + j * 16 + (lane >> 4) * 4;                        // column inside the tile
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (e.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = (n0 + nl + q < e.N) ? bf2f(e.bias[n0 + nl + q]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = wm + i * 16 + (lane & 15);
            float4 v;
            v.x = apply_act(acc[i][j][0] * e.alpha + b4[0], e.act);
            v.y = apply_act(acc[i][j][1] * e.alpha + b4[1], e.act);
            v.z = apply_act(acc[i][j][2] * e.alpha + b4[2], e.act);
            v.w = apply_act(acc[i][j][3] * e.alpha + b4[3], e.act);
            *(float4*)(smem + row * ROWB + (((nl >> 2) ^ (row & 7)) << 4)) = v;
        }
    }
    __syncthreads();
    constexpr int RPI = NWAVES * 64 / CPR;                                   // tile rows covered per iteration
    for (int it = 0; it < BM / RPI; ++it) {
        const int row = it * RPI + tid / CPR, ch = tid % CPR;
        const float4 v = *(const float4*)(smem + row * ROWB + ((ch ^ (row & 7)) << 4));
        const int m = m0 + row, n = n0 + ch * 4;
        if (m < e.M && n < e.N) store_row4(e, m, n, v);
    }
}

#include "gemm_halftile.h"

// dst[64][ld] = rows r0 .. r0+nrows-1 of src (row stride ld, `cols` valid columns), zero-filled below: the K tail tile of a
// contraction-major operand
__global__ __launch_bounds__(256) void gemm_tail_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long ld, int cols,
                                                             int r0, int nrows) {
    const int per = cols >> 3;
    const long total = 64L * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / per), c = (int)(i % per) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nrows) v = *(const uint4*)(src + (long)(r0 + r) * ld + c);
        *(uint4*)(dst + (long)r * ld + c) = v;
    }
}

}  // namespace

// ---- launch planning --------------------------------------------------------------------------------------------
constexpr long WS_SLAB_BYTES = 256L * 256 * 4, WS_MAX_SLABS = 256;
constexpr long WS_TAIL_ELEMS = 384L * 1024;          // lda + ldb of a ragged-K contraction-major launch (64 rows each, bf16)
constexpr long WS_TAIL_BYTES = WS_TAIL_ELEMS * 64 * 2;

// Tail split of the 256-tile kernel: the tiles of the last, partially filled round of the 256 CUs are each cut into
// `splits` K ranges (<= 256 blocks in total, >= 16 K tiles per range: below that the slab round trip costs more than
// the idle CUs it recovers).
static void tail_plan(long tiles, int nt, bool have_ws, int* full, int* splits) {
    *full = (int)tiles; *splits = 1;
    const int rem = (int)(tiles % 256);
    if (!have_ws || rem == 0 || rem > 176) return;
    int s = 256 / rem;
    if (s > 8) s = 8;
    while (s > 1 && nt / s < 16) --s;
    if (s < 2) return;
    *full = (int)tiles - rem; *splits = s;
}

// Tile choice, wave-quantisation aware.  Cost in units of one 256x256 tile on one CU: the 256 tile runs one workgroup
// per CU (256 slots), the 128 tile two (512 slots) and is ~25 % slower per FLOP, i.e. 0.625 per round.
// spacer_plan::gemm_tile = 128 | 256 forces one of them (tests run every shape through both).
static int choose_tile(int M, int N, int K, bool have_ws, const spacer_plan* plan) {
    if (plan && plan->gemm_tile) return plan->gemm_tile >= 256 ? 256 : 128;
    const long tiles256 = (long)cdiv(M, 256) * cdiv(N, 256);
    const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128);
    int full, splits;
    tail_plan(tiles256, K / BK, have_ws, &full, &splits);
    double cost256 = (double)(full / 256);
    if (splits > 1) cost256 += 1.0 / splits + 20.0 / (0.0268 * K);       // + ~20 us of slab traffic, in tile times
    else if (tiles256 % 256) cost256 += 1.0;
    const double cost128 = 0.625 * (double)cdiv(tiles128, 512);
    return cost256 <= cost128 ? 256 : 128;
}

extern "C" long spacer_gemm_workspace_bytes(void) { return WS_MAX_SLABS * WS_SLAB_BYTES + WS_TAIL_BYTES; }

extern "C" int spacer_gemm_tile(int M, int N, int K, int have_workspace, const spacer_plan* plan) {
    SP_PLAN_OR_ZERO(plan);
    return choose_tile(M, N, K, have_workspace != 0, plan);
}

static int launch_gemm(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, bool ta, bool tb,
                       const spacer_gemm_epilogue* epi, spacer_stream_t stream, const void* A2 = nullptr) {
    SP_REQUIRE(A && B && C, SPACER_EINVAL, "gemm: null operand");
    SP_REQUIRE(!A2 || (!ta && !tb && K % BK == 0 && ((uintptr_t)A2 % 16) == 0), SPACER_EINVAL, "gemm_pair: NT operands, K %% %d == 0, aligned A_lo", BK);
    SP_REQUIRE(M > 0 && N > 0 && K > 0, SPACER_EINVAL, "gemm: empty shape M=%d N=%d K=%d", M, N, K);
    SP_REQUIRE(!ta || tb, SPACER_EINVAL, "gemm: trans_a without trans_b is not instantiated (no caller on the hot path)");
    SP_REQUIRE(ta || K % BK == 0, SPACER_EINVAL, "gemm: K=%d must be a multiple of %d (pad the contraction dim)", K, BK);
    SP_REQUIRE(!ta || (M % 8 == 0 && M >= 8), SPACER_EINVAL, "gemm: trans_a needs M %% 8 == 0 (M=%d)", M);
    SP_REQUIRE(!tb || (N % 8 == 0 && N >= 8), SPACER_EINVAL, "gemm: trans_b needs N %% 8 == 0 (N=%d)", N);
    SP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, SPACER_EINVAL, "gemm: lda/ldb must be multiples of 8 elements");
    SP_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, SPACER_EINVAL, "gemm: A/B must be 16-byte aligned");
    GemmArgs g;
    g.swiglu_inter = 0; g.C2 = nullptr; g.ldc2 = 0;
    g.k_tail_tile = -1; g.A_tail = nullptr; g.B_tail = nullptr;
    PairArgs pa = {};
    pa.A2 = (const bf16_t*)A2; pa.kt_wrap = A2 ? K / BK : 0x7fffffff; pa.pair_mode = A2 ? PAIR_PLAIN : PAIR_NONE;
    if (A2) K *= 2;                                   // one launch walks the K tiles of both passes
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
    const bool have_ws = epi && epi->workspace && epi->workspace_bytes >= spacer_gemm_workspace_bytes();
    SP_REQUIRE(!(epi && epi->workspace) || ((uintptr_t)epi->workspace % 16) == 0, SPACER_EINVAL, "gemm: workspace misaligned");
    // the contraction-major operand forms exist on the 256 tile only
    const spacer_plan* plan = epi ? epi->plan : nullptr;
    SP_REQUIRE_PLAN(plan);
    const bool big = tb || choose_tile(M, N, K, have_ws, plan) == 256;
    SP_REQUIRE(!A2 || g.out_f32, SPACER_EINVAL, "gemm_pair: the output is fp32 (out_f32 = 1)");
    SP_REQUIRE(!A2 || big, SPACER_EINVAL, "gemm_pair: M=%d N=%d K=%d does not run on the 256 tile; use two accumulate passes", M, N, K / 2);
    hipStream_t s = (hipStream_t)stream;
    if (ta && K % BK != 0) {
        // ragged contraction length: the last K % 64 rows of both operands go to zero-padded 64-row tail buffers
        SP_REQUIRE(have_ws, SPACER_EINVAL, "gemm: trans_a with K=%d not a multiple of %d needs the workspace", K, BK);
        SP_REQUIRE(lda + ldb <= WS_TAIL_ELEMS && M <= lda && N <= ldb, SPACER_EINVAL, "gemm: lda + ldb = %ld exceeds the K-tail workspace (%ld)",
                   lda + ldb, WS_TAIL_ELEMS);
        bf16_t* ta_buf = (bf16_t*)((char*)epi->workspace + WS_MAX_SLABS * WS_SLAB_BYTES);
        bf16_t* tb_buf = ta_buf + 64 * lda;
        const int r0 = K / BK * BK, nr = K - r0;
        hipLaunchKernelGGL(gemm_tail_rows_kernel, dim3(cdiv(64L * (M / 8), 256)), dim3(256), 0, s, (const bf16_t*)A, ta_buf, lda, M, r0, nr);
        hipLaunchKernelGGL(gemm_tail_rows_kernel, dim3(cdiv(64L * (N / 8), 256)), dim3(256), 0, s, (const bf16_t*)B, tb_buf, ldb, N, r0, nr);
        g.k_tail_tile = K / BK; g.A_tail = ta_buf; g.B_tail = tb_buf;
    }
    if (big) {
        constexpr int LDS = 8 * 128 * BK * 2;
        static const int once = hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                              + hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                              + hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                              + hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                              + hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                              + hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                              + hipFuncSetAttribute((const void*)gemm_bf16_pair_256h_kernel<PAIR_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)once;
        g.tiles_m = cdiv(M, 256); g.tiles_n = cdiv(N, 256);
        const long tiles = (long)g.tiles_m * g.tiles_n;
        const bool nosplit = plan && plan->gemm_no_split;           // tests switch the K-split tail off for bit-exact comparisons
        tail_plan(tiles, cdiv(K, BK), have_ws && !nosplit, &g.full_tiles, &g.splits);
        g.slabs = have_ws ? (float*)epi->workspace : nullptr;
        const long tail_tiles = tiles - g.full_tiles;
        g.total_blocks = (int)(g.full_tiles + tail_tiles * g.splits);
        g.stage_bf16 = (!g.out_f32 && !g.resid) ? 1 : 0;
        const dim3 grid(g.stage_bf16 ? persistent_grid(g.total_blocks) : (unsigned)g.total_blocks);
        if (g.stage_bf16) {
            if (ta) hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, true, true, true>), grid, dim3(512), LDS, s, g);
            else if (tb) hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, false, true, true>), grid, dim3(512), LDS, s, g);
            else hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, false, false, true>), grid, dim3(512), LDS, s, g);
        } else {
            if (ta) hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, true, true, false>), grid, dim3(512), LDS, s, g);
            else if (tb) hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, false, true, false>), grid, dim3(512), LDS, s, g);
            else if (A2) hipLaunchKernelGGL((gemm_bf16_pair_256h_kernel<PAIR_PLAIN>), grid, dim3(512), LDS, s, g, pa);
            else hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, false, false, false>), grid, dim3(512), LDS, s, g);
        }
        if (g.splits > 1) hipLaunchKernelGGL(gemm_tail_reduce_kernel, dim3((unsigned)(tail_tiles * 32)), dim3(512), 0, s, g, pa);
    } else {
        constexpr int LDS = 2 * (128 * BK * 2 + 128 * BK * 2);
        g.tiles_m = cdiv(M, 128); g.tiles_n = cdiv(N, 128);
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<2, 2, 4, 4>), dim3(g.tiles_m * g.tiles_n), dim3(256), LDS, s, g);
    }
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M,
                                   int N, int K, const spacer_gemm_epilogue* epi, spacer_stream_t stream) {
    return launch_gemm(A, lda, B, ldb, C, ldc, M, N, K, false, false, epi, stream);
}

// Precise mode (csrc/precise.hip): C = (A_hi + A_lo) . B^T + ... as ONE launch over the K-concatenated operands [A_hi | A_lo] and
// [B | B] (the weights are exactly bf16, so only the activation is a pair) instead of two accumulate passes: the fp32 output is
// written once (no read-modify-write pass) and the launch's fixed costs are paid once.
extern "C" int spacer_gemm_pair_fused(int M, int N, int K, int have_workspace, const spacer_plan* plan) {
    SP_PLAN_OR_ZERO(plan);
    return K % BK == 0 && choose_tile(M, N, 2 * K, have_workspace != 0, plan) == 256;
}

extern "C" int spacer_gemm_bf16_pair_nt(const void* A_hi, const void* A_lo, long lda, const void* B, long ldb, void* C, long ldc, int M,
                                        int N, int K, const spacer_gemm_epilogue* epi, spacer_stream_t stream) {
    SP_REQUIRE(A_lo, SPACER_EINVAL, "gemm_pair: null A_lo");
    return launch_gemm(A_hi, lda, B, ldb, C, ldc, M, N, K, false, false, epi, stream, A_lo);
}

// ---- pair GEMMs whose epilogue is the producer that used to follow them (round 5).  One launch:
//   kind SPACER_PAIR_SWIGLU: (y_hi, y_lo)[M, inter] = pair(silu(g) * u), [g | u] = (A_hi + A_lo) . W^T + bias, W = [gate rows | up rows];
//                            tape = bf16(g | u) [M, 2 inter] or NULL
//   kind SPACER_PAIR_ROPE:   (y_hi, y_lo)[M, N] = pair(rotary((A_hi + A_lo) . W^T + bias)) on the first rope_heads heads of 128 dims
//   kind SPACER_PAIR_ACT:    (y_hi, y_lo)[M, N] = pair(act((A_hi + A_lo) . W^T + bias)); tape = bf16(pre-activation) or NULL
// Same sums as spacer_gemm_bf16_pair_nt (same K walk), same producer arithmetic as spacer_swiglu_f32_pair / spacer_rope_f32_pair /
// spacer_act_f32_pair on its fp32 output -- which is never written.
static bool pair_epilogue_ok(int kind, int M, int N, int K, int head_dim, bool have_ws, const spacer_plan* plan) {
    if (K % BK != 0 || N % 4 != 0) return false;
    if (kind == SPACER_PAIR_SWIGLU && (N % 256 != 0)) return false;            // N = 2 * inter, inter % 128 == 0
    if (kind == SPACER_PAIR_ROPE && (head_dim != 128 || N % 128 != 0)) return false;
    if (kind < SPACER_PAIR_SWIGLU || kind > SPACER_PAIR_ACT) return false;
    return choose_tile(M, N, 2 * K, have_ws, plan) == 256;
}

extern "C" int spacer_gemm_pair_epilogue_fused(int kind, int M, int N, int K, int head_dim, int have_workspace, const spacer_plan* plan) {
    SP_PLAN_OR_ZERO(plan);
    return pair_epilogue_ok(kind, M, N, K, head_dim, have_workspace != 0, plan) ? 1 : 0;
}

extern "C" int spacer_gemm_bf16_pair_epilogue(int kind, const void* A_hi, const void* A_lo, long lda, const void* W, long ldb,
                                              const void* bias, void* y_hi, void* y_lo, long ld_y, void* tape_bf16, long ld_tape,
                                              const float* rope_cos, const float* rope_sin, int rope_heads, int head_dim, int act,
                                              int M, int N, int K, void* workspace, long workspace_bytes, const spacer_plan* plan,
                                              spacer_stream_t stream) {
    SP_REQUIRE(A_hi && A_lo && W && y_hi && y_lo, SPACER_EINVAL, "gemm_pair_epilogue: null operand");
    SP_REQUIRE_PLAN(plan);
    SP_REQUIRE(M > 0 && N > 0 && K > 0, SPACER_EINVAL, "gemm_pair_epilogue: empty shape M=%d N=%d K=%d", M, N, K);
    const bool have_ws = workspace && workspace_bytes >= spacer_gemm_workspace_bytes();
    SP_REQUIRE(pair_epilogue_ok(kind, M, N, K, head_dim, have_ws, plan), SPACER_EINVAL,
               "gemm_pair_epilogue: kind %d, M=%d N=%d K=%d head_dim=%d does not run fused (spacer_gemm_pair_epilogue_fused == 0): use "
               "spacer_gemm_bf16_pair_nt + the producer kernel", kind, M, N, K, head_dim);
    SP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ld_y % 4 == 0 && (!tape_bf16 || ld_tape % 4 == 0), SPACER_EINVAL, "gemm_pair_epilogue: leading dimensions");
    SP_REQUIRE(((uintptr_t)A_hi % 16) == 0 && ((uintptr_t)A_lo % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)y_hi % 8) == 0
               && ((uintptr_t)y_lo % 8) == 0 && ((uintptr_t)tape_bf16 % 8) == 0 && ((uintptr_t)workspace % 16) == 0, SPACER_EINVAL,
               "gemm_pair_epilogue: misaligned operand");
    SP_REQUIRE(kind != SPACER_PAIR_ROPE || (rope_cos && rope_sin && rope_heads >= 0 && rope_heads * 128 <= N && ((uintptr_t)rope_cos % 16) == 0
               && ((uintptr_t)rope_sin % 16) == 0), SPACER_EINVAL, "gemm_pair_epilogue: rotary tables / rope_heads=%d", rope_heads);
    SP_REQUIRE(kind != SPACER_PAIR_ACT || (act >= SPACER_ACT_NONE && act <= SPACER_ACT_SILU), SPACER_EINVAL, "gemm_pair_epilogue: unknown activation %d", act);
    GemmArgs g;
    PairArgs pa = {};
    g.A = (const bf16_t*)A_hi; pa.A2 = (const bf16_t*)A_lo; g.B = (const bf16_t*)W; g.C = y_hi; g.C2 = y_lo; pa.C3 = tape_bf16;
    g.lda = lda; g.ldb = ldb; g.ldc = ld_y; g.ldc2 = ld_y; pa.ldc3 = ld_tape;
    g.M = M; g.N = N; g.K = 2 * K; pa.kt_wrap = K / BK;
    g.bias = (const bf16_t*)bias; g.resid = nullptr; g.ldr = 0; g.out_f32 = 1; g.act = kind == SPACER_PAIR_ACT ? act : SPACER_ACT_NONE; g.alpha = 1.f;
    g.swiglu_inter = kind == SPACER_PAIR_SWIGLU ? N / 2 : 0;
    g.k_tail_tile = -1; g.A_tail = nullptr; g.B_tail = nullptr;
    pa.pair_mode = kind == SPACER_PAIR_SWIGLU ? PAIR_SWIGLU : kind == SPACER_PAIR_ROPE ? PAIR_ROPE : PAIR_ACT;
    pa.rope_cos = rope_cos; pa.rope_sin = rope_sin; pa.rope_heads = rope_heads;
    g.tiles_m = cdiv(M, 256); g.tiles_n = N / 256 + (kind != SPACER_PAIR_SWIGLU && N % 256 ? 1 : 0);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    tail_plan(tiles, g.K / BK, have_ws && !(plan && plan->gemm_no_split), &g.full_tiles, &g.splits);
    g.slabs = have_ws ? (float*)workspace : nullptr;
    const long tail_tiles = tiles - g.full_tiles;
    g.total_blocks = (int)(g.full_tiles + tail_tiles * g.splits);
    g.stage_bf16 = 0;
    constexpr int LDS = 8 * 128 * BK * 2;
    static const int once = hipFuncSetAttribute((const void*)gemm_bf16_pair_256h_kernel<PAIR_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                          + hipFuncSetAttribute((const void*)gemm_bf16_pair_256h_kernel<PAIR_ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                          + hipFuncSetAttribute((const void*)gemm_bf16_pair_256h_kernel<PAIR_ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)once;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)g.total_blocks);
    if (kind == SPACER_PAIR_SWIGLU) hipLaunchKernelGGL((gemm_bf16_pair_256h_kernel<PAIR_SWIGLU>), grid, dim3(512), LDS, s, g, pa);
    else if (kind == SPACER_PAIR_ROPE) hipLaunchKernelGGL((gemm_bf16_pair_256h_kernel<PAIR_ROPE>), grid, dim3(512), LDS, s, g, pa);
    else hipLaunchKernelGGL((gemm_bf16_pair_256h_kernel<PAIR_ACT>), grid, dim3(512), LDS, s, g, pa);
    if (g.splits > 1) hipLaunchKernelGGL(gemm_tail_reduce_kernel, dim3((unsigned)(tail_tiles * 32)), dim3(512), 0, s, g, pa);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                int trans_a, int trans_b, const spacer_gemm_epilogue* epi, spacer_stream_t stream) {
    return launch_gemm(A, lda, B, ldb, C, ldc, M, N, K, trans_a != 0, trans_b != 0, epi, stream);
}

// act[M, I] (bf16) = silu(A . Wgate^T + bgate) * (A . Wup^T + bup) with W = [gate rows | up rows] ([2I, K]), in ONE launch on the
// 256-tile kernel: the gate and up columns of the same outputs meet in one tile (see GemmArgs::swiglu_inter).  gu (bf16
// [M, 2I], may be NULL) additionally receives the rounded gate|up values the backward pass needs.  Bit-identical to
// spacer_gemm_bf16_nt into gu followed by spacer_swiglu_fwd.  Returns SPACER_EINVAL when the problem would not run on the
// 256 tile (spacer_gemm_swiglu_fused(M, inter, K) == 0): the caller then takes the two-step path.
extern "C" int spacer_gemm_swiglu_fused(int M, int inter, int K, const spacer_plan* plan) {
    SP_PLAN_OR_ZERO(plan);
    return inter > 0 && inter % 128 == 0 && K % BK == 0 && choose_tile(M, 2 * inter, K, false, plan) == 256;
}

extern "C" int spacer_gemm_swiglu_bf16(const void* A, long lda, const void* W, long ldb, const void* bias, void* act, long ld_act,
                                       void* gu, long ld_gu, int M, int inter, int K, spacer_stream_t stream) {
    SP_REQUIRE(A && W && act, SPACER_EINVAL, "gemm_swiglu: null operand");
    SP_REQUIRE(M > 0 && inter > 0 && K > 0, SPACER_EINVAL, "gemm_swiglu: empty shape M=%d I=%d K=%d", M, inter, K);
    SP_REQUIRE(inter > 0 && inter % 128 == 0 && K % BK == 0, SPACER_EINVAL,
               "gemm_swiglu: M=%d I=%d K=%d does not run on the 256 tile (I %% 128, K %% %d); use gemm + swiglu_fwd", M, inter, K, BK);
    SP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ld_act % 4 == 0 && (!gu || ld_gu % 4 == 0), SPACER_EINVAL, "gemm_swiglu: leading dimensions");
    SP_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)act % 8) == 0 && ((uintptr_t)gu % 8) == 0, SPACER_EINVAL,
               "gemm_swiglu: misaligned operand");
    GemmArgs g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)W; g.C = act;
    g.lda = lda; g.ldb = ldb; g.ldc = ld_act; g.M = M; g.N = 2 * inter; g.K = K;
    g.bias = (const bf16_t*)bias; g.resid = nullptr; g.ldr = 0; g.out_f32 = 0; g.act = SPACER_ACT_NONE; g.alpha = 1.f;
    g.swiglu_inter = inter; g.C2 = gu; g.ldc2 = ld_gu;
    g.k_tail_tile = -1; g.A_tail = nullptr; g.B_tail = nullptr;
    g.tiles_m = cdiv(M, 256); g.tiles_n = inter / 128;
    g.full_tiles = g.tiles_m * g.tiles_n; g.splits = 1; g.slabs = nullptr;      // no K-split tail: the reduce kernel has no SwiGLU form
    g.total_blocks = g.full_tiles; g.stage_bf16 = 1;
    constexpr int LDS = 8 * 128 * BK * 2;
    static const int once = hipFuncSetAttribute((const void*)gemm_bf16_nt_256h_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)once;
    hipLaunchKernelGGL((gemm_bf16_nt_256h_kernel<true, false, false, true>), dim3(persistent_grid(g.total_blocks)), dim3(512), LDS, (hipStream_t)stream, g);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
