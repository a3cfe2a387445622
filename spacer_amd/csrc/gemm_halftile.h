// 256x256x64 bf16 NT GEMM tile, 8 waves (2x4, wave = 128x64), HALF-TILE PIPELINE.
// Included by gemm.hip inside its anonymous namespace (uses GemmArgs / EpiArgs / store_frag / BK).
//
// Why a second schedule: the simple tile loop (gemm_bf16_nt_kernel) drains its global->LDS DMA queue (vmcnt(0)) at the
// barrier that ends every K tile, so each tile pays the tail of its own prefetch.  Here the DMA queue never drains:
//
//   * LDS holds 2 parities x 4 half-tile images {A-lo, A-hi, B-lo, B-hi} of 16 KiB (128 rows x 64 k).  "A half h" is
//     the 64-row quadrant h of BOTH wave rows, "B half h" the 32-column quadrant h of all four wave columns, so one
//     C quadrant (mq, nq) of every wave needs exactly A[mq] and B[nq].
//   * a K tile is consumed in 4 phases = the 4 C quadrants in the order (0,0) (0,1) (1,1) (1,0); the A fragments stay in
//     registers for two phases, the B-lo fragments for the whole tile, so the images are read at phases
//         A-lo: 0    B-lo: 0    B-hi: 1    A-hi: 2    (phase 3 reads nothing)
//   * every phase stages ONE half-tile (2 x 1 KiB DMA per wave) of a later tile into an image last read >= 2 phases ago:
//         phase 0: B-hi(t+1)   phase 1: A-hi(t+1)   phase 2: A-lo(t+2)   phase 3: B-lo(t+2)
//     and then waits with a COUNTED vmcnt(8): 4 half-tiles (64 KiB per CU) stay in flight, each has 4 phases
//     (~2000 cycles) to land, and the half-tile a phase waits for is read one phase later (the barrier between
//     publishes it to the other waves).
//   * a phase is [ds_read fragments | issue DMA | vmcnt(8)] s_barrier [lgkmcnt(0) | 16 MFMA] s_barrier.  The two wave
//     rows run one barrier apart (row 1 takes one extra barrier up front, row 0 one at the end): the two waves sharing a
//     SIMD alternate between the LDS segment and the MFMA segment, so the matrix pipe always has a wave feeding it.
//   * K tiles past the end are staged from the last real tile (redundant, never read) so the in-flight count is the
//     same in every phase; the queue is drained once, before the epilogue.
//
// Operand layouts (TA / TB).  C[m][n] = sum_k A(m,k) B(n,k).  The default (false) is "contraction-contiguous":
// A(m,k) = A[m*lda + k].  TA = true reads A stored CONTRACTION-MAJOR, A(m,k) = A[k*lda + m] (an [K, M] row-major array),
// likewise TB; this is what the backward GEMMs need without a transpose pass:
//     dX[T, in]  = dY[T, out] . W[out, in]           A = dY (plain),   B(n = in, k = out) = W[k][n]     -> TB
//     dW[out,in] = dY[T, out]^T . X[T, in]           A(m = out, k = t) = dY[k][m], B(n = in, k = t) = X[k][n] -> TA, TB
// A contraction-major half-tile image is [64 k-rows][128 columns] (256-byte rows, same 16 KiB): the DMA writes 4 k-rows
// per 1 KiB piece, and the MFMA operand fragments are gathered with gfx950's transposing LDS read (ds_read_b64_tr_b16: a
// 16-lane group fetches a [4 k-rows][16 columns] block, lane i receives column i of the 4 rows; two reads fill the 8
// contraction slots k = kk*32 + (lane>>4)*8 + 0..7 of a 16x16x32 operand, the same slot order as the plain image, so the two
// layouts mix freely).  Bank swizzle of that image: the 32-byte column pair index ^= (row & 3) | ((row >> 3) & 1) << 2 -- the 8
// rows a 32-lane group touches in one read ({r..r+3} and {r+8..r+11}) land on 8 distinct 32-byte bank groups.
// A ragged contraction length (dW: K = tokens) costs nothing in the loop: the host copies the last K % 64 rows of both operands
// into zero-padded 64-row tail buffers (caller's workspace) and the kernel stages that one K tile from them (a scalar
// base-pointer select per DMA).
//
// Persistent form (round 2).  A launch whose work items outnumber the CUs runs min(items, CUs) workgroups, each walking the items
// b, b + G, b + 2G, ... (item = a whole tile, or one K range of a tail tile).  When the output is bf16 without a residual (or the
// SwiGLU form) the epilogue stages through the PARITY-1 half of the LDS only, as bf16 ([128 rows][512 B] per pass, 8-byte chunk
// index ^= row & 15), and the first K tile of the workgroup's NEXT item is requested into the parity-0 half before the epilogue
// starts: the ~8 us pipeline fill of every tile rides under the store phase of the tile before it.  (vmcnt also counts the
// epilogue's stores; they are older than every DMA that a later counted wait is meant for, and returns are in order, so the
// counted waits of the main loop stay conservative.)
#include <type_traits>

// pid (position in the launch's tile order) -> tile coordinates: groups of 4 tile-rows walked column by column, so the
// tiles that run together share A rows and B columns in L2.
__device__ __forceinline__ void tile_of_256(int pid, const GemmArgs& g, int& tm, int& tn) {
    constexpr int GM = 4;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    tm = first_m + (pid % per_group) % gsz;
    tn = (pid % per_group) / gsz;
}

typedef __attribute__((ext_vector_type(4))) short gemm_s16x4;
typedef __attribute__((ext_vector_type(8))) short gemm_s16x8;

// 4 x 4 transpose of one dword per lane across the four 16-lane rows of a wave: in: x[j] = fragment j's dword of lane-row g; out:
// lane-row r holds x[k] = fragment r's dword of (former) lane-row k.  With the accumulator layout of this kernel (fragment j = 16
// columns, lane-row g = columns 4g .. 4g+3 of it) a lane then owns 16 CONSECUTIVE columns of its row, so the epilogue can write
// 32 (bf16) / 64 (fp32) contiguous bytes per lane -- 128 / 256 B per row and instruction -- straight from registers, without the LDS
// staging pass (v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap
// rows {2,3} with rows {0,1}).
__device__ __forceinline__ void xpose4_rows(uint32_t (&x)[4]) {
    const auto a = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false);
    const auto c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    x[0] = c[0]; x[2] = c[1]; x[1] = d[0]; x[3] = d[1];
}

// STG16: bf16 staging epilogue + early prologue (bf16 output without a residual, SwiGLU); else the fp32 staging of round 1
// PAIR (round 5; NT operands, fp32 staging only): the K-concatenated pair form of the precise mode as its OWN instantiations, so that
// the scalar selects of the pair walk (kt_wrap / A2) are compiled out of every other launch.  Modes:
//     PAIR_PLAIN   C fp32 = [A | A2] . [B | B]^T (+ bias, + residual)                      o / down / proj / fc2 / lm_head chunks
//     PAIR_SWIGLU  (hi, lo) bf16 pair of silu(gate) * up straight from the fp32 staging rows (B rows mapped as in the bf16 SwiGLU
//                  form: tile columns 0..127 = gate, 128..255 = up of the same 128 outputs); C3 = bf16(gate | up) for the tape
//     PAIR_ROPE    (hi, lo) pair of the rotary-embedded q | k | v row (head_dim 128: a 256-column tile = two whole heads, the partner
//                  of column d is d +- 64 in the same staged row); heads >= rope_heads (v) are split as they are
//     PAIR_ACT     (hi, lo) pair of act(x) (+ C3 = bf16(x), what act_bwd differentiates at)
// The fp32 [T, N] tensor that the unfused path writes and a separate producer kernel re-reads (1.67 GB per layer for gate|up at two
// cfg3 groups) never exists.
enum { PAIR_NONE = 0, PAIR_PLAIN = 1, PAIR_SWIGLU = 2, PAIR_ROPE = 3, PAIR_ACT = 4 };

// precise-mode scalar helpers of the pair epilogues (the same expressions as csrc/precise.hip's producer kernels)
__device__ __forceinline__ float pair_sigm(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float pair_act(float v, int act) {
    if (act == SPACER_ACT_QUICK_GELU) return v * pair_sigm(1.702f * v);
    if (act == SPACER_ACT_GELU_ERF) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (act == SPACER_ACT_SILU) return v * pair_sigm(v);
    return v;
}
// 4 fp32 values -> 4 bf16 hi + 4 bf16 lo (lo = bf16(x - hi); x - hi is exact in fp32), 8-byte stores
__device__ __forceinline__ void pair_store4(bf16_t* __restrict__ yh, bf16_t* __restrict__ yl, long idx, const float v[4]) {
    const uint32_t h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]);
    const uint32_t l0 = pack_bf2(v[0] - bf_lo(h0), v[1] - bf_hi(h0)), l1 = pack_bf2(v[2] - bf_lo(h1), v[3] - bf_hi(h1));
    *(uint2*)(yh + idx) = make_uint2(h0, h1);
    *(uint2*)(yl + idx) = make_uint2(l0, l1);
}

// The kernels rocprof names.  gemm_bf16_nt_256h_kernel<BALANCED, TA, TB, STG16>: every production form (no pair selects compiled in);
// gemm_bf16_pair_256h_kernel<MODE>: the K-concatenated pair forms of the precise mode (MODE = PAIR_PLAIN .. PAIR_ACT).
template <bool BALANCED, bool TA = false, bool TB = false, bool STG16 = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_nt_256h_kernel(GemmArgs g) {
    constexpr int PAIR = PAIR_NONE;
    const PairArgs pa = {};                        // never read: every use sits behind `PAIR ?` / `if constexpr (PAIR == ...)`
#include "gemm_halftile_body.inc"
}

template <int PAIR>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pair_256h_kernel(GemmArgs g, PairArgs pa) {
    static_assert(PAIR >= PAIR_PLAIN && PAIR <= PAIR_ACT, "pair mode");
    constexpr bool BALANCED = true, TA = false, TB = false, STG16 = false;     // NT launches with the fp32 staging epilogue
#include "gemm_halftile_body.inc"
}

// Sums the K-split partial tiles of the tail (written by gemm_bf16_nt_256h_kernel) in split order -- deterministic --
// and applies the epilogue.  One block per (tail tile, accumulator fragment index): the whole chip takes part, a
// single CU could pull its tile's slabs only at ~25 GB/s.
__global__ __launch_bounds__(512) void gemm_tail_reduce_kernel(GemmArgs g, PairArgs pa) {
    constexpr int SLAB4 = 256 * 256 / 4;
    const int lt = blockIdx.x >> 5, f = blockIdx.x & 31, i = f >> 2, j = f & 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    int tm, tn;
    tile_of_256(g.full_tiles + lt, g, tm, tn);
    const float4* p = (const float4*)g.slabs + (size_t)lt * g.splits * SLAB4 + f * 512 + tid;
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the slabs of up to 8 splits are requested together, then summed in split order (round 6: as a rolled loop every split was its own
    // dependent round trip -- load, s_waitcnt vmcnt(0), add -- and the kernel took 17 us for a few MB of slabs); same order of additions
    for (int s0 = 0; s0 < g.splits; s0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (s0 + u < g.splits) ? p[(size_t)(s0 + u) * SLAB4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < g.splits) { sum[0] += v[u].x; sum[1] += v[u].y; sum[2] += v[u].z; sum[3] += v[u].w; }
    }
    const EpiArgs e = {g.C, g.bias, g.resid, g.ldc, g.ldr, g.M, g.N, g.out_f32, g.act, g.alpha};
    const int m = tm * 256 + wr * 128 + i * 16 + (lane & 15), nl = wc * 64 + j * 16 + (lane >> 4) * 4;      // row, column inside the tile
    if (pa.pair_mode <= PAIR_PLAIN) {
        store_frag(e, m, tn * 256 + nl, sum);
        return;
    }
    // ---- pair epilogues on a tail tile (the same arithmetic as the staged epilogue in gemm_halftile_body.inc): the partner value of a
    // SwiGLU output (gate <-> up: tile column +- 128) or of a rotary pair (d <-> d +- 64) sits in the same lane of another wave
    const int sw = g.swiglu_inter;
    float v[4] = {sum[0], sum[1], sum[2], sum[3]};
    if (g.bias) {
        const int nb = sw ? (nl < 128 ? 0 : sw - 128) + tn * 128 + nl : tn * 256 + nl;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += (nb + q < g.N) ? bf2f(g.bias[nb + q]) : 0.f;
    }
    __shared__ float4 ex[512];
    ex[tid] = make_float4(v[0], v[1], v[2], v[3]);
    __syncthreads();
    bf16_t* yh = (bf16_t*)g.C; bf16_t* yl = (bf16_t*)g.C2;
    if (pa.pair_mode == PAIR_SWIGLU) {
        const float4 pt = ex[tid ^ 128];
        if (m >= g.M) return;
        const long n = (long)tn * 128 + (nl & 127);
        if (nl < 128) {
            const float u[4] = {pt.x, pt.y, pt.z, pt.w};
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = v[q] * pair_sigm(v[q]) * u[q];
            pair_store4(yh, yl, (long)m * g.ldc + n, o);
        }
        if (pa.C3) *(uint2*)((bf16_t*)pa.C3 + (long)m * pa.ldc3 + (nl < 128 ? 0 : sw) + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    } else if (pa.pair_mode == PAIR_ROPE) {
        const float4 pt = ex[tid ^ 64];
        const int n = tn * 256 + nl;
        if (m >= g.M || n >= g.N) return;
        float r[4] = {v[0], v[1], v[2], v[3]};
        if ((n >> 7) < pa.rope_heads) {
            const int d = nl & 127;
            const float4 c4 = *(const float4*)(pa.rope_cos + (long)m * 128 + d), s4 = *(const float4*)(pa.rope_sin + (long)m * 128 + d);
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, p[4] = {pt.x, pt.y, pt.z, pt.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = d < 64 ? v[q] * cc[q] - p[q] * ss[q] : v[q] * cc[q] + p[q] * ss[q];
        }
        pair_store4(yh, yl, (long)m * g.ldc + n, r);
    } else {                                                                 // PAIR_ACT
        const int n = tn * 256 + nl;
        if (m >= g.M || n >= g.N) return;
        if (pa.C3) *(uint2*)((bf16_t*)pa.C3 + (long)m * pa.ldc3 + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        const float o[4] = {pair_act(v[0], g.act), pair_act(v[1], g.act), pair_act(v[2], g.act), pair_act(v[3], g.act)};
        pair_store4(yh, yl, (long)m * g.ldc + n, o);
    }
}
