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

template <bool BALANCED, bool TA, bool TB, bool STG16, int PAIR>
__device__ __forceinline__ void gemm_256h_body(const GemmArgs g) {
    static_assert(PAIR == PAIR_NONE || (!TA && !TB && !STG16), "the pair forms are NT launches with the fp32 staging epilogue");
    constexpr int BM = 256, BN = 256;
    constexpr int HALF = 128 * BK * 2;                 // one half-tile image
    enum { A_LO = 0, A_HI = 1, B_LO = 2, B_HI = 3 };
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [parity][A_LO, A_HI, B_LO, B_HI][HALF]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- block -> (tile, K range).  Blocks [0, full_tiles) own whole tiles (bijective XCD remap); the remaining
    // tiles (the last, partially filled round of the 256 CUs) are each cut into `splits` K ranges, so the tail of the
    // launch also fills the chip.  splits == 1 -> full_tiles == all tiles and there is no tail.
    int pid = 0, split = 0, kt0 = 0, nt = 0, tm = 0, tn = 0, m0 = 0, n0 = 0;
    bool tail = false;
    // ---- DMA sources: this wave owns pieces (wave*2 + i), i = 0..1, of every half-tile image.
    // plain image: piece = 8 rows x 128 B;  contraction-major image: piece = 4 k-rows x 256 B.  32-bit element offsets
    // relative to a per-K-tile scalar base (plain: + k0 elements; contraction-major: + k0 rows).
    unsigned offA[2][2], offB[2][2];                   // [half][piece] element offsets of this lane's 16-byte chunk
    auto setup = [&](int b) {                          // work item b: tile, K range, DMA offsets
        split = 0; kt0 = 0; nt = (g.K + BK - 1) / BK; tail = false;
        if (b < g.full_tiles) {
            const int x = b & 7, q = g.full_tiles >> 3, r = g.full_tiles & 7;
            pid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        } else {
            const int u = b - g.full_tiles;
            pid = g.full_tiles + u / g.splits;
            split = u % g.splits;
            tail = true;
            const int nt_all = (g.K + BK - 1) / BK;
            kt0 = (int)((long)split * nt_all / g.splits);
            nt = (int)((long)(split + 1) * nt_all / g.splits) - kt0;
        }
        tile_of_256(pid, g, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (wave * 2 + i) * 8 + (lane >> 3);                 // plain image row
            const int chunk = (lane & 7) ^ ((hr >> 1) & 7);                  // logical 16-B chunk this lane fetches
            const int kr = (wave * 2 + i) * 4 + (lane >> 4);                 // contraction-major image row
            const int pc = lane & 15;                                        // physical 16-B chunk in the 256-byte row
            const int lc = ((((pc >> 1) ^ ((kr & 3) | (((kr >> 3) & 1) << 2))) << 1) | (pc & 1));   // logical chunk (8 columns)
            if (TA) {
                int m = m0 + (lc >> 3) * 128 + h * 64 + (lc & 7) * 8;         // first of 8 consecutive tile rows
                m = m + 8 <= g.M ? m : g.M - 8;
                offA[h][i] = (unsigned)((long)kr * g.lda + m);
            } else {
                int ra = m0 + (hr >> 6) * 128 + h * 64 + (hr & 63);
                ra = ra < g.M ? ra : g.M - 1;
                offA[h][i] = (unsigned)((long)ra * g.lda + chunk * 8);
            }
            if (TB) {
                int n = n0 + (lc >> 2) * 64 + h * 32 + (lc & 3) * 8;          // first of 8 consecutive tile columns
                n = n + 8 <= g.N ? n : g.N - 8;
                offB[h][i] = (unsigned)((long)kr * g.ldb + n);
            } else {
                const int lcol = (hr >> 5) * 64 + h * 32 + (hr & 31);        // tile column this B row produces
                int rb = n0 + lcol;
                if (g.swiglu_inter) rb = (lcol < 128 ? 0 : g.swiglu_inter - 128) + tn * 128 + lcol;
                rb = rb < g.N ? rb : g.N - 1;
                offB[h][i] = (unsigned)((long)rb * g.ldb + chunk * 8);
            }
        }
    };
    // K tiles of the current item: global tiles kt0 .. kt0+nt-1
    auto stage = [&](int par, int which, int t) {
        const int kt = kt0 + (t < nt ? t : nt - 1);     // global K tile (wave-uniform)
        char* dst = smem + (par * 4 + which) * HALF + wave * 2048;
        const bf16_t* base;                             // scalar base of this K tile
        // (pair forms only: K tiles kt >= kt_wrap come from A2 / wrap around in B -- two scalar selects, compiled out of every other form)
        if (which < 2) base = TA ? (kt == g.k_tail_tile ? g.A_tail : g.A + (long)kt * BK * g.lda)
                                 : (PAIR ? (kt < g.kt_wrap ? g.A + kt * BK : g.A2 + (kt - g.kt_wrap) * BK) : g.A + kt * BK);
        else base = TB ? (kt == g.k_tail_tile ? g.B_tail : g.B + (long)kt * BK * g.ldb)
                       : g.B + (PAIR ? (kt < g.kt_wrap ? kt : kt - g.kt_wrap) : kt) * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = base + ((which < 2) ? offA[which & 1][i] : offB[which & 1][i]);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[8][4];
    const EpiArgs e = {g.C, g.bias, g.resid, g.ldc, g.ldr, g.M, g.N, g.out_f32, g.act, g.alpha};
    const int sw = g.swiglu_inter;
    int vb = blockIdx.x;
    setup(vb);
    bool early = false;                                  // K tile 0 of this item was requested during the previous epilogue
  while (true) {
    // The per-lane fragment offsets are recomputed for every work item from an opaque copy of the lane id: as loop invariants
    // they would stay live across the epilogue, and the register allocator then spills inside the K loop (scratch reloads there
    // drain the DMA queue through the compiler's vmcnt(0)).
    int fl = lane;
    asm volatile("" : "+v"(fl));
    // ---- fragment read offsets: row (lane&15) of a 16-row fragment, chunk (kk*4 + lane>>4) ^ swizzle(row)
    const int fsw = (fl & 15) >> 1;                                        // (row >> 1) & 7 for every fragment row
    const int fo0 = (fl & 15) * 128 + (((fl >> 4) ^ fsw) << 4);          // kk = 0
    const int fo1 = fo0 ^ 64;                                                // kk = 1: chunk + 4
    const int aoff = wr * 64 * 128, boff = wc * 32 * 128;
    // contraction-major image: lane i of a 16-lane group supplies the address of k-row kk*32 + g*8 + (i>>2) [+4 for the
    // second read], columns cb + 4*(i&3) .. +3 (8 bytes), and receives column cb + i of the four rows.  The swizzle term
    // (row & 3) | ((row >> 3) & 1) << 2 of those rows does not depend on kk or on the +4, so it is a per-lane constant and a
    // fragment's address is  img + [lane part] + ((block ^ hx) << 5) + kk*8192 (+1024), block = 16-column block of the image.
    const int tq = (fl & 15) >> 2, tg = fl >> 4;
    const int hx = tq | ((tg & 1) << 2);
    const int tlane = (tg * 8 + tq) * 256 + (fl & 3) * 8;
    // tlane has no bit in 5..7, so tlane + ((block ^ hx) << 5) = (tlane ^ (hx << 5)) ^ (block << 5): ONE per-lane register and
    // a scalar XOR per fragment (block = wr*4 + i / wc*2 + j is wave-uniform) instead of six offset registers
    const int tbase = tlane ^ (hx << 5);
    // Issued as inline asm: the ds_read_tr16 builtin makes hipcc drain the LDS-DMA queue (s_waitcnt vmcnt(0)) in front of
    // every read -- it cannot tell the read from the in-flight global_load_lds writes of OTHER images -- which serialises the
    // whole pipeline.  The asm form is invisible to that pass; its completion is covered by the explicit lgkmcnt(0) that
    // already precedes each MFMA segment (and the sched_barriers around it), the only consumers of these registers.
    auto frag_t16 = [&](const char* p) -> bf16x8 {
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
        u32x2_t lo, hi;
        const unsigned addr = (unsigned)(uintptr_t)p;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr));
        const uint4 both = make_uint4(lo[0], lo[1], hi[0], hi[1]);
        return __builtin_bit_cast(bf16x8, both);
    };
    auto fragA = [&](const char* img, int i, int kk) -> bf16x8 {
        if (TA) return frag_t16(img + (tbase ^ ((wr * 4 + i) << 5)) + kk * 8192);
        return *(const bf16x8*)(img + aoff + i * 2048 + (kk ? fo1 : fo0));
    };
    auto fragB = [&](const char* img, int j, int kk) -> bf16x8 {
        if (TB) return frag_t16(img + (tbase ^ ((wc * 2 + j) << 5)) + kk * 8192);
        return *(const bf16x8*)(img + boff + j * 2048 + (kk ? fo1 : fo0));
    };
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: tile 0 complete + the first two halves of tile 1, in steady-state issue order
    if (!early) { stage(0, A_LO, 0); stage(0, B_LO, 0); stage(0, B_HI, 0); stage(0, A_HI, 0); }
    stage(1, A_LO, 1); stage(1, B_LO, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                         // A-lo(0), B-lo(0) landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 a[4][2], bhi[2][2], blo_even[2][2], blo_odd[2][2];
    if (BALANCED) {                                                          // tile 0's B-lo fragments
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) blo_even[j][kk] = fragB(smem + B_LO * HALF, j, kk);
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();                               // wave row 1 runs one barrier behind

    auto tile = [&](auto PAR, const int t, bf16x8 (&blo)[2][2], bf16x8 (&blo_next)[2][2]) {
        constexpr int par = decltype(PAR)::value;
        const char* img = smem + par * 4 * HALF;
        const char* img_next = smem + (par ^ 1) * 4 * HALF;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // ------------------------------------------------ LDS segment
            if (p == 0) {
                if (!BALANCED) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < 2; ++j) blo[j][kk] = fragB(img + B_LO * HALF, j, kk);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i][kk] = fragA(img + A_LO * HALF, i, kk);
            } else if (p == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) bhi[j][kk] = fragB(img + B_HI * HALF, j, kk);
            } else if (p == 2) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i][kk] = fragA(img + A_HI * HALF, i, kk);
            } else if (BALANCED) {                                           // next tile's B-lo (waited for in phase 2)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) blo_next[j][kk] = fragB(img_next + B_LO * HALF, j, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p == 0) stage(par ^ 1, B_HI, t + 1);
            if (p == 1) stage(par ^ 1, A_HI, t + 1);
            if (p == 2) stage(par, A_LO, t + 2);
            if (p == 3) stage(par, B_LO, t + 2);
            if (BALANCED) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ------------------------------------------------ MFMA segment: C quadrant (mq, nq)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            const int mq = p >> 1;
            const bool hi = (p == 1 || p == 2);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        // swapped operands: D'[n][m] -> lane holds n = (lane>>4)*4 + r, m = lane&15
                        acc[mq * 4 + i][(hi ? 2 : 0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            hi ? bhi[j][kk] : blo[j][kk], a[i][kk], acc[mq * 4 + i][(hi ? 2 : 0) + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t0 = 0; t0 < nt; t0 += 2) {
        tile(std::integral_constant<int, 0>{}, t0, blo_even, blo_odd);
        if (t0 + 1 < nt) tile(std::integral_constant<int, 1>{}, t0 + 1, blo_odd, blo_even);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // redundant tail DMAs must land before exit

    // only the bf16-staging instantiations are persistent (the others keep one work item per workgroup and the register
    // allocation of the non-looping kernel: the fp32 read-modify-write dW form has no register to give to a loop)
    const int nvb = vb + (int)gridDim.x;
    const bool more = STG16 && nvb < g.total_blocks;
    // ---- split-K tail: leave this block's fp32 partial tile as a slab (fragment-major, 16 B per lane, coalesced);
    // gemm_tail_reduce_kernel sums a tile's slabs in split order and runs the epilogue (launched right behind).
    if (tail) {
        constexpr int SLAB4 = BM * BN / 4;                                   // float4 per slab
        float4* mine = (float4*)g.slabs + ((size_t)(pid - g.full_tiles) * g.splits + split) * SLAB4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                mine[(i * 4 + j) * 512 + tid] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (!more) return;
        vb = nvb; setup(vb); early = false;
        __syncthreads();                                                     // every wave is done with the operand images
        continue;
    }

    // ---- epilogue, staged through LDS so that every global store instruction of a wave covers ONE full output row of
    // the tile (1 KiB fp32 / 512 B bf16).  Writing accumulator fragments straight out gives 16 row segments of 32 B per
    // instruction, and those partial-line writes cost ~25 us per tile (a quarter of a K = 3584 launch).
    const int cm0 = m0, cn0 = n0, ctn = tn;                                  // this tile; m0 / n0 / tn move on to the next item below
    early = more && STG16;
    if (more) { vb = nvb; setup(vb); }
    __syncthreads();                                                         // every wave is done with the operand images
    if (early) { stage(0, A_LO, 0); stage(0, B_LO, 0); stage(0, B_HI, 0); stage(0, A_HI, 0); }   // next item's K tile 0 -> parity 0
    if (STG16 && !sw) {
        // ---- direct form (bf16 output, no residual, no SwiGLU): alpha / bias / activation, ONE rounding, then the 4 x 4 lane-row
        // transpose so that lane (l15, r) owns columns wc*64 + r*16 .. +15 of row wr*128 + i*16 + l15: two 16-byte stores per lane,
        // 128 contiguous bytes per row and fragment row -- no LDS, no barrier
        float bias16[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = cn0 + wc * 64 + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) bias16[j][q] = (e.bias && nb + q < e.N) ? bf2f(e.bias[nb + q]) : 0.f;
        }
        const int ncol = cn0 + wc * 64 + (lane >> 4) * 16;                   // first of this lane's 16 columns after the transpose
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j] * e.alpha + (f32x4){bias16[j][0], bias16[j][1], bias16[j][2], bias16[j][3]};
                if (e.act != SPACER_ACT_NONE) v = apply_act4(v, e.act);
                lo[j] = pack_bf2(v[0], v[1]); hi[j] = pack_bf2(v[2], v[3]);
            }
            xpose4_rows(lo); xpose4_rows(hi);
            const int m = cm0 + wr * 128 + i * 16 + (lane & 15);
            if (m < e.M && ncol < e.N) {
                bf16_t* c = (bf16_t*)e.C + (long)m * e.ldc + ncol;
                if (ncol + 16 <= e.N && (e.ldc % 8) == 0 && ((uintptr_t)e.C % 16) == 0) {
                    *(uint4*)c = make_uint4(lo[0], hi[0], lo[1], hi[1]);
                    *(uint4*)(c + 8) = make_uint4(lo[2], hi[2], lo[3], hi[3]);
                } else {
                    const uint32_t w[8] = {lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]};
                    for (int q = 0; q < 16 && ncol + q < e.N; ++q) c[q] = (bf16_t)((q & 1) ? (w[q >> 1] >> 16) : (w[q >> 1] & 0xffffu));
                }
            }
        }
    } else if constexpr (STG16) {
        // bf16 staging in the parity-1 half (SwiGLU form: gate and up columns of one output sit in different waves): [128 rows][256 cols] bf16 = 64 KiB per pass; 8-byte chunk (4 columns) index ^= row & 15:
        // a fragment store (16 rows x one chunk per 16-lane group) and a row read (64 chunks of one row) are both conflict-free.
        // alpha / bias / activation and the ONE rounding happen on the way in (no residual in this form).
        char* stg = smem + 4 * HALF;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {                           // (not unrolled: the body does not index registers by `pass`)
            if (wr == pass) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = wc * 64 + j * 16 + (lane >> 4) * 4;          // column inside the tile
                    float b4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (e.bias) {
                        const int nb = sw ? (nl < 128 ? 0 : sw - 128) + ctn * 128 + nl : cn0 + nl;   // weight row of that column
#pragma unroll
                        for (int q = 0; q < 4; ++q) b4[q] = (nb + q < e.N) ? bf2f(e.bias[nb + q]) : 0.f;
                    }
                    // the activation switch is hoisted out of the 32 unrolled fragment bodies: inlined per element it made the
                    // epilogue 11.7 K instructions (~70 KB, more than the instruction cache two CUs share) that every tile walked
                    // through even with act = none -- 9-12 us per tile round
                    if (e.act == SPACER_ACT_NONE) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int row = i * 16 + (lane & 15);
                            const uint32_t lo = pack_bf2(acc[i][j][0] * e.alpha + b4[0], acc[i][j][1] * e.alpha + b4[1]);
                            const uint32_t hi = pack_bf2(acc[i][j][2] * e.alpha + b4[2], acc[i][j][3] * e.alpha + b4[3]);
                            *(uint2*)(stg + row * 512 + (((nl >> 2) ^ (row & 15)) << 3)) = make_uint2(lo, hi);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int row = i * 16 + (lane & 15);
                            const f32x4 v = apply_act4(acc[i][j] * e.alpha + (f32x4){b4[0], b4[1], b4[2], b4[3]}, e.act);
                            *(uint2*)(stg + row * 512 + (((nl >> 2) ^ (row & 15)) << 3)) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                        }
                    }
                }
            }
            __syncthreads();
            if (sw) {
                // SwiGLU: a half wave per tile row; lane l reads the gate chunk l and the up chunk l + 32 of the row (bf16, what the
                // unfused path stores and swiglu_fwd_kernel reads back) and writes 4 outputs = 8 bytes
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 16 + wave * 2 + (lane >> 5), l = lane & 31;
                    const uint2 gv = *(const uint2*)(stg + row * 512 + ((l ^ (row & 15)) << 3));
                    const uint2 uv = *(const uint2*)(stg + row * 512 + (((l + 32) ^ (row & 15)) << 3));
                    const int m = cm0 + pass * 128 + row;
                    if (m < e.M) {
                        const uint32_t g01 = gv.x, g23 = gv.y, u01 = uv.x, u23 = uv.y;
                        if (g.C2) {
                            bf16_t* c2 = (bf16_t*)g.C2 + (long)m * g.ldc2 + ctn * 128 + l * 4;
                            *(uint2*)c2 = make_uint2(g01, g23);
                            *(uint2*)(c2 + sw) = make_uint2(u01, u23);
                        }
                        const float g0 = bf_lo(g01), g1 = bf_hi(g01), g2 = bf_lo(g23), g3 = bf_hi(g23);
                        const uint32_t o01 = pack_bf2(g0 * (1.f / (1.f + __expf(-g0))) * bf_lo(u01), g1 * (1.f / (1.f + __expf(-g1))) * bf_hi(u01));
                        const uint32_t o23 = pack_bf2(g2 * (1.f / (1.f + __expf(-g2))) * bf_lo(u23), g3 * (1.f / (1.f + __expf(-g3))) * bf_hi(u23));
                        *(uint2*)((bf16_t*)e.C + (long)m * e.ldc + ctn * 128 + l * 4) = make_uint2(o01, o23);
                    }
                }
            } else {
                for (int it = 0; it < 16; ++it) {
                    const int row = it * 8 + wave;                                // one wave = one tile row
                    const uint2 v = *(const uint2*)(stg + row * 512 + ((lane ^ (row & 15)) << 3));
                    const int m = cm0 + pass * 128 + row, n = cn0 + lane * 4;
                    if (m < e.M && n < e.N) {
                        bf16_t* c = (bf16_t*)e.C + (long)m * e.ldc + n;
                        if (n + 4 <= e.N && (e.ldc % 4) == 0) *(uint2*)c = v;
                        else {
                            const bf16_t t4[4] = {(bf16_t)(v.x & 0xffffu), (bf16_t)(v.x >> 16), (bf16_t)(v.y & 0xffffu), (bf16_t)(v.y >> 16)};
                            for (int q = 0; q < 4 && n + q < e.N; ++q) c[q] = t4[q];
                        }
                    }
                }
            }
            if (pass == 0) __syncthreads();
        }
    } else {
    // (the direct, register-transposed form of the bf16 case was tried here too: 1175 -> 1162 TF/s for o / down and 1093 -> 1077 for dW in
    // the step -- the residual read-modify-write wants whole 1 KiB rows per instruction; the LDS-staged form stays)
    // Two passes (wave row 0, then 1): [128 rows][256 cols] fp32 = the whole 128 KiB; 16-byte chunk index ^= row & 7 keeps
    // the fragment writes (8 rows per store group) and the row reads conflict-free.  alpha / bias / activation are applied
    // on the way in, residual + conversion on the way out (one rounding, as before).
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (wr == pass) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nl = wc * 64 + j * 16 + (lane >> 4) * 4;              // column inside the tile
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                if (e.bias) {
                    const int nb = sw ? (nl < 128 ? 0 : sw - 128) + ctn * 128 + nl : cn0 + nl;   // weight row of that column
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[q] = (nb + q < e.N) ? bf2f(e.bias[nb + q]) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = i * 16 + (lane & 15);
                    f32x4 v = acc[i][j] * e.alpha + (f32x4){b4[0], b4[1], b4[2], b4[3]};
                    if (PAIR <= PAIR_PLAIN && e.act != SPACER_ACT_NONE) v = apply_act4(v, e.act);          // out of line: see the bf16 staging form
                    *(float4*)(smem + row * 1024 + (((nl >> 2) ^ (row & 7)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();
        if constexpr (PAIR == PAIR_SWIGLU) {
            // half a wave per staged row: lane l reads gate chunk l and up chunk l + 32 (fp32, never rounded), writes silu(gate) * up
            // as a (hi, lo) pair and -- for a taped forward -- bf16(gate | up), the point swiglu_bwd differentiates at
            for (int it = 0; it < 8; ++it) {
                const int row = it * 16 + wave * 2 + (lane >> 5), l = lane & 31;
                const float4 gv = *(const float4*)(smem + row * 1024 + ((l ^ (row & 7)) << 4));
                const float4 uv = *(const float4*)(smem + row * 1024 + (((l + 32) ^ (row & 7)) << 4));
                const int m = cm0 + pass * 128 + row;
                if (m < e.M) {
                    const long n = (long)ctn * 128 + l * 4;
                    const float o[4] = {gv.x * pair_sigm(gv.x) * uv.x, gv.y * pair_sigm(gv.y) * uv.y, gv.z * pair_sigm(gv.z) * uv.z,
                                        gv.w * pair_sigm(gv.w) * uv.w};
                    pair_store4((bf16_t*)e.C, (bf16_t*)g.C2, (long)m * e.ldc + n, o);
                    if (g.C3) {
                        bf16_t* c3 = (bf16_t*)g.C3 + (long)m * g.ldc3 + n;
                        *(uint2*)c3 = make_uint2(pack_bf2(gv.x, gv.y), pack_bf2(gv.z, gv.w));
                        *(uint2*)(c3 + sw) = make_uint2(pack_bf2(uv.x, uv.y), pack_bf2(uv.z, uv.w));
                    }
                }
            }
        } else if constexpr (PAIR == PAIR_ROPE) {
            // head_dim 128: the staged row holds two whole heads; lane l of a half wave takes dims 4c .. 4c + 3 of the first half of head
            // hh and the matching dims of the second half (HF rotate_half convention, fp32 tables [tokens, 128])
            for (int it = 0; it < 8; ++it) {
                const int row = it * 16 + wave * 2 + (lane >> 5), l = lane & 31;
                const int hh = l >> 4, c = l & 15, ch1 = hh * 32 + c, ch2 = ch1 + 16;
                const float4 a = *(const float4*)(smem + row * 1024 + ((ch1 ^ (row & 7)) << 4));
                const float4 b = *(const float4*)(smem + row * 1024 + ((ch2 ^ (row & 7)) << 4));
                const int m = cm0 + pass * 128 + row, n1 = cn0 + ch1 * 4;
                if (m < e.M && n1 < e.N) {
                    const float x1[4] = {a.x, a.y, a.z, a.w}, x2[4] = {b.x, b.y, b.z, b.w};
                    float r1[4], r2[4];
                    if ((n1 >> 7) < g.rope_heads) {
                        const float4 c1 = *(const float4*)(g.rope_cos + (long)m * 128 + c * 4), c2 = *(const float4*)(g.rope_cos + (long)m * 128 + 64 + c * 4);
                        const float4 s1 = *(const float4*)(g.rope_sin + (long)m * 128 + c * 4), s2 = *(const float4*)(g.rope_sin + (long)m * 128 + 64 + c * 4);
                        const float cc1[4] = {c1.x, c1.y, c1.z, c1.w}, cc2[4] = {c2.x, c2.y, c2.z, c2.w};
                        const float ss1[4] = {s1.x, s1.y, s1.z, s1.w}, ss2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            r1[q] = x1[q] * cc1[q] - x2[q] * ss1[q];
                            r2[q] = x2[q] * cc2[q] + x1[q] * ss2[q];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { r1[q] = x1[q]; r2[q] = x2[q]; }
                    }
                    pair_store4((bf16_t*)e.C, (bf16_t*)g.C2, (long)m * e.ldc + n1, r1);
                    pair_store4((bf16_t*)e.C, (bf16_t*)g.C2, (long)m * e.ldc + n1 + 64, r2);
                }
            }
        } else if constexpr (PAIR == PAIR_ACT) {
            for (int it = 0; it < 16; ++it) {
                const int row = it * 8 + wave;                                // one wave = one tile row
                const float4 v = *(const float4*)(smem + row * 1024 + ((lane ^ (row & 7)) << 4));
                const int m = cm0 + pass * 128 + row, n = cn0 + lane * 4;
                if (m < e.M && n < e.N) {
                    if (g.C3) *(uint2*)((bf16_t*)g.C3 + (long)m * g.ldc3 + n) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
                    const float o[4] = {pair_act(v.x, e.act), pair_act(v.y, e.act), pair_act(v.z, e.act), pair_act(v.w, e.act)};
                    pair_store4((bf16_t*)e.C, (bf16_t*)g.C2, (long)m * e.ldc + n, o);
                }
            }
        } else {
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + wave;                                // one wave = one tile row
            const float4 v = *(const float4*)(smem + row * 1024 + ((lane ^ (row & 7)) << 4));
            const int m = cm0 + pass * 128 + row, n = cn0 + lane * 4;
            if (m < e.M && n < e.N) store_row4(e, m, n, v);
        }
        }
        if (pass == 0) __syncthreads();
    }
    }
    if (!more) return;
    __syncthreads();                                                         // staging reads done before the next item's DMA lands there
  }
}

// The kernels rocprof names.  gemm_bf16_nt_256h_kernel<BALANCED, TA, TB, STG16>: every production form (no pair selects compiled in);
// gemm_bf16_pair_256h_kernel<MODE>: the K-concatenated pair forms of the precise mode (MODE = PAIR_PLAIN .. PAIR_ACT).
template <bool BALANCED, bool TA = false, bool TB = false, bool STG16 = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_nt_256h_kernel(GemmArgs g) {
    gemm_256h_body<BALANCED, TA, TB, STG16, PAIR_NONE>(g);
}
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pair_256h_kernel(GemmArgs g) {
    gemm_256h_body<true, false, false, false, MODE>(g);
}

// Sums the K-split partial tiles of the tail (written by gemm_bf16_nt_256h_kernel) in split order -- deterministic --
// and applies the epilogue.  One block per (tail tile, accumulator fragment index): the whole chip takes part, a
// single CU could pull its tile's slabs only at ~25 GB/s.
__global__ __launch_bounds__(512) void gemm_tail_reduce_kernel(GemmArgs g) {
    constexpr int SLAB4 = 256 * 256 / 4;
    const int lt = blockIdx.x >> 5, f = blockIdx.x & 31, i = f >> 2, j = f & 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    int tm, tn;
    tile_of_256(g.full_tiles + lt, g, tm, tn);
    const float4* p = (const float4*)g.slabs + (size_t)lt * g.splits * SLAB4 + f * 512 + tid;
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < g.splits; ++sp) {
        const float4 v = p[(size_t)sp * SLAB4];
        sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
    }
    const EpiArgs e = {g.C, g.bias, g.resid, g.ldc, g.ldr, g.M, g.N, g.out_f32, g.act, g.alpha};
    const int m = tm * 256 + wr * 128 + i * 16 + (lane & 15), nl = wc * 64 + j * 16 + (lane >> 4) * 4;      // row, column inside the tile
    if (g.pair_mode <= PAIR_PLAIN) {
        store_frag(e, m, tn * 256 + nl, sum);
        return;
    }
    // ---- pair epilogues on a tail tile (the same arithmetic as the staged epilogue of gemm_256h_body): the partner value of a
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
    if (g.pair_mode == PAIR_SWIGLU) {
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
        if (g.C3) *(uint2*)((bf16_t*)g.C3 + (long)m * g.ldc3 + (nl < 128 ? 0 : sw) + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    } else if (g.pair_mode == PAIR_ROPE) {
        const float4 pt = ex[tid ^ 64];
        const int n = tn * 256 + nl;
        if (m >= g.M || n >= g.N) return;
        float r[4] = {v[0], v[1], v[2], v[3]};
        if ((n >> 7) < g.rope_heads) {
            const int d = nl & 127;
            const float4 c4 = *(const float4*)(g.rope_cos + (long)m * 128 + d), s4 = *(const float4*)(g.rope_sin + (long)m * 128 + d);
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, p[4] = {pt.x, pt.y, pt.z, pt.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = d < 64 ? v[q] * cc[q] - p[q] * ss[q] : v[q] * cc[q] + p[q] * ss[q];
        }
        pair_store4(yh, yl, (long)m * g.ldc + n, r);
    } else {                                                                 // PAIR_ACT
        const int n = tn * 256 + nl;
        if (m >= g.M || n >= g.N) return;
        if (g.C3) *(uint2*)((bf16_t*)g.C3 + (long)m * g.ldc3 + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        const float o[4] = {pair_act(v[0], g.act), pair_act(v[1], g.act), pair_act(v[2], g.act), pair_act(v[3], g.act)};
        pair_store4(yh, yl, (long)m * g.ldc + n, o);
    }
}
