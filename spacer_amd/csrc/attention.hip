// Flash-style attention for gfx950 (MFMA 16x16x32 bf16, LDS-staged tiles, online softmax): forward, and a
// two-kernel backward (dQ by query-block owners, dK/dV by key-block owners).
//
// One set of kernels serves the three attention shapes on the SG-RLVR hot path (include/spacer_hip.h
// "segments"):
//   * ViT per-temporal-grid non-causal attention, head_dim 80 (HF VisionAttention via cu_seqlens)
//   * LLM causal GQA prefill, head_dim 128
//   * shared-prefix scoring: K rollouts attend one prompt's keys + their own causal keys
//     (== K independent causal rows prompt+completion_k, SG_RLVR_trainer.py:527, without re-running the prompt)
//
// Forward decomposition: workgroup = 4 waves = 128 query rows of one (segment, q-head); each wave owns 32 rows
// (two 16-column MFMA blocks).  Per 64-key tile:
//   S^T = K . Q^T      (A = K frag from LDS, B = Q frag held in registers)   -> lane holds S^T[key][q = lane&15]
//   online softmax per q column: in-lane max over 16 scores + 2 cross-lane shuffles (lanes l, l^16, l^32, l^48)
//   O^T += V^T . P^T   (A = V^T frag from the transposed LDS tile, B = P^T straight from the S^T registers:
//                       the contraction order over keys is permuted identically on both operands, so no
//                       lane exchange / LDS round trip is needed for P)
// O^T keeps q on lane&15, the index the softmax statistics live on, so rescaling is a per-lane multiply.
// K and V tiles are prefetched into registers one tile ahead (global loads stay in flight under the MFMAs)
// and written to LDS after the barrier (V is transposed on the way).
#include "attn_common.h"
#include <type_traits>

namespace {

constexpr int BQ = 128;   // query rows per workgroup (fwd, dQ)
constexpr int BKV = 64;   // keys per tile

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
    const bf16_t* d_o; float* delta; bf16_t* dq; float* dk; float* dv;
    long q_stride, kv_stride, o_stride;
    const spacer_attn_segment* segs;
    int num_segs, nqb, T, Hq, Hkv, causal, lpt;
    float scale;
};

// tile t of a query block's key list: prefix tiles first, then own keys
struct KeyTile { int start_abs, len, rel0; bool own; };
__device__ __forceinline__ KeyTile key_tile(const spacer_attn_segment& seg, int n_pre, int own_len, int t) {
    KeyTile kt;
    if (t < n_pre) { kt.own = false; kt.rel0 = t * BKV; kt.start_abs = seg.pre_start; kt.len = seg.pre_len; }
    else { kt.own = true; kt.rel0 = (t - n_pre) * BKV; kt.start_abs = seg.q_start; kt.len = own_len; }
    return kt;
}

// ================================================================================================ forward, pipelined form
// Decomposition: see the file header.  K / V tiles reach LDS by global_load_lds DMA straight into TWO pairs of
// images (64 KiB per workgroup, still two workgroups per CU), issued one tile ahead, ONE barrier per tile:
//     top of tile t:  s_waitcnt vmcnt(0)  (my pieces of tile t landed)  ->  s_barrier (everyone's landed, and everyone is done
//                     with tile t-1's buffer)  ->  issue the DMA of tile t+1 into that buffer  ->  compute tile t.
// No register staging (-32 VGPRs), no ds_write pass, the swizzle sits on the DMA source address.  Rows past a ragged tile's end are
// fetched from its last valid row (finite; their scores are masked), the D = 80 image's padding chunks from chunk 0 of the row
// (finite; they meet Q's zero padding).  That alone (the `dma` stepping stone, 616 TF/s, since removed) was not the stall: every
// LDS latency is also taken off the critical path of a wave (PMC on the cfg3 layout showed the SIMDs
// idle ~45 % of the time with MFMA 29 % / other VALU 24 % busy: four exposed K-fragment round trips, two serial ds_bpermute
// levels in the row max, the first V^T batch, and ~120 address VALU per tile):
//   * K fragments: inline-asm ds_read_b128 with immediate offsets, two key blocks in flight (issued right after the barrier,
//     before the DMA issue code), refilled after each block's MFMAs; counted lgkmcnt waits.
//   * V^T fragments: first batch issued under the last QK^T MFMAs, second between the two softmax halves.
//   * row max across the four lane groups: v_permlane16_swap / v_permlane32_swap (VALU) instead of two ds_bpermute trips.
//   * addresses: per-lane LDS bases hoisted (the swizzle is an XOR on a bit field disjoint from the row and tile fields),
//     DMA source = scalar tile base + hoisted 32-bit lane offsets; only ragged tiles take the clamping path.
// (The register-staged round-1 forms of these kernels were removed in round 3; csrc/precise.hip keeps the plain structure.)
typedef __attribute__((ext_vector_type(2))) unsigned int attn_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int attn_u32x4;

template <int OFF> __device__ __forceinline__ void lds_b128(attn_u32x4& x, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_tr64(attn_u32x2& x, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// fragments of key block KF (rows KF*16 + lane&15) for every 32-wide contraction chunk
template <int DC, int KF> __device__ __forceinline__ void k_issue(attn_u32x4 (&r)[4], const unsigned (&ka)[4]) {
    lds_b128<KF * 16 * AT_RM_ROW_BYTES>(r[0], ka[0]);
    lds_b128<KF * 16 * AT_RM_ROW_BYTES>(r[1], ka[1]);
    lds_b128<KF * 16 * AT_RM_ROW_BYTES>(r[2], ka[2]);
    if constexpr (DC > 3) lds_b128<KF * 16 * AT_RM_ROW_BYTES>(r[3], ka[3]);
    __builtin_amdgcn_sched_barrier(0);
}
template <int DC> __device__ __forceinline__ void qk_mfma(f32x4 (&st)[2], const attn_u32x4 (&r)[4], const bf16x8 (&qf)[2][DC]) {
    st[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; st[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
        const bf16x8 kfr = __builtin_bit_cast(bf16x8, r[dc]);
        st[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[0][dc], st[0], 0, 0, 0);
        st[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[1][dc], st[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
// V^T fragments fr = 4B .. 4B+3 (fr = df*2 + c): two transposing reads each; va = this lane's base in the V image
template <int DF, int B, int J> __device__ __forceinline__ void v_issue1(attn_u32x2 (&lo)[4], attn_u32x2 (&hi)[4], unsigned va) {
    constexpr int fr = B * 4 + J;
    if constexpr (fr < DF * 2) {
        const unsigned ad = va ^ ((fr >> 1) << 5);
        lds_tr64<(fr & 1) * 32 * AT_RM_ROW_BYTES>(lo[J], ad);
        lds_tr64<(fr & 1) * 32 * AT_RM_ROW_BYTES + 16 * AT_RM_ROW_BYTES>(hi[J], ad);
    }
}
template <int DF, int B> __device__ __forceinline__ void v_issue(attn_u32x2 (&lo)[4], attn_u32x2 (&hi)[4], unsigned va) {
    if constexpr (B * 4 < DF * 2) {
        v_issue1<DF, B, 0>(lo, hi, va); v_issue1<DF, B, 1>(lo, hi, va); v_issue1<DF, B, 2>(lo, hi, va); v_issue1<DF, B, 3>(lo, hi, va);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int DF, int B> constexpr int v_batch_reads() { return (B * 4 >= DF * 2) ? 0 : 2 * ((DF * 2 - B * 4) < 4 ? (DF * 2 - B * 4) : 4); }
template <int DF, int B> __device__ __forceinline__ void pv_mfma(f32x4 (&oacc)[2][DF], const attn_u32x2 (&lo)[4], const attn_u32x2 (&hi)[4],
                                                                 const bf16x8 (&pf)[2][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int fr = B * 4 + j;
        if (fr < DF * 2) {
            const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo[j][0], lo[j][1], hi[j][0], hi[j][1]));
            const int df = fr >> 1, c = fr & 1;
            oacc[0][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][c], oacc[0][df], 0, 0, 0);
            oacc[1][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][c], oacc[1][df], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
// max over the four 16-lane groups (same lane&15), VALU only
__device__ __forceinline__ float group_max4(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(AttnArgs a) {
    constexpr int DC = (D + 31) / 32;
    constexpr int DF = D / 16;
    constexpr int BUF = 2 * AT_RM_BYTES;
    constexpr int NB = (DF * 2 + 3) / 4;
    static_assert(NB >= 3 && NB <= 4, "V^T batch schedule below is written for 3 or 4 batches");
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][K | V]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seg_id = a.lpt ? a.num_segs - 1 - blockIdx.x / a.nqb : blockIdx.x / a.nqb;
    const int qb = a.lpt ? a.nqb - 1 - blockIdx.x % a.nqb : blockIdx.x % a.nqb;
    const int h = blockIdx.y, hk = h / (a.Hq / a.Hkv);
    const spacer_attn_segment seg = a.segs[seg_id];
    const int qb0 = qb * BQ;
    if (qb0 >= seg.q_len) return;
    const int l15 = lane & 15, g = lane >> 4;
    const int wq0 = qb0 + wave * 32;

    bf16x8 qf[2][DC];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        int qi = wq0 + f * 16 + l15;
        qi = qi < seg.q_len ? qi : seg.q_len - 1;
        const bf16_t* qp = a.q + (long)(seg.q_start + qi) * a.q_stride + (long)h * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) qf[f][dc] = frag_global<D>(qp, dc, lane);
    }
    f32x4 oacc[2][DF];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[f][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const float c2 = a.scale * 1.4426950408889634f;

    const int own_len = a.causal ? min(seg.q_len, qb0 + BQ) : seg.q_len;
    const int n_pre = (seg.pre_len + BKV - 1) / BKV, n_tiles = n_pre + (own_len + BKV - 1) / BKV;

    // ---- hoisted addresses
    // LDS: K fragment (row kf*16 + l15, chunk dc*4+g): l15*256 + (((dc*4+g) ^ l15) << 4) = kx0 ^ (dc << 6), kf*4096 immediate.
    //      V^T fragment (rows 32c + 4g + (i>>2) [+16], chunk db*2 + ((i&3)>>1)): vx0 ^ (db << 5), c*8192 [+4096] immediate.
    const unsigned smem0 = (unsigned)(uintptr_t)smem;
    unsigned kx[4];
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) kx[dc] = smem0 + (unsigned)(l15 * AT_RM_ROW_BYTES + ((((dc < DC ? dc : 0) * 4 + g) ^ l15) << 4));
    //      The V image is swizzled by chunk ^= (row & 7) << 1 (not row & 15 as the K image): one ds_read_b64_tr_b16 cycle serves
    //      32 lanes = 8 rows x 32 contiguous bytes, and with row & 15 rows 2j and 2j+1 land on the same two chunks (2-way
    //      conflict on every read, SQ_LDS_BANK_CONFLICT = 1/3 of the LDS cycles); with (row & 7) << 1 the eight rows take eight
    //      different chunk pairs.
    const int vrr = 4 * g + (l15 >> 2);
    const unsigned vx0 = (unsigned)(vrr * AT_RM_ROW_BYTES + ((vrr & 7) << 5) + ((l15 & 3) >> 1) * 16 + (l15 & 1) * 8);
    // DMA: waves 0,1 fill the K image, waves 2,3 the V image; piece i = rows (wave&1)*32 + 4i + g, i = 0..7; pieces i and
    // i+4 share the chunk permutation, so four lane offsets + two scalar bases (rows +0 / +16) cover a full tile
    const bf16_t* gsrc = (wave >= 2 ? a.v : a.k) + (long)hk * D;
    const int img_off = (wave >= 2 ? AT_RM_BYTES : 0) + (wave & 1) * 8 * 1024;
    const int prow = (wave & 1) * 32 + g;
    const unsigned stride_b = (unsigned)a.kv_stride * 2u;
    unsigned doff[4], dchunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = prow + 4 * j;
        int c = l15 ^ (wave >= 2 ? (row & 7) << 1 : (row & 15));             // logical chunk stored at this lane's LDS position
        if (c >= D / 8) c = 0;
        dchunk[j] = (unsigned)c * 16u;
        doff[j] = (unsigned)row * stride_b + dchunk[j];
    }
    auto issue_tile = [&](int t, int buf) {
        const KeyTile kt = key_tile(seg, n_pre, own_len, t);
        const int valid = kt.len - kt.rel0;                                   // >= 1
        const char* base = (const char*)(gsrc + (long)(kt.start_abs + kt.rel0) * a.kv_stride);
        char* dst = smem + buf * BUF + img_off;
        if (valid >= BKV) {
            const char* base16 = base + 16l * stride_b;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((i < 4 ? base : base16) + doff[i & 3]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int row = prow + 4 * i;
                row = row < valid ? row : valid - 1;                          // finite filler; those scores are masked
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)row * stride_b + dchunk[i & 3]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
    };

    issue_tile(0, 0);
    for (int t = 0; t < n_tiles; ++t) {
        const KeyTile kt = key_tile(seg, n_pre, own_len, t);
        const int buf = t & 1;
        const bool active = !(kt.own && a.causal && kt.rel0 > wq0 + 31);      // else: causal tile entirely above this wave's rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        attn_u32x4 kr[2][4];
        unsigned ka[4];
        if (active) {
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) ka[dc] = kx[dc] + (unsigned)(buf * BUF);
            k_issue<DC, 0>(kr[0], ka);
            k_issue<DC, 1>(kr[1], ka);
        }
        if (t + 1 < n_tiles) issue_tile(t + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!active) continue;                                                // (the next barrier re-syncs)
        const unsigned va = smem0 + (unsigned)(buf * BUF + AT_RM_BYTES) + vx0;

        f32x4 st[4][2];
        attn_u32x2 vlo[2][4], vhi[2][4];
        wait_lgkm<DC>();
        qk_mfma<DC>(st[0], kr[0], qf);
        k_issue<DC, 2>(kr[0], ka);
        wait_lgkm<DC>();
        qk_mfma<DC>(st[1], kr[1], qf);
        k_issue<DC, 3>(kr[1], ka);
        wait_lgkm<DC>();
        qk_mfma<DC>(st[2], kr[0], qf);
        v_issue<DF, 0>(vlo[0], vhi[0], va);
        wait_lgkm<v_batch_reads<DF, 0>()>();
        qk_mfma<DC>(st[3], kr[1], qf);

        const bool need_mask = (kt.rel0 + BKV > kt.len) || (kt.own && a.causal && kt.rel0 + BKV - 1 > wq0);
        bf16x8 pf[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int qi = wq0 + f * 16 + l15;
            float mx = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kr_ = kt.rel0 + kf * 16 + g * 4 + r;
                        const bool ok = kr_ < kt.len && !(kt.own && a.causal && kr_ > qi);
                        st[kf][f][r] = ok ? st[kf][f][r] : -INFINITY;
                    }
            }
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kf][f][r]);
            mx = group_max4(mx);
            // lazy reference: m_run follows the row maximum only when it jumps by more than 2^8 (or leaves -inf), so p <= 256
            // instead of <= 1 (bf16 P keeps its relative precision, l and O are fp32) and the O rescale -- 32 v_pk_mul per
            // fragment, taken on most tiles of random data with the exact rule -- almost never runs: 759 -> 799 TF/s.  O / l and the
            // LSE m_run ln2 + log l are the same quantities; O moves by at most a bf16 ulp against an eager rescale.
            const float m_cand = mx * c2;
            const float m_new = (m_cand > m_run[f] + 8.f) ? m_cand : m_run[f];
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_use);
            float psum = 0.f;
            float p[4][4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kf][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kf][f][r], c2, -m_use));
                    psum += p[kf][r];
                }
            l_run[f] = l_run[f] * alpha + psum;
            m_run[f] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int d = 0; d < DF; ++d) oacc[f][d] *= alpha;
            }
            pf[f][0] = pack_slots(p[0], p[1]);
            pf[f][1] = pack_slots(p[2], p[3]);
            if (f == 0) {                                                     // second V^T batch rides under the other half's softmax
                __builtin_amdgcn_sched_barrier(0);
                wait_lgkm<0>();
                v_issue<DF, 1>(vlo[1], vhi[1], va);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- O^T += V^T . P^T
        wait_lgkm<v_batch_reads<DF, 1>()>();
        pv_mfma<DF, 0>(oacc, vlo[0], vhi[0], pf);
        v_issue<DF, 2>(vlo[0], vhi[0], va);
        wait_lgkm<v_batch_reads<DF, 2>()>();
        pv_mfma<DF, 1>(oacc, vlo[1], vhi[1], pf);
        v_issue<DF, 3>(vlo[1], vhi[1], va);
        wait_lgkm<v_batch_reads<DF, 3>()>();
        pv_mfma<DF, 2>(oacc, vlo[0], vhi[0], pf);
        if constexpr (NB > 3) {
            wait_lgkm<0>();
            pv_mfma<DF, 3>(oacc, vlo[1], vhi[1], pf);
        }
    }

#pragma unroll
    for (int f = 0; f < 2; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qi = wq0 + f * 16 + l15;
        if (qi >= seg.q_len) continue;
        const float inv = 1.f / l;
        const long tok = seg.q_start + qi;
        bf16_t* op = a.o + tok * a.o_stride + (long)h * D;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const f32x4 v = oacc[f][df];
            *(uint2*)(op + df * 16 + g * 4) = make_uint2(pack_bf2(v[0] * inv, v[1] * inv), pack_bf2(v[2] * inv, v[3] * inv));
        }
        if (a.lse && g == 0) a.lse[(long)h * a.T + tok] = m_run[f] * 0.6931471805599453f + logf(l);
    }
}


// ================================================================================================ backward
// delta[h][t] = sum_d dO[t,h,d] * O[t,h,d]
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a) {
    constexpr int PER = D / 8;                       // 16-byte chunks per (token, head)
    constexpr int GRP = (PER <= 8) ? 8 : 16;         // lanes cooperating on one (token, head)
    const long total = (long)a.T * a.Hq;
    const int sub = threadIdx.x % GRP;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) / GRP; i < total; i += (long)gridDim.x * 256 / GRP) {
        const long t = i / a.Hq; const int h = (int)(i % a.Hq);   // group-uniform: all GRP lanes share i
        float s = 0.f;
        if (sub < PER) {
            const uint4 x = *(const uint4*)(a.o + t * a.o_stride + (long)h * D + sub * 8);
            const uint4 y = *(const uint4*)(a.d_o + t * a.o_stride + (long)h * D + sub * 8);
            s = bf_lo(x.x) * bf_lo(y.x) + bf_hi(x.x) * bf_hi(y.x) + bf_lo(x.y) * bf_lo(y.y) + bf_hi(x.y) * bf_hi(y.y) +
                bf_lo(x.z) * bf_lo(y.z) + bf_hi(x.z) * bf_hi(y.z) + bf_lo(x.w) * bf_lo(y.w) + bf_hi(x.w) * bf_hi(y.w);
        }
#pragma unroll
        for (int o = GRP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0) a.delta[(long)h * a.T + t] = s;
    }
}

// ================================================================================================ dK/dV, pipelined form
// dK / dV by key-block owners with the forward "pipe" treatment: Q / dO tiles (and the 64 lse / delta values of the tile) go HBM -> LDS
// by global_load_lds into two buffers, one barrier per (head, query tile); row-major fragments, the row statistics and the
// transposing reads are inline-asm LDS reads issued one batch ahead of the MFMAs with counted lgkmcnt waits; addresses hoisted.
// Both images are read row-major (ds_read_b128: S, dP) AND through the transposing read (dV, dK), so they use at_swz2: the
// 16-byte chunk index is XORed with s(row) = (((row & 7) ^ ((row & 8) >> 1)) << 1) | ((row & 8) >> 3), which is conflict-free
// for both (b128 lane groups {0-3,12-15 | 20-27} see 16 distinct chunks; the 8 rows of one transposing-read cycle see 8
// distinct chunk pairs) -- with row & 15 every transposing read is a 2-way conflict.
__device__ __forceinline__ int at_swz2(int row) { return ((((row & 7) ^ ((row & 8) >> 1)) << 1) | ((row & 8) >> 3)); }

template <int DC, int QF>
__device__ __forceinline__ void dkv_rm_issue(attn_u32x4 (&qr)[4], attn_u32x4 (&dr)[4], attn_u32x4& ls, attn_u32x4& dl,
                                             unsigned qa0, unsigned sa) {
    constexpr int O = QF * 16 * AT_RM_ROW_BYTES;                             // the dO image sits AT_RM_BYTES behind the Q image
    // contraction chunk dc: the chunk field of the address (bits 4-7) ^ (dc << 6); buffer bases are 256-byte aligned
    const unsigned qa1 = qa0 ^ (1 << 6), qa2 = qa0 ^ (2 << 6), qa3 = qa0 ^ (3 << 6);
    lds_b128<O>(qr[0], qa0); lds_b128<O + AT_RM_BYTES>(dr[0], qa0);
    lds_b128<O>(qr[1], qa1); lds_b128<O + AT_RM_BYTES>(dr[1], qa1);
    lds_b128<O>(qr[2], qa2); lds_b128<O + AT_RM_BYTES>(dr[2], qa2);
    if constexpr (DC > 3) { lds_b128<O>(qr[3], qa3); lds_b128<O + AT_RM_BYTES>(dr[3], qa3); }
    lds_b128<QF * 64>(ls, sa); lds_b128<QF * 64 + 256>(dl, sa);
    __builtin_amdgcn_sched_barrier(0);
}
// transposed fragments of d block DFI: t[0..3] = dO (c = 0 lo, hi; c = 1 lo, hi), t[4..7] = Q
template <int DFI>
__device__ __forceinline__ void dkv_tr_issue(attn_u32x2 (&t)[8], unsigned qa_tr) {
    const unsigned aq = qa_tr ^ (DFI << 5);
    lds_tr64<AT_RM_BYTES>(t[0], aq); lds_tr64<AT_RM_BYTES + 16 * AT_RM_ROW_BYTES>(t[1], aq);
    lds_tr64<AT_RM_BYTES + 32 * AT_RM_ROW_BYTES>(t[2], aq); lds_tr64<AT_RM_BYTES + 48 * AT_RM_ROW_BYTES>(t[3], aq);
    lds_tr64<0>(t[4], aq); lds_tr64<16 * AT_RM_ROW_BYTES>(t[5], aq); lds_tr64<32 * AT_RM_ROW_BYTES>(t[6], aq); lds_tr64<48 * AT_RM_ROW_BYTES>(t[7], aq);
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_pipe_kernel(AttnArgs a) {
    constexpr int DC = (D + 31) / 32, DF = D / 16;
    constexpr int BUF = 2 * AT_RM_BYTES, STAT0 = 2 * BUF, RB = 2 * DC + 2;     // [2][Q | dO] images, then [2][lse 64 | delta 64] floats
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkb = a.nqb;
    const int ks_id = blockIdx.x / nkb, kb = blockIdx.x % nkb, hk = blockIdx.y, qs_id = a.causal ? (int)blockIdx.z : ks_id;
    const spacer_attn_segment ks = a.segs[ks_id], qs = a.segs[qs_id];
    const int kb0 = kb * BKV;
    if (kb0 >= ks.q_len) return;
    const int key_abs0 = ks.q_start + kb0;
    const bool own = (ks_id == qs_id);
    int kv_lo = key_abs0, kv_hi = min(key_abs0 + BKV, ks.q_start + ks.q_len);
    if (!own) {
        kv_lo = max(kv_lo, qs.pre_start); kv_hi = min(kv_hi, qs.pre_start + qs.pre_len);
        if (kv_lo >= kv_hi) return;
    }
    const int l15 = lane & 15, g = lane >> 4;
    const int my_key_abs = key_abs0 + wave * 16 + l15;
    const int my_key_c = min(my_key_abs, ks.q_start + ks.q_len - 1);
    const bool key_ok = my_key_abs >= kv_lo && my_key_abs < kv_hi;
    const int my_key_rel = kb0 + wave * 16 + l15;

    bf16x8 kf[DC], vf[DC];
    {
        const bf16_t* kp = a.k + (long)my_key_c * a.kv_stride + (long)hk * D;
        const bf16_t* vp = a.v + (long)my_key_c * a.kv_stride + (long)hk * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) { kf[dc] = frag_global<D>(kp, dc, lane); vf[dc] = frag_global<D>(vp, dc, lane); }
    }
    f32x4 dka[DF], dva[DF];
#pragma unroll
    for (int d = 0; d < DF; ++d) { dka[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dva[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int q_first = (own && a.causal) ? (kb0 / BKV) * BKV : 0;
    const int rep = a.Hq / a.Hkv;
    const float c2 = a.scale * 1.4426950408889634f, log2e = 1.4426950408889634f;
    const int q_tiles = (qs.q_len - q_first + BKV - 1) / BKV, n_iter = rep * q_tiles;
    const bool keys_full = (kv_lo == key_abs0) && (kv_hi == key_abs0 + BKV);
    if (n_iter <= 0) return;

    // ---- hoisted LDS addresses (relative to a buffer): row-major fragment (row qf*16 + l15, chunk dc*4 + g) = rx ^ (dc << 6),
    // qf*4096 immediate; transposed fragment (rows 32c + 4g + (i>>2) [+16], chunk db*2 + ((i&3)>>1)) = tx ^ (db << 5)
    const unsigned smem0 = (unsigned)(uintptr_t)smem;
    const unsigned rx = (unsigned)(l15 * AT_RM_ROW_BYTES + ((g ^ at_swz2(l15)) << 4));
    const int trr = 4 * g + (l15 >> 2);
    const unsigned tx = (unsigned)(trr * AT_RM_ROW_BYTES + ((((l15 & 3) >> 1) ^ at_swz2(trr)) << 4) + (l15 & 1) * 8);
    // ---- DMA: waves 0,1 fill the Q image, waves 2,3 the dO image; piece i = rows (wave&1)*32 + 4i + g
    const bool is_do = wave >= 2;
    const bf16_t* gsrc = is_do ? a.d_o : a.q;
    const unsigned stride_b = (unsigned)(is_do ? a.o_stride : a.q_stride) * 2u;
    const int img_off = (is_do ? AT_RM_BYTES : 0) + (wave & 1) * 8 * 1024;
    const int prow = (wave & 1) * 32 + g;
    auto dchunk = [&](int j) {                                                // logical 16-byte chunk at this lane's LDS position
        int c = l15 ^ at_swz2(prow + 4 * j);
        return (unsigned)(c >= D / 8 ? 0 : c) * 16u;
    };
    unsigned doff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) doff[j] = (unsigned)(prow + 4 * j) * stride_b + dchunk(j);
    const float* stat_src = (wave == 0) ? a.lse : a.delta;       // wave 0: lse row, wave 2: delta row (4-byte DMA per lane)
    auto issue_iter = [&](int it, int buf) {
        const int h = hk * rep + it / q_tiles, q0 = q_first + (it % q_tiles) * BKV;
        const long tok0 = qs.q_start + q0;
        const int valid = qs.q_len - q0;                                      // >= 1
        const char* base = (const char*)(gsrc + tok0 * (is_do ? a.o_stride : a.q_stride) + (long)h * D);
        char* dst = smem + buf * BUF + img_off;
        if (valid >= BKV) {
            const char* base16 = base + 16l * stride_b;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((i < 4 ? base : base16) + doff[i & 3]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int row = prow + 4 * i;
                row = row < valid ? row : valid - 1;                          // finite filler; those rows are masked
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)row * stride_b + dchunk(i & 3)),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
        if ((wave & 1) == 0) {
            const long tk = min(tok0 + lane, (long)qs.q_start + qs.q_len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stat_src + (long)h * a.T + tk),
                                             (__attribute__((address_space(3))) void*)(smem + STAT0 + buf * 512 + (wave ? 256 : 0)), 4, 0, 0);
        }
    };

    issue_iter(0, 0);
    for (int it = 0; it < n_iter; ++it) {
        const int q0 = q_first + (it % q_tiles) * BKV;
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned qb_ = smem0 + (unsigned)(buf * BUF);
        const unsigned qa = qb_ + rx;
        const unsigned sa = smem0 + (unsigned)(STAT0 + buf * 512 + g * 16);
        const unsigned qa_tr = qb_ + tx;
        attn_u32x4 qr[2][4], dr[2][4], ls[2], dl[2];
        dkv_rm_issue<DC, 0>(qr[0], dr[0], ls[0], dl[0], qa, sa);
        dkv_rm_issue<DC, 1>(qr[1], dr[1], ls[1], dl[1], qa, sa);
        if (it + 1 < n_iter) issue_iter(it + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool need_mask = !keys_full || (q0 + BKV > qs.q_len) || (own && a.causal && kb0 + BKV - 1 > q0);

        float pv[2][4], dsv[2][4];                                            // the current pair of query fragments
        bf16x8 pf0, pf1, sf0, sf1;
        attn_u32x2 tr[2][8];
        // S / dP / P / dS for the 16-row query fragment QF from batch slot QF & 1
        auto phase1 = [&](auto QFc) {
            constexpr int QF = decltype(QFc)::value;
            f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qr[QF & 1][dc]), kf[dc], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, dr[QF & 1][dc]), vf[dc], dp, 0, 0, 0);
            }
            if constexpr (QF == 3) {          // last row-major batch consumed: the first transposed batches ride under this VALU block
                __builtin_amdgcn_sched_barrier(0);
                dkv_tr_issue<0>(tr[0], qa_tr);
                dkv_tr_issue<1>(tr[1], qa_tr);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr_ = QF * 16 + g * 4 + r, qi = q0 + qr_;           // lane holds S[q = qr_][key = l15]
                float sv = st[r];
                if (need_mask) sv = (key_ok && qi < qs.q_len && !(own && a.causal && my_key_rel > qi)) ? sv : -INFINITY;
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c2, -(__uint_as_float(ls[QF & 1][r]) * log2e)));
                pv[QF & 1][r] = p;
                dsv[QF & 1][r] = p * (dp[r] - __uint_as_float(dl[QF & 1][r])) * a.scale;
            }
            if constexpr (QF == 1) { pf0 = pack_slots(pv[0], pv[1]); sf0 = pack_slots(dsv[0], dsv[1]); }
            if constexpr (QF == 3) { pf1 = pack_slots(pv[0], pv[1]); sf1 = pack_slots(dsv[0], dsv[1]); }
            __builtin_amdgcn_sched_barrier(0);
        };
        wait_lgkm<RB>();
        phase1(std::integral_constant<int, 0>{});
        dkv_rm_issue<DC, 2>(qr[0], dr[0], ls[0], dl[0], qa, sa);
        wait_lgkm<RB>();
        phase1(std::integral_constant<int, 1>{});
        dkv_rm_issue<DC, 3>(qr[1], dr[1], ls[1], dl[1], qa, sa);
        wait_lgkm<RB>();
        phase1(std::integral_constant<int, 2>{});
        wait_lgkm<0>();
        phase1(std::integral_constant<int, 3>{});

        auto phase2 = [&](auto DFc) {
            constexpr int DFI = decltype(DFc)::value;
            if constexpr (DFI < DF) {
                if constexpr (DFI + 1 < DF) wait_lgkm<8>(); else wait_lgkm<0>();
                const attn_u32x2(&t)[8] = tr[DFI & 1];
                const bf16x8 d0 = __builtin_bit_cast(bf16x8, make_uint4(t[0][0], t[0][1], t[1][0], t[1][1]));
                const bf16x8 d1 = __builtin_bit_cast(bf16x8, make_uint4(t[2][0], t[2][1], t[3][0], t[3][1]));
                const bf16x8 q0f = __builtin_bit_cast(bf16x8, make_uint4(t[4][0], t[4][1], t[5][0], t[5][1]));
                const bf16x8 q1f = __builtin_bit_cast(bf16x8, make_uint4(t[6][0], t[6][1], t[7][0], t[7][1]));
                dva[DFI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, pf0, dva[DFI], 0, 0, 0);
                dka[DFI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, sf0, dka[DFI], 0, 0, 0);
                dva[DFI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, pf1, dva[DFI], 0, 0, 0);
                dka[DFI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, sf1, dka[DFI], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DFI + 2 < DF) dkv_tr_issue<DFI + 2>(tr[DFI & 1], qa_tr);
            }
        };
        phase2(std::integral_constant<int, 0>{}); phase2(std::integral_constant<int, 1>{}); phase2(std::integral_constant<int, 2>{});
        phase2(std::integral_constant<int, 3>{}); phase2(std::integral_constant<int, 4>{}); phase2(std::integral_constant<int, 5>{});
        phase2(std::integral_constant<int, 6>{}); phase2(std::integral_constant<int, 7>{});
    }
    if (key_ok) {
        float* kp = a.dk + ((long)my_key_abs * a.Hkv + hk) * D;
        float* vp = a.dv + ((long)my_key_abs * a.Hkv + hk) * D;
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                atomicAdd(kp + df * 16 + g * 4 + r, dka[df][r]);
                atomicAdd(vp + df * 16 + g * 4 + r, dva[df][r]);
            }
    }
}

// ================================================================================================ dQ, pipelined form
// dQ by query-block owners with LDS-DMA K / V tiles (two buffer pairs, one barrier per tile) and pipelined fragment reads.
// NF = 16-row query fragments per wave.  With NF = 2 every K / V / K^T fragment read from LDS feeds two MFMAs instead of one (at
// NF = 1 the three products need 192 LDS cycles per 768 MFMA cycles per wave, i.e. at full MFMA rate the LDS would be as busy as
// the four SIMDs together), but at D = 128 that needs ~280 VGPRs (30 spilled); D = 128 runs NF = 1, D = 80 NF = 2.
// K image: row-major AND transposing reads -> at_swz2.
template <int DC, int KF, int IMG>
__device__ __forceinline__ void dq_rm_issue(attn_u32x4 (&r)[4], unsigned a0) {
    constexpr int O = KF * 16 * AT_RM_ROW_BYTES + IMG * AT_RM_BYTES;
    const unsigned a1 = a0 ^ (1 << 6), a2 = a0 ^ (2 << 6), a3 = a0 ^ (3 << 6);
    lds_b128<O>(r[0], a0); lds_b128<O>(r[1], a1); lds_b128<O>(r[2], a2);
    if constexpr (DC > 3) lds_b128<O>(r[3], a3);
    __builtin_amdgcn_sched_barrier(0);
}
// K^T fragments of d blocks 2J, 2J+1: t[4*e + 2*c + hi]
template <int DF, int J>
__device__ __forceinline__ void dq_tr_issue(attn_u32x2 (&t)[8], unsigned ta) {
    if constexpr (2 * J < DF) {
        const unsigned a0 = ta ^ ((2 * J) << 5);
        lds_tr64<0>(t[0], a0); lds_tr64<16 * AT_RM_ROW_BYTES>(t[1], a0); lds_tr64<32 * AT_RM_ROW_BYTES>(t[2], a0); lds_tr64<48 * AT_RM_ROW_BYTES>(t[3], a0);
    }
    if constexpr (2 * J + 1 < DF) {
        const unsigned a1 = ta ^ ((2 * J + 1) << 5);
        lds_tr64<0>(t[4], a1); lds_tr64<16 * AT_RM_ROW_BYTES>(t[5], a1); lds_tr64<32 * AT_RM_ROW_BYTES>(t[6], a1); lds_tr64<48 * AT_RM_ROW_BYTES>(t[7], a1);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int DF, int J> constexpr int dq_tr_reads() { return 2 * J >= DF ? 0 : (2 * J + 1 < DF ? 8 : 4); }

template <int D, int NF>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_pipe_kernel(AttnArgs a) {
    constexpr int DC = (D + 31) / 32, DF = D / 16, BQD = 64 * NF, BUF = 2 * AT_RM_BYTES, NTB = (DF + 1) / 2;
    static_assert(NTB >= 3 && NTB <= 4, "K^T batch schedule below is written for 3 or 4 batches");
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][K | V]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seg_id = a.lpt ? a.num_segs - 1 - blockIdx.x / a.nqb : blockIdx.x / a.nqb;
    const int qb = a.lpt ? a.nqb - 1 - blockIdx.x % a.nqb : blockIdx.x % a.nqb;
    const int h = blockIdx.y, hk = h / (a.Hq / a.Hkv);
    const spacer_attn_segment seg = a.segs[seg_id];
    const int qb0 = qb * BQD;
    if (qb0 >= seg.q_len) return;
    const int l15 = lane & 15, g = lane >> 4;
    const int wq0 = qb0 + wave * 16 * NF;

    bf16x8 qf[NF][DC], dof[NF][DC];
    float lse2_q[NF], dl_q[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        int qi = wq0 + f * 16 + l15;
        qi = qi < seg.q_len ? qi : seg.q_len - 1;
        const long tok = seg.q_start + qi;
        const bf16_t* qp = a.q + tok * a.q_stride + (long)h * D;
        const bf16_t* dp = a.d_o + tok * a.o_stride + (long)h * D;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) { qf[f][dc] = frag_global<D>(qp, dc, lane); dof[f][dc] = frag_global<D>(dp, dc, lane); }
        lse2_q[f] = a.lse[(long)h * a.T + tok] * 1.4426950408889634f;
        dl_q[f] = a.delta[(long)h * a.T + tok];
    }
    f32x4 acc[NF][DF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int d = 0; d < DF; ++d) acc[f][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int own_len = a.causal ? min(seg.q_len, qb0 + BQD) : seg.q_len;
    const int n_pre = (seg.pre_len + BKV - 1) / BKV, n_tiles = n_pre + (own_len + BKV - 1) / BKV;
    const float c2 = a.scale * 1.4426950408889634f;

    const unsigned smem0 = (unsigned)(uintptr_t)smem;
    const unsigned rx = (unsigned)(l15 * AT_RM_ROW_BYTES + ((g ^ at_swz2(l15)) << 4));
    const int trr = 4 * g + (l15 >> 2);
    const unsigned tx = (unsigned)(trr * AT_RM_ROW_BYTES + ((((l15 & 3) >> 1) ^ at_swz2(trr)) << 4) + (l15 & 1) * 8);
    const bf16_t* gsrc = (wave >= 2 ? a.v : a.k) + (long)hk * D;
    const int img_off = (wave >= 2 ? AT_RM_BYTES : 0) + (wave & 1) * 8 * 1024;
    const int prow = (wave & 1) * 32 + g;
    const unsigned stride_b = (unsigned)a.kv_stride * 2u;
    auto dchunk = [&](int j) {
        int c = l15 ^ at_swz2(prow + 4 * j);
        return (unsigned)(c >= D / 8 ? 0 : c) * 16u;
    };
    unsigned doff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) doff[j] = (unsigned)(prow + 4 * j) * stride_b + dchunk(j);
    auto issue_tile = [&](int t, int buf) {
        const KeyTile kt = key_tile(seg, n_pre, own_len, t);
        const int valid = kt.len - kt.rel0;
        const char* base = (const char*)(gsrc + (long)(kt.start_abs + kt.rel0) * a.kv_stride);
        char* dst = smem + buf * BUF + img_off;
        if (valid >= BKV) {
            const char* base16 = base + 16l * stride_b;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((i < 4 ? base : base16) + doff[i & 3]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int row = prow + 4 * i;
                row = row < valid ? row : valid - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)row * stride_b + dchunk(i & 3)),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
    };

    issue_tile(0, 0);
    for (int t = 0; t < n_tiles; ++t) {
        const KeyTile kt = key_tile(seg, n_pre, own_len, t);
        const int buf = t & 1;
        const bool active = !(kt.own && a.causal && kt.rel0 > wq0 + 16 * NF - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned ra = smem0 + (unsigned)(buf * BUF) + rx, ta = smem0 + (unsigned)(buf * BUF) + tx;
        attn_u32x4 kr[4], vr[4];
        if (active) {
            dq_rm_issue<DC, 0, 0>(kr, ra);
            dq_rm_issue<DC, 0, 1>(vr, ra);
        }
        if (t + 1 < n_tiles) issue_tile(t + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!active) continue;
        const bool need_mask = (kt.rel0 + BKV > kt.len) || (kt.own && a.causal && kt.rel0 + BKV - 1 > wq0);

        float dsv[NF][4];                                                     // dS of the even key block, until its odd partner is done
        bf16x8 dsf[NF][2];
        attn_u32x2 tr[2][8];
        auto step = [&](auto KFc) {
            constexpr int KF = decltype(KFc)::value;
            f32x4 st[NF], dp[NF];
            wait_lgkm<DC>();                                                  // K(KF) landed (V(KF) may still be in flight)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                st[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dc = 0; dc < DC; ++dc)
                    st[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kr[dc]), qf[f][dc], st[f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KF < 3) { dq_rm_issue<DC, KF + 1, 0>(kr, ra); wait_lgkm<DC>(); }
            else wait_lgkm<0>();
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                dp[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dc = 0; dc < DC; ++dc)
                    dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vr[dc]), dof[f][dc], dp[f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KF < 3) dq_rm_issue<DC, KF + 1, 1>(vr, ra);
            else dq_tr_issue<DF, 0>(tr[0], ta);                               // first K^T batch rides under the last VALU block
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int qi = wq0 + f * 16 + l15;
                float dso[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv = st[f][r];
                    if (need_mask) {
                        const int kr_ = kt.rel0 + KF * 16 + g * 4 + r;
                        sv = (kr_ < kt.len && !(kt.own && a.causal && kr_ > qi)) ? sv : -INFINITY;   // p = 0
                    }
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c2, -lse2_q[f]));
                    dso[r] = p * (dp[f][r] - dl_q[f]) * a.scale;
                }
                if constexpr (KF & 1) dsf[f][KF >> 1] = pack_slots(dsv[f], dso);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dsv[f][r] = dso[r];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        dq_tr_issue<DF, 1>(tr[1], ta);
        auto phase2 = [&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            if constexpr (J < NTB) {
                wait_lgkm<dq_tr_reads<DF, J + 1>()>();
                const attn_u32x2(&tt)[8] = tr[J & 1];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int df = 2 * J + e;
                    if (df < DF) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const bf16x8 ktf = __builtin_bit_cast(bf16x8, make_uint4(tt[4 * e + 2 * c][0], tt[4 * e + 2 * c][1], tt[4 * e + 2 * c + 1][0], tt[4 * e + 2 * c + 1][1]));
#pragma unroll
                            for (int f = 0; f < NF; ++f) acc[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[f][c], acc[f][df], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (J + 2 < NTB) dq_tr_issue<DF, J + 2>(tr[J & 1], ta);
            }
        };
        phase2(std::integral_constant<int, 0>{}); phase2(std::integral_constant<int, 1>{});
        phase2(std::integral_constant<int, 2>{}); phase2(std::integral_constant<int, 3>{});
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int qi = wq0 + f * 16 + l15;
        if (qi >= seg.q_len) continue;
        bf16_t* op = a.dq + (long)(seg.q_start + qi) * a.q_stride + (long)h * D;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const f32x4 v = acc[f][df];
            *(uint2*)(op + df * 16 + g * 4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    }
}

int check_common(const char* who, long q_stride, long kv_stride, long o_stride, int Hq, int Hkv, int D) {
    SP_REQUIRE(D == 80 || D == 128, SPACER_EINVAL, "%s: head_dim %d unsupported (80 or 128)", who, D);
    SP_REQUIRE(Hkv > 0 && Hq % Hkv == 0, SPACER_EINVAL, "%s: Hq=%d not a multiple of Hkv=%d", who, Hq, Hkv);
    SP_REQUIRE(q_stride % 8 == 0 && kv_stride % 8 == 0 && o_stride % 8 == 0, SPACER_EINVAL,
               "%s: token strides must keep 16-byte aligned rows", who);
    return SPACER_OK;
}

}  // namespace

extern "C" int spacer_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, long q_stride,
                               long kv_stride, long o_stride, const spacer_attn_segment* segs_dev, int num_segs,
                               int max_q_len, int T, int Hq, int Hkv, int D, int causal, float scale,
                               spacer_stream_t stream) {
    if (int rc = check_common("attn_fwd", q_stride, kv_stride, o_stride, Hq, Hkv, D)) return rc;
    if (num_segs <= 0 || max_q_len <= 0) return SPACER_OK;
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
    a.q_stride = q_stride; a.kv_stride = kv_stride; a.o_stride = o_stride; a.segs = segs_dev; a.num_segs = num_segs;
    a.nqb = cdiv(max_q_len, BQ); a.T = T; a.Hq = Hq; a.Hkv = Hkv; a.causal = causal; a.scale = scale;
    a.lpt = 1;                                         // longest blocks first
    const dim3 grid(num_segs * a.nqb, Hq);
    constexpr int LDS = 4 * AT_RM_BYTES;
    static const int once = hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                          + hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SP_REQUIRE(once == 0, SPACER_ELAUNCH, "attn_fwd: cannot raise the dynamic LDS limit to %d bytes", LDS);
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) hipLaunchKernelGGL(attn_fwd_pipe_kernel<128>, grid, dim3(256), LDS, s, a);
    else hipLaunchKernelGGL(attn_fwd_pipe_kernel<80>, grid, dim3(256), LDS, s, a);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                               const float* lse, float* delta, void* dq, float* dk, float* dv, long q_stride,
                               long kv_stride, long o_stride, const spacer_attn_segment* segs_dev, int num_segs,
                               int max_q_len, int T, int Hq, int Hkv, int D, int causal, float scale,
                               spacer_stream_t stream) {
    if (int rc = check_common("attn_bwd", q_stride, kv_stride, o_stride, Hq, Hkv, D)) return rc;
    if (num_segs <= 0 || max_q_len <= 0) return SPACER_OK;
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = (float*)lse;
    a.d_o = (const bf16_t*)d_o; a.delta = delta; a.dq = (bf16_t*)dq; a.dk = dk; a.dv = dv;
    a.q_stride = q_stride; a.kv_stride = kv_stride; a.o_stride = o_stride; a.segs = segs_dev; a.num_segs = num_segs;
    a.T = T; a.Hq = Hq; a.Hkv = Hkv; a.causal = causal; a.scale = scale;
    a.lpt = 1;
    hipStream_t s = (hipStream_t)stream;
    const int dgrid = (int)(((long)T * Hq * 16 + 255) / 256 < 4096 ? ((long)T * Hq * 16 + 255) / 256 : 4096);
    a.nqb = cdiv(max_q_len, 64);                       // dQ at D = 128: 64 query rows per workgroup (16 per wave)
    const dim3 qgrid(num_segs * a.nqb, Hq);
    AttnArgs b = a;
    b.nqb = cdiv(max_q_len, BKV);
    const dim3 kgrid(num_segs * b.nqb, Hkv, causal ? num_segs : 1);
    constexpr int PIPE_LDS = 4 * AT_RM_BYTES + 1024, DQ_LDS = 4 * AT_RM_BYTES;
    AttnArgs ap = a;                                   // dQ at D = 80: 128 query rows per workgroup
    ap.nqb = cdiv(max_q_len, 128);
    const dim3 pgrid(num_segs * ap.nqb, Hq);
    static const int once = hipFuncSetAttribute((const void*)attn_bwd_dq_pipe_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS)
                          + hipFuncSetAttribute((const void*)attn_bwd_dq_pipe_kernel<80, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS)
                          + hipFuncSetAttribute((const void*)attn_bwd_dkv_pipe_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS)
                          + hipFuncSetAttribute((const void*)attn_bwd_dkv_pipe_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
    SP_REQUIRE(once == 0, SPACER_ELAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");
    if (D == 128) {
        hipLaunchKernelGGL(attn_delta_kernel<128>, dim3(dgrid), dim3(256), 0, s, a);
        hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<128, 1>), qgrid, dim3(256), DQ_LDS, s, a);
        hipLaunchKernelGGL(attn_bwd_dkv_pipe_kernel<128>, kgrid, dim3(256), PIPE_LDS, s, b);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<80>, dim3(dgrid), dim3(256), 0, s, a);
        hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<80, 2>), pgrid, dim3(256), DQ_LDS, s, ap);
        hipLaunchKernelGGL(attn_bwd_dkv_pipe_kernel<80>, kgrid, dim3(256), PIPE_LDS, s, b);
    }
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
