// Shared LDS tile layouts for the attention kernels (forward, dQ, dKV).
//
// Two images of a 64-row x D bf16 tile:
//   row-major  RM : 256-byte rows (128 bf16 slots; D=80 uses chunks 0..9 and zero-fills 10,11), 16-byte chunk
//                   index ^= (row & 15)  -> conflict-free ds_read_b128 of MFMA operand fragments
//   transposed T  : [D rows][64 cols] (128-byte rows), 8-byte "row-quad" slot ^= vswz(d) -> conflict-free for
//                   the transposing ds_write_b64 and the fragment ds_read_b64
// stage_tile<D,RM,T> moves one tile HBM -> registers (tile_load) -> LDS (tile_store); the split lets callers
// keep the global loads in flight under MFMAs (cdna_hip_programming T14).
#pragma once
#include "common.h"

constexpr int AT_ROWS = 64;
constexpr int AT_RM_ROW_BYTES = 256;
constexpr int AT_RM_BYTES = AT_ROWS * AT_RM_ROW_BYTES;   // 16 KiB
constexpr int AT_T_ROW_BYTES = AT_ROWS * 2;              // 128 B
#define AT_T_BYTES(D) ((D) * AT_T_ROW_BYTES)

__device__ __forceinline__ int vswz(int d) {
    const int m = (d >> 1) & 7;
    return ((2 * (m ^ (m >> 2))) ^ (d >> 3)) & 15;
}

// thread -> (16-byte chunk c = tid & 15, row quad rq = tid >> 4); loads rows rq*4 .. rq*4+3.
// rows >= valid_rows are zero.  src points at row 0 of the tile for this head (row_stride in elements).
template <int D>
__device__ __forceinline__ void tile_load(uint4 (&reg)[4], const bf16_t* __restrict__ src, long row_stride,
                                          int valid_rows, int tid) {
    const int c = tid & 15, rq = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = rq * 4 + j;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < D / 8 && row < valid_rows) v = *(const uint4*)(src + (long)row * row_stride + c * 8);
        reg[j] = v;
    }
}

template <int D, bool RM, bool TR>
__device__ __forceinline__ void tile_store(const uint4 (&reg)[4], char* rm_lds, char* t_lds, int tid) {
    const int c = tid & 15, rq = tid >> 4;
    if (RM) {
        // D=80: chunks 10, 11 hold the zero padding of the third 32-wide contraction step (reg is zero there)
        if (c < (D == 80 ? 12 : D / 8)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rq * 4 + j;
                *(uint4*)(rm_lds + row * AT_RM_ROW_BYTES + ((c ^ (row & 15)) * 16)) = reg[j];
            }
        }
    }
    if (TR) {
        if (c < D / 8) {
            const uint32_t w[4][4] = {{reg[0].x, reg[0].y, reg[0].z, reg[0].w}, {reg[1].x, reg[1].y, reg[1].z, reg[1].w},
                                      {reg[2].x, reg[2].y, reg[2].z, reg[2].w}, {reg[3].x, reg[3].y, reg[3].z, reg[3].w}};
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // dim d = c*8 + i : pack rows 4rq .. 4rq+3
                const int d = c * 8 + i, wi = i >> 1, hi = i & 1;
                uint32_t e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = hi ? (w[j][wi] >> 16) : (w[j][wi] & 0xffffu);
                *(uint2*)(t_lds + d * AT_T_ROW_BYTES + ((rq ^ vswz(d)) * 8)) = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
            }
        }
    }
}

// MFMA 16x16x32 operand fragment from the row-major image: rows rb*16 + (lane&15), dims dc*32 + (lane>>4)*8 ..
__device__ __forceinline__ bf16x8 frag_rm(const char* rm_lds, int rb, int dc, int lane) {
    const int row = rb * 16 + (lane & 15), ch = dc * 4 + (lane >> 4);
    return *(const bf16x8*)(rm_lds + row * AT_RM_ROW_BYTES + ((ch ^ (row & 15)) * 16));
}
// operand fragment from the transposed image: d = db*16 + (lane&15); contraction slots j=0..3 <- rows
// (2c)*16 + 4g + j, j=4..7 <- rows (2c+1)*16 + 4g + (j-4), g = lane>>4 (matches the C-layout of two stacked
// 16x16 MFMA results, so a just-computed score tile feeds the next MFMA without any lane exchange).
__device__ __forceinline__ bf16x8 frag_t(const char* t_lds, int db, int c, int lane) {
    const int d = db * 16 + (lane & 15), g = lane >> 4, sw = vswz(d);
    const uint2 lo = *(const uint2*)(t_lds + d * AT_T_ROW_BYTES + ((((2 * c) * 4 + g) ^ sw) * 8));
    const uint2 hi = *(const uint2*)(t_lds + d * AT_T_ROW_BYTES + ((((2 * c + 1) * 4 + g) ^ sw) * 8));
    return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// The same operand fragment as frag_t, but read from the ROW-MAJOR image with gfx950's transposing LDS read
// (ds_read_b64_tr_b16): inside each 16-lane group the lanes supply the addresses of a [4 rows][16 columns] block -- lane i
// the 8 bytes of row i >> 2, columns 4 (i & 3) .. +3 -- and lane i receives column i of the 4 rows.  Group g asks for rows
// (2c)*16 + 4g .. +3 (slots 0..3) and (2c+1)*16 + 4g .. +3 (slots 4..7), columns db*16 .. +15, so no transposed image, no
// transposing store pass (VALU repacking + ds_write_b64) is needed at all.  (Semantics: scripts/probes/tr_probe.hip.)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ bf16x8 frag_tr(const char* rm_lds, int db, int c, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int chunk = db * 2 + ((i & 3) >> 1), sub = (i & 1) * 8;
    const int r0 = (2 * c) * 16 + 4 * g + (i >> 2), r1 = r0 + 16;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(rm_lds + r0 * AT_RM_ROW_BYTES + ((chunk ^ (r0 & 15)) * 16) + sub));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(rm_lds + r1 * AT_RM_ROW_BYTES + ((chunk ^ (r1 & 15)) * 16) + sub));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}
// pack two stacked C-layout tiles (rows 2c*16.., (2c+1)*16..) into the matching contraction-slot operand
__device__ __forceinline__ bf16x8 pack_slots(const float (&a)[4], const float (&b)[4]) {
    return __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])));
}
// operand fragment straight from global memory (row-per-lane, 16 B): rows row(lane&15), dims dc*32+(lane>>4)*8
template <int D>
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* __restrict__ rowp, int dc, int lane) {
    const int d0 = dc * 32 + (lane >> 4) * 8;
    uint4 t = make_uint4(0, 0, 0, 0);
    if (d0 < D) t = *(const uint4*)(rowp + d0);
    return __builtin_bit_cast(bf16x8, t);
}
