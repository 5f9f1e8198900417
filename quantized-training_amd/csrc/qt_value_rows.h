// The value pass of the table-format attention core (csrc/qt_attention_rows.hip): fq_v(V) as bf16 values, transposed to [d][key] with
// the keys of every 32-chunk permuted into the k-slot order of the core's P.V instruction (slot 8 g + 4 h + e <-> key 16 h + 4 g + e).
// One 256-thread workgroup converts a block of KB keys (128, or 64 where the launch also carries the rotary kernel and LDS per
// workgroup decides how many of them a CU holds) of one (batch, head); head_dim 128.  Shared by qt_value_t_rows and qt_rope_map_value.
#pragma once
#include <stdint.h>

#include "qt_device.h"

constexpr int kValueRowsD = 128;
template <int KB>
constexpr int value_rows_tile_elems() { return kValueRowsD * (KB + 8); }          // [d][key slot] uint16, rows padded by 16 bytes

// tile: value_rows_tile_elems<KB>() uint16 of LDS; rnd: the row form with its table in LDS; the caller has synchronised the table
template <int KB>
__device__ __forceinline__ void value_t_rows_block(uint16_t *tile, const Rounder<kFmtRows> &rnd, const uint16_t *v, uint16_t *vt, int H, long Sk,
                                                   long sb, long sh, long sk, long bh, int kb, int t) {
    constexpr int kD = kValueRowsD, kRow = KB + 8;
    const long b = bh / H, h = bh % H;
    // a thread takes TWO adjacent keys (k, k + 1: adjacent slots too) and one 8-wide d chunk, so every LDS write is a 32-bit
    // {fq(V)[k][d], fq(V)[k + 1][d]} pair into row d of the image
#pragma unroll
    for (int it = 0; it < (KB / 2) * (kD / 8) / 256; ++it) {
        const int item = it * 256 + t, kp = item % (KB / 2), dv = item / (KB / 2), key = 2 * kp;
        const uint16_t *src = v + b * sb + h * sh + ((long)kb * KB + key) * sk + dv * 8;
        const uint4 i0 = *(const uint4 *)src, i1 = *(const uint4 *)(src + sk);
        const uint32_t w0[4] = {i0.x, i0.y, i0.z, i0.w}, w1[4] = {i1.x, i1.y, i1.z, i1.w};
        uint32_t a[4], c[4];
        fq_rows_words<4, false>(w0, a, rnd);
        fq_rows_words<4, false>(w1, c, rnd);
        // key = 32 c + 16 hh + 4 gg + ee  ->  slot 32 c + 8 gg + 4 hh + ee (ee even here: slot even)
        const int p = (key & ~31) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3);
        uint32_t *base = (uint32_t *)(tile + (dv * 8) * kRow + p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            base[(2 * j) * (kRow / 2)] = (a[j] & 0xFFFFu) | (c[j] << 16);
            base[(2 * j + 1) * (kRow / 2)] = (a[j] >> 16) | (c[j] & 0xFFFF0000u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kD * KB / 8 / 256; ++it) {
        const int ci = it * 256 + t, d = ci / (KB / 8), ch = ci % (KB / 8);
        *(uint4 *)(vt + (bh * kD + d) * Sk + (long)kb * KB + ch * 8) = *(const uint4 *)(tile + d * kRow + ch * 8);
    }
}
