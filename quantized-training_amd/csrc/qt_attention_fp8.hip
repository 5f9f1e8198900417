// qt_attention_fp8.hip -- the attention core on FP8 codes for head_dim 128: Q.K^T, scaling, mask, softmax, fake-quant of the
// probabilities and P.V in ONE launch, the S x S tensors never written.
//
// Replaces, when all four fake-quantizers around the two matmuls are stateless E4M3 / E5M2 ones (unit scale),
//     av_matmul(fq_p(softmax(attn_scaling(qk_matmul(fq_q(q), fq_k(k)^T), scale) + mask)), fq_v(v))
//                                                     modules/quantizable/modeling_llama.py:228-246, modeling_bert.py:118-158
// with the module chain's rounding points kept: scores rounded to bf16 (the matmul's output dtype), bf16(score * scaling),
// bf16(. + mask), softmax in fp32 over the WHOLE row (no running rescale), probabilities rounded to bf16, their fake-quantizer
// (for a value in [0, 1] the format's round-to-nearest-even is the hardware conversion), P.V accumulated in fp32, output bf16.
//
// A workgroup = 4 waves owns 64 query rows of one (batch, head); a wave owns 16 of them and keeps ITS WHOLE score strip
// (16 x Sk, Sk <= 1024) in registers as packed bf16 -- Q.K^T is one v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile because
// head_dim = 128 is the instruction's depth.  The operands are multiplied swapped (K fragment first), so lane (r, g) ends up with
// the scores of query r against keys 16 t + 4 g + {0..3} of every tile t: the row statistics are lane-local plus two shuffles.
// Sweep 1 walks the keys in blocks of 128 (K block by LDS-DMA, double buffered) and fills the strip; the maximum and the sum of
// exponentials follow; sweep 2 walks the same blocks with V: the lane's 32 probability codes of a block ARE the B operand of the
// P.V instruction if the keys of a block are assigned to its k slots as slot 16 g + 4 t + e <-> key 16 t + 4 g + e -- so V arrives
// transposed ([128 d][Sk]) with that permutation inside every 128-key block (qt_value_codes_t writes it while fake-quantizing V),
// and its fragments are read exactly like K's.  Key blocks beyond every row's last unmasked column (qt_mask_row_live) are
// skipped in both sweeps: their probabilities are exactly 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_value_codes.h"
#include "qt_formats.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 128, kMaxBlocks = 8;                                    // one K / V^T block: 128 keys x 128 B = 16 KiB of codes
constexpr int kUnit = 127;                                                        // E8M0 2^0

struct AttnArgs {
    const uint8_t *q8, *k8, *vt8;       // [B][H][Sq][128], [B][H][Sk][128], [B][H][128][Sk] (keys permuted inside 128-blocks)
    const uint16_t *mask;               // additive bf16 or NULL; element strides for (b, h, q), columns contiguous
    long msb, msh, msq;
    const int *row_live;                // optional: one past the last unmasked column per mask row; row strides for (b, h, q)
    long lsb, lsh, lsq;
    int mask_simple;                    // the caller vouches: every mask row is exactly 0 up to its row_live entry and the bf16 minimum from
                                        // there on (causal, right padding) -- the mask is then not read at all
    uint16_t *out;                      // [B][Sq][H][128] bf16
    int H, Sq, Sk;
    float scaling;
    uint8_t *out8;                      // optional: the consumer's (output projection's input) stateless FP8 fake-quantizer applied on the way
    qt_format out_fmt;                  // out: `out` then holds fq(result), out8 its codes, same layout
    const int *mask_irregular;          // optional, device: 0 = qt_mask_row_live_checked found every row "zeros, then the minimum" (as mask_simple)
};

__device__ __forceinline__ float blo(uint32_t w) { return qt_u2f(w << 16); }
__device__ __forceinline__ float bhi(uint32_t w) { return qt_u2f(w & 0xFFFF0000u); }
__device__ __forceinline__ int chunk_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// LDS-DMA as inline asm (16 bytes per lane, 1 KiB per wave): hipcc's wait-count model then does not see a pending LDS access
// and keeps counted waits for the ordinary loads (mask, fragments); the DMA is waited for explicitly (vmcnt(0) + barrier)
__device__ __forceinline__ void dma16(const uint8_t *src, uint32_t dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
}

// F: operand format of q, k, v and p (0 = E4M3, 1 = E5M2)
//
// ---- the kernel: the two wave groups split the KEYS of one block of 64 query rows ---------------------------------------------
// Of every 128-key block group 0 takes tiles 0-3 (keys 0 .. 63) and group 1 tiles 4-7, so the two always carry the same work; an
// iteration walks a PAIR of blocks, which gives each wave 8 score tiles = the 128 probabilities one P.V instruction contracts (its
// low half from the first block, its high half from the second: with the slot <-> key permutation of the V^T blocks the group's tiles
// are exactly the low (group 0) or high (group 1) 16-byte chunks of a lane's fragment).  A wave's strip is at most 4 x 8 tiles, which
// leaves room to keep it in fp32: each exponential is evaluated once (round 2's first kernel -- two blocks of rows per workgroup,
// bf16 logits, exp evaluated in both sweeps: 35 against 21 us -- was removed in round 4).  The row
// maximum and sum and the partial P.V sums of the two groups meet in LDS.
//
// Control flow is wave-uniform throughout (extents are scalars), so the four tiles of a half-block form one basic block the
// scheduler can overlap: fragment reads, MFMA and the rounding chain of neighbouring tiles.  Under a row-extent mask a wave skips
// the tiles beyond the extents of its own 16 rows, and tiles wholly inside them take the path without mask arithmetic.
//
// Workgroups are ordered heaviest first over the whole launch (one per CU at a time: longest-first keeps the CUs evenly loaded).
typedef short v2s __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
// two packed bf16 pairs of probabilities -> four FP8 codes with the hardware conversion (non-saturating: right for values in [0, 1])
template <bool E5M2>
__device__ __forceinline__ uint32_t prob_codes(uint32_t p0, uint32_t p1) {
    v2s o = __builtin_bit_cast(v2s, p0);
    if constexpr (E5M2) {
        o = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(o, __builtin_bit_cast(v2bf, p0), 1.0f, false);
        o = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(o, __builtin_bit_cast(v2bf, p1), 1.0f, true);
    } else {
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o, __builtin_bit_cast(v2bf, p0), 1.0f, false);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o, __builtin_bit_cast(v2bf, p1), 1.0f, true);
    }
    return __builtin_bit_cast(uint32_t, o);
}

__device__ __forceinline__ float max3(float x, float y, float z) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}

// scores of the four tiles t0 .. t0 + 3 of one K block.  MODE 0: no mask arithmetic; 1: row-extent mask applied from my_live
// (tiles from `tiles` on lie beyond every row of the wave); 2: additive mask read from mrow
template <int F, int D, int MODE>
__device__ __forceinline__ void score_half(const uint8_t *blk, int t0, const v8i &qf, int f_lo, int f_hi, float scaling, float (*ev)[4], float &mx,
                                           const uint16_t *mrow, int key0, int my_live, int tiles) {
    // staged by hand -- all fragment reads, then the MFMAs back to back, then the rounding chains -- because the compiler keeps
    // source order and a read -> MFMA -> chain sequence per tile exposes the LDS and MFMA latencies four times over
    uint2 m[4];
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = *(const uint2 *)(mrow + key0 + j * 16);
    }
    u32x4 klo[4], khi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == 1 && j >= tiles) continue;
        klo[j] = *(const u32x4 *)(blk + (t0 + j) * (16 * D) + f_lo);
        if (D == 128) khi[j] = *(const u32x4 *)(blk + (t0 + j) * (16 * D) + f_hi);
        else khi[j] = u32x4{0u, 0u, 0u, 0u};                             // head_dim 64: the upper half of the 128-deep product is 0 x 0
    }
    v4f s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == 1 && j >= tiles) continue;
        const v8i kf = {(int)klo[j].x, (int)klo[j].y, (int)klo[j].z, (int)klo[j].w, (int)khi[j].x, (int)khi[j].y, (int)khi[j].z, (int)khi[j].w};
        s[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kf, qf, v4f{0.f, 0.f, 0.f, 0.f}, F, F, 0, kUnit, 0, kUnit);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == 1 && j >= tiles) {
            ev[j][0] = ev[j][1] = ev[j][2] = ev[j][3] = -INFINITY;
            continue;
        }
        uint32_t w0 = pack_bf16x2(s[j][0], s[j][1]), w1 = pack_bf16x2(s[j][2], s[j][3]);     // the matmul's bf16 output
        w0 = pack_bf16x2(blo(w0) * scaling, bhi(w0) * scaling);
        w1 = pack_bf16x2(blo(w1) * scaling, bhi(w1) * scaling);
        if (MODE == 2) {
            w0 = pack_bf16x2(blo(w0) + blo(m[j].x), bhi(w0) + bhi(m[j].x));
            w1 = pack_bf16x2(blo(w1) + blo(m[j].y), bhi(w1) + bhi(m[j].y));
        }
        float v[4] = {blo(w0), bhi(w0), blo(w1), bhi(w1)};
        if (MODE == 1) {
            // x + 0 = x; bf16(x + min) = min for every finite x (NaN stays NaN): the mask's effect without reading it
            const int left = my_live - (key0 + j * 16);                    // columns of this lane's four still inside its row's extent
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (e < left || v[e] != v[e]) ? v[e] : -3.3895313892515355e38f;
        }
        ev[j][0] = v[0]; ev[j][1] = v[1]; ev[j][2] = v[2]; ev[j][3] = v[3];
        mx = max3(mx, v[0], v[1]);
        mx = max3(mx, v[2], v[3]);
    }
}

// MB: key blocks of 128 the strip is sized for (8: up to 1024 keys, one workgroup per CU; 4, head_dim 64 only: up to 512 keys -- half the
// strip registers (113 in all) and a quarter of the LDS, so TWO workgroups share a CU and cover each other's sweeps and barriers:
// BERT-base's 384-key batches.  At head_dim 128 the 32 accumulator registers more do not fit two workgroups without spilling.)
template <int F, int D, int MB>
__global__ __launch_bounds__(512, (MB == 4 && D == 64) ? 4 : 1) void attention_fp8_split_kernel(AttnArgs a) {
    constexpr int kBuf = kBlock * D;                                       // one K / V^T block of codes: 16 KiB at head_dim 128, 8 KiB at 64
    constexpr int kPieces = D / 32;                                        // 1 KiB request pieces per wave and pair of blocks
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // 8 x 16 KiB of K, then V^T, blocks (later the partial sums) + the row statistics
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = l & 15, g = l >> 4, grp = w >> 2, wq = w & 3;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;                 // a head's blocks of rows are 8 apart in launch order: one XCD, one L2
    const int nkb = a.Sk / kBlock, nqb = (a.Sq + 63) / 64;
    const int qb = nqb - 1 - (int)blockIdx.y;                              // the heaviest blocks of rows of every head first under a causal mask
    const int q0 = qb * 64;
    const int qrow = q0 + wq * 16 + r, qc = min(qrow, a.Sq - 1);
    // extents: nlive = key blocks holding an unmasked column of one of the 64 rows; wmax / wmin = the largest / smallest extent among
    // this wave's 16 rows; my_live = this lane's row (0: a fully masked row, which attends to every key alike)
    int nlive = nkb, wmax = a.Sk, wmin = a.Sk, my_live = a.Sk;
    if (a.row_live) {
        const int qq = min(q0 + l, a.Sq - 1);
        int lv = a.row_live[b * a.lsb + h * a.lsh + qq * a.lsq];
        my_live = a.row_live[b * a.lsb + h * a.lsh + qc * a.lsq];
        if (lv <= 0) lv = a.Sk;
        int hi = my_live <= 0 ? a.Sk : my_live, lo = max(my_live, 0);       // a fully masked row is walked to the end and masked from column 0
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) lv = max(lv, __shfl_xor(lv, off, 64));
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            hi = max(hi, __shfl_xor(hi, off, 64));
            lo = min(lo, __shfl_xor(lo, off, 64));
        }
        nlive = __builtin_amdgcn_readfirstlane(min(nkb, (lv + kBlock - 1) / kBlock));
        wmax = __builtin_amdgcn_readfirstlane(hi);
        wmin = __builtin_amdgcn_readfirstlane(lo);
    }
    const int niter = (nlive + 1) / 2;                                     // iteration i: blocks 2 i and 2 i + 1
    const bool simple = a.mask && a.row_live && (a.mask_simple || (a.mask_irregular && *a.mask_irregular == 0)), full = a.mask && !simple;
    if (!simple) wmin = a.Sk;                                              // extents then only bound the walk; inside them the mask is read
    if (!simple && !a.row_live) wmax = a.Sk;
    const uint32_t l0 = lds_addr(lds);
    // DMA: an iteration brings two blocks (2 x D / 16 pieces of 1 KiB): this wave's pieces are kPieces w .. of those.  K block, head_dim
    // 128: a piece = 8 key rows x 128 bytes, 16-byte chunks XOR-swizzled by the row (chunk_off); head_dim 64: 16 key rows x 64 bytes,
    // chunk c of row r in slot c ^ 2 (r >> 3 & 1) -- with ds_read_b128's lane groups that makes the 16 fragment reads of a tile
    // conflict-free.  V^T block: D rows (d) x 128 bytes (keys), pieces of 8 rows, the same swizzle as the 128-byte K rows.
    const uint8_t *kp[kPieces], *vp[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int piece = (w * kPieces + i) % (D / 8);                    // within its block (D / 8 pieces of 1 KiB)
        if (D == 128) {
            const int row = piece * 8 + (l >> 3), slot = l & 7, sw = ((slot ^ ((row >> 1) & 7)) << 4);
            kp[i] = a.k8 + ((long)bh * a.Sk + row) * D + sw;                // + block * 128 * D
        } else {
            const int r16 = l >> 2, row = piece * 16 + r16, sw = (((l & 3) ^ (((r16 >> 3) & 1) << 1)) << 4);
            kp[i] = a.k8 + ((long)bh * a.Sk + row) * D + sw;
        }
        const int vrow = piece * 8 + (l >> 3), vslot = l & 7, vsw = ((vslot ^ ((vrow >> 1) & 7)) << 4);
        vp[i] = a.vt8 + ((long)bh * D + vrow) * a.Sk + vsw;                 // + block * 128
    }
    // the second block of the last pair may lie beyond the live blocks: it is then fetched from the last one and never multiplied
    auto issue_k = [&](int it) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int piece = w * kPieces + i, kb = min(2 * it + piece / (D / 8), nkb - 1);
            dma16(kp[i] + (long)kb * kBuf, l0 + (2 * it) * kBuf + piece * 1024);
        }
    };
    auto issue_v = [&](int it) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int piece = w * kPieces + i, kb = min(2 * it + piece / (D / 8), nkb - 1);
            dma16(vp[i] + (long)kb * kBlock, l0 + (2 * it) * kBuf + piece * 1024);
        }
    };
    // every pair has its own buffers; requests run two pairs ahead of the arithmetic (a wave that requested a whole sweep at once
    // would sit in the request queue -- 64 bytes per clock per CU -- before its first multiplication)
    auto wait_pair = [&](int it) {
        if (it + 1 < niter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory");   // the next pair (this wave's requests of it) may still be under way
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const uint8_t *qp = a.q8 + ((long)bh * a.Sq + qc) * D;
    const u32x4 qlo = *(const u32x4 *)(qp + 16 * g), qhi = D == 128 ? *(const u32x4 *)(qp + 64 + 16 * g) : u32x4{0u, 0u, 0u, 0u};
    const uint16_t *mrow = full ? a.mask + b * a.msb + h * a.msh + (long)qc * a.msq + 4 * g : nullptr;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the compiler's waits for these must not meet the block requests
    const v8i qf = {(int)qlo.x, (int)qlo.y, (int)qlo.z, (int)qlo.w, (int)qhi.x, (int)qhi.y, (int)qhi.z, (int)qhi.w};
    const int f_lo = D == 128 ? chunk_off(r, g) : r * 64 + ((g ^ (((r >> 3) & 1) << 1)) << 4), f_hi = chunk_off(r, 4 + g);
    const int f_v = chunk_off(r, 4 * grp + g);                             // V^T: this group's tiles are one 16-byte chunk per block
    constexpr int kIter = MB / 2;
    // this wave's four tiles of block kb start at key kb * 128 + 64 grp; how many of them does one of its rows reach?
    auto tiles_of = [&](int kb) { return kb < nlive ? min(4, max(0, (wmax - (kb * kBlock + 64 * grp) + 15) >> 4)) : 0; };

    float e[kIter][8][4];                                                  // this wave's part of the strip: logits, then their exponentials
    float mx = -INFINITY;
    // ---- sweep 1: scores
    issue_k(0);
    if (niter > 1) issue_k(1);
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        if (it < niter) {
            wait_pair(it);
            if (it + 2 < niter) issue_k(it + 2);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int kb = 2 * it + c, key0 = kb * kBlock + 64 * grp, nt = tiles_of(kb);
                const uint8_t *blk = lds + kb * kBuf;
                float (*ev)[4] = &e[it][4 * c];
                if (nt == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ev[j][0] = ev[j][1] = ev[j][2] = ev[j][3] = -INFINITY;
                } else if (full) {
                    score_half<F, D, 2>(blk, 4 * grp, qf, f_lo, f_hi, a.scaling, ev, mx, mrow, key0, 0, 4);
                } else if (!simple || key0 + 64 <= wmin) {
                    score_half<F, D, 0>(blk, 4 * grp, qf, f_lo, f_hi, a.scaling, ev, mx, nullptr, key0, 0, 4);
                } else {
                    score_half<F, D, 1>(blk, 4 * grp, qf, f_lo, f_hi, a.scaling, ev, mx, nullptr, key0 + 4 * g, my_live, nt);
                }
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // everyone is through sweep 1: the V^T blocks replace the K blocks while the exponentials are evaluated
    __builtin_amdgcn_s_barrier();
    issue_v(0);
    if (niter > 1) issue_v(1);
    // the two groups' maxima meet in LDS
    float *stat = (float *)(lds + MB * kBuf);                              // [max, sum][2 groups][64 rows]
    if (g == 0) stat[grp * 64 + wq * 16 + r] = mx;
    __syncthreads();
    mx = fmaxf(stat[wq * 16 + r], stat[64 + wq * 16 + r]);
    // ---- exponentials, once (a column at the bf16 minimum, or beyond the extents, gives exactly 0)
    const float kLog2e = 1.4426950408889634f;                              // (v - max) first: a fully masked row has max = the bf16 minimum
    float2_t sum2 = {0.0f, 0.0f};
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        if (it < niter) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (tiles_of(2 * it + c) != 0) {
#pragma unroll
                    for (int j = 4 * c; j < 4 * c + 4; ++j) {
                        float *v = e[it][j];
                        // packed subtract and multiply (two values per instruction), the same roundings as the scalar forms
                        const float2_t a0 = (float2_t{v[0], v[1]} - float2_t{mx, mx}) * float2_t{kLog2e, kLog2e};
                        const float2_t a1 = (float2_t{v[2], v[3]} - float2_t{mx, mx}) * float2_t{kLog2e, kLog2e};
                        v[0] = __builtin_amdgcn_exp2f(a0[0]); v[1] = __builtin_amdgcn_exp2f(a0[1]);
                        v[2] = __builtin_amdgcn_exp2f(a1[0]); v[3] = __builtin_amdgcn_exp2f(a1[1]);
                        sum2 += float2_t{v[0], v[1]};
                        sum2 += float2_t{v[2], v[3]};
                    }
                }
            }
        }
    }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0) stat[128 + grp * 64 + wq * 16 + r] = sum;
    __syncthreads();
    const float inv = 1.0f / (stat[128 + wq * 16 + r] + stat[128 + 64 + wq * 16 + r]);
    constexpr int kDT = D / 16;                                            // output tiles of 16 d
    v4f acc[kDT];
#pragma unroll
    for (int i = 0; i < kDT; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        if (it < niter) {
            wait_pair(it);
            if (it + 2 < niter) issue_v(it + 2);
            const bool live0 = tiles_of(2 * it) != 0, live1 = tiles_of(2 * it + 1) != 0;
            if (live0 | live1) {
                // staged as in sweep 1: half of the V^T fragments are requested before the codes are formed, the rest before the MFMAs
                const uint8_t *blk = lds + (2 * it) * kBuf + f_v;
                u32x4 vlo[kDT], vhi[kDT];
#pragma unroll
                for (int dt = 0; dt < kDT / 2; ++dt) {
                    vlo[dt] = *(const u32x4 *)(blk + dt * 2048);
                    vhi[dt] = *(const u32x4 *)(blk + kBuf + dt * 2048);
                }
                uint32_t pd[8];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c == 0 ? live0 : live1) {
#pragma unroll
                        for (int j = 4 * c; j < 4 * c + 4; ++j) {
                            const float *v = e[it][j];
                            const uint32_t p0 = pack_bf16x2(v[0] * inv, v[1] * inv), p1 = pack_bf16x2(v[2] * inv, v[3] * inv);   // probabilities, bf16
                            pd[j] = prob_codes<F == 1>(p0, p1);                                                               // fq_p
                        }
                    } else {
                        pd[4 * c] = pd[4 * c + 1] = pd[4 * c + 2] = pd[4 * c + 3] = 0;
                    }
                }
                const v8i pf = {(int)pd[0], (int)pd[1], (int)pd[2], (int)pd[3], (int)pd[4], (int)pd[5], (int)pd[6], (int)pd[7]};
#pragma unroll
                for (int dt = kDT / 2; dt < kDT; ++dt) {
                    vlo[dt] = *(const u32x4 *)(blk + dt * 2048);
                    vhi[dt] = *(const u32x4 *)(blk + kBuf + dt * 2048);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < kDT; ++dt) {
                    const v8i vf = {(int)vlo[dt].x, (int)vlo[dt].y, (int)vlo[dt].z, (int)vlo[dt].w, (int)vhi[dt].x, (int)vhi[dt].y, (int)vhi[dt].z, (int)vhi[dt].w};
                    acc[dt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vf, pf, acc[dt], F, F, 0, kUnit, 0, kUnit);
                }
            }
        }
    }
    // ---- the two partial sums meet in LDS: group 1 parks its accumulators ([64 rows][128 d] fp32), group 0 adds and stores
    __syncthreads();
    float *part = (float *)lds;
    constexpr int kPartRow = D + 4;                                       // floats per row: the pad spreads the 16 rows of a store over the banks
    const int prow = wq * 16 + r;
    if (grp == 1) {
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) *(float4 *)(part + prow * kPartRow + dt * 16 + 4 * g) = float4{acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]};
    }
    __syncthreads();
    if (grp == 0 && qrow < a.Sq) {
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) {
            const float4 o = *(const float4 *)(part + prow * kPartRow + dt * 16 + 4 * g);
            acc[dt][0] += o.x; acc[dt][1] += o.y; acc[dt][2] += o.z; acc[dt][3] += o.w;
        }
        const long o0 = (((long)b * a.Sq + qrow) * a.H + h) * D + 4 * g;
        uint16_t *orow = a.out + o0;
        if (a.out8) {
            const bool oe5 = a.out_fmt.p0 == 2;
#pragma unroll
            for (int dt = 0; dt < kDT; dt += 2) {
                uint32_t o[4] = {pack_bf16x2(acc[dt][0], acc[dt][1]), pack_bf16x2(acc[dt][2], acc[dt][3]),
                                 pack_bf16x2(acc[dt + 1][0], acc[dt + 1][1]), pack_bf16x2(acc[dt + 1][2], acc[dt + 1][3])};
                const uint2 codes = oe5 ? fq8_hw_vec8<true>(o, a.out_fmt) : fq8_hw_vec8<false>(o, a.out_fmt);
                *(uint2 *)(orow + dt * 16) = uint2{o[0], o[1]};
                *(uint2 *)(orow + dt * 16 + 16) = uint2{o[2], o[3]};
                *(uint32_t *)(a.out8 + o0 + dt * 16) = codes.x;
                *(uint32_t *)(a.out8 + o0 + dt * 16 + 16) = codes.y;
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < kDT; ++dt)
                *(uint2 *)(orow + dt * 16) = uint2{pack_bf16x2(acc[dt][0], acc[dt][1]), pack_bf16x2(acc[dt][2], acc[dt][3])};
        }
    }
}

// fq_v(V) as FP8 codes in the layout the P.V instruction wants: qt_value_codes.h.  One workgroup per (batch * head, key block).
template <bool E5M2, int D>
__global__ __launch_bounds__(256) void value_codes_t_kernel(const uint16_t *v, uint8_t *vt8, int H, long Sk, long sb, long sh, long sk,
                                                            qt_format fmt) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[D * kBlock];
    value_codes_block<E5M2, D>(tile, v, vt8, (int)blockIdx.x, (long)blockIdx.y, H, Sk, sb, sh, sk, fmt);
}

int status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

bool fp8_closed_form(const qt_format *f, bool &e5m2) {
    if (!f || f->kind != QT_FMT_FP_SAT) return false;
    e5m2 = f->p0 == 2 && f->p1 == -14 && f->fhi == 57344.0f;
    return e5m2 || (f->p0 == 3 && f->p1 == -6 && f->fhi == 448.0f);
}

template <int F, int D, int MB>
int launch_split_mb(const AttnArgs &a, long BH, int nqb, hipStream_t st) {
    constexpr int kLds = MB * kBlock * D + 2 * 2 * 64 * 4;               // every block of a sweep + the row statistics
    static_assert(MB * kBlock * D >= 64 * (D + 4) * 4, "the partial sums of the second wave group reuse the block buffers");
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        if (hipFuncSetAttribute((const void *)attention_fp8_split_kernel<F, D, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess)
            return QT_ERR_BAD_ARG;
        configured.done();
    }
    attention_fp8_split_kernel<F, D, MB><<<dim3((unsigned)BH, (unsigned)nqb), 512, kLds, st>>>(a);
    return status();
}

template <int F, int D>
int launch_split(const AttnArgs &a, long BH, int nqb, hipStream_t st) {
    if constexpr (D == 64) {
        if (a.Sk <= 4 * kBlock) return launch_split_mb<F, D, 4>(a, BH, nqb, st);
    }
    return launch_split_mb<F, D, 8>(a, BH, nqb, st);
}

}  // namespace

extern "C" {

int qt_value_codes_t(const uint16_t *v_dev, uint8_t *vt8_dev, long B, long H, long Sk, int head_dim, long stride_b, long stride_h, long stride_k,
                     const qt_format *fmt, void *stream) {
    if (B * H * Sk == 0) return QT_OK;
    bool e5m2 = false;
    if (!v_dev || !vt8_dev || B < 0 || H < 1 || Sk < 0 || Sk % kBlock != 0 || B * H > 65535 || (head_dim != 64 && head_dim != 128) ||
        !fp8_closed_form(fmt, e5m2))
        return QT_ERR_BAD_ARG;
    if ((((uintptr_t)v_dev | (uintptr_t)vt8_dev) & 15u) || ((stride_b | stride_h | stride_k) & 7)) return QT_ERR_UNALIGNED;
    const dim3 grid((unsigned)(Sk / kBlock), (unsigned)(B * H));
    hipStream_t st = (hipStream_t)stream;
    if (head_dim == 128) {
        if (e5m2) value_codes_t_kernel<true, 128><<<grid, 256, 0, st>>>(v_dev, vt8_dev, (int)H, Sk, stride_b, stride_h, stride_k, *fmt);
        else value_codes_t_kernel<false, 128><<<grid, 256, 0, st>>>(v_dev, vt8_dev, (int)H, Sk, stride_b, stride_h, stride_k, *fmt);
    } else {
        if (e5m2) value_codes_t_kernel<true, 64><<<grid, 256, 0, st>>>(v_dev, vt8_dev, (int)H, Sk, stride_b, stride_h, stride_k, *fmt);
        else value_codes_t_kernel<false, 64><<<grid, 256, 0, st>>>(v_dev, vt8_dev, (int)H, Sk, stride_b, stride_h, stride_k, *fmt);
    }
    return status();
}

int qt_attention_fp8(const uint8_t *q8_dev, const uint8_t *k8_dev, const uint8_t *vt8_dev, int operand_format, const uint16_t *mask_dev,
                     long mask_sb, long mask_sh, long mask_sq, const int *row_live_dev, long live_sb, long live_sh, long live_sq,
                     int mask_is_simple, const int *mask_irregular_dev, uint16_t *out_dev, uint8_t *out8_dev, const qt_format *out_format, long B,
                     int H, int Sq, int Sk, int head_dim, float scaling, void *stream) {
    if (B * H * Sq == 0) return QT_OK;
    if (!q8_dev || !k8_dev || !vt8_dev || !out_dev || B < 0 || H < 1 || Sq < 1 || Sk < kBlock || Sk % kBlock != 0 || Sk > kBlock * kMaxBlocks ||
        B * H > 65535 || operand_format < 0 || operand_format > 1 || (head_dim != 64 && head_dim != 128))
        return QT_ERR_BAD_ARG;
    if ((((uintptr_t)q8_dev | (uintptr_t)k8_dev | (uintptr_t)vt8_dev) & 15u) || ((uintptr_t)out_dev & 7u) ||
        (mask_dev && ((((uintptr_t)mask_dev) & 7u) || ((mask_sb | mask_sh | mask_sq) & 3))))
        return QT_ERR_UNALIGNED;
    bool oe5 = false;
    if (out8_dev && (!fp8_closed_form(out_format, oe5) || ((uintptr_t)out8_dev & 3u))) return QT_ERR_BAD_ARG;
    AttnArgs a{q8_dev, k8_dev, vt8_dev, mask_dev, mask_sb, mask_sh, mask_sq, row_live_dev, live_sb, live_sh, live_sq, mask_is_simple ? 1 : 0,
               out_dev, H, Sq, Sk, scaling, out8_dev, out8_dev ? *out_format : qt_format{}, mask_irregular_dev};
    const int nqb = (Sq + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    if (head_dim == 128) return operand_format == 0 ? launch_split<0, 128>(a, B * H, nqb, st) : launch_split<1, 128>(a, B * H, nqb, st);
    return operand_format == 0 ? launch_split<0, 64>(a, B * H, nqb, st) : launch_split<1, 64>(a, B * H, nqb, st);
}

}  // extern "C"
