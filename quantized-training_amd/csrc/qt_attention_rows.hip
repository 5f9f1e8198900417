// qt_attention_rows.hip -- attention core for stateless TABLE formats (posit, fpN, ...) on the bf16 matrix instruction, in one launch,
// with the whole score strip in registers (round 4).
//
// Replaces, for one attention block whose four matmul inputs are fake-quantized with a stateless table format (BASELINE configs[3]:
// posit(8,2)), everything between the hooks of `qk_matmul` and the output of `av_matmul` of the reference's quantizable attention
// (modules/quantizable/modeling_llama.py:228-246, modeling_bert.py:118-158, functional_modules.py:22-26):
//     S = matmul(fq(Q), fq(K)^T)  ->  t = bf16(bf16(S * scaling) + mask)  ->  p = bf16(softmax_fp32(t))  ->  pq = fq_P(p)  ->  O = matmul(pq, fq(V))
// with every bf16 rounding point of that chain kept (the same arithmetic as csrc/qt_attention_fp8.hip and csrc/qt_attention.hip).
//
// Round 3's kernel for these formats (attention_fq_kernel, csrc/qt_attention.hip) makes two passes over the keys and computes
// Q.K^T twice, 64-key tiles staged through registers with two barriers per tile: 88 us per LLaMA-2-13B layer (40 heads x 1024 x
// 1024, causal), an order of magnitude off both the matrix and the vector floor.  This kernel is the FP8 core's design
// (attention_fp8_split_kernel) carried over to bf16 operands:
//   * a workgroup = 8 waves takes one block of 64 query rows; its two groups of 4 waves split the KEYS of every 128-key block
//     (group 0: keys 0-63, group 1: 64-127), a wave owns 16 query rows, so a wave's strip is at most 4 pairs x 8 tiles of 16 x 16
//     scores = 128 fp32 registers: sweep 1 fills it with the logits, maximum and sum meet across the groups in LDS, each
//     exponential is evaluated ONCE, sweep 2 rounds to bf16, applies fq_P in its row form (csrc/qt_device.h, Rounder<kFmtRows>) and
//     multiplies by V;
//   * operands swapped (K fragment first): lane (r, g) holds query r against keys 16 t + 4 g + {0..3} of tile t, so the row
//     statistics are lane-local plus two shuffles, and a lane's quantized probabilities of two neighbouring tiles ARE its B
//     fragment of v_mfma_f32_16x16x32_bf16 for P.V if V arrives transposed, [D][Sk], with slot 8 g + 4 h + e <-> key 16 h + 4 g + e
//     inside every 32-key chunk -- qt_value_t_rows writes that image while fake-quantizing V (this is the `fq_v` call);
//   * K blocks and V^T blocks (32 KiB each) come by LDS-DMA into a ring of three (96 KiB; behind it 32 KiB of the value map: the
//     probabilities' fake-quantizer as a gather), two blocks in flight ahead of the arithmetic, one barrier per block; 256-byte rows with the 16-byte chunk index XORed with the row, so every fragment read
//     (sixteen rows x one chunk per lane group of ds_read_b128) is conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_value_rows.h"

namespace {

typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 128, kMaxBlocks = 8, kD = 128;
constexpr int kRowB = 256;                       // bytes of a K row (128 d) and of a V^T row segment (128 keys)
constexpr int kBuf = kBlock * kRowB;             // one block: 32 KiB
constexpr int kSlots = 3;                        // ring slots: the block being multiplied and two in flight
constexpr int kProbLut = 32768;                  // the map's entries 0 .. 0x3FFF (bf16 patterns of [0, 2)): the probabilities' fake-quantizer as a gather
constexpr int kRing = kSlots * kBuf + kProbLut;  // three blocks | the probability table
constexpr int kTbl = 8192;                       // the row table (512 rows x 16 B; 256 used by maps indexed by exponent only)
constexpr int kLds = kRing + kTbl + 2 * 2 * 64 * 4;

struct Args {
    const uint16_t *q, *k, *vt;         // [B][H][Sq][128], [B][H][Sk][128], [B][H][128][Sk] (keys permuted inside 32-chunks), bf16 values
    const uint16_t *mask;               // additive bf16 or NULL; element strides for (b, h, q), columns contiguous
    long msb, msh, msq;
    const int *row_live;                // optional: one past the last unmasked column per mask row; row strides for (b, h, q)
    long lsb, lsh, lsq;
    const int *mask_irregular;          // optional, device: 0 = every mask row is "zeros, then the bf16 minimum": the mask is not read
    uint16_t *out;                      // [B][Sq][H][128] bf16
    int H, Sq, Sk;
    float scaling;
    qt_format fmt;                      // the probabilities' format: table format with the row words behind the map (p1 bit 0)
    const uint16_t *lut;
    int out_fq;                         // 1: the output projection's input fake-quantizer (same format) applied on the way out
#ifdef QT_TUNING_BUILD
    unsigned long long *dbg;            // QT_AR_STAMPS = device address: s_memtime stamps of workgroup (0, 0), waves 0 and 4, 64 slots each
#endif
};

__device__ __forceinline__ float blo(uint32_t w) { return qt_u2f(w << 16); }
__device__ __forceinline__ float bhi(uint32_t w) { return qt_u2f(w & 0xFFFF0000u); }
// byte offset of 16-byte chunk `chunk` of row `row` inside a block image
__device__ __forceinline__ int chunk_off(int row, int chunk) { return row * kRowB + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// LDS-DMA as inline asm (hipcc's wait-count model then keeps counted waits for ordinary loads, see qt_attention_fp8.hip): a uniform
// base (SGPR pair) + a 32-bit lane offset; the 64 lanes' 16 bytes land linearly at `dst`
__device__ __forceinline__ void dma16(const void *base, uint32_t off, uint32_t dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ float max3(float x, float y, float z) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}

// scores of the four tiles t0 .. t0 + 3 of one K block (this wave's half of it).  MODE 0: no mask arithmetic; 1: row-extent mask applied
// from my_live (tiles from `tiles` on lie beyond every row of the wave); 2: additive mask read from mrow
template <int MODE>
__device__ __forceinline__ void score_half(const uint8_t *blk, int t0, const v8s (&qf)[4], int r, int g, float scaling, float (*ev)[4], float &mx,
                                           const uint16_t *mrow, int key0, int my_live, int tiles) {
    uint2 m[4];
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = *(const uint2 *)(mrow + key0 + j * 16);
    }
    v4f s[4];
    // staged by hand: all sixteen fragment reads of the four tiles, then their sixteen multiplications back to back, then the rounding
    // chains (a read -> multiply -> chain sequence per tile would expose the LDS and matrix latencies four times over)
    u32x4 kf[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == 1 && j >= tiles) continue;
        const int row = (t0 + j) * 16 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[j][ks] = *(const u32x4 *)(blk + chunk_off(row, 4 * ks + g));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == 1 && j >= tiles) continue;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8s, kf[j][ks]), qf[ks], acc, 0, 0, 0);
        s[j] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == 1 && j >= tiles) {
            ev[j][0] = ev[j][1] = ev[j][2] = ev[j][3] = -INFINITY;
            continue;
        }
        uint32_t w0 = pack_bf16x2(s[j][0], s[j][1]), w1 = pack_bf16x2(s[j][2], s[j][3]);     // the matmul's bf16 output
        w0 = pack_bf16x2(blo(w0) * scaling, bhi(w0) * scaling);                             // attn_scaling (MulFunctional), bf16
        w1 = pack_bf16x2(blo(w1) * scaling, bhi(w1) * scaling);
        if (MODE == 2) {
            w0 = pack_bf16x2(blo(w0) + blo(m[j].x), bhi(w0) + bhi(m[j].x));                 // + mask, bf16
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

__global__ __launch_bounds__(512, 1) void attention_rows_split_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // ring of four 32 KiB blocks (later the partial sums) | row table | row statistics
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = l & 15, g = l >> 4, grp = w >> 2, wq = w & 3;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;                 // a head's blocks of rows are 8 apart in launch order: one XCD, one L2
    const int nkb = a.Sk / kBlock, nqb = (a.Sq + 63) / 64;
    const int qb = nqb - 1 - (int)blockIdx.y;                              // the heaviest blocks of rows of every head first under a causal mask
    const int q0 = qb * 64;
    const int qrow = q0 + wq * 16 + r, qc = min(qrow, a.Sq - 1);
    // ---- the row table of the probabilities' (and the output's) fake-quantizer
    Rounder<kFmtRows> rnd{a.fmt, nullptr, nullptr};
    {
        uint4 *s_rows = (uint4 *)(lds + kRing);
        const uint4 *gr = (const uint4 *)(a.lut + QT_MAP_ENTRIES);
        const int nrows = (a.fmt.p1 & 2) ? 512 : 256;
        if (t < nrows) s_rows[t] = gr[t];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = a.lut;
    }
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
    const bool simple = a.mask && a.row_live && a.mask_irregular && *a.mask_irregular == 0, full = a.mask && !simple;
    if (!simple) wmin = a.Sk;                                              // extents then only bound the walk; inside them the mask is read
    if (!simple && !a.row_live) wmax = a.Sk;
    const uint32_t l0 = lds_addr(lds);
#ifdef QT_TUNING_BUILD
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && l == 0 && (w & 3) == 0) a.dbg[(w >> 2) * 64 + slot] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [](int) __attribute__((always_inline)) {};
#endif
    stamp(0);
    // the probabilities' fake-quantizer as a table: the map's first 16 384 entries (32 KiB: every bf16 pattern of [0, 2), a probability is
    // in [0, 1]) by LDS-DMA behind the ring, long before sweep 2 reads them -- one 2-byte gather per probability instead of the row form's
    // ~15 instructions (the stamps: sweep 2 was 2 300 cycles per block of a 55 k-cycle workgroup, all of it vector issue)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16((const uint8_t *)a.lut + (4 * w + i) * 1024, (uint32_t)l * 16u, l0 + kSlots * kBuf + (4 * w + i) * 1024);
    // ---- DMA geometry.  A block is 32 pieces of 1 KiB (4 rows x 256 bytes); wave w issues pieces 4 w .. 4 w + 3 of every block: piece
    // pb = 4 w + i holds rows 4 pb .. 4 pb + 3.  Lane l lands at row 4 pb + (l >> 4), 16-byte slot l & 15 of the LDS image, which must hold
    // logical chunk (l & 15) ^ (row & 15); (row & 15) = 4 i + (l >> 4), so four lane-offset registers per operand serve every piece, the
    // rest of the address is uniform.  Block kb sits in ring slot kb & 3.
    uint32_t koff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rlow = 4 * i + (l >> 4), c = (l & 15) ^ rlow;
        koff[i] = (uint32_t)((l >> 4) * kRowB + c * 16);                    // K: rows are 256 bytes apart
        voff[i] = (uint32_t)((l >> 4) * a.Sk * 2 + c * 16);                 // V^T: rows (d) are Sk * 2 bytes apart
    }
    const uint8_t *kbase = (const uint8_t *)a.k + (long)bh * a.Sk * kRowB;
    const uint8_t *vbase = (const uint8_t *)a.vt + (long)bh * kD * a.Sk * 2;
    auto issue_k = [&](int kb) __attribute__((always_inline)) {
        const uint32_t dst0 = l0 + (kb % kSlots) * kBuf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pb = 4 * w + i;
            dma16(kbase + ((long)kb * kBlock + 4 * pb) * kRowB, koff[i], dst0 + pb * 1024);
        }
    };
    auto issue_v = [&](int kb) __attribute__((always_inline)) {
        const uint32_t dst0 = l0 + (kb % kSlots) * kBuf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pb = 4 * w + i;
            dma16(vbase + ((long)(4 * pb) * a.Sk + (long)kb * kBlock) * 2, voff[i], dst0 + pb * 1024);
        }
    };
    // block kb is awaited with the requests of the next block (four per wave) still in flight
    auto await_block = [&](int kb) __attribute__((always_inline)) {
        if (nlive - 1 - kb >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // Q^T fragments (B operand): lane (r, g) holds Q[query r][32 ks + 8 g + j]
    v8s qf[4];
    {
        const uint16_t *qp = a.q + ((long)bh * a.Sq + qc) * kD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8s *)(qp + ks * 32 + g * 8);
    }
    const uint16_t *mrow = full ? a.mask + b * a.msb + h * a.msh + (long)qc * a.msq + 4 * g : nullptr;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the compiler's waits for these must not meet the block requests
    __syncthreads();                                                       // the row table is in place
    constexpr int kIter = kMaxBlocks / 2;
    // this wave's four tiles of block kb start at key kb * 128 + 64 grp; how many of them does one of its rows reach?
    auto tiles_of = [&](int kb) { return kb < nlive ? min(4, max(0, (wmax - (kb * kBlock + 64 * grp) + 15) >> 4)) : 0; };

    float e[kIter][8][4];                                                  // this wave's part of the strip: logits, then their exponentials
    float mx = -INFINITY;
    // ---- sweep 1: scores.  Two blocks are in flight ahead of the one being multiplied (the stamps put the block waits at ~180 cycles:
    // depth is not what paces the kernel); at the barrier of step kb every wave is done with block kb - 1, whose slot then takes block kb + 2
    stamp(1);
    issue_k(0);
    if (nlive > 1) issue_k(1);
#pragma unroll
    for (int kb = 0; kb < kMaxBlocks; ++kb) {
        if (kb < nlive) {
            await_block(kb);
            stamp(2 + 3 * kb);
            __builtin_amdgcn_s_barrier();
            stamp(3 + 3 * kb);
            if (kb + 2 < nlive) issue_k(kb + 2);
            const int key0 = kb * kBlock + 64 * grp, nt = tiles_of(kb);
            const uint8_t *blk = lds + (kb % kSlots) * kBuf;
            float (*ev)[4] = &e[kb >> 1][4 * (kb & 1)];
            if (nt == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j][0] = ev[j][1] = ev[j][2] = ev[j][3] = -INFINITY;
            } else if (full) {
                score_half<2>(blk, 4 * grp, qf, r, g, a.scaling, ev, mx, mrow, key0, 0, 4);
            } else if (!simple || key0 + 64 <= wmin) {
                score_half<0>(blk, 4 * grp, qf, r, g, a.scaling, ev, mx, nullptr, key0, 0, 4);
            } else {
                score_half<1>(blk, 4 * grp, qf, r, g, a.scaling, ev, mx, nullptr, key0 + 4 * g, my_live, nt);
            }
            stamp(4 + 3 * kb);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // everyone is through sweep 1: the V^T blocks replace the K blocks while the exponentials are evaluated
    stamp(26);
    __builtin_amdgcn_s_barrier();
    stamp(27);
    issue_v(0);
    if (nlive > 1) issue_v(1);
    // the two groups' maxima meet in LDS
    float *stat = (float *)(lds + kRing + kTbl);                           // [max, sum][2 groups][64 rows]
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
                        // packed subtract and multiply (v_pk_add_f32 / v_pk_mul_f32: two values per instruction), the same roundings
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
    stamp(28);
    // ---- sweep 2: probabilities (bf16, fake-quantized in the row form) x V.  Output tile dt: D[16 d x 16 queries], lane (r, g) holds
    // d = 16 dt + 4 g + {0..3} of query r.  Per 32-key chunk (two neighbouring tiles 2 c2, 2 c2 + 1 of this wave's half block) the lane's
    // eight probabilities are its B fragment; the A fragment is V^T rows 16 dt + r, slots 8 g .. 8 g + 7 of that chunk.
    constexpr int kDT = kD / 16;
    v4f acc[kDT];
#pragma unroll
    for (int i = 0; i < kDT; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < kMaxBlocks; ++kb) {
        if (kb < nlive) {
            await_block(kb);
            stamp(29 + 3 * kb);
            __builtin_amdgcn_s_barrier();
            stamp(30 + 3 * kb);
            if (kb + 2 < nlive) issue_v(kb + 2);
            if (tiles_of(kb) != 0) {
                const uint8_t *blk = lds + (kb % kSlots) * kBuf;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int chunk0 = 4 * (2 * grp + c2) + g;              // 16-byte chunk of this lane's slots inside the 256-byte row
                    u32x4 vf[kDT];
#pragma unroll
                    for (int dt = 0; dt < kDT; ++dt) vf[dt] = *(const u32x4 *)(blk + chunk_off(dt * 16 + r, chunk0));
                    uint32_t pb[4], pw[4];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float *v = e[kb >> 1][4 * (kb & 1) + 2 * c2 + hh];
                        pb[2 * hh] = pack_bf16x2(v[0] * inv, v[1] * inv);  // probabilities, bf16
                        pb[2 * hh + 1] = pack_bf16x2(v[2] * inv, v[3] * inv);
                    }
                    {                                                       // fq_p
                        const uint16_t *plut = (const uint16_t *)(lds + kSlots * kBuf);
                        const uint32_t beyond = (pb[0] | pb[1] | pb[2] | pb[3]) & 0xC000C000u;   // a pattern >= 0x4000: not a probability (NaN inputs)
                        if (__builtin_expect(beyond == 0u, 1)) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pw[i] = (uint32_t)plut[pb[i] & 0xFFFFu] | ((uint32_t)plut[pb[i] >> 16] << 16);
                        } else {
                            fq_rows_words<4, true>(pb, pw, rnd);            // the row form (and through it the map) takes anything
                        }
                    }
                    const v8s pf = __builtin_bit_cast(v8s, u32x4{pw[0], pw[1], pw[2], pw[3]});
#pragma unroll
                    for (int dt = 0; dt < kDT; ++dt)
                        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8s, vf[dt]), pf, acc[dt], 0, 0, 0);
                }
            }
            stamp(31 + 3 * kb);
        }
    }
    stamp(53);
    // ---- the two partial sums meet in LDS: group 1 parks its accumulators ([64 rows][128 d] fp32), group 0 adds and stores
    __syncthreads();
    float *part = (float *)lds;
    constexpr int kPartRow = kD + 4;                                       // floats per row: the pad spreads the 16 rows of a store over the banks
    const int prow = wq * 16 + r;
    if (grp == 1) {
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) *(float4 *)(part + prow * kPartRow + dt * 16 + 4 * g) = float4{acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]};
    }
    __syncthreads();
    if (grp == 0 && qrow < a.Sq) {
        uint16_t *orow = a.out + (((long)b * a.Sq + qrow) * a.H + h) * kD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) {
            const float4 o = *(const float4 *)(part + prow * kPartRow + dt * 16 + 4 * g);
            uint32_t o0 = pack_bf16x2(acc[dt][0] + o.x, acc[dt][1] + o.y), o1 = pack_bf16x2(acc[dt][2] + o.z, acc[dt][3] + o.w);
            if (a.out_fq) {                                                // the output projection's input fake-quantizer (same format, unit scale)
                const uint32_t ow[2] = {o0, o1};
                uint32_t oq[2];
                fq_rows_words<2, false>(ow, oq, rnd);
                o0 = oq[0];
                o1 = oq[1];
            }
            *(uint2 *)(orow + dt * 16) = uint2{o0, o1};
        }
    }
}

// fq_v(V) as bf16 values, transposed to [d][key] with the keys of every 32-chunk permuted into the k-slot order of the kernel's P.V
// instruction (csrc/qt_value_rows.h).  One 256-thread workgroup per (batch * head, block of 128 keys).
__global__ __launch_bounds__(256) void value_t_rows_kernel(const uint16_t *v, uint16_t *vt, int H, long Sk, long sb, long sh, long sk, qt_format fmt,
                                                           const uint16_t *lut) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[value_rows_tile_elems<kBlock>()];
    __shared__ uint4 s_rows[512];
    Rounder<kFmtRows> rnd{fmt, nullptr, nullptr};
    {
        const uint4 *gr = (const uint4 *)(lut + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rows[i] = gr[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = lut;
    }
    __syncthreads();
    value_t_rows_block<kBlock>(tile, rnd, v, vt, H, Sk, sb, sh, sk, (long)blockIdx.y, (int)blockIdx.x, (int)threadIdx.x);
}

int status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

bool rows_format(const qt_format *f) { return f && f->kind == QT_FMT_LUT && (f->p1 & 1); }

}  // namespace

extern "C" {

int qt_value_t_rows(const uint16_t *v_dev, uint16_t *vt_dev, long B, long H, long Sk, int head_dim, long stride_b, long stride_h, long stride_k,
                    const qt_format *fmt, const uint16_t *lut_dev, void *stream) {
    if (B * H * Sk == 0) return QT_OK;
    if (!v_dev || !vt_dev || !lut_dev || B < 0 || H < 1 || Sk < 0 || Sk % kBlock != 0 || B * H > 65535 || head_dim != kD || !rows_format(fmt))
        return QT_ERR_BAD_ARG;
    if ((((uintptr_t)v_dev | (uintptr_t)vt_dev | (uintptr_t)lut_dev) & 15u) || ((stride_b | stride_h | stride_k) & 7)) return QT_ERR_UNALIGNED;
    value_t_rows_kernel<<<dim3((unsigned)(Sk / kBlock), (unsigned)(B * H)), 256, 0, (hipStream_t)stream>>>(v_dev, vt_dev, (int)H, Sk, stride_b, stride_h,
                                                                                                         stride_k, *fmt, lut_dev);
    return status();
}

int qt_attention_rows_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *vt_dev, const uint16_t *mask_dev, long mask_sb, long mask_sh,
                           long mask_sq, const int *row_live_dev, long live_sb, long live_sh, long live_sq, const int *mask_irregular_dev,
                           uint16_t *out_dev, int out_fq, const qt_format *fmt, const uint16_t *lut_dev, long B, int H, int Sq, int Sk, int head_dim,
                           float scaling, void *stream) {
    if (B * H * Sq == 0) return QT_OK;
    if (!q_dev || !k_dev || !vt_dev || !out_dev || !lut_dev || B < 0 || H < 1 || Sq < 1 || Sk < kBlock || Sk % kBlock != 0 || Sk > kBlock * kMaxBlocks ||
        B * H > 65535 || head_dim != kD || !rows_format(fmt))
        return QT_ERR_BAD_ARG;
    if ((row_live_dev != nullptr) != (mask_irregular_dev != nullptr) || (row_live_dev && !mask_dev)) return QT_ERR_BAD_ARG;
    if ((long)Sk * kD * 2 >= (1L << 31)) return QT_ERR_BAD_ARG;           // 32-bit lane offsets of the V^T pieces
    if ((((uintptr_t)q_dev | (uintptr_t)k_dev | (uintptr_t)vt_dev | (uintptr_t)lut_dev) & 15u) || ((uintptr_t)out_dev & 7u) ||
        (mask_dev && ((((uintptr_t)mask_dev) & 7u) || ((mask_sb | mask_sh | mask_sq) & 3))))
        return QT_ERR_UNALIGNED;
    Args a{q_dev, k_dev, vt_dev, mask_dev, mask_sb, mask_sh, mask_sq, row_live_dev, live_sb, live_sh, live_sq, mask_irregular_dev,
           out_dev, H, Sq, Sk, scaling, *fmt, lut_dev, out_fq ? 1 : 0};
#ifdef QT_TUNING_BUILD
    {
        const char *e_st = getenv("QT_AR_STAMPS");
        a.dbg = e_st ? (unsigned long long *)strtoull(e_st, nullptr, 0) : nullptr;
    }
#endif
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        if (hipFuncSetAttribute((const void *)attention_rows_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return QT_ERR_BAD_ARG;
        configured.done();
    }
    attention_rows_split_kernel<<<dim3((unsigned)(B * H), (unsigned)((Sq + 63) / 64)), 512, kLds, (hipStream_t)stream>>>(a);
    return status();
}

}  // extern "C"
