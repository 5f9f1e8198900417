// qt_attention.hip -- fused fake-quant attention core on the gfx950 matrix cores.
//
// Replaces, for one attention block, everything between the hooks of `qk_matmul` and the output of
// `av_matmul` in the reference's quantizable attention (modules/quantizable/modeling_bert.py:118-158,
// modeling_llama.py:228-246, functional_modules.py:22-26):
//     S  = matmul(fq(Q), fq(K)^T)                      bf16 GEMM, S x S tensor written
//     t  = attn_scaling(S, scaling) + mask             two more passes
//     P  = softmax(t) (fp32) -> bf16                   three more passes
//     O  = matmul(fq(P), fq(V))                        fake-quant pass over P, then a bf16 GEMM
// Inputs are the already fake-quantized Q, K, V ([B, H, S, D] contiguous, written by the elementwise
// pass); the S x S scores never leave the chip.  Every rounding point of the chain above is kept:
//     S = bf16(sum_d Q K)  (fp32 MFMA accumulation),  t = bf16(bf16(S * scaling) + mask),
//     p = bf16(exp(t - m) / l),  pq = fq_P(p) (same per-element function as qt_fake_quant_bf16, amax(p)
//     observed),  O = bf16(sum_k pq V).
// Because fq_P needs the normalised probability, the kernel makes two passes over the keys: pass 1
// computes the row maxima m and sums l, pass 2 recomputes S, forms pq and accumulates O.  Not bit-defined
// (same caveat as qt_softmax.hip): MFMA accumulation order, exp(), the order of the row sum.
//
// Layout: one workgroup = 4 wavefronts = 64 query rows (16 per wave) of one (batch, head); keys / values
// stream through LDS in tiles of 64.  QK^T is computed transposed (A = K tile, B = Q^T from registers) so
// that a lane owns ONE query column and its 16 scores per tile: row max / sum are lane-local plus two
// __shfl_xor steps, and the score accumulators are already the A operand of the P.V product (the key order
// inside an MFMA k-step is permuted identically for P and V).  V is transposed into LDS while staged so that
// the permuted V fragments are 8-byte LDS reads.  O is written directly in the [B, Sq, H, D] layout the caller
// needs next (saves the transpose copy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "qt_device.h"

namespace {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kBQ = 64;    // query rows per workgroup
constexpr int kBK = 64;    // keys per tile

struct AttnArgs {
    const uint16_t *q, *k, *v, *mask;
    uint16_t *out;
    int B, H, Sq, Sk;
    long mask_sb, mask_sh, mask_sq;
    float scaling;
    qt_format fmt;
    const uint16_t *lut;
    const float *scale;
    uint32_t *amax;
    int p8;                   // probabilities' format is exactly E4M3 (1) / E5M2 (2) at unit scale: hardware conversion
    const int *row_live;      // optional (qt_mask_row_live_checked): one past the last unmasked column per mask row; strides in rows
    long lsb, lsh, lsq;
    const int *mask_irregular;// device flag behind it: 0 = every mask row is exactly "zeros, then the bf16 minimum" -- the mask is then
                              // not read at all: its value at column c of a row is (c < row_live ? 0 : minimum)
    int snake;                // > 0: workgroups dealt to `snake` slots by descending work, alternating direction (see the kernel)
    int out_fq;               // 1: the consumer's (output projection's input) fake-quantizer is the probabilities' stateless format at unit
                              // scale and is applied to the result on its way out (qt_attention_fq_out_bf16)
};

__device__ __forceinline__ float bf16_round(float f) { return qt_u2f(pack_bf16x2(f, 0.0f) << 16); }

// UNIT: the probabilities' fake-quantizer has no scale tensor (scale == 1 exactly).  OBS: its amax is observed.
template <int D, int KIND, bool UNIT, bool OBS>
__global__ __launch_bounds__(256) void attention_fq_kernel(AttnArgs a) {
    constexpr int KS = D / 32;                    // MFMA k-steps over the head dimension
    constexpr int kKRow = D * 2 + 16;             // padded K row (bytes)
    constexpr int kVRow = kBK * 2 + 16;           // padded V^T row (bytes)
    __shared__ __attribute__((aligned(16))) unsigned char lds[kBK * kKRow + D * kVRow];
    unsigned char *Ks = lds;
    unsigned char *Vt = lds + kBK * kKRow;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qc = lane & 15, g = lane >> 4;      // query column owned by the lane, lane group
    // 1-D grid, query blocks ordered so that a causal mask's heavy blocks (late queries: many live key tiles) come first
    // and the light ones mirror them: with every workgroup resident at once and round-robin placement, the two
    // workgroups a CU receives (indices c and c + total/2) then hold nq + 1 live tiles between them instead of up to 2 nq
    const int BH = a.B * a.H, nq = (a.Sq + kBQ - 1) / kBQ;
    int kq, bh, qb;
    if (a.snake > 0) {
        // the general form of the same idea (the launch fits the chip all at once, workgroup i lands on slot i mod CUs): list the
        // workgroups by descending work (query block nq - 1 first) and deal them to the slots forwards, backwards, forwards, ... so
        // that a slot's workgroups add up alike for any head count -- 40 heads x 16 blocks on 256 CUs: 24 live tiles at most per
        // slot instead of 31 (average 21)
        const int total = nq * BH, round = (int)blockIdx.x / a.snake, pos = (int)blockIdx.x % a.snake;
        const int len = min(a.snake, total - round * a.snake);
        const int idx = round * a.snake + ((round & 1) ? len - 1 - pos : pos);
        qb = nq - 1 - idx / BH;
        bh = idx % BH;
    } else {
        kq = blockIdx.x / BH; bh = blockIdx.x % BH;
        qb = kq < nq / 2 ? nq - 1 - kq : kq - nq / 2;
    }
    const int b = bh / a.H, h = bh % a.H;
    const int q0 = qb * kBQ + wave * 16;
    const int qrow = q0 + qc;
    const int qload = qrow < a.Sq ? qrow : a.Sq - 1;
    const uint16_t *Qp = a.q + ((long)bh * a.Sq + qload) * D;
    const uint16_t *Kp = a.k + (long)bh * a.Sk * D;
    const uint16_t *Vp = a.v + (long)bh * a.Sk * D;
    const uint16_t *Mp = a.mask ? a.mask + b * a.mask_sb + h * a.mask_sh + (long)qload * a.mask_sq : nullptr;
    const bool simple = a.mask && a.row_live && a.mask_irregular && *a.mask_irregular == 0;      // uniform over the launch
    const int lv = simple ? a.row_live[b * a.lsb + h * a.lsh + (long)qload * a.lsq] : 0;          // this lane's query row

    Rounder<KIND> rnd{a.fmt, a.lut};
    if constexpr (KIND == kFmtRows) {
        // table format with the row words behind the map (qt_format.p1 bit 0, csrc/qt_device.h): the probabilities' fake-quantizer is
        // arithmetic on a 4 / 8 KiB row table in LDS instead of one gather per element from the 128 KiB map in global memory
        __shared__ uint4 s_rows[512];
        const uint4 *gr = (const uint4 *)(a.lut + QT_MAP_ENTRIES);
        const int nrows = (a.fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rows[i] = gr[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = a.lut;
        __syncthreads();
    }
    const float s = UNIT ? 1.0f : qt_bf2f(qt_f2bf(*a.scale));
    const UniformDiv dv(s);

    // Q^T fragments (B operand): lane holds Q[q][32*ks + 8*g + j]
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t *)(Qp + ks * 32 + g * 8);

    const int ntiles = (a.Sk + kBK - 1) / kBK;

    // Staging is split into "issue the global loads of a tile into registers" and "write those registers to
    // LDS", so that the loads of tile t+1 are in flight while tile t is being computed (a workgroup's critical
    // path is otherwise one exposed HBM/L2 round trip per tile).
    constexpr int KCH = D / 8;                     // 16-B chunks per K row
    constexpr int KRPI = 256 / KCH;                // K rows per iteration
    constexpr int KIT = kBK / KRPI;                // iterations (4 for D = 128, 2 for D = 64)
    constexpr int VIT = (kBK / 2) * (D / 8) / 256; // (key pair, d-chunk) items per thread (2 / 1)
    // staging registers are named scalars (arrays captured by the lambdas ended up in scratch memory)
    uint4 kr0, kr1, kr2, kr3, va0, va1, vb0, vb1;
    kr0 = kr1 = kr2 = kr3 = va0 = va1 = vb0 = vb1 = uint4{0u, 0u, 0u, 0u};

    auto k_src = [&](int k0, int i) __attribute__((always_inline)) -> const uint4 * {
        const int r = tid / KCH + i * KRPI, c = tid % KCH;
        const int kr = k0 + r < a.Sk ? k0 + r : a.Sk - 1;
        return (const uint4 *)(Kp + (long)kr * D + c * 8);
    };
    auto k_dst = [&](int i) __attribute__((always_inline)) -> uint4 * {
        const int r = tid / KCH + i * KRPI, c = tid % KCH;
        return (uint4 *)(Ks + r * kKRow + c * 16);
    };
    auto load_k = [&](int k0) __attribute__((always_inline)) {
        kr0 = *k_src(k0, 0);
        kr1 = *k_src(k0, 1);
        if constexpr (KIT > 2) {
            kr2 = *k_src(k0, 2);
            kr3 = *k_src(k0, 3);
        }
    };
    auto store_k = [&]() __attribute__((always_inline)) {
        *k_dst(0) = kr0;
        *k_dst(1) = kr1;
        if constexpr (KIT > 2) {
            *k_dst(2) = kr2;
            *k_dst(3) = kr3;
        }
    };
    // V is transposed while staged: a thread takes TWO adjacent keys and one 8-wide d-chunk, so every LDS write
    // is a 32-bit {V[k][d], V[k+1][d]} pair into row d of V^T
    auto v_src = [&](int k0, int it, int which) __attribute__((always_inline)) -> const uint4 * {
        const int item = it * 256 + tid;
        const int kp = item & 31, c = item >> 5;
        int kk = k0 + 2 * kp + which;
        kk = kk < a.Sk ? kk : a.Sk - 1;
        return (const uint4 *)(Vp + (long)kk * D + c * 8);
    };
    auto load_v = [&](int k0) __attribute__((always_inline)) {
        va0 = *v_src(k0, 0, 0);
        va1 = *v_src(k0, 0, 1);
        if constexpr (VIT > 1) {
            vb0 = *v_src(k0, 1, 0);
            vb1 = *v_src(k0, 1, 1);
        }
    };
    auto put_v = [&](int it, const uint4 &x0, const uint4 &x1) __attribute__((always_inline)) {
        const int item = it * 256 + tid;
        const int kp = item & 31, c = item >> 5;
        unsigned char *base = Vt + (c * 8) * kVRow + kp * 4;
        *(uint32_t *)(base + 0 * kVRow) = (x0.x & 0xFFFFu) | (x1.x << 16);
        *(uint32_t *)(base + 1 * kVRow) = (x0.x >> 16) | (x1.x & 0xFFFF0000u);
        *(uint32_t *)(base + 2 * kVRow) = (x0.y & 0xFFFFu) | (x1.y << 16);
        *(uint32_t *)(base + 3 * kVRow) = (x0.y >> 16) | (x1.y & 0xFFFF0000u);
        *(uint32_t *)(base + 4 * kVRow) = (x0.z & 0xFFFFu) | (x1.z << 16);
        *(uint32_t *)(base + 5 * kVRow) = (x0.z >> 16) | (x1.z & 0xFFFF0000u);
        *(uint32_t *)(base + 6 * kVRow) = (x0.w & 0xFFFFu) | (x1.w << 16);
        *(uint32_t *)(base + 7 * kVRow) = (x0.w >> 16) | (x1.w & 0xFFFF0000u);
    };
    auto store_v = [&]() __attribute__((always_inline)) {
        put_v(0, va0, va1);
        if constexpr (VIT > 1) put_v(1, vb0, vb1);
    };
    auto next_live = [&](int kt, unsigned long long lv) __attribute__((always_inline)) -> int {      // first live tile index >= kt (ntiles if none)
        if (ntiles > 64) return kt;
        while (kt < ntiles && !((lv >> kt) & 1ull)) ++kt;
        return kt;
    };

    // scores of one 64-key tile for this lane's query: t[4 key-tiles][4] (keys 16*j + 4*g + r)
    // The four mask vectors of a tile are requested BEFORE the next tile's K / V prefetch is issued: vector memory
    // operations retire in order, so waiting for a mask value issued after the prefetch would drain the prefetch.
    uint2 mk0, mk1, mk2, mk3;
    mk0 = mk1 = mk2 = mk3 = uint2{0u, 0u};
    auto load_mask = [&](int k0) __attribute__((always_inline)) {
        if (Mp && !simple) {           // Sk % 4 == 0 (checked on the host): a group of 4 keys is entirely inside or outside
            const int kb = k0 + g * 4;
            mk0 = *(const uint2 *)(Mp + (kb + 4 <= a.Sk ? kb : a.Sk - 4));
            mk1 = *(const uint2 *)(Mp + (kb + 20 <= a.Sk ? kb + 16 : a.Sk - 4));
            mk2 = *(const uint2 *)(Mp + (kb + 36 <= a.Sk ? kb + 32 : a.Sk - 4));
            mk3 = *(const uint2 *)(Mp + (kb + 52 <= a.Sk ? kb + 48 : a.Sk - 4));
        }
    };
    auto scores = [&](int k0, float (&t)[4][4]) __attribute__((always_inline)) {
        const uint2 mk[4] = {mk0, mk1, mk2, mk3};
        f32x4_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *(const bf16x8_t *)(Ks + (j * 16 + qc) * kKRow + ks * 64 + g * 16);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kbase = k0 + j * 16 + g * 4;
            float mv[4] = {qt_u2f(mk[j].x << 16), qt_u2f(mk[j].x & 0xFFFF0000u), qt_u2f(mk[j].y << 16),
                           qt_u2f(mk[j].y & 0xFFFF0000u)};
            if (simple) {
                const float mn = qt_u2f(0xFF7F0000u);        // the bf16 minimum, what the mask holds from the row's extent on
#pragma unroll
                for (int r = 0; r < 4; ++r) mv[r] = kbase + r < lv ? 0.0f : mn;
            }
#pragma unroll
            for (int r = 0; r < 4; r += 2) {                   // two scores per packed conversion
                uint32_t w = pack_bf16x2(acc[j][r], acc[j][r + 1]);                                         // matmul output, bf16
                w = pack_bf16x2(qt_u2f(w << 16) * a.scaling, qt_u2f(w & 0xFFFF0000u) * a.scaling);          // attn_scaling (MulFunctional), bf16
                if (a.mask) w = pack_bf16x2(qt_u2f(w << 16) + mv[r], qt_u2f(w & 0xFFFF0000u) + mv[r + 1]);  // + mask, bf16
                t[j][r] = (kbase + r < a.Sk) ? qt_u2f(w << 16) : -INFINITY;      // keys past the end contribute nothing
                t[j][r + 1] = (kbase + r + 1 < a.Sk) ? qt_u2f(w & 0xFFFF0000u) : -INFINITY;
            }
        }
    };

    // ---- which key tiles can matter?  A tile whose additive mask is <= -1e30 for all 64 query rows of this
    // workgroup contributes exp(.) == 0 to every row sum and 0 to P.V (t equals the mask value exactly there),
    // so it is skipped -- for a causal mask that is half of all tiles.  The one case where that would change
    // the result (a row with NO unmasked key at all, which torch turns into a uniform row) is detected after
    // pass 1 and the whole thing is redone without skipping.
    unsigned long long live = ~0ull;
    if (simple && ntiles <= 64) {
        // tiles at or beyond the largest extent of the workgroup's 64 rows are dead for all of them (an extent of 0 is a row without any
        // unmasked key: it keeps every tile alive, and the "row masked everywhere" redo below is what serves it)
        int mx = (qrow < a.Sq) ? (lv <= 0 ? a.Sk : lv) : 0;
        for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
        int *s_mx = (int *)lds;                                              // LDS is not in use yet
        if (lane == 0) s_mx[wave] = mx;
        __syncthreads();
        mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        __syncthreads();
        const int nlive = min(ntiles, (mx + kBK - 1) / kBK);
        live = nlive >= 64 ? ~0ull : ((1ull << nlive) - 1ull);
        if (nlive == ntiles) live = ~0ull;
    } else if (a.mask && ntiles <= 64) {
        live = 0ull;
        const int qr = qb * kBQ + (tid >> 2);
        const uint16_t *mrow = a.mask + b * a.mask_sb + h * a.mask_sh + (long)(qr < a.Sq ? qr : a.Sq - 1) * a.mask_sq;
        // bf16 pattern >= 0xF14A  <=>  value <= -1e30 (negative, magnitude >= 1e30; -inf and NaN-free masks)
        const bool vec_ok = ((a.mask_sb | a.mask_sh | a.mask_sq) % 8 == 0) && (((uintptr_t)a.mask & 15u) == 0) && (a.Sk % 8 == 0);
        // every thread scans its 16-key slice of all tiles first (loads independent of each other, no barrier in
        // between), then ONE block-wide AND of the per-thread "tile is dead" bit sets
        unsigned long long dead_bits = 0ull;
        for (int kt = 0; kt < ntiles; ++kt) {
            bool dead = true;
            const int kb = kt * kBK + (tid & 3) * 16;
            if (vec_ok) {
#pragma unroll
                for (int hv = 0; hv < 2; ++hv) {
                    if (kb + hv * 8 < a.Sk) {
                        const uint4 w = *(const uint4 *)(mrow + kb + hv * 8);
                        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) dead &= ((ws[e] & 0xFFFFu) >= 0xF14Au) & ((ws[e] >> 16) >= 0xF14Au);
                    }
                }
            } else {
                for (int e = 0; e < 16; ++e)
                    if (kb + e < a.Sk) dead &= mrow[kb + e] >= 0xF14Au;
            }
            if (dead) dead_bits |= 1ull << kt;
        }
        // wave AND by shuffles, then across the four waves through LDS
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)dead_bits, off, 64), hi = __shfl_xor((unsigned)(dead_bits >> 32), off, 64);
            dead_bits &= ((unsigned long long)hi << 32) | lo;
        }
        unsigned long long *s_dead = (unsigned long long *)lds;              // LDS is not in use yet
        if (lane == 0) s_dead[wave] = dead_bits;
        __syncthreads();
        dead_bits = s_dead[0] & s_dead[1] & s_dead[2] & s_dead[3];
        __syncthreads();
        live = ~dead_bits & (ntiles == 64 ? ~0ull : ((1ull << ntiles) - 1ull));
        if (live == (ntiles == 64 ? ~0ull : ((1ull << ntiles) - 1ull))) live = ~0ull;     // nothing to skip
    }

    // ---- pass 1: row max and row sum ----------------------------------------------------------------
    float m = -INFINITY, l = 0.0f;
    for (int attempt = 0; attempt < 2; ++attempt) {
    m = -INFINITY; l = 0.0f;
    int kt = next_live(0, live);
    if (kt < ntiles) load_k(kt * kBK);
    while (kt < ntiles) {
        store_k();
        __syncthreads();
        const int nxt = next_live(kt + 1, live);
        load_mask(kt * kBK);
        if (nxt < ntiles) load_k(nxt * kBK);          // in flight during this tile's MFMAs
        float t[4][4];
        scores(kt * kBK, t);
        float mx = m;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, t[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float ref = mx == -INFINITY ? 0.0f : mx;   // a fully -inf prefix must not produce inf - inf
        float part = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) part += __expf(t[j][r] - ref);
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        l = (m == -INFINITY ? 0.0f : l * __expf(m - ref)) + part;
        m = mx;
        __syncthreads();
        kt = nxt;
    }
    if (live == ~0ull || !__syncthreads_or(m <= -1e30f)) break;
    live = ~0ull;                                   // some row is masked everywhere: count every tile
    }
    const float inv = 1.0f / l;

    // ---- pass 2: probabilities, fake-quant, P.V -----------------------------------------------------------
    f32x4_t o[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint32_t amax = 0;
    int kt = next_live(0, live);
    if (kt < ntiles) { load_k(kt * kBK); load_v(kt * kBK); }
    while (kt < ntiles) {
        store_k();
        store_v();
        __syncthreads();
        const int nxt = next_live(kt + 1, live);
        load_mask(kt * kBK);
        if (nxt < ntiles) { load_k(nxt * kBK); load_v(nxt * kBK); }
        float t[4][4];
        scores(kt * kBK, t);
        uint32_t pw[4][2];                            // quantised probabilities, bf16 pairs per key tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const uint32_t p = pack_bf16x2(__expf(t[j][2 * hf] - m) * inv, __expf(t[j][2 * hf + 1] - m) * inv);
                if constexpr (OBS) {
                    const uint32_t a0 = (p << 16) & 0x7FFFFFFFu, a1 = p & 0x7FFF0000u;
                    amax = amax > a0 ? amax : a0;
                    amax = amax > a1 ? amax : a1;
                }
                uint32_t lo = p << 16, hi = p & 0xFFFF0000u;
                if constexpr (UNIT && KIND == QT_FMT_FP_SAT) {
                    if (a.p8) {
                        // a probability is finite-or-NaN and in [0, 1]: no saturation, no sign, so the format's RNE is
                        // the hardware conversion and the quantized bf16 value is the decode of that byte pair
                        float2_t d;
                        if (a.p8 == 2) d = __builtin_amdgcn_cvt_pk_f32_bf8(__builtin_amdgcn_cvt_pk_bf8_f32(qt_u2f(lo), qt_u2f(hi), 0, false), false);
                        else d = __builtin_amdgcn_cvt_pk_f32_fp8(__builtin_amdgcn_cvt_pk_fp8_f32(qt_u2f(lo), qt_u2f(hi), 0, false), false);
                        pw[j][hf] = pack_bf16x2(d.x, d.y);
                        continue;
                    }
                }
                if constexpr (!UNIT) {
                    const uint32_t qd = pack_bf16x2(dv.exact(qt_u2f(lo)), dv.exact(qt_u2f(hi)));
                    lo = qd << 16;
                    hi = qd & 0xFFFF0000u;
                }
                const uint32_t r0 = rnd(lo), r1 = rnd(hi);
                if constexpr (UNIT) pw[j][hf] = (r0 >> 16) | (r1 & 0xFFFF0000u);
                else pw[j][hf] = pack_bf16x2(qt_u2f(r0) * s, qt_u2f(r1) * s);
            }
        }
        // P.V: k-step s2 covers key tiles 2*s2 and 2*s2+1; element j of lane group g is key
        // 32*s2 + (j < 4 ? 4*g + j : 16 + 4*g + j - 4) for BOTH operands
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint4 pa = {pw[2 * s2][0], pw[2 * s2][1], pw[2 * s2 + 1][0], pw[2 * s2 + 1][1]};
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pa);
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const unsigned char *row = Vt + (dt * 16 + qc) * kVRow + (32 * s2 + 4 * g) * 2;
                const uint2 v0 = *(const uint2 *)row, v1 = *(const uint2 *)(row + 32);
                const uint4 vb = {v0.x, v0.y, v1.x, v1.y};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, __builtin_bit_cast(bf16x8_t, vb), o[dt], 0, 0, 0);
            }
        }
        __syncthreads();
        kt = nxt;
    }

    // ---- output: O[q = q0 + 4*g + r][d = 16*dt + qc] in [B, Sq, H, D] ----------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qo = q0 + 4 * g + r;
        if (qo >= a.Sq) continue;
        uint16_t *dst = a.out + (((long)b * a.Sq + qo) * a.H + h) * D + qc;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            uint32_t v = pack_bf16x2(o[dt][r], 0.0f);
            if (a.out_fq) v = rnd(v << 16) >> 16;
            dst[dt * 16] = (uint16_t)v;
        }
    }
    if constexpr (OBS) {
        // rows past Sq were clamped duplicates of a valid row: they cannot raise the max
        amax = wave_max_u32(amax);
        if (lane == 0 && amax != 0u && amax > __hip_atomic_load(a.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.amax, amax);
    }
}

template <int D, int KIND>
int launch_attn_kind(const AttnArgs &a, hipStream_t st) {
    dim3 grid((unsigned)(((a.Sq + kBQ - 1) / kBQ) * a.B * a.H));
    const bool unit = a.scale == nullptr, obs = a.amax != nullptr;
    if (unit && obs) attention_fq_kernel<D, KIND, true, true><<<grid, 256, 0, st>>>(a);
    else if (unit) attention_fq_kernel<D, KIND, true, false><<<grid, 256, 0, st>>>(a);
    else if (obs) attention_fq_kernel<D, KIND, false, true><<<grid, 256, 0, st>>>(a);
    else attention_fq_kernel<D, KIND, false, false><<<grid, 256, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int D>
int launch_attn(const AttnArgs &a, hipStream_t st) {
    switch (a.fmt.kind) {
        case QT_FMT_LUT: return (a.fmt.p1 & 1) ? launch_attn_kind<D, kFmtRows>(a, st) : launch_attn_kind<D, QT_FMT_LUT>(a, st);
        case QT_FMT_FP_SAT: return launch_attn_kind<D, QT_FMT_FP_SAT>(a, st);
        case QT_FMT_INT: return launch_attn_kind<D, QT_FMT_INT>(a, st);
        case QT_FMT_IDENTITY: return launch_attn_kind<D, QT_FMT_IDENTITY>(a, st);
        default: return QT_ERR_BAD_ARG;
    }
}

}  // namespace

static int attention_fq_launch(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *mask,
                               uint16_t *out, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh,
                               long mask_sq, float scaling, const qt_format *fmt, const uint16_t *lut,
                               const float *scale, uint32_t *amax, int out_fq, void *stream, const int *row_live = nullptr,
                               long lsb = 0, long lsh = 0, long lsq = 0, const int *irregular = nullptr) {
    if (B == 0 || H == 0 || Sq == 0) return QT_OK;
    if (!q || !k || !v || !out || !fmt || B < 0 || H < 1 || Sq < 0 || Sk < 1 || (long)B * H > 65535) return QT_ERR_BAD_ARG;
    if ((D != 64 && D != 128) || (Sk & 3)) return QT_ERR_BAD_ARG;
    if (fmt->kind == QT_FMT_LUT && !lut) return QT_ERR_BAD_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15u) || ((uintptr_t)mask & 7u) ||
        (mask && ((mask_sb | mask_sh | mask_sq) & 3)))
        return QT_ERR_UNALIGNED;
    int p8 = 0;
    if (fmt->kind == QT_FMT_FP_SAT && !scale) {
        if (fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f) p8 = 1;
        else if (fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f) p8 = 2;
    }
    if (out_fq && (scale || fmt->kind == QT_FMT_IDENTITY)) return QT_ERR_BAD_ARG;      // unit scale, a real format
    // causal-style masks make late query blocks heavier: balance the slots when the grid is not a multiple that the mirrored order
    // already serves
    int snake_mode = 1;
#ifdef QT_TUNING_BUILD
    if (const char *e = getenv("QT_ATTN_SNAKE")) snake_mode = atoi(e);     // tools/ only: 0 keeps the mirrored order
#endif
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long total = (long)((Sq + kBQ - 1) / kBQ) * B * H;
    const int snake = (snake_mode && mask && total > cus && total % (2L * cus) != 0) ? cus : 0;
    if ((row_live != nullptr) != (irregular != nullptr) || (row_live && !mask)) return QT_ERR_BAD_ARG;
    AttnArgs a{q, k, v, mask, out, B, H, Sq, Sk, mask_sb, mask_sh, mask_sq, scaling, *fmt, lut, scale, amax, p8,
               row_live, lsb, lsh, lsq, irregular, snake, out_fq};
    hipStream_t st = (hipStream_t)stream;
    return D == 128 ? launch_attn<128>(a, st) : launch_attn<64>(a, st);
}

extern "C" int qt_attention_fq_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *mask,
                                    uint16_t *out, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh,
                                    long mask_sq, float scaling, const qt_format *fmt, const uint16_t *lut,
                                    const float *scale, uint32_t *amax, void *stream) {
    return attention_fq_launch(q, k, v, mask, out, B, H, Sq, Sk, D, mask_sb, mask_sh, mask_sq, scaling, fmt, lut, scale, amax, 0, stream);
}

extern "C" int qt_attention_fq_live_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *mask,
                                         uint16_t *out, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh,
                                         long mask_sq, float scaling, const qt_format *fmt, const uint16_t *lut, const float *scale,
                                         uint32_t *amax, int out_fq, const int *row_live, long live_sb, long live_sh, long live_sq,
                                         const int *mask_irregular, void *stream) {
    return attention_fq_launch(q, k, v, mask, out, B, H, Sq, Sk, D, mask_sb, mask_sh, mask_sq, scaling, fmt, lut, scale, amax, out_fq, stream,
                               row_live, live_sb, live_sh, live_sq, mask_irregular);
}

extern "C" int qt_attention_fq_out_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *mask,
                                        uint16_t *out, int B, int H, int Sq, int Sk, int D, long mask_sb, long mask_sh,
                                        long mask_sq, float scaling, const qt_format *fmt, const uint16_t *lut, void *stream) {
    return attention_fq_launch(q, k, v, mask, out, B, H, Sq, Sk, D, mask_sb, mask_sh, mask_sq, scaling, fmt, lut, nullptr, nullptr, 1, stream);
}
