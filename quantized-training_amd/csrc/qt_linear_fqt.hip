// Fake-quant Linear for ANY value map: the weight's fake-quantizer inside a bf16 matrix-core GEMM.
//
// Replaces   F.linear(input, self.weight_fake_quant(self.weight), self.bias)       modules/qat/linear.py:40-41
// for stateless weight fake-quantizers whose format is not FP8-exact (posit(n, es), intN, fp6 / fp4, ...: fake_quantize.py:31-95):
//   * the activation arrives as the bf16 VALUES of fq(x) (the elementwise pass or a producer kernel wrote them);
//   * the bf16 weight is read ONCE from HBM, unquantized; a lane loads 8 weights, sends each through the ROW FORM of the value
//     map (below) and writes the 8 quantized bf16 values into an LDS tile -- fq(W) never exists in HBM;
//   * v_mfma_f32_16x16x32_bf16 on the quantized operands: products of quantized values are exactly the reference's bf16
//     products, fp32 accumulation, bias and the bf16 rounding in the epilogue.
//
// Row form (qt_build_rowparams, csrc/qt_host.cpp): the 128 bf16 patterns that share an exponent are one row of the map, and
// within a row the map is a round-to-nearest-even onto a power-of-two grid plus a clamp,
//     t = f32(|x| bits + D);   y = med3((t + C) - C, lo, hi);   sign of x copied back            D, C, lo, hi: 16 bytes per row
// i.e. 8.5 vector instructions and one 16-byte LDS read (256 rows = 4 KiB, mostly broadcast: the weights of a tile live in a
// dozen exponents) per weight instead of a gather from the 128 KiB map, which would not fit beside the operand rings.  The
// builder verifies each row against the map on all of its inputs; rows that do not fit are flagged (bit 0 of C), lanes OR the
// C words they read, and a workgroup that met a flagged row redoes its tile with the map itself (slow_tile) -- bit-exact for
// every bf16 weight.  Flagged in practice: non-finite inputs and a few rows beyond 2^22 / below 2^-38 of some posits.
//
// Work decomposition as in qt_linear_fq8.hip (variant R): a workgroup owns 256 rows x (16 nt) columns, nt <= 15, chosen so
// that the grid is a whole number of rounds over the CUs; 8 waves = 4 row bands x 2 column halves; up to four weights sharing
// one activation are the segments of one launch; tile ids dealt so that the row tiles of a column tile share an XCD.
// LDS: row table 4 / 8 KiB | activation ring (3 x 32 KiB: 256 rows x 64 bf16, 16-byte chunks XOR-swizzled by row, filled by
// LDS-DMA) | quantized weight ring (2 x nt x 2 KiB, same layout, filled by ds_write_b128).  One raw barrier per k step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_formats.h"

namespace {

typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kTM = 256, kBK = 64, kRowBytes = 128, kMaxSeg = 4;
constexpr int kABytes = kTM * kRowBytes;            // one activation tile: 32 KiB of bf16
constexpr int kGroupBytes = 16 * kRowBytes;         // one 16-row column group of a weight tile: 2 KiB

struct Segment {
    const uint16_t *w;        // [n][K] bf16
    const uint16_t *bias;     // [n] bf16 or NULL
    int g0;                   // first 16-column group of this weight in the concatenation of all weights
};

struct Args {
    const uint16_t *x;        // [M][K] bf16 values of fq(x)
    uint16_t *y;              // [M][ldc] bf16
    const uint32_t *rows;     // [512][4] row parameters (device)
    const uint16_t *map;      // [65536] value map (device): the redo path
    uint32_t sign_mask;
    int M, K, ldc;
    int tiles_m, tiles_n, nseg;
    int gbase, gextra;        // column tile j covers gbase (+1 for gextra of them) groups of 16 columns
    int nb;                   // weight pieces (8 rows x 128 bytes) per wave and k step the widest tile needs
    Segment seg[kMaxSeg];
};

struct SegRef { const uint16_t *w, *bias; int g0; };
// by compile-time indices only: a run-time index into the kernel-argument struct makes hipcc copy it to scratch
__device__ __forceinline__ SegRef seg_lookup(const Args &a, int grp) {
    SegRef r{a.seg[0].w, a.seg[0].bias, a.seg[0].g0};
    if (a.nseg > 1 && grp >= a.seg[1].g0) r = SegRef{a.seg[1].w, a.seg[1].bias, a.seg[1].g0};
    if (a.nseg > 2 && grp >= a.seg[2].g0) r = SegRef{a.seg[2].w, a.seg[2].bias, a.seg[2].g0};
    if (a.nseg > 3 && grp >= a.seg[3].g0) r = SegRef{a.seg[3].w, a.seg[3].bias, a.seg[3].g0};
    return r;
}

// column tile tn of tiles_n: first 16-column group and group count (the wider tiles spread evenly over the tile index)
__device__ __forceinline__ void tile_span(const Args &a, int tn, int &first, int &count) {
    const long units = (long)a.gbase * a.tiles_n + a.gextra;
    first = (int)(tn * units / a.tiles_n);
    count = (int)((tn + 1) * units / a.tiles_n) - first;
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int OFF>
__device__ __forceinline__ u32x4 ds_read128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// both operand tiles: row-major 128-byte rows (64 bf16), 16-byte chunk index XOR ((row >> 1) & 7)
__device__ __forceinline__ int chunk_off(int row, int chunk) { return row * kRowBytes + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- the row form on registers ---------------------------------------------------------------------------------------------
// A table row is {D, C (bit 0: flagged), lo, hi}; the table sits at LDS address 0, so a row's address is (bits >> 7) << 4.
// ROWMASK: 0xFF0 = rows by exponent, 0x1FF0 = rows by sign and exponent (maps whose negative half is not the mirror image).
template <uint32_t ROWMASK>
__device__ __forceinline__ void row_addrs(uint32_t x, uint32_t &a0, uint32_t &a1) {
    const uint32_t s = x >> 3;
    a0 = s & ROWMASK;
    a1 = (s >> 16) & ROWMASK;
}
__device__ __forceinline__ u32x4 ds_gather128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// two packed bf16 weights and their rows -> the two packed values of the map
__device__ __forceinline__ uint32_t quant_pair(uint32_t x, const u32x4 &p0, const u32x4 &p1, uint32_t sign_mask) {
    const float t0 = qt_u2f((x << 16) + p0.x);                // the sign bit rides along; |t| below
    const float t1 = qt_u2f((x & 0xFFFF0000u) + p1.x);
    const float c0 = qt_u2f(p0.y), c1 = qt_u2f(p1.y);
    const float z0 = __builtin_amdgcn_fmed3f((__builtin_fabsf(t0) + c0) - c0, qt_u2f(p0.z), qt_u2f(p0.w));
    const float z1 = __builtin_amdgcn_fmed3f((__builtin_fabsf(t1) + c1) - c1, qt_u2f(p1.z), qt_u2f(p1.w));
    // results are exact bf16 values: the high halves are the answer
    return __builtin_amdgcn_perm(qt_f2u(z1), qt_f2u(z0), 0x07060302u) | (x & sign_mask);
}

// The LDS operations of one wave and k step, in issue order (they complete in that order, which is what makes counted waits
// possible).  A "unit" is half a weight piece of the lane: 4 weights = 2 packed words = 4 table rows.
//     RAW x NB                  the lane's 16 raw bytes of each of its pieces (weight tile t+1)
//     FA x 8, RB(0) x 2, RB(1) x 2     activation fragments of the step, weight fragments of column groups 0 and 1
//     [wait RAW(0)]  G(0) x 4   table rows of unit 0
//     per column group J:  RB(J+2) x 2;  [wait FA / RB(J)];  then per unit u the group carries:
//                          G(u+1) x 4;  [wait G(u)];  (arithmetic of unit u);  WR(u) x 1      -- and the group's 8 multiplications
// walk() replays that order and returns how many operations were issued after the last one of (from_kind, from_idx) when the
// wait (to_kind, to_idx) is reached: the lgkmcnt that wait may leave outstanding.
enum { kOpRaw = 0, kOpFA = 1, kOpRB = 2, kOpG = 3, kOpWR = 4, kWaitRaw = 10, kWaitF = 11, kWaitG = 12 };
constexpr int unit_group(int u, int NTW, int U) { return NTW > 0 ? u * NTW / U : 0; }
constexpr int walk(int NTW, int NB, int from_kind, int from_idx, int to_kind, int to_idx) {
    const int U = 2 * NB;
    int n = -1;
    bool done = false;
    int result = -1;
    auto op = [&](int kind, int idx, int count) {
        if (n >= 0) n += count;
        if (kind == from_kind && idx == from_idx) n = 0;
    };
    auto wait = [&](int kind, int idx) {
        if (!done && kind == to_kind && idx == to_idx) { result = n; done = true; }
    };
    for (int i = 0; i < NB; ++i) op(kOpRaw, i, 1);
    if (NTW > 0) {
        op(kOpFA, 0, 8);
        op(kOpRB, 0, 2);
        if (NTW > 1) op(kOpRB, 1, 2);
    }
    wait(kWaitRaw, 0);
    op(kOpG, 0, 4);
    for (int J = 0; J < (NTW > 0 ? NTW : 1); ++J) {
        if (NTW > 0) {
            if (J + 2 < NTW) op(kOpRB, J + 2, 2);
            wait(kWaitF, J);
        }
        for (int u = 0; u < U; ++u) {
            if (unit_group(u, NTW, U) != J) continue;
            if (u + 1 < U) op(kOpG, u + 1, 4);
            wait(kWaitG, u);
            op(kOpWR, u, 1);
        }
    }
    // the counter has four bits: the 16th operation issued behind one cannot issue before that one has completed
    return result > 15 ? 15 : result;
}

// ABL (timing experiments only, QT_FQT_ABLATE; results are garbage): 1 no multiplications, 2 no weight items at all, 3 no
// activation DMA, 5 weight items without the conversion (DMA, LDS round trip), 6 conversion without the table gathers
//
// NB: weight pieces (8 rows x 128 bytes of bf16) per wave and k step (1-4: tiles of up to 4, 8, 12, 15 column groups);
// SROWS: the table has 512 rows (sign and exponent).
//
// k step t of a wave (every LDS / DMA operation is inline asm, every wait a computed count):
//   top      s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      everything requested during step t-1 has landed, every wave's converted
//                                                          weights of step t are written, every wave is done with step t-1
//   first    request the activation tile of step t+1 (ring of 2) and the RAW weight tile of step t+2 (ring of 3): 4 + NB DMA
//            pieces, all in front, so that they have the whole step to land
//   then     the LDS stream above: 8 multiplications per column group on tile t, and 2 NB units of weight tile t+1 converted IN
//            PLACE, spread evenly over the groups; the rows of unit u+1 are in flight while unit u is computed
template <int NB, bool SROWS, int ABL = 0>
struct LinearFqt {
    static constexpr int kADepth = 2, kWDepth = 3, kU = 2 * NB;
    static constexpr int kMaxNT = NB * 4 < 15 ? NB * 4 : (SROWS ? 14 : 15);
    static constexpr int kTbl = SROWS ? 8192 : 4096;
    static constexpr int kWBytes = kMaxNT * kGroupBytes;
    static constexpr int kDummy = 1024;                     // where surplus pieces go (a tile of 15 groups has 30, the waves request 32)
    static constexpr int kLds = kTbl + kADepth * kABytes + kWDepth * kWBytes + kDummy;
    static_assert(kLds <= 160 * 1024, "LDS budget");
    static constexpr uint32_t kRowMask = SROWS ? 0x1FF0u : 0xFF0u;
    static constexpr int kEpiStride = 64 * ((2 * NB < 8 ? 2 * NB : 8) * 32 + 8);      // a wave's epilogue tile in LDS
    static_assert(8 * kEpiStride + 16 <= kADepth * kABytes + kWDepth * kWBytes, "epilogue tiles fit in the dead rings");

    // One wave's share: rows [wm * 64, +64) x NTW column groups starting at group jbase of the tile whose first group is tg0.
    // Returns true when a flagged row was met (the caller redoes the tile).
    template <int NTW>
    static __device__ __forceinline__ bool run(const Args &a, uint8_t *lds, int m0, int tg0, int nt, int jbase, int w, int l) {
        const int r = l & 15, g = l >> 4, wm = w & 3;
        const int nk = a.K / kBK, klast = nk - 1;
        const long krow = (long)a.K * 2;                        // bytes per row of x and W
        // LDS map (the dynamic segment starts at address 0: the kernel has no static LDS)
        constexpr uint32_t a0 = kTbl, w0 = a0 + kADepth * kABytes, dummy = w0 + kWDepth * kWBytes;
        // ---- the row table into LDS (every thread one 16-byte row)
        {
            const int t = w * 64 + l;
            if (t < (SROWS ? 512 : 256)) {
                const u32x4 v = *(const u32x4 *)(a.rows + t * 4);
                asm volatile("ds_write_b128 %0, %1" ::"v"(t * 16), "v"(v) : "memory");
            }
        }
        // ---- DMA sources: a scalar base (advanced by 128 bytes per k tile) + a 32-bit lane offset; LDS destinations are
        // wave-uniform.  Activations: 32 pieces of 8 rows x 128 bytes, four per wave.
        uint32_t ga[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (w * 4 + i) * 8 + (l >> 3), slot = l & 7;
            ga[i] = (uint32_t)((long)min(m0 + row, a.M - 1) * krow) + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        // Weight pieces: piece p = w + 8 i covers rows 8 p .. 8 p + 7 of the tile; the lane's 16 bytes land at piece base + 16 l and
        // are converted there.  Surplus pieces (p >= 2 nt) re-request piece w & 1 into the dummy kilobyte.
        const int npieces = nt * 2;
        uint32_t gw[NB];
        const uint16_t *wbase[NB];                             // wave-uniform
        uint32_t wofs[NB];                                     // piece base inside a weight stage (or the dummy), wave-uniform
        bool real[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = w + 8 * i;
            real[i] = p < npieces;
            const int pb = real[i] ? p : (w & 1);
            const int grp = tg0 + (pb >> 1);
            const int row = pb * 8 + (l >> 3), slot = l & 7;
            const SegRef sg = seg_lookup(a, grp);
            wbase[i] = sg.w;
            gw[i] = (uint32_t)((long)((grp - sg.g0) * 16 + (pb & 1) * 8 + (l >> 3)) * krow) + ((slot ^ ((row >> 1) & 7)) << 4);
            wofs[i] = real[i] ? (uint32_t)pb * 1024u : 0xFFFFFFFFu;
        }
        uint32_t flags = 0;
        const uint32_t sign_mask = a.sign_mask;
        auto dma16 = [](const void *base, uint32_t off, uint32_t dst) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
        };
        auto ds_write64 = [](uint32_t addr, u32x2 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        // all of a step's requests: activation tile ka into stage `as`, raw weight tile kb into stage `ws`
        auto request = [&](int ka, uint32_t as, int kb, uint32_t ws) __attribute__((always_inline)) {
            if constexpr (ABL != 3) {
                const uint8_t *xb = (const uint8_t *)a.x + (long)ka * kRowBytes;
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16(xb, ga[i], as + (w * 4 + i) * 1024);
            }
            if constexpr (ABL != 2) {
#pragma unroll
                for (int i = 0; i < NB; ++i) dma16((const uint8_t *)wbase[i] + (long)kb * kRowBytes, gw[i], real[i] ? ws + wofs[i] : dummy);
            }
        };
        // the lane's 16 bytes of piece i, relative to a weight stage (surplus pieces: relative address of the dummy from stage 0;
        // they are only ever read and written by this lane, whatever the stage)
        uint32_t pofs[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) pofs[i] = (real[i] ? wofs[i] : dummy - w0) + (uint32_t)l * 16u;
        u32x4 raw[NB];
        u32x4 rows[2][4];                                        // table rows of the unit being computed / of the next one
        // ---- the unit stream
        auto raw_word = [&](auto uc, int k) __attribute__((always_inline)) -> uint32_t {       // packed word k (0, 1) of unit U
            constexpr int U = decltype(uc)::value;
            return (U & 1) ? (k ? raw[U >> 1].w : raw[U >> 1].z) : (k ? raw[U >> 1].y : raw[U >> 1].x);
        };
        auto unit_gather = [&](auto uc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            if constexpr (ABL != 2 && ABL != 5 && ABL != 6) {
                uint32_t ad[4];
                row_addrs<kRowMask>(raw_word(uc, 0), ad[0], ad[1]);
                row_addrs<kRowMask>(raw_word(uc, 1), ad[2], ad[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) rows[U & 1][e] = ds_gather128(ad[e]);
            }
        };
        auto unit_finish = [&](auto uc, uint32_t wc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value, I = U >> 1;
            if constexpr (ABL != 2) {
                u32x4(&rp)[4] = rows[U & 1];
                if constexpr (ABL == 6) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rp[e] = u32x4{0u, 0x4B400000u, 0u, 0x7F000000u};
                }
                u32x2 q = {raw_word(uc, 0), raw_word(uc, 1)};
                if constexpr (ABL != 5) {
                    q.x = quant_pair(raw_word(uc, 0), rp[0], rp[1], sign_mask);
                    q.y = quant_pair(raw_word(uc, 1), rp[2], rp[3], sign_mask);
                    const uint32_t f = (rp[0].y | rp[1].y) | (rp[2].y | rp[3].y);
                    flags |= real[I] ? f : 0u;
                }
                ds_write64((real[I] ? wc : w0) + pofs[I] + (U & 1) * 8, q);
            }
        };
        // unit slot: the next unit's rows requested, this unit's rows waited for, computed, written back
        auto unit_slot = [&](auto uc, uint32_t wc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            if constexpr (U + 1 < kU) unit_gather(std::integral_constant<int, U + 1>{});
            if constexpr (ABL != 2 && ABL != 5 && ABL != 6) {
                constexpr int n = walk(NTW, NB, kOpG, U, kWaitG, U);
                static_assert(n >= 0, "schedule");
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(rows[U & 1][0]), "+v"(rows[U & 1][1]), "+v"(rows[U & 1][2]), "+v"(rows[U & 1][3]) : "n"(n));
            }
            unit_finish(uc, wc);
        };
        // the units column group J carries
        auto units_of = [&](auto jc, uint32_t wc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;      // (+ 0 * J below: the calls must depend on J, or every one is instantiated)
            if constexpr (0 < kU && unit_group(0, NTW, kU) == J) unit_slot(std::integral_constant<int, 0 + 0 * J>{}, wc);
            if constexpr (1 < kU && unit_group(1, NTW, kU) == J) unit_slot(std::integral_constant<int, 1 + 0 * J>{}, wc);
            if constexpr (2 < kU && unit_group(2, NTW, kU) == J) unit_slot(std::integral_constant<int, 2 + 0 * J>{}, wc);
            if constexpr (3 < kU && unit_group(3, NTW, kU) == J) unit_slot(std::integral_constant<int, 3 + 0 * J>{}, wc);
            if constexpr (4 < kU && unit_group(4, NTW, kU) == J) unit_slot(std::integral_constant<int, 4 + 0 * J>{}, wc);
            if constexpr (5 < kU && unit_group(5, NTW, kU) == J) unit_slot(std::integral_constant<int, 5 + 0 * J>{}, wc);
            if constexpr (6 < kU && unit_group(6, NTW, kU) == J) unit_slot(std::integral_constant<int, 6 + 0 * J>{}, wc);
            if constexpr (7 < kU && unit_group(7, NTW, kU) == J) unit_slot(std::integral_constant<int, 7 + 0 * J>{}, wc);
        };
        // the step's first LDS operations: raw pieces of the tile to convert (stage wc)
        auto read_raw = [&](uint32_t wc) __attribute__((always_inline)) {
            if constexpr (ABL != 2) {
#pragma unroll
                for (int i = 0; i < NB; ++i) raw[i] = ds_gather128((real[i] ? wc : w0) + pofs[i]);
            }
        };
        // wait for raw[0] (and hand every raw register to the compiler as defined), then the first unit's rows
        auto first_rows = [&]() __attribute__((always_inline)) {
            if constexpr (ABL != 2) {
                constexpr int n = walk(NTW, NB, kOpRaw, 0, kWaitRaw, 0);
                static_assert(n >= 0, "schedule");
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(raw[0]) : "n"(n));
                unit_gather(std::integral_constant<int, 0>{});
            }
        };
        // the other raw registers are older than the activation fragments: valid once those are (group 0's wait)
        auto raws_defined = [&]() __attribute__((always_inline)) {
            if constexpr (ABL != 2) {
                if constexpr (NB == 2) asm volatile("" : "+v"(raw[1]));
                if constexpr (NB == 3) asm volatile("" : "+v"(raw[1]), "+v"(raw[2]));
                if constexpr (NB == 4) asm volatile("" : "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]));
            }
        };

        v4f acc[4][NTW > 0 ? NTW : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < (NTW > 0 ? NTW : 1); ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // lane-constant parts of the fragment addresses: k sub-step 0 = chunks 0..3, sub-step 1 = chunks 4..7
        const uint32_t a_lo = chunk_off(wm * 64 + r, g), a_hi = chunk_off(wm * 64 + r, 4 + g);
        const uint32_t b_lo = chunk_off(jbase * 16 + r, g), b_hi = chunk_off(jbase * 16 + r, 4 + g);

        // multiplications of step t on (sa_, sb_); conversion of the raw weight tile in stage wc
        auto compute = [&](uint32_t sa_, uint32_t sb_, uint32_t wc) __attribute__((always_inline)) {
            read_raw(wc);
            if constexpr (NTW > 0) {
                u32x4 fa_lo[4], fa_hi[4], fb_lo[3], fb_hi[3];
                fa_lo[0] = ds_read128<0 * 2048>(sa_ + a_lo); fa_hi[0] = ds_read128<0 * 2048>(sa_ + a_hi);
                fa_lo[1] = ds_read128<1 * 2048>(sa_ + a_lo); fa_hi[1] = ds_read128<1 * 2048>(sa_ + a_hi);
                fa_lo[2] = ds_read128<2 * 2048>(sa_ + a_lo); fa_hi[2] = ds_read128<2 * 2048>(sa_ + a_hi);
                fa_lo[3] = ds_read128<3 * 2048>(sa_ + a_lo); fa_hi[3] = ds_read128<3 * 2048>(sa_ + a_hi);
                auto read_b = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    fb_lo[J % 3] = ds_read128<J * 2048>(sb_ + b_lo);
                    fb_hi[J % 3] = ds_read128<J * 2048>(sb_ + b_hi);
                };
                auto step = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int P = J % 3;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (J + 2 < NTW) read_b(std::integral_constant<int, J + 2>{});
                    constexpr int kAhead = walk(NTW, NB, kOpRB, J, kWaitF, J);
                    static_assert(kAhead >= 0, "schedule");
                    if constexpr (J == 0) {
                        asm volatile("s_waitcnt lgkmcnt(%10)"
                                     : "+v"(fa_lo[0]), "+v"(fa_hi[0]), "+v"(fa_lo[1]), "+v"(fa_hi[1]), "+v"(fa_lo[2]), "+v"(fa_hi[2]), "+v"(fa_lo[3]),
                                       "+v"(fa_hi[3]), "+v"(fb_lo[0]), "+v"(fb_hi[0])
                                     : "n"(kAhead));
                        raws_defined();
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fb_lo[P]), "+v"(fb_hi[P]) : "n"(kAhead));
                    }
                    units_of(jc, wc);
                    const v8s bl = __builtin_bit_cast(v8s, fb_lo[P]), bh = __builtin_bit_cast(v8s, fb_hi[P]);
                    if constexpr (ABL != 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, __builtin_bit_cast(v8s, fa_lo[i]), acc[i][J], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, __builtin_bit_cast(v8s, fa_hi[i]), acc[i][J], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                read_b(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) read_b(std::integral_constant<int, 1>{});
                first_rows();
                step(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) step(std::integral_constant<int, 1>{});
                if constexpr (NTW > 2) step(std::integral_constant<int, 2>{});
                if constexpr (NTW > 3) step(std::integral_constant<int, 3>{});
                if constexpr (NTW > 4) step(std::integral_constant<int, 4>{});
                if constexpr (NTW > 5) step(std::integral_constant<int, 5>{});
                if constexpr (NTW > 6) step(std::integral_constant<int, 6>{});
                if constexpr (NTW > 7) step(std::integral_constant<int, 7>{});
            } else {
                first_rows();
                raws_defined();                                   // (only raw[0] was waited for: the unit waits below cover the rest,
                units_of(std::integral_constant<int, 0>{}, wc);   //  every raw read being older than every gather)
            }
        };

        // prologue: table in place (barrier); activations of k tile 0, raw weights of k tiles 0 and 1 requested; this wave's pieces
        // of weight tile 0 converted as soon as they are here (only the wave that requested a piece touches it before a barrier)
        __syncthreads();
        request(0, a0, 0, w0);
        if constexpr (ABL != 2) {
#pragma unroll
            for (int i = 0; i < NB; ++i) dma16((const uint8_t *)wbase[i] + (long)min(1, klast) * kRowBytes, gw[i], real[i] ? w0 + kWBytes + wofs[i] : dummy);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
        if constexpr (ABL != 2) {
            // plain and serial: read, gather, compute, write, one piece at a time
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                u32x4 v = ds_gather128((real[i] ? w0 : w0) + pofs[i]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
                uint32_t ad[8];
                row_addrs<kRowMask>(v.x, ad[0], ad[1]); row_addrs<kRowMask>(v.y, ad[2], ad[3]);
                row_addrs<kRowMask>(v.z, ad[4], ad[5]); row_addrs<kRowMask>(v.w, ad[6], ad[7]);
                u32x4 p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = ds_gather128(ad[e]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
                u32x4 q;
                q.x = quant_pair(v.x, p[0], p[1], sign_mask); q.y = quant_pair(v.y, p[2], p[3], sign_mask);
                q.z = quant_pair(v.z, p[4], p[5], sign_mask); q.w = quant_pair(v.w, p[6], p[7], sign_mask);
                const uint32_t f = (p[0].y | p[1].y | p[2].y) | (p[3].y | p[4].y | p[5].y) | (p[6].y | p[7].y);
                if constexpr (ABL == 0) flags |= real[i] ? f : 0u;
                if constexpr (ABL != 0) q = v;
                asm volatile("ds_write_b128 %0, %1" ::"v"(w0 + pofs[i]), "v"(q) : "memory");
            }
        }
        int wmul = 0;                                              // weight stage multiplied in this step; +1: converted; +2: requested
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int wcv = wmul == 2 ? 0 : wmul + 1, wrq = wcv == 2 ? 0 : wcv + 1;
            request(min(kt + 1, klast), a0 + ((kt + 1) & 1) * kABytes, min(kt + 2, klast), w0 + wrq * kWBytes);
            compute(a0 + (kt & 1) * kABytes, w0 + wmul * kWBytes, w0 + wcv * kWBytes);
            wmul = wcv;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // A lane that read a flagged row raises the workgroup's flag (through LDS: the rings are dead here).
        __syncthreads();
        volatile int *flag = (volatile int *)(lds + a0 + 8 * kEpiStride);     // past the eight waves' epilogue tiles
        if (w == 0 && l == 0) *flag = 0;
        __syncthreads();
        if ((flags & 1u) && ABL == 0) *flag = 1;
        __syncthreads();
        if (*flag) return true;

        // ---- epilogue.  Lane (r, g) of tile (i, j) holds y[row wm*64 + i*16 + r][column group j, columns 4g .. 4g+3]; every wave
        // turns its 64 x 16 NTW tile around in its own LDS (no barrier: wave-private) and stores whole rows, 16 bytes per lane.
        if constexpr (NTW > 0) {
            constexpr int kRowB = NTW * 32 + 8;                        // + 8: rows 16 apart would otherwise share banks
            const uint32_t tbase = a0 + w * kEpiStride;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int grp = tg0 + jbase + j;
                const SegRef sg = seg_lookup(a, grp);
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (sg.bias) {
                    const uint2 b = *(const uint2 *)(sg.bias + (grp * 16 + 4 * g - sg.g0 * 16));
                    bv[0] = qt_u2f(b.x << 16); bv[1] = qt_u2f(b.x & 0xFFFF0000u);
                    bv[2] = qt_u2f(b.y << 16); bv[3] = qt_u2f(b.y & 0xFFFF0000u);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x2 o = {pack_bf16x2(acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]), pack_bf16x2(acc[i][j][2] + bv[2], acc[i][j][3] + bv[3])};
                    ds_write64(tbase + (i * 16 + r) * kRowB + j * 32 + g * 8, o);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int kChunksPerRow = NTW * 2, kChunks = 64 * kChunksPerRow;
            const long col0 = (long)(tg0 + jbase) * 16;
#pragma unroll
            for (int it = 0; it < (kChunks + 63) / 64; ++it) {
                const int c = it * 64 + l, row = c / kChunksPerRow, ch = c % kChunksPerRow;
                uint2 lo_, hi_;
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(lo_), "=&v"(hi_) : "v"(tbase + row * kRowB + ch * 16) : "memory");
                const int grow = m0 + wm * 64 + row;
                if (c < kChunks && grow < a.M) *(uint4 *)(a.y + (long)grow * a.ldc + col0 + ch * 8) = uint4{lo_.x, lo_.y, hi_.x, hi_.y};
            }
        }
        return false;
    }
};


// The redo path of a tile that met a flagged row: every weight goes through the value map itself.  Plain loops, operands
// straight from global memory, one 16 x 16 output tile at a time -- only ever taken for weights outside the rows the table
// covers (non-finite values, magnitudes at the far ends of a format).
__device__ __forceinline__ uint32_t map_pair(const uint16_t *map, uint32_t x) {
    return (uint32_t)map[x & 0xFFFFu] | ((uint32_t)map[x >> 16] << 16);
}
__device__ __forceinline__ void slow_tile(const Args &a, int m0, int tg0, int jbase, int ntw, int w, int l) {
    const int r = l & 15, g = l >> 4, wm = w & 3;
    const int nk = a.K / 32;
#pragma unroll 1
    for (int j = 0; j < ntw; ++j) {
        const int grp = tg0 + jbase + j;
        const SegRef sg = seg_lookup(a, grp);
        const uint16_t *wrow = sg.w + (long)((grp - sg.g0) * 16 + r) * a.K;
        const int col = grp * 16 + 4 * g;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (sg.bias) {
            const uint2 b = *(const uint2 *)(sg.bias + (col - sg.g0 * 16));
            bv[0] = qt_u2f(b.x << 16); bv[1] = qt_u2f(b.x & 0xFFFF0000u);
            bv[2] = qt_u2f(b.y << 16); bv[3] = qt_u2f(b.y & 0xFFFF0000u);
        }
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wm * 64 + i * 16 + r;
            const uint16_t *xrow = a.x + (long)min(row, a.M - 1) * a.K;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kt = 0; kt < nk; ++kt) {
                const u32x4 xv = *(const u32x4 *)(xrow + kt * 32 + 8 * g);
                const u32x4 wv = *(const u32x4 *)(wrow + kt * 32 + 8 * g);
                const u32x4 q = {map_pair(a.map, wv.x), map_pair(a.map, wv.y), map_pair(a.map, wv.z), map_pair(a.map, wv.w)};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8s, q), __builtin_bit_cast(v8s, xv), acc, 0, 0, 0);
            }
            if (row < a.M) {
                const uint2 o = {pack_bf16x2(acc[0] + bv[0], acc[1] + bv[1]), pack_bf16x2(acc[2] + bv[2], acc[3] + bv[3])};
                *(uint2 *)(a.y + (long)row * a.ldc + col) = o;
            }
        }
    }
}

template <int NB, bool SROWS, int ABL = 0>
__global__ __launch_bounds__(512, 1) void linear_fqt_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_t[];
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    // workgroup ids go round-robin over the 8 XCDs: XCD x gets a contiguous run of tiles, column tile = id / tiles_m, so the row
    // tiles of one column tile (one weight tile) are neighbours on one XCD
    const int ntiles = a.tiles_m * a.tiles_n;
    int id = blockIdx.x;
    {
        const int per = ntiles / 8, rem = ntiles % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;
    }
    const int tn = id / a.tiles_m, tm = id % a.tiles_m;
    int tg0, nt;
    tile_span(a, tn, tg0, nt);
    const int m0 = tm * kTM;
    const int nt0 = (nt + 1) >> 1;
    const int wn = w >> 2;
    const int ntw = wn == 0 ? nt0 : nt - nt0, jbase = wn == 0 ? 0 : nt0;
    using L = LinearFqt<NB, SROWS, ABL>;
    if (lds_addr(lds_t) != 0) __builtin_trap();            // the LDS map below is written in absolute addresses
    bool redo;
#define QT_RUN(N) L::template run<N>(a, lds_t, m0, tg0, nt, jbase, w, l)
    switch (ntw) {                                          // wave-uniform
        case 0: redo = QT_RUN(0); break;
        case 1: redo = QT_RUN(1); break;
        case 2: redo = QT_RUN(2); break;
        case 3: if constexpr (NB >= 2) { redo = QT_RUN(3); break; }
        case 4: if constexpr (NB >= 2) { redo = QT_RUN(4); break; }
        case 5: if constexpr (NB >= 3) { redo = QT_RUN(5); break; }
        case 6: if constexpr (NB >= 3) { redo = QT_RUN(6); break; }
        case 7: if constexpr (NB >= 4) { redo = QT_RUN(7); break; }
        default:
            if constexpr (NB >= 4) redo = QT_RUN(8);
            else if constexpr (NB >= 3) redo = QT_RUN(6);
            else if constexpr (NB >= 2) redo = QT_RUN(4);
            else redo = QT_RUN(2);
            break;
    }
#undef QT_RUN
    if (redo) slow_tile(a, m0, tg0, jbase, ntw, w, l);
}

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <int NB, bool SROWS, int ABL = 0>
int launch_nb(const Args &a, hipStream_t st) {
    constexpr int kLds = LinearFqt<NB, SROWS>::kLds;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void *)linear_fqt_kernel<NB, SROWS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    linear_fqt_kernel<NB, SROWS, ABL><<<a.tiles_m * a.tiles_n, 512, kLds, st>>>(a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <bool SROWS>
int launch(const Args &a, hipStream_t st) {
    if (a.nb <= 1) return launch_nb<1, SROWS>(a, st);
    if (a.nb <= 2) return launch_nb<2, SROWS>(a, st);
    if (a.nb <= 3) return launch_nb<3, SROWS>(a, st);
    if constexpr (!SROWS) {
        const char *e_abl = getenv("QT_FQT_ABLATE");       // timing experiments (tools/exp_linear_fqt.py --skip-checks): results are garbage
        switch (e_abl ? atoi(e_abl) : 0) {
            case 1: return launch_nb<4, false, 1>(a, st);
            case 2: return launch_nb<4, false, 2>(a, st);
            case 3: return launch_nb<4, false, 3>(a, st);
            case 5: return launch_nb<4, false, 5>(a, st);
            case 6: return launch_nb<4, false, 6>(a, st);
            default: break;
        }
    }
    return launch_nb<4, SROWS>(a, st);
}

}  // namespace

extern "C" {

int qt_linear_fqt_bf16(const uint16_t *x_dev, const uint16_t *const *w_devs, const uint16_t *const *bias_devs, const int *ns,
                       int count, const uint32_t *rows_dev, int signed_rows, uint32_t sign_mask, const uint16_t *map_dev,
                       uint16_t *y_dev, int M, int K, void *stream) {
    if (count < 1 || count > kMaxSeg || !w_devs || !ns) return QT_ERR_BAD_ARG;
    long ntot = 0;
    for (int i = 0; i < count; ++i) {
        if (ns[i] < 0 || ns[i] % 16 != 0) return QT_ERR_BAD_ARG;
        ntot += ns[i];
    }
    if ((long)M * ntot == 0) return QT_OK;
    if (!x_dev || !y_dev || !rows_dev || !map_dev || M < 0 || K < kBK || K % kBK != 0 || ntot > (1L << 30)) return QT_ERR_BAD_ARG;
    if (((uintptr_t)x_dev & 15u) || ((uintptr_t)y_dev & 7u) || ((uintptr_t)rows_dev & 15u) || (ntot & 3)) return QT_ERR_UNALIGNED;
    for (int i = 0; i < count; ++i) {
        if (ns[i] && (!w_devs[i] || ((uintptr_t)w_devs[i] & 15u))) return QT_ERR_UNALIGNED;
        if (bias_devs && bias_devs[i] && ((uintptr_t)bias_devs[i] & 7u)) return QT_ERR_UNALIGNED;
    }
    const long groups = ntot / 16;
    Args a{};
    a.x = x_dev; a.y = y_dev; a.rows = rows_dev; a.map = map_dev; a.sign_mask = sign_mask;
    a.M = M; a.K = K; a.ldc = (int)ntot;
    a.tiles_m = (M + kTM - 1) / kTM;
    // column tiles: as many as make whole rounds over the CUs (one 512-thread workgroup per CU), no wider than max_nt groups
    const char *e_tn = getenv("QT_FQT_TILES_N"), *e_nt = getenv("QT_FQT_MAX_NT");       // tuning / A-B switches
    const int force_tn = e_tn ? atoi(e_tn) : 0;
    const int hard_nt = signed_rows ? 14 : 15;      // LDS: table + 2 activation stages + 3 weight stages of hard_nt groups
    int max_nt = e_nt ? atoi(e_nt) : hard_nt;
    if (max_nt < 1 || max_nt > hard_nt) max_nt = hard_nt;
    const int cus = cu_count();
    const long tn_min = (groups + max_nt - 1) / max_nt;
    const long rounds = (a.tiles_m * tn_min + cus - 1) / cus;
    long tn = rounds * cus / a.tiles_m;
    if (force_tn > 0) tn = force_tn;
    if (tn < tn_min) tn = tn_min;
    if (tn > groups) tn = groups;
    a.tiles_n = (int)tn;
    a.gbase = (int)(groups / tn);
    a.gextra = (int)(groups % tn);
    const int worst_nt = a.gbase + (a.gextra ? 1 : 0);
    if (worst_nt > hard_nt) return QT_ERR_BAD_ARG;
    a.nb = (worst_nt * 2 + 7) / 8;
    int nseg = 0, g0 = 0;
    for (int i = 0; i < count; ++i) {
        if (ns[i] == 0) continue;
        a.seg[nseg].w = w_devs[i];
        a.seg[nseg].bias = bias_devs ? bias_devs[i] : nullptr;
        a.seg[nseg].g0 = g0;
        g0 += ns[i] / 16;
        ++nseg;
    }
    a.nseg = nseg;
    hipStream_t st = (hipStream_t)stream;
    return signed_rows ? launch<true>(a, st) : launch<false>(a, st);
}

}  // extern "C"
