// Fake-quant Linear for ANY value map: the weight's fake-quantizer inside a bf16 matrix-core GEMM.
//
// Replaces   F.linear(input, self.weight_fake_quant(self.weight), self.bias)       modules/qat/linear.py:40-41
// for stateless weight fake-quantizers whose format is not FP8-exact (posit(n, es), intN, fp6 / fp4, ...: fake_quantize.py:31-95):
//   * the activation arrives as the bf16 VALUES of fq(x) (the elementwise pass or a producer kernel wrote them);
//   * the bf16 weight is read ONCE from HBM, unquantized, straight into LDS (global_load_lds); the wave that requested a piece
//     sends its 16 bytes per lane through the ROW FORM of the value map (below) and writes the quantized bf16 values back IN
//     PLACE, one k step before the tile is multiplied -- fq(W) never exists in HBM;
//   * v_mfma_f32_16x16x32_bf16 on the quantized operands: products of quantized values are exactly the reference's bf16
//     products, fp32 accumulation, bias and the bf16 rounding in the epilogue.
//
// Row form (qt_build_rowparams, csrc/qt_host.cpp): the 128 bf16 patterns that share an exponent are one row of the map, and
// within a row the map is a round-to-nearest-even onto a power-of-two grid plus a clamp,
//     t = f32(|x| bits + D);   y = med3((t + C) - C, lo, hi);   sign of x copied back            D, C, lo, hi: 16 bytes per row
// i.e. 8 vector instructions and one 16-byte LDS read (256 rows = 4 KiB, mostly broadcast: the weights of a tile live in a
// dozen exponents) per weight instead of a gather from the 128 KiB map, which would not fit beside the operand rings.  The
// builder verifies each row against the map on all of its inputs; rows that do not fit are flagged (bit 0 of C), lanes OR the
// C words they read, and a workgroup that met a flagged row redoes its tile with the map itself (slow_tile) -- bit-exact for
// every bf16 weight.  Flagged in practice: non-finite inputs and a few rows beyond 2^22 / below 2^-38 of some posits.
//
// Work decomposition.  A workgroup owns TM rows x (16 nt) columns, TM = 512 (nt <= 8) where that fills the chip, else 256
// (nt <= 16): every weight is converted once per ROW TILE, so tall tiles halve the conversion work, and with k steps of 32 a
// stage is small (activations TM x 64 bytes, weights nt KiB), which buys rings deep enough to cover the memory latency -- three
// activation stages, five weight stages (multiplied | converted | three landing).  8 waves = 4 row bands x 2 column halves; the
// host picks the column widths so that the grid is a whole number of rounds over the CUs; up to four weights sharing one
// activation are the segments of one launch; tile ids are dealt so that the row tiles of a column tile share an XCD.
// LDS: row table 4 / 8 KiB | activation ring | weight ring | 1 KiB dummy; rows of 64 bytes (32 bf16), 16-byte chunk p of row R
// holds k chunk p ^ pi((R >> 2) & 3) so that a fragment's sixteen rows x one chunk spread over all 64 banks.  One raw barrier per
// k step; every global access in the loop is an LDS-DMA and every wait a counted one.
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
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBK = 32, kRowBytes = 64, kMaxSeg = 4;
constexpr int kGroupBytes = 16 * kRowBytes;         // one 16-row group of a tile = one DMA piece: 1 KiB

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
    int prio;                // static issue priority of waves 4-7 (QT_FQT_PRIO)
    unsigned long long *dbg; // ABL 9 (QT_FQT_STAMPS = device address): s_memtime stamps of workgroup 0, waves 0 and 4, k step 40
    int M, K, ldc;
    int tiles_m, tiles_n, nseg;
    int gbase, gextra;        // column tile j covers gbase (+1 for gextra of them) groups of 16 columns
    // split-K (narrow outputs: too few tiles to fill the chip): `ksplit` workgroups share a tile, each multiplies a contiguous range of
    // k steps and writes its fp32 partial sums to `ws`; the LAST one to arrive at the tile's ticket adds them in split order and writes y
    int ksplit;
    float *ws;                // [tile][split][wave][kRT x kMaxNTW fragments][64 lanes][4] fp32
    unsigned int *tickets;    // [tile] arrival counters, zero between launches (the last arriver resets its tile's)
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

// Split-K hand-off between workgroups: every byte is stored write-through (sc1) and loaded past the L1 (sc1); the stores are drained
// (s_waitcnt vmcnt(0)) in front of the workgroup barrier that precedes the ticket's atomic add.
// Addresses are a wave-uniform base (SGPR pair) + a 32-bit lane offset: no address registers per access.
__device__ __forceinline__ void store16_sc1(const float *sbase, uint32_t voff, v4f v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ v4f load16_sc1(const float *sbase, uint32_t voff) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}

// Both operand tiles: row-major 64-byte rows (32 bf16 = four 16-byte k chunks).  A fragment read is sixteen rows x one chunk per
// 16-lane group of ds_read_b128; rows four apart share a bank row, so chunk c of row R sits at position c ^ pi((R >> 2) & 3),
// pi = (0, 2, 3, 1): the four (row quad, chunk) combinations a lane group reads land on four different positions.
__device__ __forceinline__ int chunk_pos(int row, int chunk) { return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }
__device__ __forceinline__ int chunk_off(int row, int chunk) { return row * kRowBytes + (chunk_pos(row, chunk) << 4); }

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
// possible).  A "unit" is half a weight piece of the lane: 4 weights = 2 packed words = 4 table rows.  RT = row tiles of a wave.
//     FA x RT, RB(0), RB(1)     activation fragments of the step, weight fragments of column groups 0 and 1
//     RAW x NB                  the lane's 16 raw bytes of each of its pieces (weight tile t+1)
//     per column group J:  RB(J+2);  [wait FA / RB(J)];  the group's RT multiplications;  in group 0: [wait RAW] G(0) x 4, the table
//                          rows of unit 0;  then per unit u whose slot the group carries:
//                          G(u+1) x 4;  [wait G(u)];  (arithmetic of unit u, under the multiplications);  WR(u) x 1
// walk() replays that order and returns how many operations were issued after the last one of (from_kind, from_idx) when the
// wait (to_kind, to_idx) is reached: the lgkmcnt that wait may leave outstanding.
enum { kOpRaw = 0, kOpFA = 1, kOpRB = 2, kOpG = 3, kOpWR = 4, kWaitRaw = 10, kWaitF = 11, kWaitG = 12 };
constexpr int unit_group(int u, int NTW, int U) { return NTW > 0 ? u * NTW / U : 0; }
// the column group whose multiplications cover unit u's arithmetic: one behind the group that requested its rows
constexpr int slot_group(int u, int NTW, int U) { return NTW > 0 ? (unit_group(u, NTW, U) + 1 < NTW ? unit_group(u, NTW, U) + 1 : NTW - 1) : 0; }
constexpr int walk(int RT, int NTW, int NB, int from_kind, int from_idx, int to_kind, int to_idx) {
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
    if (NTW > 0) {
        op(kOpFA, 0, RT);
        op(kOpRB, 0, 1);
        if (NTW > 1) op(kOpRB, 1, 1);
    }
    for (int i = 0; i < NB; ++i) op(kOpRaw, i, 1);
    for (int J = 0; J < (NTW > 0 ? NTW : 1); ++J) {
        if (NTW > 0) {
            if (J + 2 < NTW) op(kOpRB, J + 2, 1);
            wait(kWaitF, J);
        }
        if (J == 0) {
            wait(kWaitRaw, 0);
            op(kOpG, 0, 4);
        }
        for (int u = 0; u < U; ++u) {
            if (slot_group(u, NTW, U) != J) continue;
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
// TM: rows of a workgroup's tile (512 or 256); NB: weight pieces (16 rows x 64 bytes = one column group) per wave and k step
// (1: tiles of up to 8 groups, 2: up to 16); SROWS: the table has 512 rows (sign and exponent).
//
// k step t of a wave (every LDS / DMA operation is inline asm, every wait a computed count):
//   top      s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier      the activation tile of step t and this wave's raw pieces of weight tile
//                                                          t+1 have landed (requested two / three steps ago), every wave's converted
//                                                          weights of step t are written, every wave is done with step t-1
//   first    request activation tile t+2 and RAW weight tile t+4
//   then     the LDS stream above: RT multiplications per column group on tile t, and 2 NB units of weight tile t+1 converted IN
//            PLACE, spread evenly over the groups; the rows of unit u+1 are in flight while unit u is computed
template <int TM, int NB, bool SROWS, int ABL = 0>
struct LinearFqt {
    static constexpr int kDA = 3, kDW = 5, kU = 2 * NB;
    static constexpr int kRT = TM / 64;                     // 16-row tiles of a wave's row band (TM / 4 rows)
    static constexpr int kPA = TM / 128;                    // activation DMA pieces (16 rows) per wave and step
    static constexpr int kMaxNT = 8 * NB, kMaxNTW = 4 * NB;
    static constexpr int kABytes = TM * kRowBytes, kWBytes = kMaxNT * kGroupBytes;
    static constexpr int kTbl = SROWS ? 8192 : 4096;
    static constexpr int kDummy = 1024;                     // where surplus pieces go
    static constexpr int kRings = kDA * kABytes + kDW * kWBytes;
    static constexpr int kEpiStride = 64 * (kMaxNTW * 32 + 8);           // a wave's epilogue tile (64 rows at a time) in LDS
    static constexpr int kLds = kTbl + (kRings + kDummy > 8 * kEpiStride ? kRings + kDummy : 8 * kEpiStride);
    static_assert(kLds <= 160 * 1024, "LDS budget");
    static_assert(kRT * kMaxNTW * 4 <= 128, "accumulators");
    static constexpr uint32_t kRowMask = SROWS ? 0x1FF0u : 0xFF0u;
    // What the top-of-step wait may leave in flight.  A step issues kPA activation pieces, then NB weight pieces; the activation
    // tile of step t was requested kDA - 1 steps earlier (younger: that step's weight pieces and kDA - 2 whole steps), the raw
    // weight tile t+1 kDW - 2 steps earlier (younger: kDW - 3 whole steps).
    static constexpr int kVmA = NB + (kDA - 2) * (kPA + NB), kVmW = (kDW - 3) * (kPA + NB);
    static constexpr int kVmTop = kVmA < kVmW ? kVmA : kVmW;
    static constexpr int kWeave = 5;                         // vector instructions the scheduler is asked to put behind each multiplication

    // One wave's share: rows [wm * TM / 4, + TM / 4) x NTW column groups starting at group jbase of the tile whose first group is
    // tg0.  Returns true when a flagged row was met (the caller redoes the tile).
    template <int NTW, int PH = 0>
    // kbeg / nk: this workgroup's range of k steps (all of K / 32 without split-K); split / tile_lin: its place in the hand-off.
    static __device__ __forceinline__ bool run(const Args &a, uint8_t *lds, int m0, int tg0, int nt, int jbase, int w, int l, int kbeg, int nk,
                                               int split, int tile_lin) {
        const int r = l & 15, g = l >> 4, wm = w & 3;
        const int klast = nk - 1;                               // k steps are counted from kbeg inside this function
        const long krow = (long)a.K * 2;                        // bytes per row of x and W
        // LDS map (the dynamic segment starts at address 0: the kernel has no static LDS)
        constexpr uint32_t a0 = kTbl, w0 = a0 + kDA * kABytes, dummy = w0 + kDW * kWBytes;
        // ---- the row table into LDS (every thread one 16-byte row)
        {
            const int t = w * 64 + l;
            if (t < (SROWS ? 512 : 256)) {
                u32x4 v = *(const u32x4 *)(a.rows + t * 4);
                // a flagged row (bit 0 of C) POISONS: C = lo = hi = NaN make every weight of the row a NaN, which reaches every sum it
                // enters; the sums are checked once after the k loop (no per-weight flag arithmetic in the loop)
                if (ABL == 0 && (v.y & 1u)) v = u32x4{0u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u};
                asm volatile("ds_write_b128 %0, %1" ::"v"(t * 16), "v"(v) : "memory");
            }
        }
        // ---- DMA sources: a scalar base (advanced by 64 bytes per k tile) + a 32-bit lane offset; LDS destinations are
        // wave-uniform.  A piece is 16 rows x 64 bytes: lane = (row l >> 2, position l & 3).
        uint32_t ga[kPA];
#pragma unroll
        for (int i = 0; i < kPA; ++i) {
            const int row = (w * kPA + i) * 16 + (l >> 2);
            ga[i] = (uint32_t)((long)min(m0 + row, a.M - 1) * krow) + (chunk_pos(row, l & 3) << 4);
            if constexpr (ABL == 7) {        // timing probe: whole 128-byte lines (8 rows per piece) instead of half lines
                const int row8 = (w * kPA + i) * 8 + (l >> 3);
                ga[i] = (uint32_t)((long)min(m0 + row8, a.M - 1) * krow) + ((l & 7) << 4);
            }
        }
        // Weight pieces: piece p = w + 8 i is column group p of the tile; the lane's 16 bytes land at piece base + 16 l and are
        // converted there.  Surplus pieces (p >= nt) re-request piece 0 into the dummy kilobyte.
        uint32_t gw[NB];
        const uint16_t *wbase[NB];                             // wave-uniform
        uint32_t wofs[NB];
        bool real[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = w + 8 * i;
            real[i] = p < nt;
            const int pb = real[i] ? p : 0;
            const int grp = tg0 + pb;
            const int row = pb * 16 + (l >> 2);
            const SegRef sg = seg_lookup(a, grp);
            wbase[i] = sg.w;
            gw[i] = (uint32_t)((long)((grp - sg.g0) * 16 + (l >> 2)) * krow) + (chunk_pos(row, l & 3) << 4);
            wofs[i] = (uint32_t)pb * 1024u;
        }
        const uint32_t sign_mask = a.sign_mask;
        int stamp_kt = -1;
        auto stamp = [&](int slot) __attribute__((always_inline)) {
            if constexpr (ABL == 9 || ABL == 8) {
                if (ABL == 8 && !(slot == 29 || slot == 31 || slot == 0 || slot == 30)) return;     // light: four stamps per step
                if (stamp_kt >= 40 && stamp_kt < 44 && blockIdx.x == 0 && l == 0) a.dbg[(w * 4 + (stamp_kt - 40)) * 32 + slot] = __builtin_amdgcn_s_memtime();
            }
        };
        auto dma16 = [](const void *base, uint32_t off, uint32_t dst) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
        };
        auto ds_write64 = [](uint32_t addr, u32x2 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        // a step's requests: activation tile ka into stage `as`, raw weight tile kb into stage `ws` -- kPA + NB DMA instructions,
        // always in this order (the counted waits rely on it)
        auto req_piece = [&](auto ic, int ka, uint32_t as, int kb, uint32_t ws) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < kPA) {
                const uint8_t *xb = (const uint8_t *)a.x + (ABL == 7 ? (long)((ka + kbeg) >> 1) * 128 : (long)(ka + kbeg) * kRowBytes);
                if constexpr (ABL != 3) dma16(xb, ga[I], as + (w * kPA + I) * 1024);
                else dma16(xb, ga[I], dummy);
            } else {
                constexpr int i = I - kPA;
                dma16((const uint8_t *)wbase[i] + (long)(kb + kbeg) * kRowBytes, gw[i], (real[i] && ABL != 2) ? ws + wofs[i] : dummy);
            }
        };
        auto req_range = [&](auto lo, auto hi, int ka, uint32_t as, int kb, uint32_t ws) __attribute__((always_inline)) {
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            static_assert(HI - LO <= 6, "at most six pieces");
            if constexpr (LO + 0 < HI) req_piece(std::integral_constant<int, LO + 0>{}, ka, as, kb, ws);
            if constexpr (LO + 1 < HI) req_piece(std::integral_constant<int, LO + 1>{}, ka, as, kb, ws);
            if constexpr (LO + 2 < HI) req_piece(std::integral_constant<int, LO + 2>{}, ka, as, kb, ws);
            if constexpr (LO + 3 < HI) req_piece(std::integral_constant<int, LO + 3>{}, ka, as, kb, ws);
            if constexpr (LO + 4 < HI) req_piece(std::integral_constant<int, LO + 4>{}, ka, as, kb, ws);
            if constexpr (LO + 5 < HI) req_piece(std::integral_constant<int, LO + 5>{}, ka, as, kb, ws);
        };
        constexpr int kReq = kPA + NB;
        auto request = [&](int ka, uint32_t as, int kb, uint32_t ws) __attribute__((always_inline)) {
            req_range(std::integral_constant<int, 0>{}, std::integral_constant<int, kReq>{}, ka, as, kb, ws);
        };
        // the lane's 16 bytes of piece i, relative to a weight stage (surplus pieces: the dummy, whatever the stage)
        uint32_t pofs[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) pofs[i] = wofs[i] + (uint32_t)l * 16u;
        auto piece_addr = [&](int i, uint32_t ws) __attribute__((always_inline)) -> uint32_t {
            return real[i] ? ws + pofs[i] : dummy + (uint32_t)l * 16u;
        };
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
                }
                ds_write64(piece_addr(I, wc) + (U & 1) * 8, q);
            }
        };
        // unit slot: the next unit's rows requested, this unit's rows waited for, computed, written back
        auto unit_slot = [&](auto uc, uint32_t wc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            if constexpr (U + 1 < kU) unit_gather(std::integral_constant<int, U + 1>{});
            if constexpr (ABL != 2 && ABL != 5 && ABL != 6) {
                constexpr int n = walk(kRT, NTW, NB, kOpG, U, kWaitG, U);
                static_assert(n >= 0, "schedule");
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(rows[U & 1][0]), "+v"(rows[U & 1][1]), "+v"(rows[U & 1][2]), "+v"(rows[U & 1][3]) : "n"(n));
            }
            unit_finish(uc, wc);
        };
        // the units column group J carries
        auto units_of = [&](auto jc, uint32_t wc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;      // (+ 0 * J below: the calls must depend on J, or every one is instantiated)
            if constexpr (0 < kU && slot_group(0, NTW, kU) == J) unit_slot(std::integral_constant<int, 0 + 0 * J>{}, wc);
            if constexpr (1 < kU && slot_group(1, NTW, kU) == J) unit_slot(std::integral_constant<int, 1 + 0 * J>{}, wc);
            if constexpr (2 < kU && slot_group(2, NTW, kU) == J) unit_slot(std::integral_constant<int, 2 + 0 * J>{}, wc);
            if constexpr (3 < kU && slot_group(3, NTW, kU) == J) unit_slot(std::integral_constant<int, 3 + 0 * J>{}, wc);
        };
        static_assert(kU <= 4, "units_of lists four units");
        // the step's first LDS operations: raw pieces of the tile to convert (stage wc)
        auto read_raw = [&](uint32_t wc) __attribute__((always_inline)) {
            if constexpr (ABL != 2) {
#pragma unroll
                for (int i = 0; i < NB; ++i) raw[i] = ds_gather128(piece_addr(i, wc));
            }
        };
        // wait for raw[0], then the first unit's rows
        auto first_rows = [&]() __attribute__((always_inline)) {
            if constexpr (ABL != 2) {
                constexpr int n = walk(kRT, NTW, NB, kOpRaw, NB - 1, kWaitRaw, 0);
                static_assert(n >= 0, "schedule");
                if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(raw[0]), "+v"(raw[1]) : "n"(n));
                else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(raw[0]) : "n"(n));
                unit_gather(std::integral_constant<int, 0>{});
            }
        };
        static_assert(NB <= 2, "first_rows lists two pieces");

        v16f acc32[4][2];                                         // ABL 10 only
        if constexpr (ABL == 10) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
        }
        v4f acc[kRT][NTW > 0 ? NTW : 1];
#pragma unroll
        for (int i = 0; i < kRT; ++i)
#pragma unroll
            for (int j = 0; j < (NTW > 0 ? NTW : 1); ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // lane-constant parts of the fragment addresses (row tiles and column groups are 16 rows = 1 KiB apart)
        const uint32_t a_frag = chunk_off(wm * (TM / 4) + r, g), b_frag = chunk_off(jbase * 16 + r, g);

        // multiplications of step t on (sa_, sb_); conversion of the raw weight tile in stage wc
        auto compute = [&](uint32_t sa_, uint32_t sb_, uint32_t wc, int ka, uint32_t as, int kb, uint32_t ws) __attribute__((always_inline)) {
            stamp(1);
            if constexpr (NTW > 0) {
                u32x4 fa[kRT], fb[3];
                fa[0] = ds_read128<0 * 1024>(sa_ + a_frag);
                fa[1] = ds_read128<1 * 1024>(sa_ + a_frag);
                fa[2] = ds_read128<2 * 1024>(sa_ + a_frag);
                fa[3] = ds_read128<3 * 1024>(sa_ + a_frag);
                if constexpr (kRT > 4) {
                    fa[4] = ds_read128<4 * 1024>(sa_ + a_frag);
                    fa[5] = ds_read128<5 * 1024>(sa_ + a_frag);
                    fa[6] = ds_read128<6 * 1024>(sa_ + a_frag);
                    fa[7] = ds_read128<7 * 1024>(sa_ + a_frag);
                }
                auto read_b = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    fb[J % 3] = ds_read128<J * 1024>(sb_ + b_frag);
                };
                auto step = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int P = J % 3;
                    __builtin_amdgcn_sched_barrier(0);
                    // the younger wave of a SIMD (column half 1) loses the issue arbitration to the older one and would set the pace of
                    // the step; it runs the first part of the step at priority 1 and the rest at 0, so that both finish together
                    if constexpr (PH == 1) {
                        if (a.prio == 4) {
                            if constexpr (J == 0) __builtin_amdgcn_s_setprio(1);
                            if constexpr (J == (NTW + 1) / 2) __builtin_amdgcn_s_setprio(0);
                        }
                    }
                    if constexpr (J + 2 < NTW) read_b(std::integral_constant<int, J + 2>{});
                    constexpr int kAhead = walk(kRT, NTW, NB, kOpRB, J, kWaitF, J);
                    static_assert(kAhead >= 0, "schedule");
                    if constexpr (J == 0) {
                        if constexpr (kRT > 4) {
                            asm volatile("s_waitcnt lgkmcnt(%9)"
                                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]), "+v"(fb[0])
                                         : "n"(kAhead));
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]) : "n"(kAhead));
                        }
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[P]) : "n"(kAhead));
                    }
                    stamp(4 + 3 * J);
                    // operands swapped: D rows = W rows (output columns), D columns = x rows -- a lane ends up with four
                    // consecutive output columns of one row.  The multiplications come first in program order; the vector work
                    // of the group (a unit's arithmetic, the next unit's row addresses) is woven between them below.
                    const v8s bf = __builtin_bit_cast(v8s, fb[P]);
                    if constexpr (ABL == 10) {
                        // timing probe (results are garbage): the same flops as v_mfma_f32_32x32x16_bf16 -- half the instructions, each
                        // holding the matrix pipe twice as long: per PAIR of column groups 4 row tiles x 2 k halves
                        if constexpr ((J & 1) == 0 && kRT == 8) {
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                acc32[i >> 1][J >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, __builtin_bit_cast(v8s, fa[i]), acc32[i >> 1][J >> 1], 0, 0, 0);
                        }
                    } else if constexpr (ABL != 1) {
#pragma unroll
                        for (int i = 0; i < kRT; ++i)
                            acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, __builtin_bit_cast(v8s, fa[i]), acc[i][J], 0, 0, 0);
                    }
                    stamp(5 + 3 * J);
                    if constexpr (J == 0) first_rows();
                    units_of(jc, wc);
                    // this group's share of the step's requests
                    req_range(std::integral_constant<int, J * kReq / NTW>{}, std::integral_constant<int, (J + 1) * kReq / NTW>{}, ka, as, kb, ws);
                    if constexpr (ABL != 1 && kWeave > 0) {
#pragma unroll
                        for (int i = 0; i < kRT; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one multiplication
                            __builtin_amdgcn_sched_group_barrier(0x002, kWeave, 0);     // kWeave vector instructions
                        }
                    }
                    stamp(6 + 3 * J);
                    __builtin_amdgcn_sched_barrier(0);
                };
                read_b(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) read_b(std::integral_constant<int, 1>{});
                read_raw(wc);
                stamp(2);
                step(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) step(std::integral_constant<int, 1>{});
                if constexpr (NTW > 2) step(std::integral_constant<int, 2>{});
                if constexpr (NTW > 3) step(std::integral_constant<int, 3>{});
                if constexpr (NTW > 4) step(std::integral_constant<int, 4>{});
                if constexpr (NTW > 5) step(std::integral_constant<int, 5>{});
                if constexpr (NTW > 6) step(std::integral_constant<int, 6>{});
                if constexpr (NTW > 7) step(std::integral_constant<int, 7>{});
            } else {
                request(ka, as, kb, ws);
                read_raw(wc);
                first_rows();
                units_of(std::integral_constant<int, 0>{}, wc);
            }
        };

        // prologue: table in place (barrier); activation tiles 0 .. kDA-2 and raw weight tiles 0 .. kDW-2 requested and landed; this
        // wave's pieces of weight tile 0 converted (only the wave that requested a piece touches it before a barrier)
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kDW - 1; ++s)
            request(min(s < kDA - 1 ? s : kDA - 2, klast), a0 + (s < kDA - 1 ? s : kDA - 2) * kABytes, min(s, klast), w0 + s * kWBytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (ABL != 2) {
            // plain and serial: read, gather, compute, write, one piece at a time
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                u32x4 v = ds_gather128(piece_addr(i, w0));
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
                if constexpr (ABL != 0 && ABL != 9 && ABL != 8) q = v;
                asm volatile("ds_write_b128 %0, %1" ::"v"(piece_addr(i, w0)), "v"(q) : "memory");
            }
        }
        int sa = 0, sw = 0;                                        // stages multiplied in this step (activations, weights)
        {
            for (int kt = 0; kt < nk; ++kt) {
                stamp_kt = kt; stamp(29);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kVmTop) : "memory");
                stamp(31);
                __builtin_amdgcn_s_barrier();
                stamp_kt = kt;
                stamp(0);
                // the stages requested now are the ones multiplied in step t - 1
                const int sa_req = sa == 0 ? kDA - 1 : sa - 1, sw_req = sw == 0 ? kDW - 1 : sw - 1, sw_cv = sw == kDW - 1 ? 0 : sw + 1;
                compute(a0 + sa * kABytes, w0 + sw * kWBytes, w0 + sw_cv * kWBytes, min(kt + kDA - 1, klast), a0 + sa_req * kABytes,
                        min(kt + kDW - 1, klast), w0 + sw_req * kWBytes);
                stamp(30);
                sa = sa == kDA - 1 ? 0 : sa + 1;
                sw = sw_cv;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // A lane that read a flagged row raises the workgroup's flag (through LDS: the table is dead here).
        __syncthreads();
        volatile int *flag = (volatile int *)lds;
        if (w == 0 && l == 0) *flag = 0;
        __syncthreads();
        {
            // a NaN among the sums: a weight of a flagged row (or a non-finite operand the reference would turn into NaN as well) --
            // the tile is redone through the map itself, which gives the exact answer in either case
            bool bad = false;
            if constexpr (NTW > 0 && ABL == 0) {
#pragma unroll
                for (int i = 0; i < kRT; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) bad |= (acc[i][j][0] != acc[i][j][0]) | (acc[i][j][1] != acc[i][j][1]) | (acc[i][j][2] != acc[i][j][2]) | (acc[i][j][3] != acc[i][j][3]);
            }
            if (bad) *flag = 1;
        }
        __syncthreads();
        const bool flagged = *flag != 0;
        if (a.ksplit <= 1) {
            if (flagged) return true;
        if constexpr (ABL == 10 && NTW > 0) {
    #pragma unroll
                for (int i = 0; i < 4; ++i)
    #pragma unroll
                    for (int j = 0; j < 2; ++j)
    #pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][0][e & 3] += acc32[i][j][e];
            }
        } else {
            // ---- split-K hand-off.  Partial sums leave in fragment order (a wave's store instruction writes one contiguous KiB),
            // write-through; every wave drains its stores; behind the workgroup barrier ONE lane takes the tile's ticket.  Whoever
            // draws the last ticket owns the tile: it adds the partial sums IN SPLIT ORDER (so the result does not depend on who
            // arrived last) and runs the ordinary epilogue.  Nobody ever waits for another workgroup.  A
            // flagged split says so in the ticket's upper half and the owner redoes the whole tile through the map.
            constexpr long kSlab = (long)kRT * kMaxNTW * 256;                    // floats per wave
            float *const mine = a.ws + (((long)tile_lin * a.ksplit + split) * 8 + w) * kSlab;
            if constexpr (NTW > 0) {
                if (!flagged) {
#pragma unroll
                    for (int i = 0; i < kRT; ++i)
#pragma unroll
                        for (int j = 0; j < NTW; ++j) store16_sc1(mine + (i * kMaxNTW + j) * 256, (uint32_t)l * 16u, acc[i][j]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // (RELAXED on purpose: the ordering is the sc1 stores + s_waitcnt + barrier in front and the sc1 loads behind -- valid on gfx942 /
            // gfx950, which qt_device.h asserts at compile time; an agent-scope acquire / release here writes back and invalidates the
            // whole L2, behind 20 MB of partial sums)
            if (w == 0 && l == 0) {
                const unsigned int old = __hip_atomic_fetch_add(a.tickets + tile_lin, 1u + (flagged ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = (int)old;
            }
            __syncthreads();
            const unsigned int old = (unsigned int)*flag;
            if ((int)(old & 0xFFFFu) != a.ksplit - 1) return false;               // not the last one: done
            if (w == 0 && l == 0) __hip_atomic_store(a.tickets + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old >> 16) != 0u || flagged) return true;                        // some split met a flagged row: redo the tile
            // reader's side: whatever this XCD's L2 still holds of the workspace from an earlier launch is dropped (acquire = invalidate,
            // nothing is written back; only the owner of a tile pays it)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if constexpr (NTW > 0) {
                // Every split's partial sums come back from the workspace, this workgroup's own included (no instruction depends
                // on which split arrived last): t = p_0 + p_1 + ... in split order, + bias, rounded to bf16, stored straight from
                // the fragment layout (8 bytes per lane: the owner's stores are a small part of the launch).  A rolled loop over
                // batches of four fragments with nothing but the batch live: the unrolled form over the whole accumulator array made
                // hipcc keep two copies of it and spill; S x 8 (S = 4: x 4) loads of 16 bytes in flight per lane.
                const float *const base = a.ws + ((long)tile_lin * a.ksplit * 8 + w) * kSlab;
                const uint32_t voff = (uint32_t)l * 16u;
                const int row0 = m0 + wm * (TM / 4) + r;
                auto reduce = [&](auto sc) __attribute__((always_inline)) {
                    constexpr int S = decltype(sc)::value, kB = 4, kFrag = kRT * NTW;
#pragma unroll 1
                    for (int f0 = 0; f0 < kFrag; f0 += kB) {
                        v4f part[S][kB];
#pragma unroll
                        for (int q = 0; q < S; ++q)
#pragma unroll
                            for (int b = 0; b < kB; ++b) {
                                const int f = f0 + b < kFrag ? f0 + b : kFrag - 1;      // (a ragged last batch re-reads the last fragment)
                                part[q][b] = load16_sc1(base + (long)q * 8 * kSlab + ((f / NTW) * kMaxNTW + f % NTW) * 256, voff);
                            }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int q = 0; q < S; ++q)
#pragma unroll
                            for (int b = 0; b < kB; ++b) asm volatile("" : "+v"(part[q][b]));
#pragma unroll
                        for (int b = 0; b < kB; ++b) {
                            const int f = f0 + b;
                            if (f < kFrag) {
                                v4f t = part[0][b] + part[1][b];
                                if constexpr (S > 2) t += part[2][b];
                                if constexpr (S > 3) t += part[3][b];
                                const int i = f / NTW, j = f % NTW;
                                const int grp = tg0 + jbase + j, grow = row0 + i * 16;
                                const SegRef sg = seg_lookup(a, grp);
                                if (sg.bias) {
                                    const uint2 bb = *(const uint2 *)(sg.bias + (grp * 16 + 4 * g - sg.g0 * 16));
                                    t[0] += qt_u2f(bb.x << 16); t[1] += qt_u2f(bb.x & 0xFFFF0000u);
                                    t[2] += qt_u2f(bb.y << 16); t[3] += qt_u2f(bb.y & 0xFFFF0000u);
                                }
                                if (grow < a.M) *(uint2 *)(a.y + (long)grow * a.ldc + grp * 16 + 4 * g) = uint2{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
                            }
                        }
                    }
                };
                if (a.ksplit == 2) reduce(std::integral_constant<int, 2>{});
                else if (a.ksplit == 3) reduce(std::integral_constant<int, 3>{});
                else reduce(std::integral_constant<int, 4>{});
            }
            return false;
        }
        // ---- epilogue.  Lane (r, g) of tile (i, j) holds y[row wm*TM/4 + i*16 + r][column group j, columns 4g .. 4g+3]; every wave
        // turns its tile around in its own LDS, 64 rows at a time (no barrier: wave-private), and stores whole rows, 16 bytes per lane.
        if constexpr (NTW > 0) {
            constexpr int kRowB = NTW * 32 + 8;                        // + 8: rows 16 apart would otherwise share banks
            const uint32_t tbase = a0 + w * kEpiStride;
            float bv[NTW][4];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int grp = tg0 + jbase + j;
                const SegRef sg = seg_lookup(a, grp);
                bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
                if (sg.bias) {
                    const uint2 b = *(const uint2 *)(sg.bias + (grp * 16 + 4 * g - sg.g0 * 16));
                    bv[j][0] = qt_u2f(b.x << 16); bv[j][1] = qt_u2f(b.x & 0xFFFF0000u);
                    bv[j][2] = qt_u2f(b.y << 16); bv[j][3] = qt_u2f(b.y & 0xFFFF0000u);
                }
            }
            constexpr int kChunksPerRow = NTW * 2, kChunks = 64 * kChunksPerRow;
            const long col0 = (long)(tg0 + jbase) * 16;
#pragma unroll
            for (int h = 0; h < kRT / 4; ++h) {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = h * 4 + ii;
                        const u32x2 o = {pack_bf16x2(acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1]),
                                         pack_bf16x2(acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3])};
                        ds_write64(tbase + (ii * 16 + r) * kRowB + j * 32 + g * 8, o);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < (kChunks + 63) / 64; ++it) {
                    const int c = it * 64 + l, row = c / kChunksPerRow, ch = c % kChunksPerRow;
                    uint2 lo_, hi_;
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(lo_), "=&v"(hi_) : "v"(tbase + row * kRowB + ch * 16) : "memory");
                    const int grow = m0 + wm * (TM / 4) + h * 64 + row;
                    if (c < kChunks && grow < a.M) *(uint4 *)(a.y + (long)grow * a.ldc + col0 + ch * 8) = uint4{lo_.x, lo_.y, hi_.x, hi_.y};
                }
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
template <int TM>
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
        for (int i = 0; i < TM / 64; ++i) {
            const int row = m0 + wm * (TM / 4) + i * 16 + r;
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

template <int TM, int NB, bool SROWS, int ABL = 0>
__global__ __launch_bounds__(512, 1) void linear_fqt_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_t[];
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    // workgroup ids go round-robin over the 8 XCDs: XCD x gets a contiguous run of tiles, column tile = id / tiles_m, so the row
    // tiles of one column tile (one weight tile) are neighbours on one XCD
    const int ntiles = a.tiles_m * a.tiles_n;
    int id = blockIdx.x;
    {
        const int total = ntiles * a.ksplit;
        const int per = total / 8, rem = total % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;
    }
    // with split-K the splits of one k range are neighbours (they walk the same columns of x and W): id = split * ntiles + tile
    const int split = id / ntiles;
    id -= split * ntiles;
    const int tile_lin = id;
    const int nk_all = a.K / kBK;
    const int kbeg = (int)((long)split * nk_all / a.ksplit), nk = (int)((long)(split + 1) * nk_all / a.ksplit) - kbeg;
    const int tn = id / a.tiles_m, tm = id % a.tiles_m;
    int tg0, nt;
    tile_span(a, tn, tg0, nt);
    const int m0 = tm * TM;
    const int nt0 = (nt + 1) >> 1;
    const int wn = w >> 2;
    const int ntw = wn == 0 ? nt0 : nt - nt0, jbase = wn == 0 ? 0 : nt0;
    using L = LinearFqt<TM, NB, SROWS, ABL>;
    if (lds_addr(lds_t) != 0) __builtin_trap();            // the LDS map is written in absolute addresses
    // The two waves of a SIMD are w and w + 4; the SIMD's issue arbitration prefers the older one (waves 0-3), and the younger half then
    // sets the pace of every k step (measured, tools/exp_fqt_stamps.py: 1750 against 1230 cycles of work per step, the older half
    // waiting 600 cycles at the barrier).  A static priority for the younger half evens that out (MI355X_MICROARCH.md, "Two waves per
    // SIMD", item 4).  a.prio (QT_FQT_PRIO): 0 none, 1-3 static priority of the younger half, 4 (default) the younger half at priority
    // 1 for the first half of every step only (set inside the step; 183 against 190 us at 1024 x 13824 x 5120).
    if (a.prio == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    if (a.prio == 2 && w >= 4) __builtin_amdgcn_s_setprio(2);
    if (a.prio == 3 && w >= 4) __builtin_amdgcn_s_setprio(3);
    if (a.prio == -1 && w < 4) __builtin_amdgcn_s_setprio(1);
    bool redo;
#define QT_RUN(N) (wn == 1 ? L::template run<N, 1>(a, lds_t, m0, tg0, nt, jbase, w, l, kbeg, nk, split, tile_lin) \
                           : L::template run<N, 0>(a, lds_t, m0, tg0, nt, jbase, w, l, kbeg, nk, split, tile_lin))
    switch (ntw) {                                          // wave-uniform
        case 0: redo = QT_RUN(0); break;
        case 1: redo = QT_RUN(1); break;
        case 2: redo = QT_RUN(2); break;
        case 3: redo = QT_RUN(3); break;
        case 4: redo = QT_RUN(4); break;
        case 5: if constexpr (NB >= 2) { redo = QT_RUN(5); break; }
        case 6: if constexpr (NB >= 2) { redo = QT_RUN(6); break; }
        case 7: if constexpr (NB >= 2) { redo = QT_RUN(7); break; }
        default:
            if constexpr (NB >= 2) redo = QT_RUN(8);
            else redo = QT_RUN(4);
            break;
    }
#undef QT_RUN
    if (redo) slow_tile<TM>(a, m0, tg0, jbase, ntw, w, l);
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

template <int TM, int NB, bool SROWS, int ABL = 0>
int launch_one(const Args &a, hipStream_t st) {
    constexpr int kLds = LinearFqt<TM, NB, SROWS>::kLds;
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)linear_fqt_kernel<TM, NB, SROWS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    linear_fqt_kernel<TM, NB, SROWS, ABL><<<a.tiles_m * a.tiles_n * a.ksplit, 512, kLds, st>>>(a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <bool SROWS>
int launch(const Args &a, hipStream_t st, int tm) {
    if (tm == 512) {
#ifdef QT_TUNING_BUILD
        if constexpr (!SROWS) {
            const char *e_abl = getenv("QT_FQT_ABLATE");       // timing experiments (tools/exp_linear_fqt.py --skip-checks): results are garbage
            switch (e_abl ? atoi(e_abl) : 0) {
                case 1: return launch_one<512, 1, false, 1>(a, st);
                case 2: return launch_one<512, 1, false, 2>(a, st);
                case 3: return launch_one<512, 1, false, 3>(a, st);
                case 5: return launch_one<512, 1, false, 5>(a, st);
                case 6: return launch_one<512, 1, false, 6>(a, st);
                case 7: return launch_one<512, 1, false, 7>(a, st);
                case 10: return launch_one<512, 1, false, 10>(a, st);
                case 8: return launch_one<512, 1, false, 8>(a, st);
                case 9: return launch_one<512, 1, false, 9>(a, st);
                default: break;
            }
        }
#endif
        return launch_one<512, 1, SROWS>(a, st);
    }
    const int worst_nt = a.gbase + (a.gextra ? 1 : 0);
    if (worst_nt <= 8) return launch_one<256, 1, SROWS>(a, st);
    return launch_one<256, 2, SROWS>(a, st);
}

// column tiles for `tiles_m` row tiles: as many as make whole rounds over the CUs, no wider than max_nt groups
void plan(long groups, int tiles_m, int max_nt, int cus, long &tn, long &rounds) {
    const long tn_min = (groups + max_nt - 1) / max_nt;
    rounds = (tiles_m * tn_min + cus - 1) / cus;
    tn = rounds * cus / tiles_m;
    if (tn < tn_min) tn = tn_min;
    if (tn > groups) tn = groups;
}

// How a problem is cut: row tile height, column tiles, and over how many workgroups the K loop of a tile is split.
struct Tiling {
    int tm, tiles_m, tiles_n, gbase, gextra, ksplit;
    size_t ws_bytes() const { return ksplit > 1 ? (size_t)tiles_m * tiles_n * ksplit * 8 * 32 * 1024 : 0; }      // 32 KiB per wave and split
    size_t tickets() const { return ksplit > 1 ? (size_t)tiles_m * tiles_n : 0; }
};

constexpr int kMinSplitSteps = 24;     // a split shorter than this does not pay for its prologue and the hand-off

Tiling make_tiling(int M, long groups, int K, bool may_split) {
    const int cus = cu_count();
    int force_tn = 0, force_tm = 0, force_ks = 0;
#ifdef QT_TUNING_BUILD
    const char *e_tn = getenv("QT_FQT_TILES_N"), *e_tm = getenv("QT_FQT_TM"), *e_ks = getenv("QT_FQT_KSPLIT");          // tools/ only
    force_tn = e_tn ? atoi(e_tn) : 0; force_tm = e_tm ? atoi(e_tm) : 0; force_ks = e_ks ? atoi(e_ks) : 0;
#endif
    // 512-row tiles (half the conversion work per multiplication) where they fill the chip with tiles at least five groups wide;
    // else 256-row tiles
    long tn5, r5, tn2, r2;
    plan(groups, (M + 511) / 512, 8, cus, tn5, r5);
    plan(groups, (M + 255) / 256, 16, cus, tn2, r2);
    int tm = 256;
    if (M > 256) {
        const double width5 = (double)groups / (double)tn5, fill5 = (double)((M + 511) / 512) * tn5 / ((double)r5 * cus);
        const double fill2 = (double)((M + 255) / 256) * tn2 / ((double)r2 * cus);
        if (width5 >= 5.0 && fill5 >= 0.9 * fill2) tm = 512;
    }
    // Narrow outputs: the widest 512-row tiles leave most of the chip idle, and narrower ones convert every weight for a handful of
    // multiplications (1024 x 5120 x 13824: 256 x 80 tiles ran at 0.20 of the bf16 peak).  Keep 512 x 128 tiles and split K over the
    // idle CUs instead: ksplit workgroups per tile, partial sums through the workspace, the last one to arrive writes the tile.
    int ksplit = 1;
    const int nk = K / kBK;
    if (may_split && M > 256) {
        const int tiles_m5 = (M + 511) / 512;
        const long tn_wide = (groups + 7) / 8, tiles = tiles_m5 * tn_wide;
        if (tiles * 4 <= (long)cus * 3) {                       // under three quarters of the chip
            int ks = (int)(cus / tiles);
            if (ks > 4) ks = 4;
            while (ks > 1 && nk / ks < kMinSplitSteps) --ks;
            if (ks > 1) { ksplit = ks; tm = 512; tn5 = tn_wide; }
        }
    }
    if (force_tm == 256 || force_tm == 512) { tm = force_tm; if (force_tm == 256) ksplit = 1; }
    Tiling t{};
    t.tm = tm;
    t.tiles_m = (M + tm - 1) / tm;
    long tn = tm == 512 ? tn5 : tn2;
    {
        const int max_nt = tm == 512 ? 8 : 16;
        const long tn_min = (groups + max_nt - 1) / max_nt;
        if (force_tn > 0) tn = force_tn;
        if (tn < tn_min) tn = tn_min;
        if (tn > groups) tn = groups;
    }
    if (force_ks >= 1 && force_ks <= 4 && may_split && nk / force_ks >= 1) ksplit = force_ks;
    if (tm != 512) ksplit = 1;
    t.tiles_n = (int)tn;
    t.gbase = (int)(groups / tn);
    t.gextra = (int)(groups % tn);
    t.ksplit = ksplit;
    return t;
}

int linear_fqt(const uint16_t *x_dev, const uint16_t *const *w_devs, const uint16_t *const *bias_devs, const int *ns, int count,
               const uint32_t *rows_dev, int signed_rows, uint32_t sign_mask, const uint16_t *map_dev, uint16_t *y_dev, int M, int K,
               float *ws_dev, size_t ws_bytes, uint32_t *tickets_dev, size_t n_tickets, bool may_split, void *stream) {
    if (count < 1 || count > kMaxSeg || !w_devs || !ns) return QT_ERR_BAD_ARG;
    long ntot = 0;
    for (int i = 0; i < count; ++i) {
        if (ns[i] < 0 || ns[i] % 16 != 0) return QT_ERR_BAD_ARG;
        ntot += ns[i];
    }
    if ((long)M * ntot == 0) return QT_OK;
    if (!x_dev || !y_dev || !rows_dev || !map_dev || M < 0 || K < kBK || K % kBK != 0 || ntot > (1L << 30)) return QT_ERR_BAD_ARG;
    // DMA sources are a 64-bit base + a 32-bit byte offset
    if ((long)M * K * 2 >= (1L << 32)) return QT_ERR_BAD_ARG;
    for (int i = 0; i < count; ++i)
        if ((long)ns[i] * K * 2 >= (1L << 32)) return QT_ERR_BAD_ARG;
    // y rows are stored 16 bytes per lane: y_dev and the row pitch (2 * sum n bytes) must keep them aligned
    if (((uintptr_t)x_dev & 15u) || ((uintptr_t)y_dev & 15u) || ((uintptr_t)rows_dev & 15u) || (ntot & 7)) return QT_ERR_UNALIGNED;
    for (int i = 0; i < count; ++i) {
        if (ns[i] && (!w_devs[i] || ((uintptr_t)w_devs[i] & 15u))) return QT_ERR_UNALIGNED;
        if (bias_devs && bias_devs[i] && ((uintptr_t)bias_devs[i] & 7u)) return QT_ERR_UNALIGNED;
    }
    const long groups = ntot / 16;
    const Tiling t = make_tiling(M, groups, K, may_split);
    if (t.ksplit > 1) {
        if (!ws_dev || !tickets_dev || ws_bytes < t.ws_bytes() || n_tickets < t.tickets()) return QT_ERR_BAD_ARG;
        if (((uintptr_t)ws_dev & 15u) || ((uintptr_t)tickets_dev & 3u)) return QT_ERR_UNALIGNED;
    }
    Args a{};
    a.x = x_dev; a.y = y_dev; a.rows = rows_dev; a.map = map_dev; a.sign_mask = sign_mask;
    a.M = M; a.K = K; a.ldc = (int)ntot;
    a.prio = 4;
    a.dbg = nullptr;
#ifdef QT_TUNING_BUILD
    {
        const char *e_pr = getenv("QT_FQT_PRIO");
        a.prio = e_pr ? atoi(e_pr) : 4;
        const char *e_st = getenv("QT_FQT_STAMPS");
        a.dbg = e_st ? (unsigned long long *)strtoull(e_st, nullptr, 0) : nullptr;
    }
#endif
    a.tiles_m = t.tiles_m; a.tiles_n = t.tiles_n; a.gbase = t.gbase; a.gextra = t.gextra;
    a.ksplit = t.ksplit; a.ws = ws_dev; a.tickets = tickets_dev;
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
    return signed_rows ? launch<true>(a, st, t.tm) : launch<false>(a, st, t.tm);
}

}  // namespace

extern "C" {

int qt_linear_fqt_bf16(const uint16_t *x_dev, const uint16_t *const *w_devs, const uint16_t *const *bias_devs, const int *ns,
                       int count, const uint32_t *rows_dev, int signed_rows, uint32_t sign_mask, const uint16_t *map_dev,
                       uint16_t *y_dev, int M, int K, void *stream) {
    return linear_fqt(x_dev, w_devs, bias_devs, ns, count, rows_dev, signed_rows, sign_mask, map_dev, y_dev, M, K, nullptr, 0, nullptr, 0,
                      false, stream);
}

int qt_linear_fqt_plan(int M, long n_total, int K, int *ksplit, size_t *ws_bytes, size_t *n_tickets) {
    if (M < 0 || n_total < 0 || n_total % 16 != 0 || K < kBK || K % kBK != 0) return QT_ERR_BAD_ARG;
    Tiling t{};
    t.ksplit = 1;
    if ((long)M * n_total != 0) t = make_tiling(M, n_total / 16, K, true);
    if (ksplit) *ksplit = t.ksplit;
    if (ws_bytes) *ws_bytes = t.ws_bytes();
    if (n_tickets) *n_tickets = t.tickets();
    return QT_OK;
}

int qt_linear_fqt_ws_bf16(const uint16_t *x_dev, const uint16_t *const *w_devs, const uint16_t *const *bias_devs, const int *ns,
                          int count, const uint32_t *rows_dev, int signed_rows, uint32_t sign_mask, const uint16_t *map_dev,
                          uint16_t *y_dev, int M, int K, float *ws_dev, size_t ws_bytes, uint32_t *tickets_dev, size_t n_tickets,
                          void *stream) {
    return linear_fqt(x_dev, w_devs, bias_devs, ns, count, rows_dev, signed_rows, sign_mask, map_dev, y_dev, M, K, ws_dev, ws_bytes,
                      tickets_dev, n_tickets, true, stream);
}

}  // extern "C"
