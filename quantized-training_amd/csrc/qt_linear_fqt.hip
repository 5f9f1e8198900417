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
// a table row, as an ordinary LDS load: hipcc schedules these and places their waits itself
__device__ __forceinline__ u32x4 lds_row(uint32_t addr) {
    return *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)addr;
}

// both operand tiles: row-major 128-byte rows (64 bf16), 16-byte chunk index XOR ((row >> 1) & 7)
__device__ __forceinline__ int chunk_off(int row, int chunk) { return row * kRowBytes + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Two packed bf16 weights -> the two packed values of the map, through the row form.  ROWMASK: 0xFF0 = rows by exponent,
// 0x1FF0 = rows by sign and exponent (maps whose negative half is not the mirror image: intN, uintN).
template <uint32_t ROWMASK>
__device__ __forceinline__ uint32_t quant_pair(uint32_t x, uint32_t tbl, uint32_t sign_mask, uint32_t &flags) {
    const uint32_t s = x >> 3;                                // (bits >> 7) << 4 of both halves
    const u32x4 p0 = lds_row(tbl + (s & ROWMASK));
    const u32x4 p1 = lds_row(tbl + ((s >> 16) & ROWMASK));
    const float t0 = qt_u2f(((x & 0x7FFFu) << 16) + p0.x);
    const float t1 = qt_u2f((x & 0x7FFF0000u) + p1.x);
    const float c0 = qt_u2f(p0.y), c1 = qt_u2f(p1.y);
    const float z0 = __builtin_amdgcn_fmed3f((t0 + c0) - c0, qt_u2f(p0.z), qt_u2f(p0.w));
    const float z1 = __builtin_amdgcn_fmed3f((t1 + c1) - c1, qt_u2f(p1.z), qt_u2f(p1.w));
    flags |= p0.y | p1.y;
    // results are exact bf16 values: the high halves are the answer
    return __builtin_amdgcn_perm(qt_f2u(z1), qt_f2u(z0), 0x07060302u) | (x & sign_mask);
}

// NB: weight pieces (8 rows x 128 bytes of bf16) per wave and k step (2, 3 or 4: tiles of up to 8, 12, 15 column groups);
// SROWS: the table has 512 rows (sign and exponent)
template <int NB, bool SROWS>
struct LinearFqt {
    static constexpr int kADepth = 3;
    static constexpr int kMaxNT = NB * 4 < 15 ? NB * 4 : (SROWS ? 14 : 15);
    static constexpr int kTbl = SROWS ? 8192 : 4096;
    static constexpr int kWBytes = kMaxNT * kGroupBytes;
    static constexpr int kLds = kTbl + kADepth * kABytes + 2 * kWBytes;
    static_assert(kLds <= 160 * 1024, "LDS budget");
    static constexpr uint32_t kRowMask = SROWS ? 0x1FF0u : 0xFF0u;
    static constexpr int kItems = 4 + NB;
    static constexpr int kEpiStride = 64 * ((2 * NB < 8 ? 2 * NB : 8) * 32 + 8);      // a wave's epilogue tile in LDS
    // items of column group J of NTW: the four activation DMA pieces ride on the first half of the groups, the weight items
    // (quantize + ds_write + reload) on the second half
    static constexpr int item_lo(int J, int NTW) {
        const int h = NTW / 2;
        if (h == 0) return 0;
        return J < h ? J * 4 / h : 4 + (J - h) * NB / (NTW - h);
    }
    static constexpr int item_hi(int J, int NTW) {
        const int h = NTW / 2;
        if (h == 0) return kItems;
        return J < h ? (J + 1) * 4 / h : 4 + (J - h + 1) * NB / (NTW - h);
    }
    static constexpr int writes_in(int J, int NTW) {
        const int lo = item_lo(J, NTW), hi = item_hi(J, NTW);
        return (hi > 4 ? hi : 4) - (lo > 4 ? lo : 4);
    }

    // One wave's share: rows [wm * 64, +64) x NTW column groups starting at group jbase of the tile whose first group is tg0.
    // Returns true when a flagged row was met (the caller redoes the tile).
    template <int NTW>
    static __device__ __forceinline__ bool run(const Args &a, uint8_t *lds, int m0, int tg0, int nt, int jbase, int w, int l) {
        const int r = l & 15, g = l >> 4, wm = w & 3;
        const int nk = a.K / kBK, klast = nk - 1;
        const long krow = (long)a.K * 2;                        // bytes per row of x and W
        const uint32_t l0 = lds_addr(lds);
        const uint32_t tbl = l0, a0 = l0 + kTbl, w0 = a0 + kADepth * kABytes;
        // ---- the row table into LDS (every thread one 16-byte row)
        {
            const int t = w * 64 + l;
            if (t < (SROWS ? 512 : 256)) {
                const u32x4 v = *(const u32x4 *)(a.rows + t * 4);
                asm volatile("ds_write_b128 %0, %1" ::"v"(tbl + t * 16), "v"(v) : "memory");
            }
        }
        // ---- activation DMA sources (k tile 0); LDS destinations are wave-uniform
        const uint8_t *ga[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (w * 4 + i) * 8 + (l >> 3), slot = l & 7;
            ga[i] = (const uint8_t *)a.x + (long)min(m0 + row, a.M - 1) * krow + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        // ---- weight pieces (8 rows x 128 bytes): piece p = w + 8 i; lane = (row l >> 3, 16-byte chunk l & 7 = k 8c .. 8c+7)
        const int npieces = nt * 2;
        const uint8_t *gw[NB];
        uint32_t wdst[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = w + 8 * i, pb = p < npieces ? p : (w & 1);          // surplus pieces repeat one (same bytes, same place)
            const int grp = tg0 + (pb >> 1);
            const int row = pb * 8 + (l >> 3), c = l & 7;
            const SegRef sg = seg_lookup(a, grp);
            gw[i] = (const uint8_t *)sg.w + (long)((grp - sg.g0) * 16 + (pb & 1) * 8 + (l >> 3)) * krow + c * 16;
            wdst[i] = chunk_off(row, c);
        }
        u32x4 wr[NB];
        uint32_t flags = 0;
        auto load_w = [&](auto ic, int kt) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            wr[I] = *(const u32x4 *)(gw[I] + (long)kt * kRowBytes);
        };
        // the activation DMA as inline asm, hidden from hipcc's wait-count model (qt_linear_fq8.hip, variant R)
        auto dma16 = [](const uint8_t *src, uint32_t dst) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
        };
        auto ds_write128 = [](uint32_t addr, u32x4 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        auto ds_write64 = [](uint32_t addr, u32x2 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        const uint32_t sign_mask = a.sign_mask;
        auto store_w = [&](auto ic, uint32_t wbase) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            u32x4 q;
            q.x = quant_pair<kRowMask>(wr[I].x, tbl, sign_mask, flags);
            q.y = quant_pair<kRowMask>(wr[I].y, tbl, sign_mask, flags);
            q.z = quant_pair<kRowMask>(wr[I].z, tbl, sign_mask, flags);
            q.w = quant_pair<kRowMask>(wr[I].w, tbl, sign_mask, flags);
            ds_write128(wbase + wdst[I], q);
        };
        // item 0-3: activation pieces of k tile ka into `as`; item 4 + i: weight piece i -- its registers (k tile kb - 1) are
        // quantized and written to the weight tile at LDS address `ws`, then reloaded with k tile kb
        auto item = [&](auto ic, int ka, uint32_t as, uint32_t ws, int kb) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < 4) {
                dma16(ga[I] + (long)ka * kRowBytes, as + (w * 4 + I) * 1024);
            } else {
                store_w(std::integral_constant<int, I - 4>{}, ws);
                load_w(std::integral_constant<int, I - 4>{}, kb);
            }
        };
        auto items = [&](auto lo, auto hi, int ka, uint32_t as, uint32_t ws, int kb) __attribute__((always_inline)) {
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            static_assert(HI - LO <= 8, "at most eight items");
            if constexpr (LO + 0 < HI) item(std::integral_constant<int, LO + 0>{}, ka, as, ws, kb);
            if constexpr (LO + 1 < HI) item(std::integral_constant<int, LO + 1>{}, ka, as, ws, kb);
            if constexpr (LO + 2 < HI) item(std::integral_constant<int, LO + 2>{}, ka, as, ws, kb);
            if constexpr (LO + 3 < HI) item(std::integral_constant<int, LO + 3>{}, ka, as, ws, kb);
            if constexpr (LO + 4 < HI) item(std::integral_constant<int, LO + 4>{}, ka, as, ws, kb);
            if constexpr (LO + 5 < HI) item(std::integral_constant<int, LO + 5>{}, ka, as, ws, kb);
            if constexpr (LO + 6 < HI) item(std::integral_constant<int, LO + 6>{}, ka, as, ws, kb);
            if constexpr (LO + 7 < HI) item(std::integral_constant<int, LO + 7>{}, ka, as, ws, kb);
        };
        constexpr auto kI0 = std::integral_constant<int, 0>{};
        constexpr auto kIA = std::integral_constant<int, 4>{};
        constexpr auto kIN = std::integral_constant<int, kItems>{};

        v4f acc[4][NTW > 0 ? NTW : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < (NTW > 0 ? NTW : 1); ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // lane-constant parts of the fragment addresses: k sub-step 0 = chunks 0..3, sub-step 1 = chunks 4..7
        const uint32_t a_lo = chunk_off(wm * 64 + r, g), a_hi = chunk_off(wm * 64 + r, 4 + g);
        const uint32_t b_lo = chunk_off(jbase * 16 + r, g), b_hi = chunk_off(jbase * 16 + r, 4 + g);

        auto compute = [&](uint32_t sa_, uint32_t sb_, int ka, uint32_t as, uint32_t ws, int kb) __attribute__((always_inline)) {
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
                    // LDS operations that may stay in flight (they return in order): the reads of the next two groups and the
                    // ds_writes of the weight items the two groups in front of this one carried; the table reads of those items
                    // only add to what is younger than this group's fragments, so the count stays a lower bound
                    constexpr int kAhead = (J + 1 < NTW ? 2 : 0) + (J + 2 < NTW ? 2 : 0) + (J >= 2 ? writes_in(J - 2, NTW) : 0) +
                                           (J >= 1 ? writes_in(J - 1, NTW) : 0);
                    if constexpr (J == 0) {
                        asm volatile("s_waitcnt lgkmcnt(%10)"
                                     : "+v"(fa_lo[0]), "+v"(fa_hi[0]), "+v"(fa_lo[1]), "+v"(fa_hi[1]), "+v"(fa_lo[2]), "+v"(fa_hi[2]), "+v"(fa_lo[3]),
                                       "+v"(fa_hi[3]), "+v"(fb_lo[0]), "+v"(fb_hi[0])
                                     : "n"(kAhead));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fb_lo[P]), "+v"(fb_hi[P]) : "n"(kAhead));
                    }
                    // operands swapped: D rows = W rows (output columns), D columns = x rows -- a lane ends up with four
                    // consecutive output columns of one row
                    const v8s bl = __builtin_bit_cast(v8s, fb_lo[P]), bh = __builtin_bit_cast(v8s, fb_hi[P]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, __builtin_bit_cast(v8s, fa_lo[i]), acc[i][J], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, __builtin_bit_cast(v8s, fa_hi[i]), acc[i][J], 0, 0, 0);
                    items(std::integral_constant<int, item_lo(J, NTW)>{}, std::integral_constant<int, item_hi(J, NTW)>{}, ka, as, ws, kb);
                    __builtin_amdgcn_sched_barrier(0);
                };
                read_b(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) read_b(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) step(std::integral_constant<int, 1>{});
                if constexpr (NTW > 2) step(std::integral_constant<int, 2>{});
                if constexpr (NTW > 3) step(std::integral_constant<int, 3>{});
                if constexpr (NTW > 4) step(std::integral_constant<int, 4>{});
                if constexpr (NTW > 5) step(std::integral_constant<int, 5>{});
                if constexpr (NTW > 6) step(std::integral_constant<int, 6>{});
                if constexpr (NTW > 7) step(std::integral_constant<int, 7>{});
            } else {
                items(kI0, kIN, ka, as, ws, kb);
            }
        };

        // prologue: the table is in place for every wave (barrier); activations of k tiles 0 and 1 on their way; weights of k tile
        // 0 quantized into weight tile 0, those of k tile 1 in registers
        __syncthreads();
        items(kI0, kIA, 0, a0, w0, 0);
        items(kI0, kIA, min(1, klast), a0 + kABytes, w0, 0);
        load_w(std::integral_constant<int, 0>{}, 0);
        if constexpr (NB > 1) load_w(std::integral_constant<int, 1>{}, 0);
        if constexpr (NB > 2) load_w(std::integral_constant<int, 2>{}, 0);
        if constexpr (NB > 3) load_w(std::integral_constant<int, 3>{}, 0);
        items(kIA, kIN, 0, a0, w0, min(1, klast));
        int a_slot = 0, a_tgt = 2;
        for (int kt = 0; kt < nk; ++kt) {
            // this wave's weight values of step kt are written (lgkmcnt) and its activation pieces have landed: they are older in
            // the vector-memory queue than the weight loads of step kt, which the conversions of the previous step waited for
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 + 2 * NB) : "memory");
            __builtin_amdgcn_s_barrier();                    // ... every wave's; and every wave is done with step kt - 1
            const int ka = min(kt + 2, klast), kb = min(kt + 2, klast);
            const uint32_t sa_ = a0 + a_slot * kABytes, sb_ = w0 + (kt & 1) * kWBytes, ws = w0 + ((kt + 1) & 1) * kWBytes;
            compute(sa_, sb_, ka, a0 + a_tgt * kABytes, ws, kb);
            a_tgt = a_slot; a_slot = a_slot == 2 ? 0 : a_slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // A lane that read a flagged row raises the workgroup's flag (through LDS: the rings are dead here).
        __syncthreads();
        volatile int *flag = (volatile int *)(lds + kTbl + 8 * kEpiStride);     // past the eight waves' epilogue tiles
        if (w == 0 && l == 0) *flag = 0;
        __syncthreads();
        if (flags & 1u) *flag = 1;
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

template <int NB, bool SROWS>
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
    using L = LinearFqt<NB, SROWS>;
    bool redo;
    switch (ntw) {                                          // wave-uniform
        case 0: redo = L::template run<0>(a, lds_t, m0, tg0, nt, jbase, w, l); break;
        case 1: redo = L::template run<1>(a, lds_t, m0, tg0, nt, jbase, w, l); break;
        case 2: redo = L::template run<2>(a, lds_t, m0, tg0, nt, jbase, w, l); break;
        case 3: if constexpr (NB >= 2) { redo = L::template run<3>(a, lds_t, m0, tg0, nt, jbase, w, l); break; }
        case 4: if constexpr (NB >= 2) { redo = L::template run<4>(a, lds_t, m0, tg0, nt, jbase, w, l); break; }
        case 5: if constexpr (NB >= 3) { redo = L::template run<5>(a, lds_t, m0, tg0, nt, jbase, w, l); break; }
        case 6: if constexpr (NB >= 3) { redo = L::template run<6>(a, lds_t, m0, tg0, nt, jbase, w, l); break; }
        case 7: if constexpr (NB >= 4) { redo = L::template run<7>(a, lds_t, m0, tg0, nt, jbase, w, l); break; }
        default:
            if constexpr (NB >= 4) redo = L::template run<8>(a, lds_t, m0, tg0, nt, jbase, w, l);
            else if constexpr (NB >= 3) redo = L::template run<6>(a, lds_t, m0, tg0, nt, jbase, w, l);
            else if constexpr (NB >= 2) redo = L::template run<4>(a, lds_t, m0, tg0, nt, jbase, w, l);
            else redo = L::template run<2>(a, lds_t, m0, tg0, nt, jbase, w, l);
            break;
    }
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

template <int NB, bool SROWS>
int launch_nb(const Args &a, hipStream_t st) {
    constexpr int kLds = LinearFqt<NB, SROWS>::kLds;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void *)linear_fqt_kernel<NB, SROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    linear_fqt_kernel<NB, SROWS><<<a.tiles_m * a.tiles_n, 512, kLds, st>>>(a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <bool SROWS>
int launch(const Args &a, hipStream_t st) {
    if (a.nb <= 1) return launch_nb<1, SROWS>(a, st);
    if (a.nb <= 2) return launch_nb<2, SROWS>(a, st);
    if (a.nb <= 3) return launch_nb<3, SROWS>(a, st);
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
    const int hard_nt = signed_rows ? 14 : 15;
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
