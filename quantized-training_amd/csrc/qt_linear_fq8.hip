// Fake-quant Linear with the weight's E4M3 / E5M2 fake-quantizer fused into the GEMM's operand path.
//
// Replaces   F.linear(input, self.weight_fake_quant(self.weight), self.bias)       modules/qat/linear.py:40-41
// for stateless FP8 fake-quantizers (e4m3 / e5m2 specs without `qs`: scale 1, fp8.py:10-67) on both operands:
//   * the activation arrives as the FP8 codes the producer passes already emit (qt_fake_quant_bf16_fp8 & co.);
//   * the bf16 weight is read ONCE from HBM, unquantized, straight into LDS (global_load_lds), and each wave converts the
//     fragments it multiplies on their way from LDS to the matrix core: v_cvt_scalef32_pk_{fp8,bf8}_bf16 at scale 1.0 is
//     the format's round-to-nearest-even on every bf16 input whose result is finite (tools/exp_cvt_bf16.hip: all 65 536
//     patterns; subnormals and the flush to zero included; the sign of a zero result differs, which no product sees).
//     Inputs that overflow the format (|w| > 464 / 61440: the reference saturates) and non-finite ones (the reference
//     maps +-Inf to NaN) show up as an all-ones exponent in the converted code; a lane that sees one redoes its 32
//     elements with the closed form (qt_fp_sat_u32) -- so fq(W) never exists in HBM and the codes are bit-exact;
//   * products of the codes are exactly the reference's bf16 products; v_mfma_scale_f32_16x16x128_f8f6f4 at unit E8M0
//     scales (the 2x-rate FP8 instruction), fp32 accumulation, bias and the bf16 rounding in the epilogue.
//
// Work decomposition (M = rows of x, N = rows of W = output columns).  A workgroup owns 256 rows x (16 nt) columns,
// nt <= 12; 8 waves = 4 row bands (64 rows) x 2 column halves (ceil(nt/2) and floor(nt/2) 16-column groups: the two
// waves that share a SIMD are one of each, so an odd nt still balances per SIMD).  The host picks the column widths so
// that the grid is a whole number of rounds over the CUs -- 1024 x 11008: 4 x 64 tiles of 176 / 160 columns = 256
// workgroups, not 4 x 43 of 256 (172 workgroups, a third of the chip idle).  Tile ids are dealt so that the row tiles of
// one column tile run on the same XCD at the same time: a weight tile leaves HBM once and is re-read from that XCD's L2.
// Up to four weight tensors sharing one activation (q / k / v projections) form the "segments" of one launch.
//
// What runs (LinearFq8R / LinearFq8R2 below): a lane loads 8 bf16 weights (16 B) into registers, converts them a
// step later and writes the 8 FP8 codes into a ring of FP8 weight tiles in LDS; the activations come by LDS-DMA into a ring three
// deep; both operands' fragments are two ds_read_b128 each; counted vmcnt + one raw s_barrier per k step (fragment reads are inline
// asm so that hipcc does not drain the DMA queue in front of them, see qt_mx_gemm.hip).  R2 (tiles of up to four column groups: the
// o and down projections) makes a step two k tiles deep.  qt_mlp_fq8_bf16 is R's pair mode (gate and up segments interleaved, the
// epilogue computes fq(silu(gate) * up)).  Round 2's first variant (raw bf16 weight tiles by LDS-DMA, converted by every wave that
// multiplies them: 78 against 56 us at 1024 x 11008 x 4096) was removed in round 4.
//
// Measured on MI355X (tools/exp_linear_fq8.py): 1024 x 11008 x 4096 in 54-57 us against 66-70 us for the weight pass +
// library GEMM pair.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_formats.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // native vector: usable as a "+v" asm operand
typedef short v2s __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int kTM = 256, kBK = 128, kMaxSeg = 4, kMaxNT = 12;
constexpr int kABytes = kTM * kBK;                  // one activation tile: 32 KiB of FP8 codes
constexpr int kUnitE8M0 = 127;                      // 2^0

struct Segment {
    const uint16_t *w;        // [n][K] bf16
    const uint16_t *bias;     // [n] bf16 or NULL
    int g0;                   // first 16-column group of this weight in the concatenation of all weights
};

struct Args {
    const uint8_t *x8;        // [M][K] FP8 codes
    uint16_t *y;              // [M][ldc] bf16
    int M, K, ldc;
    int tiles_m, tiles_n, nseg;
    int gbase, gextra;        // column tile j covers gbase + (j < gextra) groups of 16 columns
    int nb;                   // ceil(4 * widest tile / 8): weight DMA pieces the widest tile needs per wave
    int dbg;                  // tuning switches (QT_FQ8_DEBUG): 2 = no multiplications, 128 = no issue stagger
    Segment seg[kMaxSeg];
    // pair mode (qt_mlp_fq8_bf16): seg[0] = gate, seg[1] = up weights [N][K]; the virtual column groups alternate gate / up
    // (group 2 P: gate rows 16 P .., group 2 P + 1: the same rows of up), gbase / gextra count PAIRS, and the epilogue writes
    // h = fq(silu(bf16(gate)) * bf16(up)) as bf16 values (y, [M][ldc]) and FP8 codes (y8, [M][ldc])
    int pair;
    uint8_t *y8;
    qt_format out_fmt;        // the output fake-quantizer (E4M3 / E5M2 closed form, unit scale)
#ifdef QT_TUNING_BUILD
    unsigned long long *stamps;   // tools/ only (QT_FQ8_STAMPS = device address): per workgroup {cycles, 100 MHz ticks} around the tile
#endif
};

// Column tile tn of tiles_n: first unit (16-column group, or gate / up pair) and unit count.  The gextra tiles that are one unit wider
// are spread evenly over the tile index (tile tn starts at floor(tn * units / tiles_n)): with all of them in front, the XCDs that own
// the first tiles (tile ids are contiguous per XCD) carry up to a fifth more work than the others and the launch waits for them.
__device__ __forceinline__ void tile_span(const Args &a, int tn, int &first, int &count);

// The weight holding column group `grp`, by compile-time indices only: a run-time index into the kernel-argument struct makes
// hipcc copy the whole struct to scratch memory and read its fields from there.
struct SegRef { const uint16_t *w, *bias; int g0; };
__device__ __forceinline__ SegRef seg_lookup(const Args &a, int grp) {
    SegRef r{a.seg[0].w, a.seg[0].bias, a.seg[0].g0};
    if (a.nseg > 1 && grp >= a.seg[1].g0) r = SegRef{a.seg[1].w, a.seg[1].bias, a.seg[1].g0};
    if (a.nseg > 2 && grp >= a.seg[2].g0) r = SegRef{a.seg[2].w, a.seg[2].bias, a.seg[2].g0};
    if (a.nseg > 3 && grp >= a.seg[3].g0) r = SegRef{a.seg[3].w, a.seg[3].bias, a.seg[3].g0};
    return r;
}

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

// activation tile: row-major 128-byte rows, chunk index XOR ((row >> 1) & 7)
__device__ __forceinline__ int a_chunk_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// two packed bf16 pairs -> four FP8 codes (the first conversion's untouched half is overwritten by the second)
template <bool E5M2>
__device__ __forceinline__ uint32_t cvt_bf16x4(uint32_t p0, uint32_t p1) {
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

// 32 bf16 weights of one lane (four 16-byte chunks) -> the lane's 32-byte B fragment of fq(W) codes: the hardware conversion,
// right for every input whose result is finite.  Overflow and non-finite inputs come out as NaN / Inf codes, which poison
// the accumulator and are caught after the k loop (slow_tile redoes such a tile with the closed form).
template <bool E5M2>
__device__ __forceinline__ v8i convert_frag(const u32x4 (&raw)[4]) {
    const uint32_t in[16] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w,
                             raw[2].x, raw[2].y, raw[2].z, raw[2].w, raw[3].x, raw[3].y, raw[3].z, raw[3].w};
    v8i f;
#pragma unroll
    for (int d = 0; d < 8; ++d) f[d] = (int)cvt_bf16x4<E5M2>(in[2 * d], in[2 * d + 1]);
    return f;
}

// four bf16 values (two packed words) -> their four exact codes (closed form of the value map, fp8.py:10-67)
template <bool E5M2>
__device__ __forceinline__ uint32_t exact_bf16x4(uint32_t p0, uint32_t p1) {
    constexpr int mb = E5M2 ? 2 : 3, emin = E5M2 ? -14 : -6;
    constexpr float fmax = E5M2 ? 57344.0f : 448.0f;
    return qt_pack_fp8x4<E5M2>(qt_u2f(qt_fp_sat_u32(p0 << 16, mb, emin, fmax)), qt_u2f(qt_fp_sat_u32(p0 & 0xFFFF0000u, mb, emin, fmax)),
                               qt_u2f(qt_fp_sat_u32(p1 << 16, mb, emin, fmax)), qt_u2f(qt_fp_sat_u32(p1 & 0xFFFF0000u, mb, emin, fmax)));
}

#define QT_LDS_WAIT4(n, v) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))

// FX / FW: 0 = E4M3, 1 = E5M2 (the instruction's operand format codes); NB: weight pieces per wave and step (2, 4 or 6: tiles of up
// to 4, 8, 12 column groups)
// ---- variant R: weights converted in registers on the way in ------------------------------------------------------------
// The raw bf16 weight tile is what keeps the ring above two deep (76 KiB per stage).  Here a lane loads 8 bf16 weights (16 bytes)
// into registers with an ordinary global load -- hipcc places the vmcnt waits of those itself, copies of the registers included,
// which is what makes loop-carried loads safe (an inline-asm load in flight is not) -- converts them a step later and writes the
// 8 FP8 codes into a two-deep LDS ring of FP8 weight tiles (nt x 2 KiB); the activations keep coming by LDS-DMA into a ring
// three deep.  So every weight is converted once per workgroup (not once per row band), the fragments of both operands are two
// ds_read_b128 each, and a step's loads have a whole step to land.  No branch inside a step (hipcc would sink the
// multiplications behind it); past the last k tile the last one is requested again.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_round(float f) { return qt_u2f(pack_bf16x2(f, 0.0f) << 16); }      // round to bf16, keep as float
// silu(g) * u on four (gate, up) pairs, with the roundings of the module chain (LlamaMLP: bf16 GEMM outputs, SiLU in fp32 rounded
// to bf16, product rounded to bf16 -- the arithmetic of silu_mul_kernel, csrc/qt_model_ops.hip): two packed bf16 words
__device__ __forceinline__ void silu_mul4(const float (&gt)[4], const float (&up)[4], uint32_t &w0, uint32_t &w1) {
    float p[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g = bf_round(gt[e]), u = bf_round(up[e]);
        p[e] = bf_round(g / (1.0f + expf(-g))) * u;
    }
    w0 = pack_bf16x2(p[0], p[1]);
    w1 = pack_bf16x2(p[2], p[3]);
}

// ABL (timing experiments only, QT_FQ8_ABLATE; results are garbage): 1 no multiplications, 2 no weight items (load / convert /
// ds_write), 3 no activation DMA, 4 no fragment reads and no multiplications.  Measured at 1024 x 11008 x 4096 (55.6 us whole):
// 48.8 / 39.5 / 47.9 / 48.5 us -- the k loop is paced by the operand streams (76 KB per step and CU), not by the matrix core.
// PF > 0 (round 6): every wave also touches, PF k tiles ahead, its share of the weight lines this column tile will need -- one 4-byte
// load per 128-byte line, result never read.  The tiles_m workgroups that share a column tile run in lock step on one XCD, so today all
// of them wait out the HBM miss of each weight line together (TCC: 19 % misses, but every weight request sees the miss latency);
// brought into that XCD's L2 ahead of time the demand loads are L2 hits.  Measured ceiling of the idea: -12 % (every column tile
// reading the first tile's weights, QT_FQ8_DEBUG=256, round 5).  Each workgroup prefetches 1 / tiles_m of the tile's lines.
template <int FX, int FW, int NB, bool PAIR = false, int ABL = 0, int PF = 0>
struct LinearFq8R {
    static constexpr int kADepth = 3;
    // ABL == 20 (not an ablation: the two-register-set variant): every weight piece has TWO register sets, so a weight request has two
    // steps to land instead of one.  The requests are inline-asm loads with hand-counted waits: hipcc cannot count the (hidden)
    // activation DMA entries that share the wave's in-order queue, and its own waits for ordinary loads came out up to eight entries
    // too strict -- which is what kept a second register set from paying off in round 2.  The k loop is unrolled by two so that each
    // set lives in fixed registers (nothing rotates: a register written by a load in flight must never be copied).
    static constexpr bool W2 = (ABL == 20);
    static constexpr int kWSets = W2 ? 2 : 1;
    static constexpr int kWWait = 2 * NB + 3;               // queue entries allowed behind a weight request when its registers are read
    // PF kernels count exactly.  A weight request of set S is read two steps after it was issued; behind it in the queue are the rest
    // of its own step's requests, one whole step (4 activation pieces + NB requests) and the reading step's 4 pieces and first requests:
    // 2 NB + 7, plus the two prefetch loads issued at the two step starts in between.  (2 NB + 3 above is that count for step 0, whose
    // requests were issued back to back in the prologue; kept for every step it is four entries too strict -- harmless while every
    // entry is an L2 hit that lands within a step, but a prefetch load is an HBM miss by design and must never be waited for.)
    static constexpr int kWWaitFirst = 2 * NB + 3 + (PF > 0 ? 1 : 0);
    static constexpr int kWWaitSteady = PF > 0 ? 2 * NB + 9 : kWWait;
    static constexpr int kStepWait = 4 + 2 * NB + (PF > 0 ? 1 : 0);      // behind the activation pieces of the step that starts
    static constexpr int kWBytes = NB * 4 * 1024;           // FP8 weight tile: up to 8 NB pieces of 4 rows x 128 bytes
    static constexpr int kLds = kADepth * kABytes + 2 * kWBytes;
    static constexpr int kItems = 4 + NB;
    // Which items column group J of NTW carries: the four activation DMA pieces ride on the first half of the groups, the weight
    // items (convert + ds_write + reload) on the second half -- a weight register is then reloaded about one step before its
    // next conversion, and the waits hipcc puts in front of the conversions do not catch this step's DMA pieces young.
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
    static constexpr int writes_in(int J, int NTW) {        // weight items (one ds_write each) among them
        const int lo = item_lo(J, NTW), hi = item_hi(J, NTW);
        return (hi > 4 ? hi : 4) - (lo > 4 ? lo : 4);
    }

    template <int NTW>
    static __device__ __forceinline__ bool run(const Args &a, uint8_t *lds, int m0, int tg0, int nt, int jbase, int w, int l) {
        const int r = l & 15, g = l >> 4, wm = w & 3;
        const int nk = a.K / kBK, klast = nk - 1;
        // ---- activation DMA sources (k tile 0); LDS destinations are wave-uniform
        const uint8_t *ga[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (w * 4 + i) * 8 + (l >> 3), slot = l & 7;
            ga[i] = a.x8 + (long)min(m0 + row, a.M - 1) * a.K + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        // ---- weight pieces (4 rows x 256 bytes of bf16): piece p = w + 8 i; lane = (row l >> 4, 16-byte chunk l & 15 = k 8c .. 8c+7)
        const int npieces = nt * 4;
        const uint8_t *gw[NB];
        const uint8_t *ub[NB];                                 // wave-uniform part of gw (first row of the piece): W2 keeps these in SGPRs
        const uint32_t w_lane = (uint32_t)(l >> 4) * (uint32_t)a.K * 2u + (uint32_t)(l & 15) * 16u;      // the lane's part, same for every piece
        uint32_t wdst[NB];                                     // where the lane's 8 codes go inside an FP8 weight tile
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = w + 8 * i, pb = p < npieces ? p : (w & 3);          // surplus pieces repeat one (same bytes, same place)
#ifdef QT_TUNING_BUILD
            // timing probe (QT_FQ8_DEBUG=256, results are garbage): every column tile reads the FIRST tile's weights, so the weight
            // stream hits in L2 -- the upper bound of anything that would bring the weights into L2 ahead of the demand loads
            const int grp = ((a.dbg & 256) ? 0 : tg0) + (pb >> 2);
#else
            const int grp = tg0 + (pb >> 2);
#endif
            const int row = pb * 4 + (l >> 4), c = l & 15;
            if constexpr (PAIR) {
                const uint16_t *wb = (grp & 1) ? a.seg[1].w : a.seg[0].w;
                ub[i] = (const uint8_t *)wb + ((long)((grp >> 1) * 16 + (pb & 3) * 4) * a.K) * 2;
            } else {
                const SegRef sg = seg_lookup(a, grp);
                ub[i] = (const uint8_t *)sg.w + ((long)((grp - sg.g0) * 16 + (pb & 3) * 4) * a.K) * 2;
            }
            gw[i] = ub[i] + w_lane;
            wdst[i] = row * 128 + (((c >> 1) ^ ((row >> 1) & 7)) << 4) + (c & 1) * 8;
        }
        // ---- L2 prefetch of the weight stream (PF > 0): this wave's lines of the tile's 32 nt lines per k tile (two per weight row)
        const uint8_t *pf_base = nullptr;                     // wave-uniform: first row this wave touches, k tile 0
        uint32_t pf_lane = 0;                                  // the lane's line inside the wave's span
        if constexpr (PF > 0) {
            const int tm_ = m0 / kTM;
            if constexpr (PAIR) {
                // waves 0-3: the gate rows of the tile's pairs, waves 4-7: the up rows; rows [P0 16, (P0 + nt / 2) 16) of each matrix
                const int lines = 16 * nt;                                        // per matrix and k tile
                const int lc = (lines + a.tiles_m - 1) / a.tiles_m, lw = (lc + 3) / 4;
                int first = tm_ * lc + (w & 3) * lw;
                first = first < lines - 1 ? first : lines - 1;
                const int mine = min(lw, lines - first);
                const int ln = first + min(l, mine - 1);
                const uint16_t *wb = (w >> 2) ? a.seg[1].w : a.seg[0].w;
                pf_base = (const uint8_t *)wb + (long)((tg0 / 2) * 16) * a.K * 2;
                pf_lane = (uint32_t)(ln >> 1) * (uint32_t)a.K * 2u + (uint32_t)(ln & 1) * 128u;
            } else {
                const int lines = 32 * nt;
                const int lc = (lines + a.tiles_m - 1) / a.tiles_m, lw = (lc + 7) / 8;
                int first = tm_ * lc + w * lw;
                first = first < lines - 1 ? first : lines - 1;
                // the span stays inside the weight that holds its first row (a tile may straddle two weights of a q / k / v launch)
                const int grp0 = tg0 + (first >> 5);
                const SegRef sg = seg_lookup(a, grp0);
                int seg_end_grp = tg0 + nt;                                       // first group past this weight inside the tile
                if (a.nseg > 1 && a.seg[1].g0 > grp0 && a.seg[1].g0 < seg_end_grp) seg_end_grp = a.seg[1].g0;
                if (a.nseg > 2 && a.seg[2].g0 > grp0 && a.seg[2].g0 < seg_end_grp) seg_end_grp = a.seg[2].g0;
                if (a.nseg > 3 && a.seg[3].g0 > grp0 && a.seg[3].g0 < seg_end_grp) seg_end_grp = a.seg[3].g0;
                const int last = (seg_end_grp - tg0) * 32 - 1;
                const int mine = min(lw, last - first + 1);
                const int ln = first + min(l, mine - 1) - (grp0 - tg0) * 32;       // line relative to group grp0's first row
                pf_base = (const uint8_t *)sg.w + (long)((grp0 - sg.g0) * 16) * a.K * 2;
                pf_lane = (uint32_t)(ln >> 1) * (uint32_t)a.K * 2u + (uint32_t)(ln & 1) * 128u;
            }
        }
        uint32_t pf_sink = 0;
        auto prefetch_w = [&](int kt) __attribute__((always_inline)) {
            if constexpr (PF > 0)
                asm volatile("global_load_dword %0, %1, %2" : "=v"(pf_sink) : "v"(pf_lane), "s"(pf_base + (long)kt * (2 * kBK)) : "memory");
        };
        u32x4 wr[kWSets][NB];
        auto load_w_asm = [](u32x4 &dst, uint32_t lane_off, const uint8_t *base) __attribute__((always_inline)) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(base) : "memory");
        };
        auto wait_w_asm = [](u32x4 &reg, auto first) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(decltype(first)::value ? kWWaitFirst : kWWaitSteady) : "memory");
        };
        // (sc: the register set, + 2 when the step is the k loop's first -- its wait count differs, see kWWaitFirst)
        auto load_w = [&](auto ic, auto sc, int kt) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value, S = decltype(sc)::value & 1;
            if constexpr (W2) load_w_asm(wr[S][I], w_lane, ub[I] + (long)kt * (2 * kBK));
            else wr[S][I] = *(const u32x4 *)(gw[I] + (long)kt * (2 * kBK));
        };
        // The activation DMA as inline asm: with the builtin, hipcc's wait-count model sees a pending "flat" access to LDS and turns
        // every wait for a weight register into vmcnt(0), which also waits for the step's own DMA pieces.  Hidden from the model the
        // waits become counted ones that are at most four entries too strict (never too lax: uncounted entries only make the real
        // queue longer than the one hipcc waits on).
        // (asm statements sit in non-generic lambdas: inside a generic one clang rejects operands captured by reference)
        auto dma16 = [](const uint8_t *src, uint32_t dst) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
        };
        auto ds_write64 = [](uint32_t addr, u32x2 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        auto ds_write32 = [](uint32_t addr, uint32_t v) __attribute__((always_inline)) {
            asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        auto store_w = [&](auto ic, auto sc, uint32_t wbase) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value, S = decltype(sc)::value & 1;
            if constexpr (W2) wait_w_asm(wr[S][I], std::integral_constant<bool, (decltype(sc)::value >= 2)>{});
            const u32x2 codes = {cvt_bf16x4<FW == 1>(wr[S][I].x, wr[S][I].y), cvt_bf16x4<FW == 1>(wr[S][I].z, wr[S][I].w)};
            const uint32_t addr = wbase + wdst[I];
            ds_write64(addr, codes);
        };
        // item 0-3: activation pieces of k tile ka into `as`; item 4 + i: weight piece i -- its registers (k tile kb - 1) are
        // converted and written to the FP8 tile at LDS address `ws`, then reloaded with k tile kb
        auto item = [&](auto ic, auto sc, int ka, uint32_t as, uint32_t ws, int kb) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < 4) {
                if constexpr (ABL != 3 && ABL != 5 && ABL != 6 && ABL != 8 && ABL != 10) dma16(ga[I] + (long)ka * kBK, as + (w * 4 + I) * 1024);
            } else if constexpr (ABL == 7 || ABL == 8 || ABL == 10 || ABL == 11) {
                // transport probe: the weight piece travels by LDS-DMA (raw bf16 into the FP8 ring's space, results are garbage) instead of
                // through registers + conversion + ds_write
                dma16(gw[I - 4] + (long)kb * (2 * kBK), lds_addr(lds) + kADepth * kABytes + (w * NB + (I - 4)) * 1024);
            } else if constexpr (ABL != 2 && ABL != 5 && ABL != 6 && ABL != 9) {
                store_w(std::integral_constant<int, I - 4>{}, sc, ws);
                load_w(std::integral_constant<int, I - 4>{}, sc, kb);
            }
        };
        auto items = [&](auto lo, auto hi, auto sc, int ka, uint32_t as, uint32_t ws, int kb) __attribute__((always_inline)) {
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            static_assert(HI - LO <= 10, "at most ten items");
            if constexpr (LO + 0 < HI) item(std::integral_constant<int, LO + 0>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 1 < HI) item(std::integral_constant<int, LO + 1>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 2 < HI) item(std::integral_constant<int, LO + 2>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 3 < HI) item(std::integral_constant<int, LO + 3>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 4 < HI) item(std::integral_constant<int, LO + 4>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 5 < HI) item(std::integral_constant<int, LO + 5>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 6 < HI) item(std::integral_constant<int, LO + 6>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 7 < HI) item(std::integral_constant<int, LO + 7>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 8 < HI) item(std::integral_constant<int, LO + 8>{}, sc, ka, as, ws, kb);
            if constexpr (LO + 9 < HI) item(std::integral_constant<int, LO + 9>{}, sc, ka, as, ws, kb);
        };
        constexpr auto kI0 = std::integral_constant<int, 0>{};
        constexpr auto kIA = std::integral_constant<int, 4>{};
        constexpr auto kIN = std::integral_constant<int, kItems>{};

        v4f acc[4][NTW > 0 ? NTW : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < (NTW > 0 ? NTW : 1); ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // lane-constant parts of the fragment addresses (both operands are FP8 tiles of 128-byte rows, same swizzle)
        const uint32_t a_lo = a_chunk_off(wm * 64 + r, g), a_hi = a_chunk_off(wm * 64 + r, 4 + g);
        const uint32_t b_lo = a_chunk_off(jbase * 16 + r, g), b_hi = a_chunk_off(jbase * 16 + r, 4 + g);

        auto compute = [&](auto sc, uint32_t sa_, uint32_t sb_, int ka, uint32_t as, uint32_t ws, int kb) __attribute__((always_inline)) {
            if constexpr (NTW > 0) {
                u32x4 fa_lo[4], fa_hi[4], fb_lo[3], fb_hi[3];
                if constexpr (ABL != 4 && ABL != 6 && (ABL < 9 || ABL >= 20)) {
                    fa_lo[0] = ds_read128<0 * 2048>(sa_ + a_lo); fa_hi[0] = ds_read128<0 * 2048>(sa_ + a_hi);
                    fa_lo[1] = ds_read128<1 * 2048>(sa_ + a_lo); fa_hi[1] = ds_read128<1 * 2048>(sa_ + a_hi);
                    fa_lo[2] = ds_read128<2 * 2048>(sa_ + a_lo); fa_hi[2] = ds_read128<2 * 2048>(sa_ + a_hi);
                    fa_lo[3] = ds_read128<3 * 2048>(sa_ + a_lo); fa_hi[3] = ds_read128<3 * 2048>(sa_ + a_hi);
                }
                auto read_b = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    if constexpr (ABL != 4 && ABL != 6 && (ABL < 9 || ABL >= 20)) {
                        fb_lo[J % 3] = ds_read128<J * 2048>(sb_ + b_lo);
                        fb_hi[J % 3] = ds_read128<J * 2048>(sb_ + b_hi);
                    }
                };
                v8i fa[4];
                auto step = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int P = J % 3;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (J + 2 < NTW) read_b(std::integral_constant<int, J + 2>{});
                    // LDS operations still allowed in flight (they return in order): everything issued behind this group's fragment
                    // reads -- the reads of the next two groups (two each) and the ds_writes of the weight items that the two
                    // groups in front of this one carried
                    constexpr int kAhead = (J + 1 < NTW ? 2 : 0) + (J + 2 < NTW ? 2 : 0) + (J >= 2 ? writes_in(J - 2, NTW) : 0) +
                                           (J >= 1 ? writes_in(J - 1, NTW) : 0);
                    if constexpr (J == 0) {
                        asm volatile("s_waitcnt lgkmcnt(%10)"
                                     : "+v"(fa_lo[0]), "+v"(fa_hi[0]), "+v"(fa_lo[1]), "+v"(fa_hi[1]), "+v"(fa_lo[2]), "+v"(fa_hi[2]), "+v"(fa_lo[3]),
                                       "+v"(fa_hi[3]), "+v"(fb_lo[0]), "+v"(fb_hi[0])
                                     : "n"(kAhead));
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            fa[i] = v8i{(int)fa_lo[i].x, (int)fa_lo[i].y, (int)fa_lo[i].z, (int)fa_lo[i].w,
                                        (int)fa_hi[i].x, (int)fa_hi[i].y, (int)fa_hi[i].z, (int)fa_hi[i].w};
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fb_lo[P]), "+v"(fb_hi[P]) : "n"(kAhead));
                    }
                    const v8i fb = v8i{(int)fb_lo[P].x, (int)fb_lo[P].y, (int)fb_lo[P].z, (int)fb_lo[P].w,
                                       (int)fb_hi[P].x, (int)fb_hi[P].y, (int)fb_hi[P].z, (int)fb_hi[P].w};
                    if constexpr (ABL != 1 && ABL != 4 && ABL != 6 && (ABL < 9 || ABL >= 20)) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[i], acc[i][J], FW, FX, 0, kUnitE8M0, 0, kUnitE8M0);
                    }
                    items(std::integral_constant<int, item_lo(J, NTW)>{}, std::integral_constant<int, item_hi(J, NTW)>{}, sc, ka, as, ws, kb);
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
            } else {
                items(kI0, kIN, sc, ka, as, ws, kb);
            }
        };

        const uint32_t l0 = lds_addr(lds), w0 = l0 + kADepth * kABytes;
        // prologue: activations of k tiles 0 and 1 on their way; weights of k tile 0 converted into FP8 tile 0, those of k tile 1
        // in registers
        constexpr auto kS0 = std::integral_constant<int, 0>{};
        constexpr auto kS1 = std::integral_constant<int, kWSets - 1>{};
        auto load_all = [&](auto sc, int kt) __attribute__((always_inline)) {
            load_w(std::integral_constant<int, 0>{}, sc, kt);
            if constexpr (NB > 1) load_w(std::integral_constant<int, 1>{}, sc, kt);
            if constexpr (NB > 2) load_w(std::integral_constant<int, 2>{}, sc, kt);
            if constexpr (NB > 3) load_w(std::integral_constant<int, 3>{}, sc, kt);
            if constexpr (NB > 4) load_w(std::integral_constant<int, 4>{}, sc, kt);
            if constexpr (NB > 5) load_w(std::integral_constant<int, 5>{}, sc, kt);
        };
        items(kI0, kIA, kS0, 0, l0, w0, 0);
        items(kI0, kIA, kS0, min(1, klast), l0 + kABytes, w0, 0);
        load_all(kS0, 0);
        int a_slot = 0, a_tgt = 2;
        auto one_step = [&](auto sc, int kt, int ahead) __attribute__((always_inline)) {
            // this wave's FP8 codes of step kt are written (lgkmcnt) and its activation pieces have landed: they are older in the
            // vector-memory queue than the weight loads of step kt, which the conversions of the previous step waited for
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W2 ? kStepWait : 4 + 2 * NB) : "memory");
            __builtin_amdgcn_s_barrier();                    // ... every wave's; and every wave is done with step kt - 1
            if constexpr (PF > 0) prefetch_w(min(kt + PF, klast));
            const int ka = min(kt + 2, klast), kb = min(kt + ahead, klast);
            const uint32_t sa_ = l0 + a_slot * kABytes, sb_ = w0 + (kt & 1) * kWBytes, ws = w0 + ((kt + 1) & 1) * kWBytes;
            compute(sc, sa_, sb_, ka, l0 + a_tgt * kABytes, ws, kb);
            a_tgt = a_slot; a_slot = a_slot == 2 ? 0 : a_slot + 1;
        };
        if constexpr (!W2) {
            // weights of k tile 0 converted into FP8 tile 0, those of k tile 1 in registers
            items(kIA, kIN, kS0, 0, l0, w0, min(1, klast));
            for (int kt = 0; kt < nk; ++kt) one_step(kS0, kt, 2);
        } else {
            // k tile 0: requested, awaited, converted into FP8 tile 0; then set 1 <- k tile 1, set 0 <- k tile 2.  Step kt converts k tile
            // kt + 1 out of set (kt + 1) & 1 into FP8 tile (kt + 1) & 1 and refills that set with k tile kt + 3.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (ABL != 2) {
                auto conv0 = [&](auto ic) __attribute__((always_inline)) {
                    constexpr int I = decltype(ic)::value;
                    asm volatile("" : "+v"(wr[0][I]));
                    const u32x2 codes = {cvt_bf16x4<FW == 1>(wr[0][I].x, wr[0][I].y), cvt_bf16x4<FW == 1>(wr[0][I].z, wr[0][I].w)};
                    ds_write64(w0 + wdst[I], codes);
                };
                conv0(std::integral_constant<int, 0>{});
                if constexpr (NB > 1) conv0(std::integral_constant<int, 1>{});
                if constexpr (NB > 2) conv0(std::integral_constant<int, 2>{});
                if constexpr (NB > 3) conv0(std::integral_constant<int, 3>{});
                if constexpr (NB > 4) conv0(std::integral_constant<int, 4>{});
                if constexpr (NB > 5) conv0(std::integral_constant<int, 5>{});
            }
            load_all(kS1, min(1, klast));
            load_all(kS0, min(2, klast));
            int kt = 0;
            one_step(std::integral_constant<int, 2 + (kWSets - 1)>{}, kt, 3);          // (set 1, first step)
            for (kt = 1; kt + 1 < nk; kt += 2) {
                one_step(kS0, kt, 3);
                one_step(kS1, kt + 1, 3);
            }
            if (kt < nk) one_step(kS0, kt, 3);
            // requests past the last k tile are still in flight into the (now unused) weight registers: they stay reserved until the
            // queue has drained
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(wr[0][i]), "+v"(wr[1][i]));
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (W2) {
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(wr[0][i]), "+v"(wr[kWSets - 1][i]));
        }
        if constexpr (PF > 0) asm volatile("" : "+v"(pf_sink));          // the prefetch loads' landing register stays reserved until here
        // Overflowed or non-finite weights (and NaN activations) leave NaN / Inf in the accumulators: such a tile is redone
        // by slow_tile.  The workgroup-wide vote goes through LDS (the rings are dead here).
        bool bad = false;
        if constexpr (NTW > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bad |= (qt_f2u(acc[i][j][e]) & 0x7F800000u) == 0x7F800000u;
        }
        // (ABL 20 is the shipped two-register-set loop, not an ablation.  Rounds 3 and 4 cleared `bad` for it as well, so the widest
        // tiles -- gate / up, q / k / v -- never took the redo path: an overflowing weight came out as NaN instead of saturating.  No
        // test held such a weight in a twelve-group tile; tests/test_gpu_parity.py::test_linear_fq8_wide_tiles_redo_overflowing_weights
        // does now.)
        if constexpr (ABL != 0 && ABL != 20) bad = false;
        __syncthreads();
        volatile int *flag = (volatile int *)(lds + 8 * 64 * (6 * 32 + 8));     // past the eight waves' epilogue tiles
        if (w == 0 && l == 0) *flag = 0;
        __syncthreads();
        if (bad) *flag = 1;
        __syncthreads();
        if (*flag) return true;

        // ---- epilogue.  Lane (r, g) of tile (i, j) holds y[row wm*64 + i*16 + r][column group j, columns 4g .. 4g+3]: stored
        // straight from the registers that is 32 contiguous bytes per row and instruction.  The rings are dead, so every wave
        // turns its 64 x 16 NTW tile around in its own 13 KiB of LDS (no barrier: wave-private) and stores whole rows of it,
        // 32 NTW contiguous bytes each, 16 bytes per lane.
        if constexpr (PAIR && NTW > 0) {
            // gate / up pairs: column group 2 jp holds gate, 2 jp + 1 up, both for output columns 16 P .. 16 P + 15
            static_assert(!PAIR || NTW % 2 == 0, "pair mode gives every wave whole pairs");
            constexpr int kRowV = NTW * 16 + 8, kRowC = NTW * 8 + 8;   // bytes per tile row: bf16 values, FP8 codes
            const uint32_t tv = l0 + w * (64 * (6 * 32 + 8)), tc = tv + 64 * kRowV;
            const bool oe5 = a.out_fmt.p0 == 2;
#pragma unroll
            for (int jp = 0; jp < NTW / 2; ++jp) {
                const int col = ((tg0 + jbase) / 2 + jp) * 16 + 4 * g;
                float bg[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.seg[0].bias) {
                    const uint2 b = *(const uint2 *)(a.seg[0].bias + col);
                    bg[0] = qt_u2f(b.x << 16); bg[1] = qt_u2f(b.x & 0xFFFF0000u); bg[2] = qt_u2f(b.y << 16); bg[3] = qt_u2f(b.y & 0xFFFF0000u);
                }
                if (a.seg[1].bias) {
                    const uint2 b = *(const uint2 *)(a.seg[1].bias + col);
                    bu[0] = qt_u2f(b.x << 16); bu[1] = qt_u2f(b.x & 0xFFFF0000u); bu[2] = qt_u2f(b.y << 16); bu[3] = qt_u2f(b.y & 0xFFFF0000u);
                }
#pragma unroll
                for (int ih = 0; ih < 2; ++ih) {
                    uint32_t o[4];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = 2 * ih + ii;
                        const float gt[4] = {acc[i][2 * jp][0] + bg[0], acc[i][2 * jp][1] + bg[1], acc[i][2 * jp][2] + bg[2], acc[i][2 * jp][3] + bg[3]};
                        const float up[4] = {acc[i][2 * jp + 1][0] + bu[0], acc[i][2 * jp + 1][1] + bu[1], acc[i][2 * jp + 1][2] + bu[2],
                                             acc[i][2 * jp + 1][3] + bu[3]};
                        silu_mul4(gt, up, o[2 * ii], o[2 * ii + 1]);
                    }
                    const uint2 codes = oe5 ? fq8_hw_vec8<true>(o, a.out_fmt) : fq8_hw_vec8<false>(o, a.out_fmt);   // o: now fq(product)
                    if (a.y) {                                         // NULL: the codes only (they decode to exactly these values)
                        ds_write64(tv + ((2 * ih) * 16 + r) * kRowV + jp * 32 + g * 8, u32x2{o[0], o[1]});
                        ds_write64(tv + ((2 * ih + 1) * 16 + r) * kRowV + jp * 32 + g * 8, u32x2{o[2], o[3]});
                    }
                    ds_write32(tc + ((2 * ih) * 16 + r) * kRowC + jp * 16 + g * 4, codes.x);
                    ds_write32(tc + ((2 * ih + 1) * 16 + r) * kRowC + jp * 16 + g * 4, codes.y);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long col0 = (long)((tg0 + jbase) / 2) * 16;
            constexpr int kVChunks = 64 * NTW, kCChunks = 64 * (NTW / 2);       // 16-byte chunks of the values / of the codes
            if (a.y) {
#pragma unroll
                for (int it = 0; it < (kVChunks + 63) / 64; ++it) {
                    const int c = it * 64 + l, row = c / NTW, ch = c % NTW;
                    uint2 lo_, hi_;
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(lo_), "=&v"(hi_) : "v"(tv + row * kRowV + ch * 16) : "memory");
                    const int grow = m0 + wm * 64 + row;
                    if (c < kVChunks && grow < a.M) *(uint4 *)(a.y + (long)grow * a.ldc + col0 + ch * 8) = uint4{lo_.x, lo_.y, hi_.x, hi_.y};
                }
            }
#pragma unroll
            for (int it = 0; it < (kCChunks + 63) / 64; ++it) {
                const int c = it * 64 + l, row = c / (NTW / 2), ch = c % (NTW / 2);
                uint2 lo_, hi_;
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(lo_), "=&v"(hi_) : "v"(tc + row * kRowC + ch * 16) : "memory");
                const int grow = m0 + wm * 64 + row;
                if (c < kCChunks && grow < a.M) *(uint4 *)(a.y8 + (long)grow * a.ldc + col0 + ch * 16) = uint4{lo_.x, lo_.y, hi_.x, hi_.y};
            }
        }
        if constexpr (!PAIR && NTW > 0) {
            constexpr int kRowB = NTW * 32 + 8;                        // + 8: rows 16 apart would otherwise share banks
            const uint32_t tbase = l0 + w * (64 * (6 * 32 + 8));
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

// The redo path of a tile whose fast pass saw NaN / Inf: every weight goes through the closed form of the value map.  Plain
// loops, operands straight from global memory, one 16 x 16 output tile at a time -- only ever taken for weights beyond the
// format's range or non-finite values.
template <int FX, int FW>
__device__ __forceinline__ void slow_tile(const Args &a, int m0, int tg0, int jbase, int ntw, int w, int l) {
    const int r = l & 15, g = l >> 4, wm = w & 3;
    const int nk = a.K / kBK;
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
            const uint8_t *xrow = a.x8 + (long)min(row, a.M - 1) * a.K;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kt = 0; kt < nk; ++kt) {
                const uint4 xl = *(const uint4 *)(xrow + kt * kBK + 16 * g), xh = *(const uint4 *)(xrow + kt * kBK + 64 + 16 * g);
                const uint4 *wl = (const uint4 *)(wrow + kt * kBK + 16 * g), *wh = (const uint4 *)(wrow + kt * kBK + 64 + 16 * g);
                const uint4 w0 = wl[0], w1 = wl[1], w2 = wh[0], w3 = wh[1];
                const v8i fa = {(int)xl.x, (int)xl.y, (int)xl.z, (int)xl.w, (int)xh.x, (int)xh.y, (int)xh.z, (int)xh.w};
                v8i fb;
                fb[0] = (int)exact_bf16x4<FW == 1>(w0.x, w0.y); fb[1] = (int)exact_bf16x4<FW == 1>(w0.z, w0.w);
                fb[2] = (int)exact_bf16x4<FW == 1>(w1.x, w1.y); fb[3] = (int)exact_bf16x4<FW == 1>(w1.z, w1.w);
                fb[4] = (int)exact_bf16x4<FW == 1>(w2.x, w2.y); fb[5] = (int)exact_bf16x4<FW == 1>(w2.z, w2.w);
                fb[6] = (int)exact_bf16x4<FW == 1>(w3.x, w3.y); fb[7] = (int)exact_bf16x4<FW == 1>(w3.z, w3.w);
                acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa, acc, FW, FX, 0, kUnitE8M0, 0, kUnitE8M0);
            }
            if (row < a.M) {
                const uint2 o = {pack_bf16x2(acc[0] + bv[0], acc[1] + bv[1]), pack_bf16x2(acc[2] + bv[2], acc[3] + bv[3])};
                *(uint2 *)(a.y + (long)row * a.ldc + col) = o;
            }
        }
    }
}

// Pair mode's redo path: gate and up tiles with exactly fake-quantized weights, then the same epilogue arithmetic, straight stores.
template <int FX, int FW>
__device__ __forceinline__ void slow_tile_pair(const Args &a, int m0, int tg0, int jbase, int ntw, int w, int l) {
    const int r = l & 15, g = l >> 4, wm = w & 3;
    const int nk = a.K / kBK;
    const bool oe5 = a.out_fmt.p0 == 2;
#pragma unroll 1
    for (int jp = 0; jp < ntw / 2; ++jp) {
        const int P = (tg0 + jbase) / 2 + jp, col = P * 16 + 4 * g;
        const uint16_t *grow_ = a.seg[0].w + (long)(P * 16 + r) * a.K, *urow_ = a.seg[1].w + (long)(P * 16 + r) * a.K;
        float bg[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.seg[0].bias) {
            const uint2 b = *(const uint2 *)(a.seg[0].bias + col);
            bg[0] = qt_u2f(b.x << 16); bg[1] = qt_u2f(b.x & 0xFFFF0000u); bg[2] = qt_u2f(b.y << 16); bg[3] = qt_u2f(b.y & 0xFFFF0000u);
        }
        if (a.seg[1].bias) {
            const uint2 b = *(const uint2 *)(a.seg[1].bias + col);
            bu[0] = qt_u2f(b.x << 16); bu[1] = qt_u2f(b.x & 0xFFFF0000u); bu[2] = qt_u2f(b.y << 16); bu[3] = qt_u2f(b.y & 0xFFFF0000u);
        }
#pragma unroll 1
        for (int ih = 0; ih < 2; ++ih) {
            uint32_t o[4];
#pragma unroll 1
            for (int ii = 0; ii < 2; ++ii) {
                const int row = m0 + wm * 64 + (2 * ih + ii) * 16 + r;
                const uint8_t *xrow = a.x8 + (long)min(row, a.M - 1) * a.K;
                v4f ag = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int kt = 0; kt < nk; ++kt) {
                    const uint4 xl = *(const uint4 *)(xrow + kt * kBK + 16 * g), xh = *(const uint4 *)(xrow + kt * kBK + 64 + 16 * g);
                    const v8i fa = {(int)xl.x, (int)xl.y, (int)xl.z, (int)xl.w, (int)xh.x, (int)xh.y, (int)xh.z, (int)xh.w};
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const uint16_t *wrow = m ? urow_ : grow_;
                        const uint4 *wl = (const uint4 *)(wrow + kt * kBK + 16 * g), *wh = (const uint4 *)(wrow + kt * kBK + 64 + 16 * g);
                        const uint4 w0 = wl[0], w1 = wl[1], w2 = wh[0], w3 = wh[1];
                        v8i fb;
                        fb[0] = (int)exact_bf16x4<FW == 1>(w0.x, w0.y); fb[1] = (int)exact_bf16x4<FW == 1>(w0.z, w0.w);
                        fb[2] = (int)exact_bf16x4<FW == 1>(w1.x, w1.y); fb[3] = (int)exact_bf16x4<FW == 1>(w1.z, w1.w);
                        fb[4] = (int)exact_bf16x4<FW == 1>(w2.x, w2.y); fb[5] = (int)exact_bf16x4<FW == 1>(w2.z, w2.w);
                        fb[6] = (int)exact_bf16x4<FW == 1>(w3.x, w3.y); fb[7] = (int)exact_bf16x4<FW == 1>(w3.z, w3.w);
                        if (m) au = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa, au, FW, FX, 0, kUnitE8M0, 0, kUnitE8M0);
                        else ag = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa, ag, FW, FX, 0, kUnitE8M0, 0, kUnitE8M0);
                    }
                }
                const float gt[4] = {ag[0] + bg[0], ag[1] + bg[1], ag[2] + bg[2], ag[3] + bg[3]};
                const float up[4] = {au[0] + bu[0], au[1] + bu[1], au[2] + bu[2], au[3] + bu[3]};
                if (ii == 0) silu_mul4(gt, up, o[0], o[1]);
                else silu_mul4(gt, up, o[2], o[3]);
            }
            const uint2 codes = oe5 ? fq8_hw_vec8<true>(o, a.out_fmt) : fq8_hw_vec8<false>(o, a.out_fmt);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int row = m0 + wm * 64 + (2 * ih + ii) * 16 + r;
                if (row < a.M) {
                    if (a.y) *(uint2 *)(a.y + (long)row * a.ldc + col) = uint2{o[2 * ii], o[2 * ii + 1]};
                    *(uint32_t *)(a.y8 + (long)row * a.ldc + col) = ii ? codes.y : codes.x;
                }
            }
        }
    }
}

// ---- variant R2: variant R with TWO k tiles per step, for narrow tiles (at most four column groups) -----------------------
// A 256 x 64 tile multiplies 8 times per wave and k tile; the step's fixed costs -- barrier, the waits for the DMA pieces and
// the weight registers, the LDS latency in front of the first multiplication -- are then ~0.7 of a 1.0 us step (1024 x 4096 x
// 11008: 85 us for 86 steps).  Here a step is 256 deep: half as many of those, twice the work to cover them.  Activations two
// steps of 64 KiB (the DMA of step s + 1 lands during step s), FP8 weight tiles 2 x 2 x 8 KiB: 160 KiB of LDS.
// ABL (timing experiments, QT_FQ8_R2_ABLATE): 2 no weight items, 3 no activation DMA, 5 neither (fragment reads + multiplications only)
template <int FX, int FW, int ABL = 0>
struct LinearFq8R2 {
    static constexpr int NB = 2;                            // weight pieces per wave and k tile
    static constexpr int kAStage = 2 * kABytes, kWTile = NB * 4 * 1024, kWStage = 2 * kWTile;
    static constexpr int kLds = 2 * kAStage + 2 * kWStage;
    static constexpr int kItems = 8 + 2 * NB;               // per step and wave: 8 activation DMA pieces, 4 weight items
    // ABL == 30 (not an ablation -- variant RX, "register-extended rings"): the kernel uses half the register file, and its k loop is
    // paced by how many operand bytes a CU has in flight (one step of each operand: 96 KB against a latency of 1.1 - 2.2 us).  RX
    // requests BOTH operands into registers two steps ahead (two register sets each: 64 + 32 VGPRs; inline-asm loads, hand-counted
    // waits, the loop unrolled by two so that nothing rotates) and moves them into the LDS stage of the next step with ds_write
    // (activations as they are, weights converted) -- twice the bytes in flight with the same 160 KiB of LDS.  EXPERIMENT, opt-in
    // (QT_FQ8_R2_ABLATE=30; exact, all parity checks pass): it is NOT faster -- 1024 x 4096 x 11008 74.0 us against 71.9, 4096^3-shaped o
    // projection 30.3 against 28.8 -- so the narrow-tile loop is not limited by the bytes it has in flight.
    static constexpr bool RX = (ABL == 30);
    static constexpr int kRxWait = 2 * kItems - 1;          // queue entries behind a request issued two steps earlier at the same place
    // items of group gi (k half h = gi / NTW, column group J = gi % NTW) out of G = 2 NTW: DMA pieces on the first half of the
    // groups, weight items (convert + ds_write + reload) on the second half
    static constexpr int item_lo(int gi, int G) { return gi < G / 2 ? gi * 8 / (G / 2) : 8 + (gi - G / 2) * (2 * NB) / (G - G / 2); }
    static constexpr int item_hi(int gi, int G) { return gi < G / 2 ? (gi + 1) * 8 / (G / 2) : 8 + (gi - G / 2 + 1) * (2 * NB) / (G - G / 2); }
    static constexpr int writes_in(int gi, int G) {
        const int lo = item_lo(gi, G), hi = item_hi(gi, G);
        if (RX) return hi - lo;                              // every item ends in one ds_write
        return (hi > 8 ? hi : 8) - (lo > 8 ? lo : 8);
    }

    template <int NTW>
    static __device__ __forceinline__ bool run(const Args &a, uint8_t *lds, int m0, int tg0, int nt, int jbase, int w, int l) {
        static_assert(NTW <= 2, "narrow tiles only");
        const int r = l & 15, g = l >> 4, wm = w & 3;
        const int ns = a.K / (2 * kBK), slast = ns - 1;       // steps of two k tiles
        const uint8_t *ga[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (w * 4 + i) * 8 + (l >> 3), slot = l & 7;
            ga[i] = a.x8 + (long)min(m0 + row, a.M - 1) * a.K + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        uint32_t aoff[4];                                      // RX: the lane's part of ga[i] (the base a.x8 + k offset is wave-uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] = (uint32_t)(ga[i] - a.x8);
        const int npieces = nt * 4;
        const uint8_t *gw[NB];
        const uint8_t *ub[NB];                                 // wave-uniform part of gw
        const uint32_t w_lane = (uint32_t)(l >> 4) * (uint32_t)a.K * 2u + (uint32_t)(l & 15) * 16u;
        uint32_t wdst[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = w + 8 * i, pb = p < npieces ? p : (w & 3);
            const int grp = tg0 + (pb >> 2);
            const int row = pb * 4 + (l >> 4), c = l & 15;
            const SegRef sg = seg_lookup(a, grp);
            ub[i] = (const uint8_t *)sg.w + ((long)((grp - sg.g0) * 16 + (pb & 3) * 4) * a.K) * 2;
            gw[i] = ub[i] + w_lane;
            wdst[i] = row * 128 + (((c >> 1) ^ ((row >> 1) & 7)) << 4) + (c & 1) * 8;
        }
        u32x4 wr[2 * NB];                                      // [k half][piece]
        u32x4 ar[RX ? 2 : 1][RX ? 8 : 1], wx[RX ? 2 : 1][RX ? 2 * NB : 1];      // RX: [set][item]
        auto load_asm = [](u32x4 &dst, uint32_t lane_off, const uint8_t *base) __attribute__((always_inline)) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(base) : "memory");
        };
        auto wait_asm = [](u32x4 &reg) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(kRxWait) : "memory");
        };
        auto ds_write128 = [](uint32_t addr, u32x4 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        const uint32_t a_lane = (uint32_t)l * 16u;
        auto dma16 = [](const uint8_t *src, uint32_t dst) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
        };
        auto ds_write64 = [](uint32_t addr, u32x2 v) __attribute__((always_inline)) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        };
        // item 0-7: activation piece (k half I / 4, piece I % 4) of step sa into `as`; item 8 + i: weight registers i (k half i / NB,
        // piece i % NB, holding step sb - 1) converted into the FP8 tiles at `ws`, then reloaded with step sb
        auto item = [&](auto ic, auto sc, int sa, uint32_t as, uint32_t ws, int sb) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value, S = decltype(sc)::value;
            if constexpr (RX) {
                // sb: the step these registers are refilled with (two ahead of the one they hold); the set in hand holds step sa's operands
                if constexpr (I < 8) {
                    wait_asm(ar[S][I]);
                    ds_write128(as + (I >> 2) * kABytes + (w * 4 + (I & 3)) * 1024 + a_lane, ar[S][I]);
                    load_asm(ar[S][I], aoff[I & 3], a.x8 + (long)(2 * sb + (I >> 2)) * kBK);
                } else {
                    constexpr int Wi = I - 8, H = Wi / NB, P = Wi % NB;
                    wait_asm(wx[S][Wi]);
                    const u32x2 codes = {cvt_bf16x4<FW == 1>(wx[S][Wi].x, wx[S][Wi].y), cvt_bf16x4<FW == 1>(wx[S][Wi].z, wx[S][Wi].w)};
                    ds_write64(ws + H * kWTile + wdst[P], codes);
                    load_asm(wx[S][Wi], w_lane, ub[P] + (long)(2 * sb + H) * (2 * kBK));
                }
            } else if constexpr (I < 8) {
                if constexpr (ABL != 3 && ABL != 5) dma16(ga[I & 3] + (long)(2 * sa + (I >> 2)) * kBK, as + (I >> 2) * kABytes + (w * 4 + (I & 3)) * 1024);
            } else if constexpr (ABL != 2 && ABL != 5) {
                constexpr int Wi = I - 8, H = Wi / NB, P = Wi % NB;
                const u32x2 codes = {cvt_bf16x4<FW == 1>(wr[Wi].x, wr[Wi].y), cvt_bf16x4<FW == 1>(wr[Wi].z, wr[Wi].w)};
                const uint32_t addr = ws + H * kWTile + wdst[P];
                ds_write64(addr, codes);
                wr[Wi] = *(const u32x4 *)(gw[P] + (long)(2 * sb + H) * (2 * kBK));
            }
        };
        auto items = [&](auto lo, auto hi, auto sc, int sa, uint32_t as, uint32_t ws, int sb) __attribute__((always_inline)) {
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            static_assert(HI - LO <= 12, "at most twelve items");
            if constexpr (LO + 0 < HI) item(std::integral_constant<int, LO + 0>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 1 < HI) item(std::integral_constant<int, LO + 1>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 2 < HI) item(std::integral_constant<int, LO + 2>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 3 < HI) item(std::integral_constant<int, LO + 3>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 4 < HI) item(std::integral_constant<int, LO + 4>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 5 < HI) item(std::integral_constant<int, LO + 5>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 6 < HI) item(std::integral_constant<int, LO + 6>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 7 < HI) item(std::integral_constant<int, LO + 7>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 8 < HI) item(std::integral_constant<int, LO + 8>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 9 < HI) item(std::integral_constant<int, LO + 9>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 10 < HI) item(std::integral_constant<int, LO + 10>{}, sc, sa, as, ws, sb);
            if constexpr (LO + 11 < HI) item(std::integral_constant<int, LO + 11>{}, sc, sa, as, ws, sb);
        };
        constexpr auto kI0 = std::integral_constant<int, 0>{};
        constexpr auto kIA = std::integral_constant<int, 8>{};
        constexpr auto kIN = std::integral_constant<int, kItems>{};

        v4f acc[4][NTW > 0 ? NTW : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < (NTW > 0 ? NTW : 1); ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        const uint32_t a_lo = a_chunk_off(wm * 64 + r, g), a_hi = a_chunk_off(wm * 64 + r, 4 + g);
        const uint32_t b_lo = a_chunk_off(jbase * 16 + r, g), b_hi = a_chunk_off(jbase * 16 + r, 4 + g);

        // one k half (activations at sa_, FP8 weights at sb_) of a step, carrying the step's items of its groups
        auto half = [&](auto hc, auto sc, uint32_t sa_, uint32_t sb_, int sa, uint32_t as, uint32_t ws, int sb) __attribute__((always_inline)) {
            constexpr int H = decltype(hc)::value, G = 2 * NTW;
            u32x4 fa_lo[4], fa_hi[4], fb_lo[2], fb_hi[2];
            fa_lo[0] = ds_read128<0 * 2048>(sa_ + a_lo); fa_hi[0] = ds_read128<0 * 2048>(sa_ + a_hi);
            fa_lo[1] = ds_read128<1 * 2048>(sa_ + a_lo); fa_hi[1] = ds_read128<1 * 2048>(sa_ + a_hi);
            fa_lo[2] = ds_read128<2 * 2048>(sa_ + a_lo); fa_hi[2] = ds_read128<2 * 2048>(sa_ + a_hi);
            fa_lo[3] = ds_read128<3 * 2048>(sa_ + a_lo); fa_hi[3] = ds_read128<3 * 2048>(sa_ + a_hi);
            fb_lo[0] = ds_read128<0>(sb_ + b_lo); fb_hi[0] = ds_read128<0>(sb_ + b_hi);
            if constexpr (NTW > 1) { fb_lo[1] = ds_read128<2048>(sb_ + b_lo); fb_hi[1] = ds_read128<2048>(sb_ + b_hi); }
            v8i fa[4];
            auto step = [&](auto jc) __attribute__((always_inline)) {
                constexpr int J = decltype(jc)::value, gi = H * NTW + J;
                __builtin_amdgcn_sched_barrier(0);
                // LDS operations issued behind this group's fragment reads: the next group's reads and the ds_writes of the group in front
                constexpr int kAhead = (J + 1 < NTW ? 2 : 0) + (J >= 1 ? writes_in(gi - 1, G) : 0);
                if constexpr (J == 0) {
                    asm volatile("s_waitcnt lgkmcnt(%10)"
                                 : "+v"(fa_lo[0]), "+v"(fa_hi[0]), "+v"(fa_lo[1]), "+v"(fa_hi[1]), "+v"(fa_lo[2]), "+v"(fa_hi[2]), "+v"(fa_lo[3]),
                                   "+v"(fa_hi[3]), "+v"(fb_lo[0]), "+v"(fb_hi[0])
                                 : "n"(kAhead));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        fa[i] = v8i{(int)fa_lo[i].x, (int)fa_lo[i].y, (int)fa_lo[i].z, (int)fa_lo[i].w,
                                    (int)fa_hi[i].x, (int)fa_hi[i].y, (int)fa_hi[i].z, (int)fa_hi[i].w};
                } else {
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fb_lo[J]), "+v"(fb_hi[J]) : "n"(kAhead));
                }
                const v8i fb = v8i{(int)fb_lo[J].x, (int)fb_lo[J].y, (int)fb_lo[J].z, (int)fb_lo[J].w,
                                   (int)fb_hi[J].x, (int)fb_hi[J].y, (int)fb_hi[J].z, (int)fb_hi[J].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[i], acc[i][J], FW, FX, 0, kUnitE8M0, 0, kUnitE8M0);
                items(std::integral_constant<int, item_lo(gi, G)>{}, std::integral_constant<int, item_hi(gi, G)>{}, sc, sa, as, ws, sb);
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});
            if constexpr (NTW > 1) step(std::integral_constant<int, 1>{});
        };

        const uint32_t l0 = lds_addr(lds), w0 = l0 + 2 * kAStage;
        constexpr auto kS0 = std::integral_constant<int, 0>{};
        constexpr auto kS1 = std::integral_constant<int, RX ? 1 : 0>{};
        if constexpr (!RX) {
        // prologue: activations of step 0 on their way; weights of step 0 converted into FP8 stage 0, those of step 1 in registers
        items(kI0, kIA, kS0, 0, l0, w0, 0);
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) wr[i] = *(const u32x4 *)(gw[i % NB] + (long)(i / NB) * (2 * kBK));
        items(kIA, kIN, kS0, 0, l0, w0, min(1, slast));
        for (int s = 0; s < ns; ++s) {
            // the weight loads of step s + 1 are the newest entries of this wave's queue; its DMA pieces of step s are older
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NB) : "memory");
            __builtin_amdgcn_s_barrier();
            const int sa = min(s + 1, slast), sb = min(s + 2, slast);
            const uint32_t sa_ = l0 + (s & 1) * kAStage, sb_ = w0 + (s & 1) * kWStage;
            const uint32_t as = l0 + ((s + 1) & 1) * kAStage, ws = w0 + ((s + 1) & 1) * kWStage;
            if constexpr (NTW > 0) {
                half(std::integral_constant<int, 0>{}, kS0, sa_, sb_, sa, as, ws, sb);
                half(std::integral_constant<int, 1>{}, kS0, sa_ + kABytes, sb_ + kWTile, sa, as, ws, sb);
            } else {
                items(kI0, kIN, kS0, sa, as, ws, sb);
            }
        }
        } else {
            // RX prologue: step 0's operands requested into set 0, awaited, moved into stage 0; then set 1 <- step 1, set 0 <- step 2, in
            // the order the loop issues them (activation items, then weight items), so that the loop's constant wait count holds from
            // its first step.  Step s moves step s + 1 out of set (s + 1) & 1 into stage (s + 1) & 1 and refills that set with step s + 3.
            auto request = [&](auto sc, int step) __attribute__((always_inline)) {
                constexpr int S = decltype(sc)::value;
#pragma unroll
                for (int i = 0; i < 8; ++i) load_asm(ar[S][i], aoff[i & 3], a.x8 + (long)(2 * step + (i >> 2)) * kBK);
#pragma unroll
                for (int i = 0; i < 2 * NB; ++i) load_asm(wx[S][i], w_lane, ub[i % NB] + (long)(2 * step + i / NB) * (2 * kBK));
            };
            request(kS0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("" : "+v"(ar[0][i]));
                ds_write128(l0 + (i >> 2) * kABytes + (w * 4 + (i & 3)) * 1024 + a_lane, ar[0][i]);
            }
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) {
                asm volatile("" : "+v"(wx[0][i]));
                const u32x2 codes = {cvt_bf16x4<FW == 1>(wx[0][i].x, wx[0][i].y), cvt_bf16x4<FW == 1>(wx[0][i].z, wx[0][i].w)};
                ds_write64(w0 + (i / NB) * kWTile + wdst[i % NB], codes);
            }
            request(kS1, min(1, slast));
            request(kS0, min(2, slast));
            auto rx_step = [&](auto sc, int s) __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's ds_writes into the stage about to be read
                __builtin_amdgcn_s_barrier();                               // ... every wave's; and every wave is done with step s - 1
                const int sa = min(s + 1, slast), sb = min(s + 3, slast);
                const uint32_t sa_ = l0 + (s & 1) * kAStage, sb_ = w0 + (s & 1) * kWStage;
                const uint32_t as = l0 + ((s + 1) & 1) * kAStage, ws = w0 + ((s + 1) & 1) * kWStage;
                if constexpr (NTW > 0) {
                    half(std::integral_constant<int, 0>{}, sc, sa_, sb_, sa, as, ws, sb);
                    half(std::integral_constant<int, 1>{}, sc, sa_ + kABytes, sb_ + kWTile, sa, as, ws, sb);
                } else {
                    items(kI0, kIN, sc, sa, as, ws, sb);
                }
            };
            int s = 0;
            for (; s + 1 < ns; s += 2) {
                rx_step(kS1, s);
                rx_step(kS0, s + 1);
            }
            if (s < ns) rx_step(kS1, s);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (RX) {                                   // requests past the last step were still landing in these registers
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(ar[0][i]), "+v"(ar[1][i]));
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) asm volatile("" : "+v"(wx[0][i]), "+v"(wx[1][i]));
        }
        bool bad = false;
        if constexpr (NTW > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bad |= (qt_f2u(acc[i][j][e]) & 0x7F800000u) == 0x7F800000u;
        }
        __syncthreads();
        volatile int *flag = (volatile int *)(lds + 8 * 64 * (6 * 32 + 8));
        if (w == 0 && l == 0) *flag = 0;
        __syncthreads();
        if (bad) *flag = 1;
        __syncthreads();
        if ((ABL == 0 || RX) && *flag) return true;
        if constexpr (NTW > 0) {
            constexpr int kRowB = NTW * 32 + 8;
            const uint32_t tbase = l0 + w * (64 * (6 * 32 + 8));
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

// tools/ only: the clock the chip holds inside a tile = cycles (s_memtime) per 100 MHz tick (s_memrealtime), per workgroup
#ifdef QT_TUNING_BUILD
#define QT_FQ8_STAMP_BEGIN                                                              \
    unsigned long long qt_c0 = 0, qt_r0 = 0;                                            \
    if (a.stamps) { qt_c0 = __builtin_amdgcn_s_memtime(); qt_r0 = __builtin_amdgcn_s_memrealtime(); }
#define QT_FQ8_STAMP_END                                                                \
    if (a.stamps && t == 0) {                                                           \
        a.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - qt_c0;                \
        a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - qt_r0;        \
    }
#else
#define QT_FQ8_STAMP_BEGIN
#define QT_FQ8_STAMP_END
#endif

template <int FX, int FW, int ABL = 0>
__global__ __launch_bounds__(512, 1) void linear_fq8r2_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_r2[];
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
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
    using L = LinearFq8R2<FX, FW, ABL>;
    QT_FQ8_STAMP_BEGIN
    bool redo;
    switch (ntw) {                                          // wave-uniform
        case 0: redo = L::template run<0>(a, lds_r2, m0, tg0, nt, jbase, w, l); break;
        case 1: redo = L::template run<1>(a, lds_r2, m0, tg0, nt, jbase, w, l); break;
        default: redo = L::template run<2>(a, lds_r2, m0, tg0, nt, jbase, w, l); break;
    }
    if (redo) slow_tile<FX, FW>(a, m0, tg0, jbase, ntw, w, l);
    QT_FQ8_STAMP_END
}

template <int FX, int FW, int NB, bool PAIR, int ABL = 0>
__global__ __launch_bounds__(512, 1) void linear_fq8r_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_r[];
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntiles = a.tiles_m * a.tiles_n;
    int id = blockIdx.x;
    {
        const int per = ntiles / 8, rem = ntiles % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;
    }
    const int tn = id / a.tiles_m, tm = id % a.tiles_m;
    // column groups of this tile; in pair mode gbase / gextra count gate / up pairs and each wave half gets whole pairs
    const int unit = PAIR ? 2 : 1;
    int u0, nu;
    tile_span(a, tn, u0, nu);
    const int nt = unit * nu;
    const int tg0 = unit * u0;
    const int m0 = tm * kTM;
    const int nt0 = unit * ((nu + 1) >> 1);
    const int wn = w >> 2;
    const int ntw = wn == 0 ? nt0 : nt - nt0, jbase = wn == 0 ? 0 : nt0;
    using L = LinearFq8R<FX, FW, NB, PAIR, ABL>;
    QT_FQ8_STAMP_BEGIN
    bool redo;
    switch (ntw) {                                          // wave-uniform
        case 0: redo = L::template run<0>(a, lds_r, m0, tg0, nt, jbase, w, l); break;
        case 1: if constexpr (!PAIR) { redo = L::template run<1>(a, lds_r, m0, tg0, nt, jbase, w, l); break; }
        case 2: redo = L::template run<2>(a, lds_r, m0, tg0, nt, jbase, w, l); break;
        case 3: if constexpr (!PAIR) { redo = L::template run<3>(a, lds_r, m0, tg0, nt, jbase, w, l); break; }
        case 4: redo = L::template run<4>(a, lds_r, m0, tg0, nt, jbase, w, l); break;
        case 5: if constexpr (!PAIR) { redo = L::template run<5>(a, lds_r, m0, tg0, nt, jbase, w, l); break; }
        default: redo = L::template run<6>(a, lds_r, m0, tg0, nt, jbase, w, l); break;
    }
    if (redo) {
        if constexpr (PAIR) slow_tile_pair<FX, FW>(a, m0, tg0, jbase, ntw, w, l);
        else slow_tile<FX, FW>(a, m0, tg0, jbase, ntw, w, l);
    }
    QT_FQ8_STAMP_END
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

template <int FX, int FW, int NB, bool PAIR = false, int ABL = 0>
int launch_r_nb(const Args &a, hipStream_t st) {
    constexpr int kLds = LinearFq8R<FX, FW, NB, PAIR>::kLds;
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)linear_fq8r_kernel<FX, FW, NB, PAIR, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    linear_fq8r_kernel<FX, FW, NB, PAIR, ABL><<<a.tiles_m * a.tiles_n, 512, kLds, st>>>(a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int FX, int FW, int ABL = 0>
int launch_r2(const Args &a, hipStream_t st) {
    constexpr int kLds = LinearFq8R2<FX, FW>::kLds;
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)linear_fq8r2_kernel<FX, FW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    linear_fq8r2_kernel<FX, FW, ABL><<<a.tiles_m * a.tiles_n, 512, kLds, st>>>(a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int FX, int FW>
int launch(const Args &a, hipStream_t st) {
    // Two weight register sets (LinearFq8R::W2 = "ABL 20": a weight request has two steps to land) on the widest tiles, where it
    // measured 2 - 6 % faster.  Narrow tiles (nb <= 2, K a multiple of two k tiles) run variant R2, two k tiles per step.
    bool w2 = true, r2 = true;
#ifdef QT_TUNING_BUILD
    // tools/ only: timing ablations (results are garbage), the register-extended variant RX of R2 (exact, not faster) and A / B switches
    w2 = !(getenv("QT_FQ8_W2") && atoi(getenv("QT_FQ8_W2")) == 0);
    r2 = !(getenv("QT_FQ8_R2") && atoi(getenv("QT_FQ8_R2")) == 0);
#endif
    if (a.pair) {
        if (a.nb <= 2) return launch_r_nb<FX, FW, 2, true>(a, st);
        if (a.nb <= 4) return launch_r_nb<FX, FW, 4, true>(a, st);
        if (w2) return launch_r_nb<FX, FW, 6, true, 20>(a, st);
        return launch_r_nb<FX, FW, 6, true>(a, st);
    }
    if (a.nb <= 2 && a.K % (2 * kBK) == 0 && r2) {
#ifdef QT_TUNING_BUILD
        if constexpr (FX == 0 && FW == 0) {
            const char *e_ra = getenv("QT_FQ8_R2_ABLATE");
            switch (e_ra ? atoi(e_ra) : 0) {
                case 30: return launch_r2<0, 0, 30>(a, st);
                case 2: return launch_r2<0, 0, 2>(a, st);
                case 3: return launch_r2<0, 0, 3>(a, st);
                case 5: return launch_r2<0, 0, 5>(a, st);
                default: break;
            }
        }
#endif
        return launch_r2<FX, FW>(a, st);
    }
    if (a.nb <= 2) return launch_r_nb<FX, FW, 2>(a, st);
    if (a.nb <= 4) return launch_r_nb<FX, FW, 4>(a, st);
#ifdef QT_TUNING_BUILD
    if constexpr (FX == 0 && FW == 0) {
        const char *e_abl = getenv("QT_FQ8_ABLATE");
        switch (e_abl ? atoi(e_abl) : 0) {
            case 1: return launch_r_nb<0, 0, 6, false, 1>(a, st);
            case 2: return launch_r_nb<0, 0, 6, false, 2>(a, st);
            case 3: return launch_r_nb<0, 0, 6, false, 3>(a, st);
            case 4: return launch_r_nb<0, 0, 6, false, 4>(a, st);
            case 5: return launch_r_nb<0, 0, 6, false, 5>(a, st);     // no operand traffic: fragment reads + multiplications + barriers
            case 6: return launch_r_nb<0, 0, 6, false, 6>(a, st);     // barriers only
            case 7: return launch_r_nb<0, 0, 6, false, 7>(a, st);     // transport probe: weights by LDS-DMA (raw, unconverted)
            case 8: return launch_r_nb<0, 0, 6, false, 8>(a, st);     // the same without the activation DMA
            case 9: return launch_r_nb<0, 0, 6, false, 9>(a, st);     // bare transport: activation DMA only
            case 10: return launch_r_nb<0, 0, 6, false, 10>(a, st);   // bare transport: weight DMA only
            case 11: return launch_r_nb<0, 0, 6, false, 11>(a, st);   // bare transport: both by DMA
            default: break;
        }
    }
#endif
    if (w2) return launch_r_nb<FX, FW, 6, false, 20>(a, st);
    return launch_r_nb<FX, FW, 6>(a, st);
}

// How a problem is cut (host): column tiles as many as make whole rounds over the CUs (one 512-thread workgroup per CU), no wider than
// kMaxNT groups; the weights are treated as one concatenated [sum n][K] matrix, a tile may span two of them.
// (Round 5 built a split along K for the down projection -- 256 x 128 tiles over K / 2, fp32 partial sums through a workspace, ticket
// per tile -- and removed it: -6 % alone, nothing inside the window; the weight bytes per CU depend on the row tile only and the weight
// stream is what paces the loop.  profiles/r05_fq8_splitk.txt.)
struct Fq8Plan {
    int tiles_m, tiles_n, gbase, gextra, nb;
};

int plan_fq8(int M, long groups, Fq8Plan &p) {
    int force_tn = 0, max_nt = kMaxNT;
#ifdef QT_TUNING_BUILD
    {
        const char *e_tn = getenv("QT_FQ8_TILES_N"), *e_nt = getenv("QT_FQ8_MAX_NT");   // tools/ only
        force_tn = e_tn ? atoi(e_tn) : 0;
        max_nt = e_nt ? atoi(e_nt) : kMaxNT;
        if (max_nt < 1 || max_nt > kMaxNT) max_nt = kMaxNT;
    }
#endif
    const int cus = cu_count();
    p.tiles_m = (M + kTM - 1) / kTM;
    const long tn_min = (groups + max_nt - 1) / max_nt;
    const long rounds = (p.tiles_m * tn_min + cus - 1) / cus;
    long tn = rounds * cus / p.tiles_m;
    if (force_tn > 0) tn = force_tn;
    if (tn < tn_min) tn = tn_min;
    if (tn > groups) tn = groups;
    p.tiles_n = (int)tn;
    p.gbase = (int)(groups / tn);
    p.gextra = (int)(groups % tn);
    const int worst_nt = p.gbase + (p.gextra ? 1 : 0);
    if (worst_nt > kMaxNT) return QT_ERR_BAD_ARG;
    p.nb = (worst_nt * 4 + 7) / 8;
    return QT_OK;
}

// pair mode (qt_mlp_fq8_bf16): column tiles in gate / up pairs, at most six pairs (twelve column groups) each, whole rounds over the
// CUs; gbase / gextra count PAIRS
int plan_mlp(int M, long pairs, Fq8Plan &p) {
    const int cus = cu_count();
    p.tiles_m = (M + kTM - 1) / kTM;
    const long tn_min = (pairs + 5) / 6;
    const long rounds = (p.tiles_m * tn_min + cus - 1) / cus;
    long tn = rounds * cus / p.tiles_m;
    if (tn < tn_min) tn = tn_min;
    if (tn > pairs) tn = pairs;
    p.tiles_n = (int)tn;
    p.gbase = (int)(pairs / tn);
    p.gextra = (int)(pairs % tn);
    const int worst_nt = 2 * (p.gbase + (p.gextra ? 1 : 0));
    if (worst_nt > kMaxNT) return QT_ERR_BAD_ARG;
    p.nb = (worst_nt * 4 + 7) / 8;
    return QT_OK;
}

// which kernel launch<>() picks for a plan: 0 = R2 (two k tiles per step), 2 / 4 = R with that many weight pieces per wave,
// 6 = R with six pieces and two register sets per piece (the widest tiles)
int variant_of(const Fq8Plan &p, int K, bool pair) {
    if (!pair && p.nb <= 2 && K % (2 * kBK) == 0) return 0;
    return p.nb <= 2 ? 2 : (p.nb <= 4 ? 4 : 6);
}

}  // namespace

extern "C" {

int qt_linear_fq8_plan(int M, long n_total, int K, int pair, int *tiles_m, int *tiles_n, int *groups_lo, int *groups_hi, int *variant) {
    if (M < 1 || n_total < 16 || n_total % 16 != 0 || K < kBK || K % kBK != 0) return QT_ERR_BAD_ARG;
    Fq8Plan p;
    if (const int rc = pair ? plan_mlp(M, n_total / 16, p) : plan_fq8(M, n_total / 16, p)) return rc;
    const int unit = pair ? 2 : 1;
    if (tiles_m) *tiles_m = p.tiles_m;
    if (tiles_n) *tiles_n = p.tiles_n;
    if (groups_lo) *groups_lo = unit * p.gbase;
    if (groups_hi) *groups_hi = unit * (p.gbase + (p.gextra ? 1 : 0));
    if (variant) *variant = variant_of(p, K, pair != 0);
    return QT_OK;
}

int qt_linear_fq8_bf16(const uint8_t *x8_dev, int x_format, const uint16_t *const *w_devs, const uint16_t *const *bias_devs,
                       const int *ns, int count, int w_format, uint16_t *y_dev, int M, int K, void *stream) {
    if (count < 1 || count > kMaxSeg || !w_devs || !ns) return QT_ERR_BAD_ARG;
    if (x_format < 0 || x_format > 1 || w_format < 0 || w_format > 1) return QT_ERR_BAD_DTYPE;
    long ntot = 0, groups = 0;
    for (int i = 0; i < count; ++i) {
        if (ns[i] < 0 || ns[i] % 16 != 0) return QT_ERR_BAD_ARG;
        ntot += ns[i];
    }
    if ((long)M * ntot == 0) return QT_OK;
    if (!x8_dev || !y_dev || M < 0 || K < kBK || K % kBK != 0 || ntot > (1L << 30)) return QT_ERR_BAD_ARG;
    if (((uintptr_t)x8_dev & 15u) || ((uintptr_t)y_dev & 7u)) return QT_ERR_UNALIGNED;
    for (int i = 0; i < count; ++i) {
        if (ns[i] && (!w_devs[i] || ((uintptr_t)w_devs[i] & 15u))) return QT_ERR_UNALIGNED;
        if (bias_devs && bias_devs[i] && ((uintptr_t)bias_devs[i] & 7u)) return QT_ERR_UNALIGNED;
    }
    groups = ntot / 16;
    Fq8Plan p;
    if (const int rc = plan_fq8(M, groups, p)) return rc;
    Args a{};
    a.x8 = x8_dev; a.y = y_dev; a.M = M; a.K = K; a.ldc = (int)ntot;
    a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.gbase = p.gbase; a.gextra = p.gextra; a.nb = p.nb;
    a.dbg = 0;
#ifdef QT_TUNING_BUILD
    if (const char *e_dbg = getenv("QT_FQ8_DEBUG")) a.dbg = atoi(e_dbg);
    if (const char *e_st = getenv("QT_FQ8_STAMPS")) a.stamps = (unsigned long long *)strtoull(e_st, nullptr, 0);
#endif
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
    if (x_format == 0 && w_format == 0) return launch<0, 0>(a, st);
    if (x_format == 1 && w_format == 1) return launch<1, 1>(a, st);
    if (x_format == 0 && w_format == 1) return launch<0, 1>(a, st);
    if (x_format == 1 && w_format == 0) return launch<1, 0>(a, st);
    return QT_ERR_BAD_DTYPE;
}

int qt_mlp_fq8_bf16(const uint8_t *x8_dev, int x_format, const uint16_t *w_gate_dev, const uint16_t *w_up_dev, const uint16_t *bias_gate_dev,
                    const uint16_t *bias_up_dev, int N, int w_format, uint16_t *h_dev, uint8_t *h8_dev, const qt_format *out_format, int M, int K,
                    void *stream) {
    if (x_format < 0 || x_format > 1 || w_format < 0 || w_format > 1) return QT_ERR_BAD_DTYPE;
    if (!out_format || out_format->kind != QT_FMT_FP_SAT || !(out_format->p0 == 2 || out_format->p0 == 3)) return QT_ERR_BAD_DTYPE;
    if (N < 0 || N % 16 != 0) return QT_ERR_BAD_ARG;
    if ((long)M * N == 0) return QT_OK;
    if (!x8_dev || !w_gate_dev || !w_up_dev || !h8_dev || M < 0 || K < kBK || K % kBK != 0) return QT_ERR_BAD_ARG;       // h_dev NULL: codes only
    if (((uintptr_t)x8_dev | (uintptr_t)w_gate_dev | (uintptr_t)w_up_dev | (uintptr_t)h_dev | (uintptr_t)h8_dev) & 15u) return QT_ERR_UNALIGNED;
    if ((bias_gate_dev && ((uintptr_t)bias_gate_dev & 7u)) || (bias_up_dev && ((uintptr_t)bias_up_dev & 7u))) return QT_ERR_UNALIGNED;
    Args a{};
    a.x8 = x8_dev; a.y = h_dev; a.y8 = h8_dev; a.M = M; a.K = K; a.ldc = N; a.pair = 1; a.out_fmt = *out_format;
    Fq8Plan p;
    if (const int rc = plan_mlp(M, N / 16, p)) return rc;
    a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.gbase = p.gbase; a.gextra = p.gextra; a.nb = p.nb;
    a.dbg = 0;
#ifdef QT_TUNING_BUILD
    if (const char *e_dbg = getenv("QT_FQ8_DEBUG")) a.dbg = atoi(e_dbg);
    if (const char *e_st = getenv("QT_FQ8_STAMPS")) a.stamps = (unsigned long long *)strtoull(e_st, nullptr, 0);
#endif
    a.seg[0].w = w_gate_dev; a.seg[0].bias = bias_gate_dev; a.seg[0].g0 = 0;
    a.seg[1].w = w_up_dev; a.seg[1].bias = bias_up_dev; a.seg[1].g0 = 0;
    a.nseg = 2;
    hipStream_t st = (hipStream_t)stream;
    if (x_format == 0 && w_format == 0) return launch<0, 0>(a, st);
    if (x_format == 1 && w_format == 1) return launch<1, 1>(a, st);
    if (x_format == 0 && w_format == 1) return launch<0, 1>(a, st);
    return launch<1, 0>(a, st);
}

}  // extern "C"
