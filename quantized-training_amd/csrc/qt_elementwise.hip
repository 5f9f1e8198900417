// qt_elementwise.hip -- gfx950 kernels for the elementwise fake-quantization hot path.
//
// What the reference does per call (fake_quantize.py:217-248 -> decomposed.py:146-163):
//   amax = max|x|                      one full read
//   y = qmap[bits(x / s)] * s          Python loop over 65 536-element chunks, ~6 int32 temporaries
// Here one launch reads x ONCE (16 B per lane, coalesced), max-accumulates |x| for the observer
// (wavefront reduction -> one atomic per workgroup), rounds through either a closed form or the
// 65 536-entry value map staged in LDS (128 KiB of the CU's 160 KiB), and writes y once.
// Algorithmic traffic: 4 B/element bf16, 8 B/element fp32 -> HBM-bound; see DESIGN.md.
//
// Numerics contract (bit-exact against the reference CPU path):
//   bf16 tensors: q = bf16(f32(x) / f32(s_bf16)); r = map[q]; y = bf16(f32(r) * f32(s_bf16))
//   fp32 tensors: q = x / s (IEEE);  idx = hi16(q) | (lo16(q) != 0);  y = f32(map[idx]) * s
//   s == 1 skips both the division and the multiplication (x/1 and r*1 are exact).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "qt_device.h"
#include "qt_chain.h"

extern "C" float qt_internal_posit_threshold(int nbits, int es);
extern "C" int qt_internal_fp8_emin(float fp8_min);

namespace {

constexpr int kLutBlock = 1024;   // one workgroup per CU (128 KiB LDS), 16 waves
constexpr int kAluBlock = 256;
constexpr int kUnroll = 1;        // 16-B loads per lane per tile (measured best: many small tiles)
constexpr size_t kRowsDirectMaxVecs = (size_t)256 * 8 * 256 * 2;   // row form: up to two vectors per lane of a full chip read the table from global memory

// scalar element (tails, unaligned tensors)
template <int IO, int KIND, bool OBS>
__device__ __forceinline__ void fq_one(const void *x, void *y, size_t i, float s, bool unit, const Rounder<KIND> &rnd,
                                       uint32_t &amax) {
    if constexpr (IO == kIoBf16) {
        uint32_t img = (uint32_t)((const uint16_t *)x)[i] << 16;
        if constexpr (OBS) {
            uint32_t a = img & 0x7FFFFFFFu;
            amax = amax > a ? amax : a;
        }
        if (!y) return;
        if (!unit) img = pack_bf16x2(qt_u2f(img) / s, 0.0f) << 16;
        uint32_t r = rnd(img);
        ((uint16_t *)y)[i] = unit ? (uint16_t)(r >> 16) : (uint16_t)pack_bf16x2(qt_u2f(r) * s, 0.0f);
    } else {
        uint32_t w = ((const uint32_t *)x)[i];
        if constexpr (OBS) {
            uint32_t a = w & 0x7FFFFFFFu;
            amax = amax > a ? amax : a;
        }
        if (!y) return;
        float q = unit ? qt_u2f(w) : qt_u2f(w) / s;
        float r = qt_u2f(rnd(qt_fold_img(qt_f2u(q))));
        ((float *)y)[i] = unit ? r : r * s;
    }
}

template <int BLOCK>
__device__ __forceinline__ void block_amax_commit(uint32_t amax, uint32_t *out) {
    __shared__ uint32_t s_part[BLOCK / 64];
    amax = wave_max_u32(amax);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_part[wave] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = s_part[0];
#pragma unroll
        for (int i = 1; i < BLOCK / 64; ++i) m = m > s_part[i] ? m : s_part[i];
        // history[0] was zeroed by qt_scale_update.  Thousands of same-address atomics serialise (~12 ns each),
        // so a block only issues one when it can raise the running max (a stale read just costs an atomic).
        if (m != 0u && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
    }
}

__device__ __forceinline__ uint4 ld_vec(const uint4 *p, bool nt) {
    if (nt) {
        uint4 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        v.z = __builtin_nontemporal_load(&p->z);
        v.w = __builtin_nontemporal_load(&p->w);
        return v;
    }
    return *p;
}
__device__ __forceinline__ void st_vec(uint4 *p, uint4 v, bool nt) {
    if (nt) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z);
        __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

template <int IO, int KIND, int DIV, bool OBS, int BLOCK, int kUnroll, int NT>
__device__ __forceinline__ void fq_stream(const uint4 *__restrict__ x, uint4 *__restrict__ y, size_t nvec, const UniformDiv &dv,
                                          const Rounder<KIND> &rnd, uint32_t &amax) {
    constexpr size_t kTile = (size_t)BLOCK * kUnroll;
    const size_t nfull = nvec / kTile;
    for (size_t t = blockIdx.x; t < nfull; t += gridDim.x) {
        const size_t base = t * kTile + threadIdx.x;
        uint4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = ld_vec(x + base + (size_t)u * BLOCK, NT & 1);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            uint4 r = fq_vec<IO, KIND, DIV, OBS>(v[u], dv, rnd, amax);
            if (y) st_vec(y + base + (size_t)u * BLOCK, r, NT & 2);
        }
    }
    // ragged last tile: the block that would own it in the grid-stride order
    if (nfull % gridDim.x == blockIdx.x) {
        for (size_t i = nfull * kTile + threadIdx.x; i < nvec; i += BLOCK) {
            uint4 r = fq_vec<IO, KIND, DIV, OBS>(x[i], dv, rnd, amax);
            if (y) y[i] = r;
        }
    }
}

// Fused observe + fake-quantize, per-tensor scale.
//   x, y      : 16-B aligned base; nvec 16-B vectors followed by `ntail` scalar elements
//   scale     : fp32 scale on device (NULL = 1); cast to the input dtype like scale.to(X.dtype)
//   amax_out  : uint32 view of amax_history[0] (NULL when OBS is false)
template <int IO, int KIND, bool OBS, int BLOCK, int kUnroll = 1, int NT = 0>
__global__ __launch_bounds__(BLOCK) void fq_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t nvec,
                                                  size_t n, qt_format fmt, const uint16_t *__restrict__ lut,
                                                  const float *__restrict__ scale, uint32_t *amax_out) {
    Rounder<KIND> rnd{fmt, nullptr};
    if constexpr (KIND == QT_FMT_LUT || KIND == kFmtLutHalf) {
        constexpr int kVecs = (KIND == kFmtLutHalf ? QT_MAP_ENTRIES / 2 : QT_MAP_ENTRIES) * 2 / 16;   // 64 KiB: two workgroups per CU
        __shared__ uint4 s_lut[kVecs];
        if (yv) {
            const uint4 *g = (const uint4 *)lut;
#pragma unroll
            for (int i = 0; i < kVecs / BLOCK; ++i) s_lut[i * BLOCK + threadIdx.x] = g[i * BLOCK + threadIdx.x];
        }
        rnd.lds = (const uint16_t *)s_lut;
        __syncthreads();
    }
    if constexpr (KIND == kFmtRows) {
        // the row words sit behind the 65 536 map entries: 4 KiB (rows by exponent) or 8 KiB in LDS, many workgroups per CU
        __shared__ uint4 s_rows[512];
        const uint4 *g = (const uint4 *)(lut + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += BLOCK) s_rows[i] = g[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = lut;
        __syncthreads();
    }
    float s = scale ? *scale : 1.0f;
    if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
    const bool unit = (s == 1.0f);
    uint32_t amax = 0;
    const uint4 *x = (const uint4 *)xv;
    uint4 *y = (uint4 *)yv;
    const UniformDiv dv(s);
    if (unit)
        fq_stream<IO, KIND, kDivUnit, OBS, BLOCK, kUnroll, NT>(x, y, nvec, dv, rnd, amax);
    else if (dv.safe)
        fq_stream<IO, KIND, kDivFast, OBS, BLOCK, kUnroll, NT>(x, y, nvec, dv, rnd, amax);
    else
        fq_stream<IO, KIND, kDivExact, OBS, BLOCK, kUnroll, NT>(x, y, nvec, dv, rnd, amax);
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    if (blockIdx.x == gridDim.x - 1) {
        for (size_t i = nvec * kPer + threadIdx.x; i < n; i += BLOCK) fq_one<IO, KIND, OBS>(xv, yv, i, s, unit, rnd, amax);
    }
    if constexpr (OBS) block_amax_commit<BLOCK>(amax, amax_out);
}

// Element-granular variant with the table read from global memory (L1/L2-resident, 128 KiB):
// tensors that are unaligned or too small to amortise a 128 KiB LDS fill per workgroup.
template <int IO, int KIND, bool OBS>
__global__ __launch_bounds__(256) void fq_gather_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t n,
                                                       qt_format fmt, const uint16_t *__restrict__ lut,
                                                       const float *__restrict__ scale, uint32_t *amax_out) {
    Rounder<KIND> rnd{fmt, lut};
    float s = scale ? *scale : 1.0f;
    if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
    const bool unit = (s == 1.0f);
    uint32_t amax = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        fq_one<IO, KIND, OBS>(xv, yv, i, s, unit, rnd, amax);
    if constexpr (OBS) block_amax_commit<256>(amax, amax_out);
}

// The row form for SHORT passes (a lane has one or two vectors: the [2048, 768 .. 3072] activations and gradients of a training step):
// the 4 - 8 KiB row table is read where it lies, in global memory behind the map (L1 / L2 hits), eight gathers in flight per vector --
// no staging into LDS and no barrier in front of the first load.  An OBSERVED short pass runs 1024-thread workgroups: every workgroup
// that can raise the running maximum issues an atomicMax on ONE address, the slot starts each step at zero, and same-address atomics
// serialise at ~12 ns each -- 768 workgroups of 256 threads spent most of a [2048, 768] launch's 10.9 us queueing there.
template <int IO, bool OBS, int BLOCK, int UNR = 1>
__global__ __launch_bounds__(BLOCK) void fq_rows_direct_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t nvec, size_t n,
                                                            qt_format fmt, const uint16_t *__restrict__ lut, const float *__restrict__ scale,
                                                            uint32_t *amax_out) {
    Rounder<kFmtRows> rnd{fmt, lut + QT_MAP_ENTRIES, lut};
    float s = scale ? *scale : 1.0f;
    if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
    const bool unit = (s == 1.0f);
    uint32_t amax = 0;
    const uint4 *x = (const uint4 *)xv;
    uint4 *y = (uint4 *)yv;
    const UniformDiv dv(s);
    if (unit)
        fq_stream<IO, kFmtRows, kDivUnit, OBS, BLOCK, UNR, 0>(x, y, nvec, dv, rnd, amax);
    else if (dv.safe)
        fq_stream<IO, kFmtRows, kDivFast, OBS, BLOCK, UNR, 0>(x, y, nvec, dv, rnd, amax);
    else
        fq_stream<IO, kFmtRows, kDivExact, OBS, BLOCK, UNR, 0>(x, y, nvec, dv, rnd, amax);
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    if (blockIdx.x == gridDim.x - 1) {
        for (size_t i = nvec * kPer + threadIdx.x; i < n; i += BLOCK) fq_one<IO, kFmtRows, OBS>(xv, yv, i, s, unit, rnd, amax);
    }
    if constexpr (OBS) block_amax_commit<BLOCK>(amax, amax_out);
}

// Vector-granular variant of the same (aligned tensors too small for the LDS table: a [2048, 768] gradient is 1.5 M
// elements): 16-B loads and stores, eight gathers per lane from the L2-resident table.  The element-granular kernel
// above moves 2 B per lane per access -- 12.4 us for such a gradient; this one is bandwidth-shaped again.
template <int IO, bool OBS>
__global__ __launch_bounds__(256) void fq_gather_vec_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t nvec,
                                                           size_t n, qt_format fmt, const uint16_t *__restrict__ lut,
                                                           const float *__restrict__ scale, uint32_t *amax_out) {
    Rounder<QT_FMT_LUT> rnd{fmt, lut};
    float s = scale ? *scale : 1.0f;
    if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
    const bool unit = (s == 1.0f);
    uint32_t amax = 0;
    const uint4 *x = (const uint4 *)xv;
    uint4 *y = (uint4 *)yv;
    const UniformDiv dv(s);
    if (unit)
        fq_stream<IO, QT_FMT_LUT, kDivUnit, OBS, 256, 1, 0>(x, y, nvec, dv, rnd, amax);
    else if (dv.safe)
        fq_stream<IO, QT_FMT_LUT, kDivFast, OBS, 256, 1, 0>(x, y, nvec, dv, rnd, amax);
    else
        fq_stream<IO, QT_FMT_LUT, kDivExact, OBS, 256, 1, 0>(x, y, nvec, dv, rnd, amax);
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    if (blockIdx.x == gridDim.x - 1) {
        for (size_t i = nvec * kPer + threadIdx.x; i < n; i += 256) fq_one<IO, QT_FMT_LUT, OBS>(xv, yv, i, s, unit, rnd, amax);
    }
    if constexpr (OBS) block_amax_commit<256>(amax, amax_out);
}

// ---- several fake-quantizers of one training step over ONE tensor, as one launch (qt_fake_quant_chain_bf16) -----------------------------
// The reference's hooks call fake-quantizers back to back on the same tensor (quantize.py:116-179): a gradient leaving a LayerNorm
// goes through the residual add's backward-pre quantizer, its two backward quantizers and the dense layer's backward-pre quantizer
// -- four observed E5M2 passes over [tokens, hidden]; a LayerNorm's output goes through the input quantizers of query / key / value.
// Inside a replayed step every such pass is a launch of its own (>= 4.5 us each, 8 us observed).  Here stage i reads x or the result
// of an earlier stage, applies ITS fake-quantizer (own scale, own amax slot: the per-tensor state machines stay the reference's) and
// writes its result; optionally the fp32 column sums of one stage's result are produced on the way (the bias gradient of the Linear
// behind it: grad_output.sum(0), run_glue_no_trainer.py:660-667) -- rows are dealt to workgroups in bands, a lane adds its rows in
// row order, the lanes of a column in lane order, the workgroups in workgroup order (last arriver, through a workspace): deterministic.
// Tuning build only (tools/exp_train_stamps.py): s_memrealtime (100 MHz, one counter for the whole chip) of wave 0 and of the last wave
// of a workgroup at a few points of the chain / LayerNorm-backward kernels.  QT_EW_STAMPS = device address of [launch][512 workgroups][32]
// slots; every launch the host issues takes the next region (a captured graph keeps the region of its capture).
#ifdef QT_TUNING_BUILD
#define QT_EW_STAMP_FIELD unsigned long long *dbg; int ablate;
#define QT_EW_STAMP(a, slot)                                                                                                     \
    do {                                                                                                                         \
        if ((a).dbg && blockIdx.x < 512 && (threadIdx.x & 63) == 0 && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64))      \
            (a).dbg[(size_t)blockIdx.x * 32 + (threadIdx.x ? 16 : 0) + (slot)] = __builtin_amdgcn_s_memrealtime();                \
    } while (0)
#define QT_EW_STAMP_HEAD(a, tag)                                                                                                 \
    do {                                                                                                                         \
        if ((a).dbg && blockIdx.x < 512 && threadIdx.x == 0) (a).dbg[(size_t)blockIdx.x * 32 + 7] = ((unsigned long long)(tag) << 32) | gridDim.x; \
    } while (0)
static unsigned long long *ew_stamp_region() {
    const char *e = getenv("QT_EW_STAMPS");
    if (!e) return nullptr;
    static int launch = 0;
    return (unsigned long long *)strtoull(e, nullptr, 0) + (size_t)(launch++ % 256) * 512 * 32;
}
#else
#define QT_EW_STAMP_FIELD
#define QT_EW_STAMP(a, slot)
#define QT_EW_STAMP_HEAD(a, tag)
#endif
constexpr int kChainBlock = 512, kChainStripV = 8, kChainRowLanes = kChainBlock / kChainStripV;
// Work decomposition: a workgroup (512 threads: up to 256 registers, the row form keeps many live) owns a STRIP of 64 columns (8 vectors =
// one 128-byte line per row) x a BAND of rows; lane (rl = t / 8, v = t % 8) walks rows rl, rl + 64, ... of the band, two or four loads in flight.  Column sums: a lane adds its rows in row order,
// the 64 row lanes of a column meet in a fixed-order tree in LDS, and the bands of a strip meet in fixed-point accumulators (below).
struct ChainArgs {
    const uint4 *x;
    long rows;
    int cv;                   // 16-byte vectors per row
    int nstage;
    ChainStageDev st[kChainMax];
    int strips, bands;
    long band_rows;           // rows per band (a multiple of kChainRowLanes)
    int colsum_stage;         // -1: none
    float colsum_max;         // largest magnitude the format produces at scale 1 (bounds the fixed-point range)
    long long *acc;           // [strip][64] fixed-point column sums, zero between launches
    unsigned int *ticket;     // [strip]
    uint16_t *colsum_out;     // [cols] bf16
    // prologue: the value the stages see is not x itself but an elementwise function of it, computed in fp32 and rounded to bf16 as the
    // torch kernel it replaces does -- 1: erf GELU of x (BertIntermediate's activation in front of the output dense's input quantizer);
    // 2: its backward, x = grad_output, x2 = the forward's input: dy * (Phi(z) + z phi(z)) (in front of the intermediate dense's
    // backward-pre quantizer).  pre_out (nullable): where that value goes.
    int pre_op;
    const uint4 *x2;
    uint4 *pre_out;
    int table_in_lds;         // table formats: stage the row words in LDS (else gather them from global memory)
    QT_EW_STAMP_FIELD
};

__device__ __forceinline__ float gelu_erf_f(float x) { return (x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float dy, float x) {
    // torch's GeluBackward (erf form): cdf = 0.5 (1 + erf(x / sqrt 2)), pdf = exp(-x^2 / 2) / sqrt(2 pi)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
    return dy * (cdf + x * pdf);
}
__device__ __forceinline__ uint4 chain_prologue(int op, uint4 q, uint4 q2) {
    if (op == 0) return q;
    const uint32_t a[4] = {q.x, q.y, q.z, q.w}, b[4] = {q2.x, q2.y, q2.z, q2.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = qt_u2f(a[j] << 16), a1 = qt_u2f(a[j] & 0xFFFF0000u);
        if (op == 1) o[j] = pack_bf16x2(gelu_erf_f(a0), gelu_erf_f(a1));
        else o[j] = pack_bf16x2(gelu_erf_grad_f(a0, qt_u2f(b[j] << 16)), gelu_erf_grad_f(a1, qt_u2f(b[j] & 0xFFFF0000u)));
    }
    return uint4{o[0], o[1], o[2], o[3]};
}

template <int KIND, int NS>
__global__ __launch_bounds__(kChainBlock) void fq_chain_kernel(ChainArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    QT_EW_STAMP(a, 0);
    QT_EW_STAMP_HEAD(a, 0x100 + NS * 16 + (a.colsum_stage >= 0 ? 1 : 0) + (a.pre_op << 1));
    // (a run-time choice of where the row table lies makes its pointer generic: every row gather becomes a flat_load.  The product build
    // always stages the table; the tuning build keeps the switch.)
#ifdef QT_TUNING_BUILD
    const bool rows_in_lds = a.table_in_lds != 0;
#else
    constexpr bool rows_in_lds = true;
#endif
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, kChainBlock, rows_in_lds);
    QT_EW_STAMP(a, 1);
    const int t = threadIdx.x;
    const int v = t % kChainStripV, rl = t / kChainStripV;
    const int strip = blockIdx.x % a.strips, band = blockIdx.x / a.strips;
    const int cvec = strip * kChainStripV + v;
    const bool live = cvec < a.cv;
    float sc[NS];
    uint32_t amax[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        sc[i] = a.st[i].scale ? qt_bf2f(qt_f2bf(*a.st[i].scale)) : 1.0f;       // scale.to(X.dtype), as the single passes do
        amax[i] = 0u;
    }
    float col[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long r_begin = (long)band * a.band_rows, r_end = min(a.rows, r_begin + a.band_rows);
    constexpr int kChainUnroll = KIND == kFmtRows ? 2 : 4;      // rows in flight per lane (the row form keeps eight table rows live per vector)
    if (live) {
        for (long r0 = r_begin + rl; r0 < r_end; r0 += (long)kChainRowLanes * kChainUnroll) {
            uint4 q[kChainUnroll], q2[kChainUnroll];
#pragma unroll
            for (int u = 0; u < kChainUnroll; ++u) {
                const long r = r0 + (long)u * kChainRowLanes;
                if (r < r_end) {
                    q[u] = a.x[r * a.cv + cvec];
                    if (a.pre_op == 2) q2[u] = a.x2[r * a.cv + cvec];
                }
            }
#ifdef QT_TUNING_BUILD
            if (a.dbg) {                                             // (diagnosis: the first two rounds split into load wait and the rest)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (r0 == r_begin + rl) QT_EW_STAMP(a, 8);
                else if (r0 == r_begin + rl + (long)kChainRowLanes * kChainUnroll) QT_EW_STAMP(a, 10);
            }
#endif
#pragma unroll
            for (int u = 0; u < kChainUnroll; ++u) {
                const long r = r0 + (long)u * kChainRowLanes;
                if (r >= r_end) continue;
                const size_t idx = (size_t)(r * a.cv + cvec);
#ifdef QT_TUNING_BUILD
                // QT_CHAIN_ABLATE (timing only, results are garbage): 1 = no GELU arithmetic, 2 = no fake-quantizer stages
                const uint4 val = (a.ablate & 1) ? q[u] : chain_prologue(a.pre_op, q[u], q2[u]);
                if (a.pre_op && a.pre_out) a.pre_out[idx] = val;
                uint4 res[NS];
                if (a.ablate & 2) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        res[i] = val;
                        if (a.st[i].out) a.st[i].out[idx] = val;
                    }
                } else {
                    chain_stages<KIND, NS>(a.st, sc, rnd, val, idx, amax, res);
                }
#else
                const uint4 val = chain_prologue(a.pre_op, q[u], q2[u]);
                if (a.pre_op && a.pre_out) a.pre_out[idx] = val;
                uint4 res[NS];
                chain_stages<KIND, NS>(a.st, sc, rnd, val, idx, amax, res);
#endif
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    if (a.colsum_stage == i) {
                        col[0] += qt_u2f(res[i].x << 16); col[1] += qt_u2f(res[i].x & 0xFFFF0000u);
                        col[2] += qt_u2f(res[i].y << 16); col[3] += qt_u2f(res[i].y & 0xFFFF0000u);
                        col[4] += qt_u2f(res[i].z << 16); col[5] += qt_u2f(res[i].z & 0xFFFF0000u);
                        col[6] += qt_u2f(res[i].w << 16); col[7] += qt_u2f(res[i].w & 0xFFFF0000u);
                    }
                }
            }
#ifdef QT_TUNING_BUILD
            if (a.dbg) {
                if (r0 == r_begin + rl) QT_EW_STAMP(a, 9);
                else if (r0 == r_begin + rl + (long)kChainRowLanes * kChainUnroll) QT_EW_STAMP(a, 11);
            }
#endif
        }
    }
    // ---- amax of every stage's input: wave, then workgroup, then at most one atomic per stage and workgroup
    __shared__ uint32_t s_amax[NS][kChainBlock / 64];
    __shared__ float s_col[kChainRowLanes][kChainStripV * 8 + 1];
    QT_EW_STAMP(a, 2);
    if (a.colsum_stage >= 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_col[rl][v * 8 + e] = col[e];
    }
    chain_amax_commit<NS, kChainBlock>(a.st, amax, s_amax);          // (its barrier also covers s_col)
    QT_EW_STAMP(a, 3);
    if (a.colsum_stage < 0) return;
    // ---- column sums: the row lanes of a column in a fixed-order tree; then the bands of a strip meet in 64-bit FIXED-POINT accumulators
    // by agent-scope atomic adds -- integer addition is associative, so the result does not depend on the arrival order, and atomics are
    // coherent across the XCDs' L2s without the release fence a plain hand-off would need (on this part that fence writes back the whole
    // L2: + 30 us behind 12 MB of results, measured).  The band that draws the strip's last ticket reads the sums (atomically), rounds
    // them to bf16 and leaves the accumulators zero.  Fixed point (fx_plan, qt_chain.h): one unit = 2^(E - 42) up to 8192 rows, E = exponent of the largest value the stage
    // can produce (format maximum x scale): a band's fp32 partial sum converts exactly unless it is below 2^-42 of that.
    constexpr int kC = kChainStripV * 8;
    for (int half = kChainRowLanes / 2; half >= 1; half >>= 1) {
        for (int i = t; i < half * kC; i += kChainBlock) {
            const int r = i / kC, c = i % kC;
            s_col[r][c] += s_col[r + half][c];
        }
        lds_only_barrier();                                         // (six rounds: none of them waits for the result stores in flight)
    }
    QT_EW_STAMP(a, 4);
    float s_cs = 1.0f;
#pragma unroll
    for (int i = 0; i < NS; ++i)
        if (a.colsum_stage == i) s_cs = sc[i];
    int E;
    (void)frexpf(a.colsum_max * s_cs, &E);                          // max value < 2^E
    long long *acc = a.acc + (size_t)strip * kC;
    const FxPlan fxp = fx_plan(a.rows, a.bands);
    if (a.bands == 1) {
        const int c = strip * kC + t;
        if (t < kC && c < a.cv * 8) a.colsum_out[c] = (uint16_t)(pack_bf16x2(s_col[0][t], 0.0f) & 0xFFFFu);
        return;
    }
    if (t < kC) {
        const float part = s_col[0][t];
        const long long fx = fx_encode(part, E, fxp);                // NaN / Inf: the poison term (qt_chain.h)
        (void)__hip_atomic_fetch_add(acc + t, fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // returning form: performed when it returns
    }
    __syncthreads();
    QT_EW_STAMP(a, 5);
    __shared__ unsigned int s_old;
    if (t == 0) s_old = __hip_atomic_fetch_add(a.ticket + strip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    QT_EW_STAMP(a, 6);
    if (s_old != (unsigned)a.bands - 1u) return;
    if (t == 0) __hip_atomic_store(a.ticket + strip, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < kC) {
        const int c = strip * kC + t;
        const long long fx = __hip_atomic_exchange(acc + t, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read and re-zero
        const float sum = fx_decode(fx, E, fxp);
        if (c < a.cv * 8) a.colsum_out[c] = (uint16_t)(pack_bf16x2(sum, 0.0f) & 0xFFFFu);
    }
}

// ---- LayerNorm of a training step with the fake-quantizer calls that follow it (qt_layernorm_train_*) -----------------------------------
// Forward: y = LayerNorm(x) (torch's rounding points: fp32 statistics, one rounding to bf16), mean / rstd kept for the backward, and the
// input quantizers of the Linears that read y (query / key / value, or the intermediate dense: quantize.py:128-140) evaluated on the row
// while it is in registers.  Backward: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma (torch's layer_norm_grad_input), the
// gradient chain behind it (the residual add's three quantizers and the dense layer's, quantize.py:116-179) on dx in registers, and
// per-workgroup partial sums of dgamma = sum dy xhat, dbeta = sum dy and the bias gradient (column sums of one stage's result), which a
// second small launch adds in workgroup order (qt_layernorm_train_reduce).  One wave per row, rows up to 1024 columns.
constexpr int kLnMaxVec = 2;
struct LnTrainArgs {
    const uint4 *x, *w, *b;
    const uint4 *x2;          // optional: the residual connection -- the row normalised is bf16(x + x2) (torch's add), written to `sum`
    uint4 *sum;
    uint4 *y;
    float *mean, *rstd;
    long rows;
    int nvec;
    float inv_cols, eps;
    ChainStageDev st[kChainMax];
};

template <int KIND, int NS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ln_train_fwd_kernel(LnTrainArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, BLOCK);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * (BLOCK / 64) + wave;
    float sc[NS];
    uint32_t amax[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        sc[i] = a.st[i].scale ? qt_bf2f(qt_f2bf(*a.st[i].scale)) : 1.0f;
        amax[i] = 0u;
    }
    if (row < a.rows) {
        const size_t base = (size_t)row * (size_t)a.nvec;
        uint4 v[kLnMaxVec];
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < kLnMaxVec; ++i) {
            const int c = lane + i * 64;
            if (c < a.nvec) {
                v[i] = a.x[base + c];
                if (a.x2) {                                        // (uniform)
                    const uint4 r = a.x2[base + c];
                    v[i] = uint4{pack_bf16x2(bf_lo(v[i].x) + bf_lo(r.x), bf_hi(v[i].x) + bf_hi(r.x)), pack_bf16x2(bf_lo(v[i].y) + bf_lo(r.y), bf_hi(v[i].y) + bf_hi(r.y)),
                                 pack_bf16x2(bf_lo(v[i].z) + bf_lo(r.z), bf_hi(v[i].z) + bf_hi(r.z)), pack_bf16x2(bf_lo(v[i].w) + bf_lo(r.w), bf_hi(v[i].w) + bf_hi(r.w))};
                    a.sum[base + c] = v[i];
                }
                sum += (bf_lo(v[i].x) + bf_hi(v[i].x)) + (bf_lo(v[i].y) + bf_hi(v[i].y)) + (bf_lo(v[i].z) + bf_hi(v[i].z)) + (bf_lo(v[i].w) + bf_hi(v[i].w));
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum * a.inv_cols;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < kLnMaxVec; ++i) {
            const int c = lane + i * 64;
            if (c < a.nvec) {
                const uint32_t q[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d0 = bf_lo(q[j]) - mean, d1 = bf_hi(q[j]) - mean;
                    sq += d0 * d0;
                    sq += d1 * d1;
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = rsqrtf(sq * a.inv_cols + a.eps);
        if (lane == 0) {
            a.mean[row] = mean;
            a.rstd[row] = rstd;
        }
#pragma unroll
        for (int i = 0; i < kLnMaxVec; ++i) {
            const int c = lane + i * 64;
            if (c < a.nvec) {
                const uint4 ww = a.w[c], bb = a.b[c];
                const uint32_t q[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, g[4] = {ww.x, ww.y, ww.z, ww.w}, h[4] = {bb.x, bb.y, bb.z, bb.w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf16x2(bf_lo(g[j]) * (rstd * (bf_lo(q[j]) - mean)) + bf_lo(h[j]), bf_hi(g[j]) * (rstd * (bf_hi(q[j]) - mean)) + bf_hi(h[j]));
                const uint4 yv = uint4{o[0], o[1], o[2], o[3]};
                a.y[base + c] = yv;
                uint4 res[NS];
                chain_stages<KIND, NS>(a.st, sc, rnd, yv, base + c, amax, res);
            }
        }
    }
    __shared__ uint32_t s_amax[NS][BLOCK / 64];
    chain_amax_commit<NS, BLOCK>(a.st, amax, s_amax);
}

struct LnBwdArgs {
    const uint4 *dy, *x, *w;
    const float *mean, *rstd;
    uint4 *dx;
    long rows;
    int nvec;
    float inv_cols;
    int rows_per_wave;        // a wave walks rows wave, wave + BLOCK / 64, ... of its workgroup's band
    float *part;              // [workgroup][3][cols] fp32: dgamma, dbeta, column sums of stage `colsum_stage`
    int colsum_stage;         // -1: none (that slab is not written)
    ChainStageDev st[kChainMax];
    // gradients that arrive for the LayerNorm's result next to dy (qt_grad_fanin_bf16's sum, formed while dy is loaded):
    // dy := (((dy + y_0) + y_1) + y_2), y_i = fq_i(fan_x[i]) (fan[i].scale / .amax; .src != 0) or fan_x[i] itself, every add in bf16
    int nfan;
    const uint4 *fan_x[3];
    ChainStageDev fan[kChainMax];   // [0..2]: scale, amax; out unused; src: 1 = through the fake-quantizer, 0 = plain
    QT_EW_STAMP_FIELD
};

template <int KIND, int NS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ln_train_bwd_kernel(LnBwdArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    constexpr int RPB = BLOCK / 64;
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    QT_EW_STAMP(a, 0);
    QT_EW_STAMP_HEAD(a, 0x200 + NS * 16 + a.nfan * 2 + (a.colsum_stage >= 0 ? 1 : 0));
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, BLOCK);
    QT_EW_STAMP(a, 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sc[NS];
    uint32_t amax[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        sc[i] = a.st[i].scale ? qt_bf2f(qt_f2bf(*a.st[i].scale)) : 1.0f;
        amax[i] = 0u;
    }
    float fsc[3];
    uint32_t famax[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fsc[i] = (i < a.nfan && a.fan[i].src && a.fan[i].scale) ? qt_bf2f(qt_f2bf(*a.fan[i].scale)) : 1.0f;
        famax[i] = 0u;
    }
    // One row per wave (qt_layernorm_train_backward_groups).  The row's dy, x^ and the column-sum stage's result stay in registers until
    // the waves' column partials meet in LDS below: rounds 4-5 also carried three loop accumulators for "rows_per_wave" rows that the host
    // never asked for -- 48 registers that pushed the four-stage kernel to 256 VGPRs and 18 spilled ones (scratch traffic in the hot loop).
    float xh[kLnMaxVec][8], dyv[kLnMaxVec][8], csv[kLnMaxVec][8];
#pragma unroll
    for (int i = 0; i < kLnMaxVec; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[i][e] = dyv[i][e] = csv[i][e] = 0.0f;
    uint4 gam[kLnMaxVec];
#pragma unroll
    for (int i = 0; i < kLnMaxVec; ++i)
        if (lane + i * 64 < a.nvec) gam[i] = a.w[lane + i * 64];
    const long row = (long)blockIdx.x * RPB + wave;
    if (row < a.rows) {                                            // (wave-uniform)
        const size_t base = (size_t)row * (size_t)a.nvec;
        const float mean = a.mean[row], rstd = a.rstd[row];
        float g[kLnMaxVec][8];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < kLnMaxVec; ++i) {
            const int c = lane + i * 64;
            if (c < a.nvec) {
                uint4 d = a.dy[base + c];
                const uint4 x = a.x[base + c];
                if (a.nfan > 0) {                                  // (uniform) the other arrivals, in the engine's order
                    uint4 fx[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (t < a.nfan) fx[t] = a.fan_x[t][base + c];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        if (t >= a.nfan) break;
                        uint4 y = fx[t];
                        if (a.fan[t].src) {
                            const UniformDiv fdv(fsc[t]);
                            y = chain_apply<KIND>(fx[t], fsc[t], fdv, rnd, famax[t]);
                        }
                        const uint32_t aw[4] = {d.x, d.y, d.z, d.w}, bw[4] = {y.x, y.y, y.z, y.w};
                        d = uint4{pack_bf16x2(bf_lo(aw[0]) + bf_lo(bw[0]), bf_hi(aw[0]) + bf_hi(bw[0])),
                                  pack_bf16x2(bf_lo(aw[1]) + bf_lo(bw[1]), bf_hi(aw[1]) + bf_hi(bw[1])),
                                  pack_bf16x2(bf_lo(aw[2]) + bf_lo(bw[2]), bf_hi(aw[2]) + bf_hi(bw[2])),
                                  pack_bf16x2(bf_lo(aw[3]) + bf_lo(bw[3]), bf_hi(aw[3]) + bf_hi(bw[3]))};
                    }
                }
                const uint32_t dw[4] = {d.x, d.y, d.z, d.w}, xw[4] = {x.x, x.y, x.z, x.w}, gw[4] = {gam[i].x, gam[i].y, gam[i].z, gam[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float dyf = h ? bf_hi(dw[j]) : bf_lo(dw[j]), xf = h ? bf_hi(xw[j]) : bf_lo(xw[j]), gf = h ? bf_hi(gw[j]) : bf_lo(gw[j]);
                        const int e = 2 * j + h;
                        dyv[i][e] = dyf;
                        xh[i][e] = (xf - mean) * rstd;
                        g[i][e] = dyf * gf;
                        s1 += g[i][e];
                        s2 += g[i][e] * xh[i][e];
                    }
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        const float c1 = s1 * a.inv_cols, c2 = s2 * a.inv_cols;
        QT_EW_STAMP(a, 2);
#pragma unroll
        for (int i = 0; i < kLnMaxVec; ++i) {
            const int c = lane + i * 64;
            if (c < a.nvec) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf16x2(rstd * (g[i][2 * j] - c1 - xh[i][2 * j] * c2), rstd * (g[i][2 * j + 1] - c1 - xh[i][2 * j + 1] * c2));
                const uint4 dxv = uint4{o[0], o[1], o[2], o[3]};
                a.dx[base + c] = dxv;
                uint4 res[NS];
                chain_stages<KIND, NS>(a.st, sc, rnd, dxv, base + c, amax, res);
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) {
                    if (a.colsum_stage == s_) {
                        const uint32_t rw[4] = {res[s_].x, res[s_].y, res[s_].z, res[s_].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            csv[i][2 * j] = bf_lo(rw[j]);
                            csv[i][2 * j + 1] = bf_hi(rw[j]);
                        }
                    }
                }
            }
        }
    }
    QT_EW_STAMP(a, 3);
    // ---- the waves' column partials meet in LDS, quantity by quantity, and are added in wave order
    __shared__ float s_red[RPB][kLnMaxVec * 64 * 8];
    __shared__ uint32_t s_amax[NS][BLOCK / 64];
    const int cols = a.nvec * 8;
    float *const mine = a.part + (size_t)blockIdx.x * 3 * cols;
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
        if (qn == 2 && a.colsum_stage < 0) break;
#pragma unroll
        for (int i = 0; i < kLnMaxVec; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s_red[wave][(lane + i * 64) * 8 + e] = qn == 0 ? dyv[i][e] * xh[i][e] : (qn == 1 ? dyv[i][e] : csv[i][e]);
        lds_only_barrier();                                         // (not __syncthreads: the row's result stores stay in flight)
        for (int c = threadIdx.x; c < cols; c += BLOCK) {
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < RPB; ++w) sum += s_red[w][c];
            mine[(size_t)qn * cols + c] = sum;
        }
        lds_only_barrier();
    }
    QT_EW_STAMP(a, 4);
    chain_amax_commit<NS, BLOCK>(a.st, amax, s_amax);
    if (a.nfan > 0) {                                              // (uniform) the amax slots of the arrivals' fake-quantizers
        __syncthreads();                                           // s_amax is read by the commit above
        __shared__ uint32_t s_famax[3][BLOCK / 64];
        chain_amax_commit<3, BLOCK>(a.fan, famax, s_famax);
    }
    QT_EW_STAMP(a, 5);
}

// dgamma, dbeta (and the bias gradient) of one LayerNorm backward: the workgroups' partial sums added in a fixed order -- a workgroup owns
// 64 columns of one quantity, its four waves each add a quarter of the partials (eight loads in flight), the quarters meet in wave order
__global__ __launch_bounds__(256) void ln_train_reduce_kernel(const float *__restrict__ part, int nwg, int cols, uint16_t *dgamma, uint16_t *dbeta,
                                                              uint16_t *colsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, qn = blockIdx.y;
    uint16_t *out = qn == 0 ? dgamma : (qn == 1 ? dbeta : colsum);
    __shared__ float s_q[4][64];
    float sum = 0.0f;
    if (c < cols && out) {
        const int per = (nwg + 3) / 4, w0 = wave * per, w1 = min(nwg, w0 + per);
        const float *p = part + (size_t)qn * cols + c;
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(w + u) * 3 * cols];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
        for (; w < w1; ++w) sum += p[(size_t)w * 3 * cols];
    }
    s_q[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && c < cols && out) out[c] = (uint16_t)(pack_bf16x2((s_q[0][lane] + s_q[1][lane]) + (s_q[2][lane] + s_q[3][lane]), 0.0f) & 0xFFFFu);
}

// ---- OCP FP8 side output ------------------------------------------------------------------------
// For e4m3 / e5m2 every fake-quantized value q = map[x / s] is exactly representable as an OCP FP8
// byte (gfx950's v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 convert an on-grid value exactly), so the pass
// can hand the GEMM one byte per element: y8 = fp8(q), and optionally the usual y = bf16(q * s).
// (q, s) is what the reference's converted graphs feed to the GEMM: quantize -> GEMM ->
// dequantize(s_x * s_w), quantize_pt2e.py:323-446.
// Quantise one 16-B vector of bf16 to eight fp32 images of q = map[x / s].
__device__ __forceinline__ void fq8_vec(const uint4 v, const qt_format &fmt, const UniformDiv &dv, bool unit, bool obs,
                                        uint32_t &amax, float (&q)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t img[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        img[2 * j] = w[j] << 16;
        img[2 * j + 1] = w[j] & 0xFFFF0000u;
        if (obs) {
            uint32_t a0 = img[2 * j] & 0x7FFFFFFFu, a1 = img[2 * j + 1] & 0x7FFFFFFFu;
            amax = amax > a0 ? amax : a0;
            amax = amax > a1 ? amax : a1;
        }
    }
    if (!unit) {
        bool bad = !dv.safe;
        uint32_t qd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            qd[j] = pack_bf16x2(dv.fast16(qt_u2f(img[2 * j]), bad), dv.fast16(qt_u2f(img[2 * j + 1]), bad));
        if (__builtin_expect(bad, 0)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qd[j] = pack_bf16x2(dv.exact(qt_u2f(img[2 * j])), dv.exact(qt_u2f(img[2 * j + 1])));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            img[2 * j] = qd[j] << 16;
            img[2 * j + 1] = qd[j] & 0xFFFF0000u;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = qt_u2f(qt_fp_sat_u32(img[j], fmt.p0, fmt.p1, fmt.fhi));
}

__device__ __forceinline__ uint4 bf16_vec_from(const float (&q)[8], float s, bool unit) {
    uint4 o;
    if (unit) {
        o.x = (qt_f2u(q[0]) >> 16) | (qt_f2u(q[1]) & 0xFFFF0000u);
        o.y = (qt_f2u(q[2]) >> 16) | (qt_f2u(q[3]) & 0xFFFF0000u);
        o.z = (qt_f2u(q[4]) >> 16) | (qt_f2u(q[5]) & 0xFFFF0000u);
        o.w = (qt_f2u(q[6]) >> 16) | (qt_f2u(q[7]) & 0xFFFF0000u);
    } else {
        o.x = pack_bf16x2(q[0] * s, q[1] * s);
        o.y = pack_bf16x2(q[2] * s, q[3] * s);
        o.z = pack_bf16x2(q[4] * s, q[5] * s);
        o.w = pack_bf16x2(q[6] * s, q[7] * s);
    }
    return o;
}

// FP8-only output at unit scale (the weight pass of the FP8 GEMM route): for a FINITE input the format's saturating
// round-to-nearest-even is the hardware conversion of the value clamped to +-fmax (v_med3_f32 + v_cvt_pk_fp8_f32 /
// v_cvt_pk_bf8_f32, subnormals and the flush of |x| <= half the smallest subnormal included), followed by the
// reference's "+0 for zero results" (fp8.py:33-35) done on the packed bytes.  That is ~4.5 VALU operations per element
// instead of the ~13.5 of the closed form + pack -- the pass was issue-limited at ~0.6 of its VALU rate while a
// no-arithmetic 2:1 narrowing copy with the same geometry streams 5.6 TB/s (tools/exp_rw_mix.hip).  Non-finite inputs
// (the reference maps them to NaN; the hardware keeps E5M2 infinities and the clamp would hide them) are found from the
// packed magnitude maximum that also serves the observer, and that vector pair takes the closed form.
// 16 elements (two 16-B vectors) -> 16 FP8 bytes; mag accumulates the packed |x| maxima (observer); returns false when
// a non-finite input was seen (the caller redoes this pair with the closed form)
template <bool E5M2>
__device__ __forceinline__ bool fq8_fast16(const uint4 v0, const uint4 v1, uint32_t &mag, uint4 &o8) {
    uint32_t m = pk_max_u16(v0.x & 0x7FFF7FFFu, v0.y & 0x7FFF7FFFu);
    m = pk_max_u16(m, v0.z & 0x7FFF7FFFu);
    m = pk_max_u16(m, v0.w & 0x7FFF7FFFu);
    m = pk_max_u16(m, v1.x & 0x7FFF7FFFu);
    m = pk_max_u16(m, v1.y & 0x7FFF7FFFu);
    m = pk_max_u16(m, v1.z & 0x7FFF7FFFu);
    m = pk_max_u16(m, v1.w & 0x7FFF7FFFu);
    mag = pk_max_u16(mag, m);
    o8.x = fq8_fast_word4<E5M2>(v0.x, v0.y);
    o8.y = fq8_fast_word4<E5M2>(v0.z, v0.w);
    o8.z = fq8_fast_word4<E5M2>(v1.x, v1.y);
    o8.w = fq8_fast_word4<E5M2>(v1.z, v1.w);
    return ((m & 0xFFFFu) < 0x7F80u) & ((m >> 16) < 0x7F80u);
}

template <bool E5M2>
__device__ __forceinline__ uint4 fq8_closed16(const uint4 v0, const uint4 v1, const qt_format &fmt) {
    const UniformDiv dv(1.0f);
    uint32_t unused = 0;
    float q0[8], q1[8];
    fq8_vec(v0, fmt, dv, true, false, unused, q0);
    fq8_vec(v1, fmt, dv, true, false, unused, q1);
    uint4 o8;
    o8.x = qt_pack_fp8x4<E5M2>(q0[0], q0[1], q0[2], q0[3]);
    o8.y = qt_pack_fp8x4<E5M2>(q0[4], q0[5], q0[6], q0[7]);
    o8.z = qt_pack_fp8x4<E5M2>(q1[0], q1[1], q1[2], q1[3]);
    o8.w = qt_pack_fp8x4<E5M2>(q1[4], q1[5], q1[6], q1[7]);
    return o8;
}

// Each lane handles 16 consecutive elements: two 16-B loads -> one 16-B FP8 store (+ two bf16 stores).
template <bool OBS, bool BOTH, bool E5M2>
__global__ __launch_bounds__(256) void fq8_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, uint4 *__restrict__ y8,
                                                  size_t npair, qt_format fmt, const float *__restrict__ scale,
                                                  uint32_t *amax_out, bool hw) {
    float s = scale ? qt_bf2f(qt_f2bf(*scale)) : 1.0f;
    const bool unit = (s == 1.0f);
    const UniformDiv dv(s);
    uint32_t amax = 0;
    if constexpr (!BOTH) {
        if (unit && hw) {                             // FP8 code only, scale 1: hardware conversion (see fq8_fast16)
            uint32_t mag = 0;
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
                const uint4 v0 = x[2 * i], v1 = x[2 * i + 1];
                uint4 o8;
                if (__builtin_expect(!fq8_fast16<E5M2>(v0, v1, mag, o8), 0)) o8 = fq8_closed16<E5M2>(v0, v1, fmt);
                y8[i] = o8;
            }
            if constexpr (OBS) {
                const uint32_t lo = mag << 16, hi = mag & 0xFFFF0000u;
                amax = lo > hi ? lo : hi;
                block_amax_commit<256>(amax, amax_out);
            }
            return;
        }
    }
    if constexpr (BOTH && !OBS) {
        if (unit && hw) {                             // bf16 + FP8 code, scale 1: same conversion, decoded back for the bf16 copy
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
                const uint4 v0 = x[2 * i], v1 = x[2 * i + 1];
                uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, b[4] = {v1.x, v1.y, v1.z, v1.w};
                const uint2 c0 = fq8_hw_vec8<E5M2>(a, fmt), c1 = fq8_hw_vec8<E5M2>(b, fmt);
                y8[i] = uint4{c0.x, c0.y, c1.x, c1.y};
                y[2 * i] = uint4{a[0], a[1], a[2], a[3]};
                y[2 * i + 1] = uint4{b[0], b[1], b[2], b[3]};
            }
            return;
        }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
        const uint4 v0 = x[2 * i], v1 = x[2 * i + 1];
        float q0[8], q1[8];
        fq8_vec(v0, fmt, dv, unit, OBS, amax, q0);
        fq8_vec(v1, fmt, dv, unit, OBS, amax, q1);
        uint4 o8;
        o8.x = qt_pack_fp8x4<E5M2>(q0[0], q0[1], q0[2], q0[3]);
        o8.y = qt_pack_fp8x4<E5M2>(q0[4], q0[5], q0[6], q0[7]);
        o8.z = qt_pack_fp8x4<E5M2>(q1[0], q1[1], q1[2], q1[3]);
        o8.w = qt_pack_fp8x4<E5M2>(q1[4], q1[5], q1[6], q1[7]);
        y8[i] = o8;
        if constexpr (BOTH) {
            y[2 * i] = bf16_vec_from(q0, s, unit);
            y[2 * i + 1] = bf16_vec_from(q1, s, unit);
        }
    }
    if constexpr (OBS) block_amax_commit<256>(amax, amax_out);
}

// Several tensors, one launch: the FP8-only weight pass of sibling Linears (q / k / v projections) whose FP8 codes are
// written back to back into one buffer, so that one GEMM can multiply by all of them.  The three 4096 x 4096 passes of
// a LLaMA-2-7B layer are 12.5 us launches at 4.0 TB/s each; together they stream at the large-tensor rate.
struct MultiArgs {
    const uint4 *x[4];
    size_t npair[4];          // 32-byte input pairs per tensor
    size_t first[5];          // prefix sums of npair
};

template <bool E5M2>
__global__ __launch_bounds__(256) void fq8_multi_kernel(MultiArgs a, uint4 *__restrict__ y8, qt_format fmt) {
    uint32_t mag = 0;
    const size_t total = a.first[4];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (i >= a.first[1]) + (i >= a.first[2]) + (i >= a.first[3]);
        const size_t j = i - a.first[t];
        const uint4 v0 = a.x[t][2 * j], v1 = a.x[t][2 * j + 1];
        uint4 o8;
        if (__builtin_expect(!fq8_fast16<E5M2>(v0, v1, mag, o8), 0)) o8 = fq8_closed16<E5M2>(v0, v1, fmt);
        y8[i] = o8;
    }
}

// ---- many tensors, one launch: every weight fake-quantizer of a training step (harness.GraphedTrainStep) ---------------
// A RoBERTa-base step fake-quantizes 74 weights of 0.6 - 2.4 M elements, each a ~10 us launch inside the replayed graph; the
// weights do not change between the start of a step and the optimizer at its end, so all of them can go first, as one launch:
// tensor i has its own scale and amax slot (the per-tensor state machine of fake_quantize.py:230-246 is untouched), the format is
// shared.  A workgroup takes one tile of kMultiTile 16-byte vectors of one tensor (tiles never straddle tensors).
struct qt_fq_item_dev {
    const uint4 *x;
    uint4 *y;
    const float *scale;
    uint32_t *amax;              // NULL: not observed
    unsigned long long nvec;     // 16-byte vectors (8 bf16 each)
    unsigned long long first;    // index of the tensor's first tile in the launch
};
constexpr int kMultiBlock = 256, kMultiUnroll = 4, kMultiTile = kMultiBlock * kMultiUnroll;

template <int KIND>
__global__ __launch_bounds__(kMultiBlock) void fq_multi_kernel(const qt_fq_item_dev *__restrict__ items, int count, qt_format fmt,
                                                              const uint16_t *__restrict__ lut) {
    Rounder<KIND> rnd{fmt, nullptr};
    if constexpr (KIND == kFmtRows) {
        __shared__ uint4 s_rows[512];
        const uint4 *g = (const uint4 *)(lut + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += kMultiBlock) s_rows[i] = g[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = lut;
        __syncthreads();
    }
    // the tensor this tile belongs to: the last item whose first tile is <= blockIdx.x.  Wave 0 looks at all items at once (lane l at
    // items l, l + 64, ...: independent loads, one round trip) -- a binary search is log2(count) DEPENDENT loads, 3-5 us in front of a
    // tile that takes 3
    __shared__ int s_item;
    if (threadIdx.x < 64) {
        int best = 0;
        for (int j = (int)threadIdx.x; j < count; j += 64)
            if (items[j].first <= (unsigned long long)blockIdx.x) best = j;          // (`first` ascends with j)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
        if (threadIdx.x == 0) s_item = best;
    }
    __syncthreads();
    const qt_fq_item_dev it = items[s_item];
    const size_t v0 = ((size_t)blockIdx.x - it.first) * kMultiTile;
    const float s = qt_bf2f(qt_f2bf(*it.scale));
    // (pointers that come out of memory are generic to the compiler: say that they are global, or every access is a flat_ instruction)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4_t *gload_t;
    typedef __attribute__((address_space(1))) u32x4_t *gstore_t;
    const gload_t gx = (gload_t)(const void *)it.x;
    const gstore_t gy = (gstore_t)(void *)it.y;
    const UniformDiv dv(s);
    uint32_t amax = 0;
    auto run = [&](auto divc, auto obsc) __attribute__((always_inline)) {
        constexpr int DIV = decltype(divc)::value;
        constexpr bool OBS = decltype(obsc)::value;
        uint4 v[kMultiUnroll];
#pragma unroll
        for (int u = 0; u < kMultiUnroll; ++u) {
            const size_t i = v0 + (size_t)u * kMultiBlock + threadIdx.x;
            v[u] = uint4{0u, 0u, 0u, 0u};
            if (i < it.nvec) {
                const u32x4_t t = gx[i];
                v[u] = uint4{t.x, t.y, t.z, t.w};
            }
        }
#pragma unroll
        for (int u = 0; u < kMultiUnroll; ++u) {
            const size_t i = v0 + (size_t)u * kMultiBlock + threadIdx.x;
            const uint4 r = fq_vec<kIoBf16, KIND, DIV, OBS>(v[u], dv, rnd, amax);
            if (i < it.nvec) gy[i] = u32x4_t{r.x, r.y, r.z, r.w};
        }
    };
    const bool obs = it.amax != nullptr;                      // uniform
    if (s == 1.0f) {
        if (obs) run(std::integral_constant<int, kDivUnit>{}, std::true_type{});
        else run(std::integral_constant<int, kDivUnit>{}, std::false_type{});
    } else if (dv.safe) {
        if (obs) run(std::integral_constant<int, kDivFast>{}, std::true_type{});
        else run(std::integral_constant<int, kDivFast>{}, std::false_type{});
    } else {
        if (obs) run(std::integral_constant<int, kDivExact>{}, std::true_type{});
        else run(std::integral_constant<int, kDivExact>{}, std::false_type{});
    }
    if (obs) block_amax_commit<kMultiBlock>(amax, it.amax);
}

// FP8 codes of many weights in one launch: every weight pass of an evaluation forward whose GEMMs take the pair route (weight pass +
// library FP8 GEMM: BERT-base's 36 Linears of 0.6 - 2.4 M elements are 36 launches of ~5 us inside the replayed graph).  Stateless
// E4M3 / E5M2 at unit scale only; item i is the codes-only qt_fake_quant_bf16_fp8 of tensor i.  A workgroup takes one tile of
// 1024 pairs (16 384 elements) of one tensor.
struct qt_fq8_item_dev {
    const uint4 *x;
    uint4 *y8;
    unsigned long long npair;     // 32-byte input pairs (16 elements each)
    unsigned long long first;     // index of the tensor's first tile in the launch
};
constexpr int kMulti8Tile = 256 * 4;

template <bool E5M2>
__global__ __launch_bounds__(256) void fq8_items_kernel(const qt_fq8_item_dev *__restrict__ items, int count, qt_format fmt) {
    // (the tile's tensor by one round of independent loads, its pointers declared global: see fq_multi_kernel)
    __shared__ int s_item;
    if (threadIdx.x < 64) {
        int best = 0;
        for (int j = (int)threadIdx.x; j < count; j += 64)
            if (items[j].first <= (unsigned long long)blockIdx.x) best = j;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
        if (threadIdx.x == 0) s_item = best;
    }
    __syncthreads();
    const qt_fq8_item_dev it = items[s_item];
    const size_t p0 = ((size_t)blockIdx.x - it.first) * kMulti8Tile;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) u32x4_t *const gx = (const __attribute__((address_space(1))) u32x4_t *)(const void *)it.x;
    uint32_t mag = 0;
    uint4 v0[4], v1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = p0 + (size_t)u * 256 + threadIdx.x;
        v0[u] = v1[u] = uint4{0u, 0u, 0u, 0u};
        if (i < it.npair) {
            const u32x4_t a = gx[2 * i], b = gx[2 * i + 1];
            v0[u] = uint4{a.x, a.y, a.z, a.w};
            v1[u] = uint4{b.x, b.y, b.z, b.w};
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = p0 + (size_t)u * 256 + threadIdx.x;
        uint4 o8;
        if (__builtin_expect(!fq8_fast16<E5M2>(v0[u], v1[u], mag, o8), 0)) o8 = fq8_closed16<E5M2>(v0[u], v1[u], fmt);
        if (i < it.npair) it.y8[i] = o8;
    }
}

// ---- strided rows -> contiguous ----------------------------------------------------------------------
// Attention hands the hooks permuted views (q / k / v are [B, S, H, D] storage seen as [B, H, S, D]).  The
// reference's vmap returns a contiguous tensor (decomposed.py:155), i.e. the layout change is part of the
// op; doing it inside the pass saves the separate .contiguous() copy (2 + 2 B/element) in front of it.
// x is a logical [d0, d1, d2, inner] tensor with element strides (s0, s1, s2, 1); y is contiguous.
struct RowsArgs {
    const uint16_t *x;
    uint16_t *y;
    long d1, d2, vpr;       // vpr = inner / 8 (16-B vectors per row)
    long s0, s1, s2;
    size_t nvec;
};

template <int KIND, bool OBS, int FP8 = 0, int BLOCK = 256>          // FP8: 0 none, 1 also write E4M3 bytes, 2 E5M2 bytes (unit scale, FP_SAT kinds)
__global__ __launch_bounds__(BLOCK) void fq_rows_kernel(RowsArgs a, qt_format fmt, const uint16_t *__restrict__ lut,
                                                        const float *__restrict__ scale, uint32_t *amax_out,
                                                        uint2 *__restrict__ y8 = nullptr) {
    Rounder<KIND> rnd{fmt, lut};
    if constexpr (KIND == kFmtRows) {                    // the row words behind the map, as in fq_kernel
        __shared__ uint4 s_rows[512];
        const uint4 *g = (const uint4 *)(lut + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += BLOCK) s_rows[i] = g[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = lut;
        __syncthreads();
    }
    float s = scale ? qt_bf2f(qt_f2bf(*scale)) : 1.0f;
    const bool unit = (s == 1.0f);
    const UniformDiv dv(s);
    uint32_t amax = 0;
    for (size_t v = (size_t)blockIdx.x * BLOCK + threadIdx.x; v < a.nvec; v += (size_t)gridDim.x * BLOCK) {
        const long row = (long)(v / (size_t)a.vpr), c = (long)(v % (size_t)a.vpr);
        const long i2 = row % a.d2, t = row / a.d2;
        const long i1 = t % a.d1, i0 = t / a.d1;
        const uint4 in = *(const uint4 *)(a.x + i0 * a.s0 + i1 * a.s1 + i2 * a.s2 + c * 8);
        uint4 r;
        if constexpr (FP8 != 0 && !OBS) {                // unit scale, exact E4M3 / E5M2 (host-checked): hardware conversion
            if (unit) {
                uint32_t o[4] = {in.x, in.y, in.z, in.w};
                const uint2 codes = fq8_hw_vec8<FP8 == 2>(o, fmt);
                if (a.y) ((uint4 *)a.y)[v] = uint4{o[0], o[1], o[2], o[3]};       // NULL: only the codes are wanted
                y8[v] = codes;
                continue;
            }
        }
        if (unit) r = fq_vec<kIoBf16, KIND, kDivUnit, OBS>(in, dv, rnd, amax);
        else if (dv.safe) r = fq_vec<kIoBf16, KIND, kDivFast, OBS>(in, dv, rnd, amax);
        else r = fq_vec<kIoBf16, KIND, kDivExact, OBS>(in, dv, rnd, amax);
        ((uint4 *)a.y)[v] = r;
        if constexpr (FP8 != 0) {                        // r holds on-grid values: their FP8 bytes are exact
            y8[v] = uint2{qt_pack_fp8x4<FP8 == 2>(qt_u2f(r.x << 16), qt_u2f(r.x & 0xFFFF0000u), qt_u2f(r.y << 16), qt_u2f(r.y & 0xFFFF0000u)),
                          qt_pack_fp8x4<FP8 == 2>(qt_u2f(r.z << 16), qt_u2f(r.z & 0xFFFF0000u), qt_u2f(r.w << 16), qt_u2f(r.w & 0xFFFF0000u))};
        }
    }
    if constexpr (OBS) block_amax_commit<BLOCK>(amax, amax_out);
}

// ---- microscaling: one scale per block of `bs` consecutive elements of the last axis -----------------
// MXFakeQuantFunction (fake_quantize.py:98-133) = calculate_mx_qparam + quantize + multiply
// (decomposed.py:365-448): s = amax(block) / quant_max [-> scale table], s = (s > 0 ? s : 1),
// y = map[x / s] * s, every op in the tensor's dtype.  A block is bs/8 (bf16) or bs/4 (fp32) adjacent
// lanes of one wavefront, so the block amax is a sub-wave __shfl_xor reduction: one read, one write,
// plus bs-times-smaller scale output.
template <int IO, int KIND>
__global__ __launch_bounds__(256) void fq_mx_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, void *__restrict__ sf_out,
                                                    size_t nvec, int group, qt_format fmt, const uint16_t *__restrict__ lut,
                                                    float quant_max, const uint16_t *__restrict__ scale_lut) {
    Rounder<KIND> rnd{fmt, lut};
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t iters = (nvec + stride - 1) / stride;
    for (size_t it = 0; it < iters; ++it) {
        const size_t v = it * stride + (size_t)blockIdx.x * 256 + threadIdx.x;
        const bool live = v < nvec;
        uint4 in = {0u, 0u, 0u, 0u};
        if (live) in = x[v];
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        uint32_t am = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (IO == kIoBf16) {
                uint32_t a0 = (w[j] << 16) & 0x7FFFFFFFu, a1 = w[j] & 0x7FFF0000u;
                am = am > a0 ? am : a0;
                am = am > a1 ? am : a1;
            } else {
                uint32_t a0 = w[j] & 0x7FFFFFFFu;
                am = am > a0 ? am : a0;
            }
        }
        for (int off = 1; off < group; off <<= 1) {           // block amax over `group` adjacent lanes
            uint32_t o = (uint32_t)__shfl_xor((int)am, off, 64);
            am = am > o ? am : o;
        }
        // torch.amax propagates NaN: the integer max already does (NaN patterns are the largest)
        float s = qt_u2f(am) / quant_max;                      // amax / quant_max   (decomposed.py:415)
        if constexpr (IO == kIoBf16) s = qt_u2f(pack_bf16x2(s, 0.0f) << 16);
        if (scale_lut) {                                       // :418-419
            const uint32_t img = IO == kIoBf16 ? qt_f2u(s) : qt_fold_img(qt_f2u(s));
            s = qt_bf2f(scale_lut[img >> 16]);
        }
        s = s > 0.0f ? s : 1.0f;                               // :421
        if (live && (threadIdx.x & (group - 1)) == 0) {
            const size_t blk = v / (size_t)group;
            if constexpr (IO == kIoBf16) ((uint16_t *)sf_out)[blk] = (uint16_t)(qt_f2u(s) >> 16);
            else ((float *)sf_out)[blk] = s;
        }
        const UniformDiv dv(s);                                // per-lane divisor; same fast/exact contract
        uint32_t unused = 0;
        uint4 r;
        if (dv.safe) r = fq_vec<IO, KIND, kDivFast, false>(in, dv, rnd, unused);
        else r = fq_vec<IO, KIND, kDivExact, false>(in, dv, rnd, unused);
        if (live) y[v] = r;
    }
}

// ---- per-channel: x viewed as [outer][C][inner], scale[c], amax[c] ---------------------------
template <int IO, int KIND>
__global__ __launch_bounds__(256) void fq_pc_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t outer,
                                                   size_t C, size_t inner, qt_format fmt,
                                                   const uint16_t *__restrict__ lut, const float *__restrict__ scale,
                                                   uint32_t *amax_out) {
    // one workgroup per (outer, c) row segment group: rows = outer*C, each `inner` long
    Rounder<KIND> rnd{fmt, lut};
    const size_t rows = outer * C;
    __shared__ uint32_t s_part[4];
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const size_t c = row % C;
        float s = scale ? scale[c] : 1.0f;
        if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
        const bool unit = (s == 1.0f);
        uint32_t amax = 0;
        const size_t base = row * inner;
        if (amax_out) {
            for (size_t i = threadIdx.x; i < inner; i += 256) fq_one<IO, KIND, true>(xv, yv, base + i, s, unit, rnd, amax);
            amax = wave_max_u32(amax);
            if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = amax;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t m = s_part[0];
                for (int i = 1; i < 4; ++i) m = m > s_part[i] ? m : s_part[i];
                if (m) atomicMax(amax_out + c, m);
            }
            __syncthreads();
        } else {
            for (size_t i = threadIdx.x; i < inner; i += 256) fq_one<IO, KIND, false>(xv, yv, base + i, s, unit, rnd, amax);
        }
    }
}

// Vectorised form for rows made of whole 16-byte vectors (weights [C, K] with K % 8 == 0, conv weights, ...): one
// workgroup per (outer, c) row, 16-B loads / stores, the row's scale in a register, per-channel amax by one atomicMax.
template <int IO, int KIND>
__global__ __launch_bounds__(256) void fq_pc_vec_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, size_t rows,
                                                       size_t C, size_t vpr, qt_format fmt, const uint16_t *__restrict__ lut,
                                                       const float *__restrict__ scale, uint32_t *amax_out) {
    Rounder<KIND> rnd{fmt, lut};
    __shared__ uint32_t s_part[4];
    if constexpr (KIND == kFmtRows) {
        __shared__ uint4 s_rows[512];               // the row words behind the map (qt_format.p1 bit 0), see fq_kernel
        const uint4 *g = (const uint4 *)(lut + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rows[i] = g[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = lut;
        __syncthreads();
    }
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const size_t c = row % C;
        float s = scale ? scale[c] : 1.0f;
        if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
        const bool unit = (s == 1.0f);
        const UniformDiv dv(s);
        uint32_t amax = 0;
        const uint4 *xr = x + row * vpr;
        uint4 *yr = y ? y + row * vpr : nullptr;
        for (size_t i = threadIdx.x; i < vpr; i += 256) {
            const uint4 in = xr[i];
            uint4 r;
            if (amax_out) {
                if (unit) r = fq_vec<IO, KIND, kDivUnit, true>(in, dv, rnd, amax);
                else if (dv.safe) r = fq_vec<IO, KIND, kDivFast, true>(in, dv, rnd, amax);
                else r = fq_vec<IO, KIND, kDivExact, true>(in, dv, rnd, amax);
            } else {
                if (unit) r = fq_vec<IO, KIND, kDivUnit, false>(in, dv, rnd, amax);
                else if (dv.safe) r = fq_vec<IO, KIND, kDivFast, false>(in, dv, rnd, amax);
                else r = fq_vec<IO, KIND, kDivExact, false>(in, dv, rnd, amax);
            }
            if (yr) yr[i] = r;
        }
        if (amax_out) {
            amax = wave_max_u32(amax);
            if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = amax;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t m = s_part[0];
                for (int i = 1; i < 4; ++i) m = m > s_part[i] ? m : s_part[i];
                if (m) atomicMax(amax_out + c, m);
            }
            __syncthreads();
        }
    }
}

// Table formats on large per-channel tensors: the value map staged in LDS (the whole 128 KiB, or its non-negative 64 KiB half
// for odd-symmetric maps: then two workgroups per CU) by a persistent 1024-thread workgroup, ONE WAVE PER ROW -- the row's
// scale in a register, amax reduced inside the wave, no workgroup barrier in the row loop.  The kernel above gathers from
// the L2-resident table instead (3.0-3.1 TB/s on a [4096, 11008] weight).
template <int IO, int KINDL>
__global__ __launch_bounds__(1024) void fq_pc_vec_lds_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, size_t rows,
                                                             size_t C, size_t vpr, qt_format fmt, const uint16_t *__restrict__ lut,
                                                             const float *__restrict__ scale, uint32_t *amax_out) {
    constexpr int kVecs = (KINDL == kFmtLutHalf ? QT_MAP_ENTRIES / 2 : QT_MAP_ENTRIES) * 2 / 16;
    __shared__ uint4 s_lut[kVecs];
    {
        const uint4 *g = (const uint4 *)lut;
#pragma unroll
        for (int i = 0; i < kVecs / 1024; ++i) s_lut[i * 1024 + threadIdx.x] = g[i * 1024 + threadIdx.x];
    }
    __syncthreads();
    Rounder<KINDL> rnd{fmt, (const uint16_t *)s_lut};
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 16;
    for (size_t row = wave; row < rows; row += nwaves) {
        const size_t c = row % C;
        float s = scale ? scale[c] : 1.0f;
        if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
        const bool unit = (s == 1.0f);
        const UniformDiv dv(s);
        uint32_t amax = 0;
        const uint4 *xr = x + row * vpr;
        uint4 *yr = y + row * vpr;
        for (size_t i = lane; i < vpr; i += 64) {
            const uint4 in = xr[i];
            uint4 r;
            if (amax_out) {
                if (unit) r = fq_vec<IO, KINDL, kDivUnit, true>(in, dv, rnd, amax);
                else if (dv.safe) r = fq_vec<IO, KINDL, kDivFast, true>(in, dv, rnd, amax);
                else r = fq_vec<IO, KINDL, kDivExact, true>(in, dv, rnd, amax);
            } else {
                if (unit) r = fq_vec<IO, KINDL, kDivUnit, false>(in, dv, rnd, amax);
                else if (dv.safe) r = fq_vec<IO, KINDL, kDivFast, false>(in, dv, rnd, amax);
                else r = fq_vec<IO, KINDL, kDivExact, false>(in, dv, rnd, amax);
            }
            yr[i] = r;
        }
        if (amax_out) {
            amax = wave_max_u32(amax);
            if (lane == 0 && amax) atomicMax(amax_out + c, amax);
        }
    }
}

// inner == 1 (channel is the fastest dim): element i belongs to channel i % C
template <int IO, int KIND>
__global__ __launch_bounds__(256) void fq_pc_last_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t n,
                                                        size_t C, qt_format fmt, const uint16_t *__restrict__ lut,
                                                        const float *__restrict__ scale, uint32_t *amax_out) {
    Rounder<KIND> rnd{fmt, lut};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t c = i % C;
        float s = scale ? scale[c] : 1.0f;
        if constexpr (IO == kIoBf16) s = qt_bf2f(qt_f2bf(s));
        uint32_t amax = 0;
        if (amax_out) {
            fq_one<IO, KIND, true>(xv, yv, i, s, s == 1.0f, rnd, amax);
            if (amax) atomicMax(amax_out + c, amax);
        } else {
            fq_one<IO, KIND, false>(xv, yv, i, s, s == 1.0f, rnd, amax);
        }
    }
}

// ---- delayed-scaling state machine (fake_quantize.py:230-242), one thread per channel ----------
__device__ __forceinline__ void scale_update_channel(float *__restrict__ hist, int L, int C, float *__restrict__ scale,
                                                     float quant_max, int pow2, int c) {
    // amax = torch.amax(history, dim=0): NaN propagates
    float amax = hist[c];
    bool nan = amax != amax;
    for (int l = 1; l < L; ++l) {
        float h = hist[(size_t)l * C + c];
        nan |= (h != h);
        amax = h > amax ? h : amax;
    }
    if (nan) amax = qt_u2f(QT_NAN32);
    if (L > 1) {   // roll(history, -1, 0): new[i] = old[i+1], new[L-1] = old[0]
        float first = hist[c];
        for (int l = 0; l + 1 < L; ++l) hist[(size_t)l * C + c] = hist[(size_t)(l + 1) * C + c];
        hist[(size_t)(L - 1) * C + c] = first;
    }
    hist[c] = 0.0f;   // slot 0 <- amax_cur, max-accumulated by the fused pass
    float sf = amax / quant_max;
    const float old = scale[c];
    sf = (amax > 0.0f) ? sf : old;
    sf = (__builtin_isfinite(amax)) ? sf : old;
    if (pow2) {
        // torch.pow(2, torch.ceil(torch.log2(sf))) in fp32
        float lg = (float)log2((double)sf);
        sf = (float)exp2((double)ceilf(lg));
    }
    scale[c] = sf;
}

__global__ void scale_update_kernel(float *__restrict__ hist, int L, int C, float *__restrict__ scale, float quant_max,
                                    int pow2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) scale_update_channel(hist, L, C, scale, quant_max, pow2, c);
}

// The same update for MANY fake-quantizers in one launch (workgroup i = fake-quantizer i): a captured training step runs
// it once up front instead of one 4.7 us launch in front of each of its few hundred fake-quant passes.  Valid because
// the scale a call applies depends only on amaxes of EARLIER calls (fake_quantize.py:230-242).
__global__ void scale_update_multi_kernel(float *const *__restrict__ hist, const int *__restrict__ L, const int *__restrict__ C,
                                          float *const *__restrict__ scale, const float *__restrict__ quant_max,
                                          const int *__restrict__ pow2) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < C[i]; c += blockDim.x) scale_update_channel(hist[i], L[i], C[i], scale[i], quant_max[i], pow2[i], c);
}

// ---- generic quantize / dequantize (decomposed.py:166-262); tables read from global memory ------
struct QdqArgs {
    const uint16_t *in_lut;    // applied first (dequantize's input_qmap) or NULL
    const uint16_t *out_lut;   // applied last (quantize's qmap / dequantize's output_qmap) or NULL
    qt_format out_fmt;         // closed form for out_lut when kind != LUT and out_lut == NULL
    int use_out;               // 1 = apply out rounding
    int mode;                  // 0 = x / s [+ zp], 1 = (x [- zp]) * s
};

template <int IO>
__global__ __launch_bounds__(256) void qdq_kernel(const void *__restrict__ xv, void *__restrict__ yv, size_t n, QdqArgs a,
                                                 const void *__restrict__ scale, const void *__restrict__ zp) {
    float s, z = 0.0f;
    if constexpr (IO == kIoBf16) {
        s = qt_bf2f(*(const uint16_t *)scale);
        if (zp) z = qt_bf2f(*(const uint16_t *)zp);
    } else {
        s = *(const float *)scale;
        if (zp) z = *(const float *)zp;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // value `v` is kept in the tensor dtype after every op (bf16 ops round to bf16)
        float v;
        if constexpr (IO == kIoBf16) v = qt_bf2f(((const uint16_t *)xv)[i]);
        else v = ((const float *)xv)[i];
        auto round_io = [](float f) -> float {
            if constexpr (IO == kIoBf16) return qt_u2f(pack_bf16x2(f, 0.0f) << 16);
            else return f;
        };
        auto lookup = [&](const uint16_t *t, float f) -> float {
            uint32_t img = (IO == kIoBf16) ? qt_f2u(f) : qt_fold_img(qt_f2u(f));
            return qt_bf2f(t[img >> 16]);
        };
        if (a.in_lut) v = lookup(a.in_lut, v);
        if (a.mode == 0) {
            v = round_io(v / s);
            if (zp) v = round_io(v + z);
        } else {
            if (zp) v = round_io(v - z);
            v = round_io(v * s);
        }
        if (a.use_out) {
            if (a.out_lut) v = lookup(a.out_lut, v);
            else {
                uint32_t img = (IO == kIoBf16) ? qt_f2u(v) : qt_fold_img(qt_f2u(v));
                v = qt_u2f(qt_apply_format_img(a.out_fmt, img));
            }
        }
        if constexpr (IO == kIoBf16) ((uint16_t *)yv)[i] = (uint16_t)(qt_f2u(v) >> 16);
        else ((float *)yv)[i] = v;
    }
}

// fp16 vmap: index through the fp32 image, looked-up bf16 value cast to fp16 (decomposed.py:151-161)
__global__ __launch_bounds__(256) void vmap_f16_kernel(const _Float16 *__restrict__ x, _Float16 *__restrict__ y, size_t n,
                                                      qt_format fmt, const uint16_t *__restrict__ lut) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t img = qt_fold_img(qt_f2u((float)x[i]));
        uint32_t r = lut ? ((uint32_t)lut[img >> 16] << 16) : qt_apply_format_img(fmt, img);
        y[i] = (_Float16)qt_u2f(r);
    }
}

// exported rounding functions on fp32 tensors
__global__ __launch_bounds__(256) void round_fp8_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                       int mbits, int emin, float fmax) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = qt_u2f(qt_fp_sat_u32(qt_f2u(x[i]), mbits, emin, fmax));
}
__global__ __launch_bounds__(256) void round_posit_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                         int nbits, int es, float thr) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = qt_u2f(qt_posit_u32(qt_f2u(x[i]), nbits, es, thr));
}

__global__ __launch_bounds__(256) void posit_bits_kernel(const float *__restrict__ x, float *__restrict__ y, int32_t *__restrict__ pbits,
                                                        size_t n, int nbits, int es, float thr) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int32_t pb;
        y[i] = qt_u2f(qt_posit_bits_u32(qt_f2u(x[i]), nbits, es, thr, &pb));
        if (pbits) pbits[i] = pb;
    }
}

// ---- launch helpers --------------------------------------------------------------------------
int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return cus;
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

inline unsigned grid_for(size_t work_items, size_t per_block, int blocks_per_cu) {
    size_t want = (work_items + per_block - 1) / per_block;
    size_t cap = (size_t)num_cus() * blocks_per_cu;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

// Tensors below this many elements use the global-table gather kernel instead of staging 128 KiB per CU.
constexpr size_t kLutLdsMinElems = (size_t)1 << 21;

#ifdef QT_TUNING_BUILD
int g_variant = 0;        // tools/exp_stream.py: selects a launch geometry for bf16 closed-form passes
#endif
// Observed SHORT passes (the [2048, 768 .. 3072] tensors of a training step): 16-byte loads in flight per lane, i.e. how few
// workgroups -- and same-address atomics on the freshly zeroed amax slot -- the launch has.  (Measured in the training step, round 5:
// 1 or 2 makes no difference, 9.0 against 9.1 us per launch.)
inline int obs_unroll() {
#ifdef QT_TUNING_BUILD
    static const int v = getenv("QT_OBS_UNR") ? atoi(getenv("QT_OBS_UNR")) : 1;
    return v;
#else
    return 1;
#endif
}
int g_blocks_per_cu = 32;

template <int IO, int KIND, int BLOCK, int UNR, int NT>
int launch_variant(const void *x, void *y, size_t n, const qt_format &fmt, const uint16_t *lut, const float *scale,
                   uint32_t *amax, hipStream_t st) {
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    const size_t nvec = n / kPer;
    unsigned grid = grid_for(nvec, (size_t)BLOCK * UNR, g_blocks_per_cu);
    if (amax)
        fq_kernel<IO, KIND, true, BLOCK, UNR, NT><<<grid, BLOCK, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
    else
        fq_kernel<IO, KIND, false, BLOCK, UNR, NT><<<grid, BLOCK, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
    return launch_status();
}

template <int IO, int KIND>
int launch_fq_kind(const void *x, void *y, size_t n, const qt_format &fmt, const uint16_t *lut, const float *scale,
                   uint32_t *amax, hipStream_t st) {
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15u) == 0;
    const bool gather = !aligned || (KIND == QT_FMT_LUT && (n < kLutLdsMinElems || y == nullptr)) || n < 4096;
    if constexpr (KIND == QT_FMT_LUT) {
        // The row form (qt_format.p1 bit 0): no 128 KiB staging, so the pass runs at the closed
        // forms' occupancy and bandwidth, small tensors included.
        if ((fmt.p1 & 1) && aligned && y != nullptr && n >= 4096) {
            const size_t nv = n / kPer;
            int row_blocks = 0;
#ifdef QT_TUNING_BUILD
            static const int e_row_blocks = getenv("QT_ROW_BLOCKS") ? atoi(getenv("QT_ROW_BLOCKS")) : 0;   // tools/ only: workgroups per CU
            static const int row_wide = getenv("QT_ROW_WIDE") ? atoi(getenv("QT_ROW_WIDE")) : 0;           // tools/ only: 1024-thread workgroups
            row_blocks = e_row_blocks;
            if (row_wide) {
                unsigned grid = grid_for(nv, (size_t)1024, row_blocks ? row_blocks : 4);
                if (amax) fq_kernel<IO, kFmtRows, true, 1024><<<grid, 1024, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
                else fq_kernel<IO, kFmtRows, false, 1024><<<grid, 1024, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
                return launch_status();
            }
#endif
            if (IO == kIoBf16 && nv <= kRowsDirectMaxVecs) {           // short pass: the table where it lies (fq_rows_direct_kernel)
                if (amax && obs_unroll() == 2) fq_rows_direct_kernel<IO, true, 1024, 2><<<grid_for(nv, 2048, 2), 1024, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
                else if (amax) fq_rows_direct_kernel<IO, true, 1024><<<grid_for(nv, 1024, 2), 1024, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
                else fq_rows_direct_kernel<IO, false, 256><<<grid_for(nv, 256, 8), 256, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
                return launch_status();
            }
            unsigned grid = grid_for(nv, (size_t)kAluBlock * kUnroll, row_blocks ? row_blocks : 8);
            if (amax)
                fq_kernel<IO, kFmtRows, true, kAluBlock><<<grid, kAluBlock, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
            else
                fq_kernel<IO, kFmtRows, false, kAluBlock><<<grid, kAluBlock, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
            return launch_status();
        }
        if (aligned && y != nullptr && n >= 4096 && n < kLutLdsMinElems) {
            const size_t nv = n / kPer;
            unsigned grid = grid_for(nv, 256, 8);
            if (amax) fq_gather_vec_kernel<IO, true><<<grid, 256, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
            else fq_gather_vec_kernel<IO, false><<<grid, 256, 0, st>>>(x, y, nv, n, fmt, lut, scale, amax);
            return launch_status();
        }
    }
    if (gather) {
        // observe-only never touches the table, so KIND is irrelevant there
        unsigned grid = grid_for(n, 256 * 8, 8);
        if (amax)
            fq_gather_kernel<IO, KIND, true><<<grid, 256, 0, st>>>(x, y, n, fmt, lut, scale, amax);
        else
            fq_gather_kernel<IO, KIND, false><<<grid, 256, 0, st>>>(x, y, n, fmt, lut, scale, amax);
        return launch_status();
    }
    const size_t nvec = n / kPer;
    if constexpr (KIND == QT_FMT_LUT) {
        // (Odd-symmetric maps could stage only their non-negative half -- 64 KiB, two workgroups per CU.  Measured SLOWER on bf16
        // [4096, 11008]: posit8_1 4.37 against 4.79 TB/s with the whole table: putting the sign back costs three more VALU operations
        // per element than the second workgroup's waves recover.  Tuning build only.)
#ifdef QT_TUNING_BUILD
        static const int half_mode = getenv("QT_LUT_HALF") ? atoi(getenv("QT_LUT_HALF")) : 0;
        if (fmt.p0 == 1 && half_mode) {
            unsigned grid = grid_for(nvec, (size_t)kLutBlock * 4 * 4, 2);
            if (amax)
                fq_kernel<IO, kFmtLutHalf, true, kLutBlock, 4><<<grid, kLutBlock, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
            else
                fq_kernel<IO, kFmtLutHalf, false, kLutBlock, 4><<<grid, kLutBlock, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
            return launch_status();
        }
#endif
        unsigned grid = grid_for(nvec, (size_t)kLutBlock * 4 * 4, 1);
        if (amax)
            fq_kernel<IO, KIND, true, kLutBlock, 4><<<grid, kLutBlock, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
        else
            fq_kernel<IO, KIND, false, kLutBlock, 4><<<grid, kLutBlock, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
    } else {
#ifdef QT_TUNING_BUILD
        if constexpr (IO == kIoBf16 && KIND == QT_FMT_FP_SAT) {
            switch (g_variant) {
                case 1: return launch_variant<IO, KIND, 256, 4, 2>(x, y, n, fmt, lut, scale, amax, st);
                case 2: return launch_variant<IO, KIND, 256, 4, 3>(x, y, n, fmt, lut, scale, amax, st);
                case 3: return launch_variant<IO, KIND, 256, 8, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 4: return launch_variant<IO, KIND, 512, 4, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 5: return launch_variant<IO, KIND, 256, 2, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 6: return launch_variant<IO, KIND, 256, 1, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 7: return launch_variant<IO, KIND, 1024, 2, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 8: return launch_variant<IO, KIND, 256, 8, 3>(x, y, n, fmt, lut, scale, amax, st);
                case 9: return launch_variant<IO, KIND, 256, 2, 3>(x, y, n, fmt, lut, scale, amax, st);
                case 10: return launch_variant<IO, KIND, 256, 4, 1>(x, y, n, fmt, lut, scale, amax, st);
                case 11: return launch_variant<IO, KIND, 256, 1, 2>(x, y, n, fmt, lut, scale, amax, st);
                case 12: return launch_variant<IO, KIND, 256, 1, 3>(x, y, n, fmt, lut, scale, amax, st);
                case 13: return launch_variant<IO, KIND, 512, 1, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 14: return launch_variant<IO, KIND, 1024, 1, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 15: return launch_variant<IO, KIND, 128, 1, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 16: return launch_variant<IO, KIND, 128, 2, 0>(x, y, n, fmt, lut, scale, amax, st);
                case 17: return launch_variant<IO, KIND, 64, 1, 0>(x, y, n, fmt, lut, scale, amax, st);
                default: break;
            }
        }
#endif
        unsigned grid = grid_for(nvec, (size_t)kAluBlock * kUnroll, g_blocks_per_cu);
        if (amax && nvec <= kRowsDirectMaxVecs && obs_unroll() == 2)     // observed short pass: an eighth of the workgroups, two loads in flight per lane
            fq_kernel<IO, KIND, true, 1024, 2><<<grid_for(nvec, 2048, 2), 1024, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
        else if (amax && nvec <= kRowsDirectMaxVecs)     // observed short pass: a quarter of the workgroups, i.e. of the same-address atomics
            fq_kernel<IO, KIND, true, 1024><<<grid_for(nvec, 1024, 2), 1024, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
        else if (amax)
            fq_kernel<IO, KIND, true, kAluBlock><<<grid, kAluBlock, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
        else
            fq_kernel<IO, KIND, false, kAluBlock><<<grid, kAluBlock, 0, st>>>(x, y, nvec, n, fmt, lut, scale, amax);
    }
    return launch_status();
}

template <int IO>
int launch_fq(const void *x, void *y, size_t n, const qt_format *fmt, const uint16_t *lut, const float *scale,
              uint32_t *amax, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !fmt || (!y && !amax)) return QT_ERR_BAD_ARG;
    constexpr size_t esz = IO == kIoBf16 ? 2 : 4;
    if (((uintptr_t)x % esz) || ((uintptr_t)y % esz)) return QT_ERR_UNALIGNED;
    hipStream_t st = (hipStream_t)stream;
    if (!y) {   // observe only: no rounding, no scale
        qt_format ident = {QT_FMT_IDENTITY, 0, 0, 0.0f, 0.0f};
        return launch_fq_kind<IO, QT_FMT_IDENTITY>(x, nullptr, n, ident, nullptr, nullptr, amax, st);
    }
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut && y) return QT_ERR_BAD_ARG;
            return launch_fq_kind<IO, QT_FMT_LUT>(x, y, n, *fmt, lut, scale, amax, st);
        case QT_FMT_FP_SAT: return launch_fq_kind<IO, QT_FMT_FP_SAT>(x, y, n, *fmt, lut, scale, amax, st);
        case QT_FMT_INT: return launch_fq_kind<IO, QT_FMT_INT>(x, y, n, *fmt, lut, scale, amax, st);
        case QT_FMT_IDENTITY: return launch_fq_kind<IO, QT_FMT_IDENTITY>(x, y, n, *fmt, lut, scale, amax, st);
        default: return QT_ERR_BAD_ARG;
    }
}

template <int IO, int KIND>
int launch_pc_kind(const void *x, void *y, size_t outer, size_t C, size_t inner, const qt_format &fmt,
                   const uint16_t *lut, const float *scale, uint32_t *amax, hipStream_t st) {
    const size_t n = outer * C * inner;
    constexpr size_t kPer = IO == kIoBf16 ? 8 : 4;
    if (inner == 1) {
        unsigned grid = grid_for(n, 256 * 4, 8);
        fq_pc_last_kernel<IO, KIND><<<grid, 256, 0, st>>>(x, y, n, C, fmt, lut, scale, amax);
    } else if (inner % kPer == 0 && inner >= 64 * kPer && ((((uintptr_t)x | (uintptr_t)y) & 15u) == 0)) {
        if constexpr (KIND == QT_FMT_LUT) {
                if ((fmt.p1 & 1) && y) {
                unsigned grid = grid_for(outer * C, 1, 16);
                fq_pc_vec_kernel<IO, kFmtRows><<<grid, 256, 0, st>>>((const uint4 *)x, (uint4 *)y, outer * C, C, inner / kPer, fmt, lut, scale, amax);
                return launch_status();
            }
            if (y && outer * C * inner >= kLutLdsMinElems) {          // the 128 KiB table in LDS, a wave per row (3.0 -> 3.8 TB/s)
                const size_t rows = outer * C;
                unsigned grid = grid_for(rows, 16, 1);
                fq_pc_vec_lds_kernel<IO, QT_FMT_LUT><<<grid, 1024, 0, st>>>((const uint4 *)x, (uint4 *)y, rows, C, inner / kPer, fmt,
                                                                          lut, scale, amax);
                return launch_status();
            }
        }
        unsigned grid = grid_for(outer * C, 1, 16);
        fq_pc_vec_kernel<IO, KIND><<<grid, 256, 0, st>>>((const uint4 *)x, (uint4 *)y, outer * C, C, inner / kPer, fmt, lut,
                                                         scale, amax);
    } else {
        unsigned grid = grid_for(outer * C, 1, 16);
        fq_pc_kernel<IO, KIND><<<grid, 256, 0, st>>>(x, y, outer, C, inner, fmt, lut, scale, amax);
    }
    return launch_status();
}

template <int IO>
int launch_pc(const void *x, void *y, size_t outer, size_t C, size_t inner, const qt_format *fmt, const uint16_t *lut,
              const float *scale, uint32_t *amax, void *stream) {
    if (outer * C * inner == 0) return QT_OK;
    if (!x || !fmt || (!y && !amax)) return QT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut && y) return QT_ERR_BAD_ARG;
            return launch_pc_kind<IO, QT_FMT_LUT>(x, y, outer, C, inner, *fmt, lut, scale, amax, st);
        case QT_FMT_FP_SAT: return launch_pc_kind<IO, QT_FMT_FP_SAT>(x, y, outer, C, inner, *fmt, lut, scale, amax, st);
        case QT_FMT_INT: return launch_pc_kind<IO, QT_FMT_INT>(x, y, outer, C, inner, *fmt, lut, scale, amax, st);
        case QT_FMT_IDENTITY:
            return launch_pc_kind<IO, QT_FMT_IDENTITY>(x, y, outer, C, inner, *fmt, lut, scale, amax, st);
        default: return QT_ERR_BAD_ARG;
    }
}

template <int IO>
int launch_qdq(const void *x, void *y, size_t n, const QdqArgs &a, const void *scale, const void *zp, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !y || !scale) return QT_ERR_BAD_ARG;
    unsigned grid = grid_for(n, 256 * 4, 8);
    qdq_kernel<IO><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, a, scale, zp);
    return launch_status();
}


template <int IO>
int launch_mx(const void *x, void *y, void *sf, size_t rows, size_t cols, int bs, const qt_format *fmt, const uint16_t *lut,
              float quant_max, const uint16_t *scale_lut, void *stream) {
    if (rows * cols == 0) return QT_OK;
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    if (!x || !y || !sf || !fmt || bs < kPer || (bs & (bs - 1)) || bs > 64 * kPer) return QT_ERR_BAD_ARG;
    if (fmt->kind == QT_FMT_LUT && !lut) return QT_ERR_BAD_ARG;
    if ((cols % (size_t)bs) || (((uintptr_t)x | (uintptr_t)y) & 15u)) return QT_ERR_UNALIGNED;
    const size_t nvec = rows * cols / kPer;
    const int group = bs / kPer;
    const unsigned grid = grid_for(nvec, 256, 32);
    const uint4 *xv = (const uint4 *)x;
    uint4 *yv = (uint4 *)y;
    hipStream_t st = (hipStream_t)stream;
    switch (fmt->kind) {
        case QT_FMT_LUT: fq_mx_kernel<IO, QT_FMT_LUT><<<grid, 256, 0, st>>>(xv, yv, sf, nvec, group, *fmt, lut, quant_max, scale_lut); break;
        case QT_FMT_FP_SAT: fq_mx_kernel<IO, QT_FMT_FP_SAT><<<grid, 256, 0, st>>>(xv, yv, sf, nvec, group, *fmt, lut, quant_max, scale_lut); break;
        case QT_FMT_INT: fq_mx_kernel<IO, QT_FMT_INT><<<grid, 256, 0, st>>>(xv, yv, sf, nvec, group, *fmt, lut, quant_max, scale_lut); break;
        case QT_FMT_IDENTITY: fq_mx_kernel<IO, QT_FMT_IDENTITY><<<grid, 256, 0, st>>>(xv, yv, sf, nvec, group, *fmt, lut, quant_max, scale_lut); break;
        default: return QT_ERR_BAD_ARG;
    }
    return launch_status();
}

}  // namespace

extern "C" {

#ifdef QT_TUNING_BUILD
void qt_internal_set_variant(int variant, int blocks_per_cu) {       // tools/exp_stream.py, tools/exp_fq8_grid.py: launch-geometry sweeps
    g_variant = variant;
    g_blocks_per_cu = blocks_per_cu > 0 ? blocks_per_cu : 32;
}
#endif

int qt_scale_update(float *history_dev, int L, int C, float *scale_dev, float quant_max, int force_pow2, void *stream) {
    if (!history_dev || !scale_dev || L < 1 || C < 1) return QT_ERR_BAD_ARG;
    unsigned grid = (unsigned)((C + 127) / 128);
    scale_update_kernel<<<grid, 128, 0, (hipStream_t)stream>>>(history_dev, L, C, scale_dev, quant_max, force_pow2);
    return launch_status();
}

int qt_scale_update_multi(float *const *history_ptrs_dev, const int *L_dev, const int *C_dev, float *const *scale_ptrs_dev,
                          const float *quant_max_dev, const int *force_pow2_dev, int count, void *stream) {
    if (count == 0) return QT_OK;
    if (!history_ptrs_dev || !L_dev || !C_dev || !scale_ptrs_dev || !quant_max_dev || !force_pow2_dev || count < 0) return QT_ERR_BAD_ARG;
    scale_update_multi_kernel<<<(unsigned)count, 128, 0, (hipStream_t)stream>>>(history_ptrs_dev, L_dev, C_dev, scale_ptrs_dev,
                                                                               quant_max_dev, force_pow2_dev);
    return launch_status();
}

int qt_fake_quant_bf16(const uint16_t *x, uint16_t *y, size_t n, const qt_format *fmt, const uint16_t *lut,
                       const float *scale, uint32_t *amax, void *stream) {
    return launch_fq<kIoBf16>(x, y, n, fmt, lut, scale, amax, stream);
}
int qt_fake_quant_f32(const float *x, float *y, size_t n, const qt_format *fmt, const uint16_t *lut, const float *scale,
                      uint32_t *amax, void *stream) {
    return launch_fq<kIoF32>(x, y, n, fmt, lut, scale, amax, stream);
}
int qt_fake_quant_pc_bf16(const uint16_t *x, uint16_t *y, size_t outer, size_t C, size_t inner, const qt_format *fmt,
                          const uint16_t *lut, const float *scale, uint32_t *amax, void *stream) {
    return launch_pc<kIoBf16>(x, y, outer, C, inner, fmt, lut, scale, amax, stream);
}
int qt_fake_quant_pc_f32(const float *x, float *y, size_t outer, size_t C, size_t inner, const qt_format *fmt,
                         const uint16_t *lut, const float *scale, uint32_t *amax, void *stream) {
    return launch_pc<kIoF32>(x, y, outer, C, inner, fmt, lut, scale, amax, stream);
}

int qt_fake_quant_bf16_fp8(const uint16_t *x, uint16_t *y, uint8_t *y8, size_t n, const qt_format *fmt,
                           const float *scale, uint32_t *amax, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !y8 || !fmt || fmt->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if ((n & 15) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)y8) & 15u)) return QT_ERR_UNALIGNED;
    const size_t nvec = n / 16;
    const unsigned grid = grid_for(nvec, 256, g_blocks_per_cu);
    hipStream_t st = (hipStream_t)stream;
    const uint4 *xv = (const uint4 *)x;
    uint4 *yv = (uint4 *)y;
    uint4 *y8v = (uint4 *)y8;
    bool hw = true;
#ifdef QT_TUNING_BUILD
    hw = getenv("QT_FQ8_CLOSED_FORM") == nullptr;                        // tools/ only: A/B for DESIGN.md's measurement
#endif
#define QT_FQ8(OBS, BOTH, E5)                                                                                  \
    fq8_kernel<OBS, BOTH, E5><<<grid, 256, 0, st>>>(xv, yv, y8v, nvec, *fmt, scale, amax, hw)
    if (e5m2) {
        if (amax) { if (y) QT_FQ8(true, true, true); else QT_FQ8(true, false, true); }
        else      { if (y) QT_FQ8(false, true, true); else QT_FQ8(false, false, true); }
    } else {
        if (amax) { if (y) QT_FQ8(true, true, false); else QT_FQ8(true, false, false); }
        else      { if (y) QT_FQ8(false, true, false); else QT_FQ8(false, false, false); }
    }
#undef QT_FQ8
    return launch_status();
}

// strips x bands of one chain launch: about one workgroup per CU, bands of whole 64-row groups, at most 32 bands
static void chain_geometry(long rows, long cols, int &strips, int &bands, long &band_rows, int pre_op = 0) {
    strips = (int)((cols / 8 + kChainStripV - 1) / kChainStripV);
    const long groups = (rows + kChainRowLanes - 1) / kChainRowLanes;           // 64-row groups
    // measured: 96 and 384 workgroups are both slower than 192 on [2048, 768] (profiles/r05_chain_geometry.txt; 192 and 256 cut that shape
    // the same way: 12 strips x 16 bands).  On [2048, 3072] -- the GELU launches, bound by their erf / exp arithmetic, not by memory
    // (tools/exp_train_stamps.py: 20 of 26 us between the loads and the last store) -- 192 left a quarter of the CUs idle: 48 strips x 6
    // bands instead of x 4 took 0.12 ms off the configs[4] step (profiles/r06_train_step_ab.txt).  Tried on the GELU launches afterwards
    // without gain in the replayed step (profiles/r06_chain_experiments.txt): four rows in flight per lane instead of two (133 registers:
    // 22.9 -> 29.7 us), a strip-less variant for launches without column sums, GELU(x) / GELU'(x) gathered from tables (global memory;
    // a 19 KiB slice in LDS).  One wave's arithmetic runs under the load latency of the SIMD's other wave: what such a launch spends is
    // its dependent load -> arithmetic -> store rounds plus ~7 us of fixed start and end.
    int target = 256;
#ifdef QT_TUNING_BUILD
    if (const char *e = getenv("QT_CHAIN_WGS")) target = atoi(e) > 0 ? atoi(e) : target;          // tools/ only
    if (pre_op != 0)
        if (const char *e = getenv("QT_CHAIN_WGS_PRE")) target = atoi(e) > 0 ? atoi(e) : target;  // the GELU launches alone
#endif
    (void)pre_op;
    long want = (target + strips - 1) / strips;
    if (want < 1) want = 1;
    if (want > 32) want = 32;
    if (want > groups) want = groups;
    const long per = (groups + want - 1) / want;
    band_rows = per * kChainRowLanes;
    bands = (int)((groups + per - 1) / per);
}

static int chain_launch(const uint16_t *x_dev, const uint16_t *x2_dev, int pre_op, uint16_t *pre_out_dev, long rows, long cols,
                        const qt_chain_stage *stages, int nstage, const qt_format *fmt, const uint16_t *lut_dev, int colsum_stage, float colsum_max,
                        uint16_t *colsum_out_dev, void *ws_dev, size_t ws_bytes, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x_dev || !stages || !fmt || nstage < 1 || nstage > kChainMax || rows < 0 || cols < 8 || cols % 8 != 0) return QT_ERR_BAD_ARG;
    if (pre_op < 0 || pre_op > 2 || (pre_op == 2 && !x2_dev)) return QT_ERR_BAD_ARG;
    if (((uintptr_t)x_dev | (uintptr_t)x2_dev | (uintptr_t)pre_out_dev) & 15u) return QT_ERR_UNALIGNED;
    ChainArgs a{};
    a.x = (const uint4 *)x_dev; a.rows = rows; a.cv = (int)(cols / 8); a.nstage = nstage;
    a.pre_op = pre_op; a.x2 = (const uint4 *)x2_dev; a.pre_out = (uint4 *)pre_out_dev;
    a.table_in_lds = 1;
#ifdef QT_TUNING_BUILD
    if (const char *e = getenv("QT_CHAIN_LDS")) a.table_in_lds = atoi(e);                          // tools/ only
    if (const char *e = getenv("QT_CHAIN_ABLATE")) a.ablate = atoi(e);                             // tools/ only: timing, garbage results
    a.dbg = ew_stamp_region();
#endif
    for (int i = 0; i < nstage; ++i) {
        if (stages[i].src >= i || stages[i].src < -1) return QT_ERR_BAD_ARG;
        if ((uintptr_t)stages[i].out_dev & 15u) return QT_ERR_UNALIGNED;
        a.st[i] = ChainStageDev{stages[i].scale_f32_dev, stages[i].amax_bits_dev, (uint4 *)stages[i].out_dev, stages[i].src};
    }
    chain_geometry(rows, cols, a.strips, a.bands, a.band_rows, pre_op);
    a.colsum_stage = -1;
    if (colsum_stage >= 0) {
        if (colsum_stage >= nstage || !colsum_out_dev || !(colsum_max > 0.0f) || !(colsum_max < 3.0e38f)) return QT_ERR_BAD_ARG;
        if (a.bands > 1 && (!ws_dev || ws_bytes < qt_fake_quant_chain_ws_bytes(rows, cols))) return QT_ERR_BAD_ARG;
        if (((uintptr_t)ws_dev & 15u) || ((uintptr_t)colsum_out_dev & 1u)) return QT_ERR_UNALIGNED;
        a.colsum_stage = colsum_stage; a.colsum_max = colsum_max; a.colsum_out = colsum_out_dev;
        a.acc = (long long *)ws_dev;
        a.ticket = (unsigned int *)((char *)ws_dev + (size_t)a.strips * kChainStripV * 8 * sizeof(long long));
    }
    const unsigned grid = (unsigned)(a.strips * a.bands);
    hipStream_t st = (hipStream_t)stream;
#define QT_CHAIN(K, NS) fq_chain_kernel<K, NS><<<grid, kChainBlock, 0, st>>>(a, *fmt, lut_dev)
#define QT_CHAIN_NS(K)                                                                  \
    switch (nstage) {                                                                   \
        case 1: QT_CHAIN(K, 1); break;                                                  \
        case 2: QT_CHAIN(K, 2); break;                                                  \
        case 3: QT_CHAIN(K, 3); break;                                                  \
        default: QT_CHAIN(K, 4); break;                                                 \
    }
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut_dev || !(fmt->p1 & 1)) return QT_ERR_BAD_DTYPE;             // table formats in their row form only
            QT_CHAIN_NS(kFmtRows)
            break;
        case QT_FMT_FP_SAT: QT_CHAIN_NS(QT_FMT_FP_SAT) break;
        case QT_FMT_INT: QT_CHAIN_NS(QT_FMT_INT) break;
        default: return QT_ERR_BAD_DTYPE;
    }
#undef QT_CHAIN_NS
#undef QT_CHAIN
    return launch_status();
}

int qt_fake_quant_chain_bf16(const uint16_t *x_dev, long rows, long cols, const qt_chain_stage *stages, int nstage, const qt_format *fmt,
                             const uint16_t *lut_dev, int colsum_stage, float colsum_max, uint16_t *colsum_out_dev, void *ws_dev, size_t ws_bytes,
                             void *stream) {
    return chain_launch(x_dev, nullptr, 0, nullptr, rows, cols, stages, nstage, fmt, lut_dev, colsum_stage, colsum_max, colsum_out_dev, ws_dev,
                        ws_bytes, stream);
}

int qt_gelu_chain_bf16(const uint16_t *x_dev, uint16_t *y_dev, long rows, long cols, const qt_chain_stage *stages, int nstage,
                       const qt_format *fmt, const uint16_t *lut_dev, void *stream) {
    return chain_launch(x_dev, nullptr, 1, y_dev, rows, cols, stages, nstage, fmt, lut_dev, -1, 0.0f, nullptr, nullptr, 0, stream);
}

int qt_gelu_backward_chain_bf16(const uint16_t *grad_out_dev, const uint16_t *x_dev, uint16_t *grad_in_dev, long rows, long cols,
                                const qt_chain_stage *stages, int nstage, const qt_format *fmt, const uint16_t *lut_dev, int colsum_stage,
                                float colsum_max, uint16_t *colsum_out_dev, void *ws_dev, size_t ws_bytes, void *stream) {
    return chain_launch(grad_out_dev, x_dev, 2, grad_in_dev, rows, cols, stages, nstage, fmt, lut_dev, colsum_stage, colsum_max, colsum_out_dev,
                        ws_dev, ws_bytes, stream);
}

static int chain_stage_args(const qt_chain_stage *stages, int nstage, ChainStageDev (&st)[kChainMax]) {
    if (nstage < 0 || nstage > kChainMax || (nstage > 0 && !stages)) return QT_ERR_BAD_ARG;
    for (int i = 0; i < nstage; ++i) {
        if (stages[i].src >= i || stages[i].src < -1) return QT_ERR_BAD_ARG;
        if ((uintptr_t)stages[i].out_dev & 15u) return QT_ERR_UNALIGNED;
        st[i] = ChainStageDev{stages[i].scale_f32_dev, stages[i].amax_bits_dev, (uint4 *)stages[i].out_dev, stages[i].src};
    }
    return QT_OK;
}

#define QT_LN_DISPATCH(KERNEL, ARGS, GRID)                                                                                      \
    switch (fmt->kind) {                                                                                                        \
        case QT_FMT_LUT:                                                                                                        \
            if (!lut_dev || !(fmt->p1 & 1)) return QT_ERR_BAD_DTYPE;                                                            \
            switch (nstage) {                                                                                                   \
                case 1: KERNEL<kFmtRows, 1, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;          \
                case 2: KERNEL<kFmtRows, 2, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;          \
                case 3: KERNEL<kFmtRows, 3, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;          \
                default: KERNEL<kFmtRows, 4, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;         \
            }                                                                                                                   \
            break;                                                                                                              \
        case QT_FMT_FP_SAT:                                                                                                     \
            switch (nstage) {                                                                                                   \
                case 1: KERNEL<QT_FMT_FP_SAT, 1, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;     \
                case 2: KERNEL<QT_FMT_FP_SAT, 2, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;     \
                case 3: KERNEL<QT_FMT_FP_SAT, 3, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;     \
                default: KERNEL<QT_FMT_FP_SAT, 4, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;    \
            }                                                                                                                   \
            break;                                                                                                              \
        case QT_FMT_INT:                                                                                                        \
            switch (nstage) {                                                                                                   \
                case 1: KERNEL<QT_FMT_INT, 1, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;        \
                case 2: KERNEL<QT_FMT_INT, 2, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;        \
                case 3: KERNEL<QT_FMT_INT, 3, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;        \
                default: KERNEL<QT_FMT_INT, 4, 512><<<(unsigned)((GRID)(512)), 512, 0, st>>>(ARGS, *fmt, lut_dev); break;       \
            }                                                                                                                   \
            break;                                                                                                              \
        default: return QT_ERR_BAD_DTYPE;                                                                                       \
    }

int qt_layernorm_train_bf16(const uint16_t *x_dev, const uint16_t *weight_dev, const uint16_t *bias_dev, uint16_t *y_dev, float *mean_dev,
                            float *rstd_dev, long rows, long cols, float eps, const qt_chain_stage *stages, int nstage, const qt_format *fmt,
                            const uint16_t *lut_dev, const uint16_t *residual_dev, uint16_t *sum_dev, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x_dev || !weight_dev || !bias_dev || !y_dev || !mean_dev || !rstd_dev || !fmt || rows < 0 || cols < 8 || cols % 8 || cols > 64 * 8 * kLnMaxVec ||
        nstage < 1)
        return QT_ERR_BAD_ARG;
    if (((uintptr_t)x_dev | (uintptr_t)weight_dev | (uintptr_t)bias_dev | (uintptr_t)y_dev | (uintptr_t)residual_dev | (uintptr_t)sum_dev) & 15u) return QT_ERR_UNALIGNED;
    if ((residual_dev != nullptr) != (sum_dev != nullptr)) return QT_ERR_BAD_ARG;
    LnTrainArgs a{};
    a.x = (const uint4 *)x_dev; a.w = (const uint4 *)weight_dev; a.b = (const uint4 *)bias_dev; a.y = (uint4 *)y_dev;
    a.x2 = (const uint4 *)residual_dev; a.sum = (uint4 *)sum_dev;
    a.mean = mean_dev; a.rstd = rstd_dev; a.rows = rows; a.nvec = (int)(cols / 8); a.inv_cols = 1.0f / (float)cols; a.eps = eps;
    if (const int rc = chain_stage_args(stages, nstage, a.st)) return rc;
    hipStream_t st = (hipStream_t)stream;
    auto grid = [&](int block) { return (rows + block / 64 - 1) / (block / 64); };
    QT_LN_DISPATCH(ln_train_fwd_kernel, a, grid)
    return launch_status();
}

// workgroups of the backward launch (512 threads = 8 waves, one row per wave): its partial sums are part_dev[that many][3][cols] fp32
long qt_layernorm_train_backward_groups(long rows) { return rows <= 0 ? 0 : (rows + 7) / 8; }

int qt_layernorm_train_backward_bf16(const uint16_t *grad_out_dev, const uint16_t *x_dev, const uint16_t *weight_dev, const float *mean_dev,
                                     const float *rstd_dev, uint16_t *grad_in_dev, long rows, long cols, const qt_chain_stage *stages, int nstage,
                                     const qt_format *fmt, const uint16_t *lut_dev, int colsum_stage, float *part_dev, size_t part_bytes,
                                     uint16_t *grad_weight_dev, uint16_t *grad_bias_dev, uint16_t *colsum_out_dev, const qt_fanin_item *fan_items,
                                     int fan_count, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (fan_count < 0 || fan_count > 3 || (fan_count > 0 && !fan_items)) return QT_ERR_BAD_ARG;
    if (!grad_out_dev || !x_dev || !weight_dev || !mean_dev || !rstd_dev || !grad_in_dev || !fmt || !part_dev || !grad_weight_dev || !grad_bias_dev ||
        rows < 0 || cols < 8 || cols % 8 || cols > 64 * 8 * kLnMaxVec || nstage < 1 || colsum_stage >= nstage || (colsum_stage >= 0 && !colsum_out_dev))
        return QT_ERR_BAD_ARG;
    if (((uintptr_t)grad_out_dev | (uintptr_t)x_dev | (uintptr_t)weight_dev | (uintptr_t)grad_in_dev | (uintptr_t)part_dev) & 15u) return QT_ERR_UNALIGNED;
    const long groups = qt_layernorm_train_backward_groups(rows);
    if (part_bytes < (size_t)groups * 3 * cols * sizeof(float)) return QT_ERR_BAD_ARG;
    LnBwdArgs a{};
    a.dy = (const uint4 *)grad_out_dev; a.x = (const uint4 *)x_dev; a.w = (const uint4 *)weight_dev; a.mean = mean_dev; a.rstd = rstd_dev;
    a.dx = (uint4 *)grad_in_dev; a.rows = rows; a.nvec = (int)(cols / 8); a.inv_cols = 1.0f / (float)cols; a.rows_per_wave = 1;
    a.part = part_dev; a.colsum_stage = colsum_stage < 0 ? -1 : colsum_stage;
    if (const int rc = chain_stage_args(stages, nstage, a.st)) return rc;
    a.nfan = fan_count;
    for (int i = 0; i < fan_count; ++i) {
        if (!fan_items[i].x_dev) return QT_ERR_BAD_ARG;
        if ((uintptr_t)fan_items[i].x_dev & 15u) return QT_ERR_UNALIGNED;
        a.fan_x[i] = (const uint4 *)fan_items[i].x_dev;
        a.fan[i] = ChainStageDev{fan_items[i].fq ? fan_items[i].scale_f32_dev : nullptr, fan_items[i].fq ? fan_items[i].amax_bits_dev : nullptr, nullptr,
                                 fan_items[i].fq ? 1 : 0};
    }
    hipStream_t st = (hipStream_t)stream;
    auto grid = [&](int) { return groups; };
#ifdef QT_TUNING_BUILD
    a.dbg = ew_stamp_region();
#endif
    QT_LN_DISPATCH(ln_train_bwd_kernel, a, grid)
    if (const int rc = launch_status()) return rc;
    ln_train_reduce_kernel<<<dim3((unsigned)((cols + 63) / 64), colsum_stage >= 0 ? 3u : 2u), 256, 0, st>>>(part_dev, (int)groups, (int)cols, grad_weight_dev,
                                                                                                      grad_bias_dev, colsum_out_dev);
    return launch_status();
}

size_t qt_fake_quant_chain_ws_bytes(long rows, long cols) {
    if (rows <= 0 || cols < 8 || cols % 8) return 0;
    const size_t strips = (size_t)((cols / 8 + kChainStripV - 1) / kChainStripV);
    return strips * kChainStripV * 8 * sizeof(long long) + strips * sizeof(unsigned int);      // accumulators, then tickets
}

int qt_fake_quant_rows_bf16(const uint16_t *x, uint16_t *y, long d0, long d1, long d2, long inner, long s0, long s1,
                            long s2, const qt_format *fmt, const uint16_t *lut, const float *scale, uint32_t *amax,
                            void *stream) {
    if (d0 * d1 * d2 * inner == 0) return QT_OK;
    if (!x || !y || !fmt || d0 < 0 || d1 < 0 || d2 < 0 || inner < 0) return QT_ERR_BAD_ARG;
    if (fmt->kind == QT_FMT_LUT && !lut) return QT_ERR_BAD_ARG;
    if ((inner & 7) || ((s0 | s1 | s2) & 7) || (((uintptr_t)x | (uintptr_t)y) & 15u)) return QT_ERR_UNALIGNED;
    RowsArgs a{x, y, d1, d2, inner / 8, s0, s1, s2, (size_t)(d0 * d1 * d2 * (inner / 8))};
    const unsigned grid = grid_for(a.nvec, 256, 32);
    hipStream_t st = (hipStream_t)stream;
#define QT_ROWS(K)                                                                           \
    do {                                                                                     \
        if (amax && a.nvec <= kRowsDirectMaxVecs)                                            \
            fq_rows_kernel<K, true, 0, 1024><<<grid_for(a.nvec, 1024, 2), 1024, 0, st>>>(a, *fmt, lut, scale, amax);  \
        else if (amax) fq_rows_kernel<K, true><<<grid, 256, 0, st>>>(a, *fmt, lut, scale, amax);  \
        else fq_rows_kernel<K, false><<<grid, 256, 0, st>>>(a, *fmt, lut, scale, amax);      \
    } while (0)
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (fmt->p1 & 1) QT_ROWS(kFmtRows);          // row form behind the map
            else QT_ROWS(QT_FMT_LUT);
            break;
        case QT_FMT_FP_SAT: QT_ROWS(QT_FMT_FP_SAT); break;
        case QT_FMT_INT: QT_ROWS(QT_FMT_INT); break;
        case QT_FMT_IDENTITY: QT_ROWS(QT_FMT_IDENTITY); break;
        default: return QT_ERR_BAD_ARG;
    }
#undef QT_ROWS
    return launch_status();
}

int qt_fake_quant_bf16_fp8_multi(const uint16_t *const *xs, const size_t *ns, int count, uint8_t *y8, const qt_format *fmt,
                                 void *stream) {
    if (!xs || !ns || !y8 || !fmt || count < 1 || count > 4 || fmt->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    MultiArgs a{};
    size_t run = 0;
    for (int i = 0; i < 4; ++i) {
        a.first[i] = run;
        if (i < count) {
            if (!xs[i]) return QT_ERR_BAD_ARG;
            if ((ns[i] & 15) || ((uintptr_t)xs[i] & 15u)) return QT_ERR_UNALIGNED;
            a.x[i] = (const uint4 *)xs[i];
            a.npair[i] = ns[i] / 16;
            run += a.npair[i];
        }
    }
    a.first[4] = run;
    if (run == 0) return QT_OK;
    if ((uintptr_t)y8 & 15u) return QT_ERR_UNALIGNED;
    const unsigned grid = grid_for(run, 256, g_blocks_per_cu);
    hipStream_t st = (hipStream_t)stream;
    if (e5m2) fq8_multi_kernel<true><<<grid, 256, 0, st>>>(a, (uint4 *)y8, *fmt);
    else fq8_multi_kernel<false><<<grid, 256, 0, st>>>(a, (uint4 *)y8, *fmt);
    return launch_status();
}

int qt_fake_quant_multi_bf16_fp8(const qt_fq8_item *items_dev, int count, unsigned long long total_tiles, const qt_format *fmt, void *stream) {
    if (count == 0 || total_tiles == 0) return QT_OK;
    if (!items_dev || !fmt || count < 0 || total_tiles > 0x7FFFFFFFull || fmt->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if ((uintptr_t)items_dev & 7u) return QT_ERR_UNALIGNED;
    static_assert(sizeof(qt_fq8_item) == sizeof(qt_fq8_item_dev), "layout of qt_fq8_item");
    const qt_fq8_item_dev *it = (const qt_fq8_item_dev *)items_dev;
    hipStream_t st = (hipStream_t)stream;
    if (e5m2) fq8_items_kernel<true><<<(unsigned)total_tiles, 256, 0, st>>>(it, count, *fmt);
    else fq8_items_kernel<false><<<(unsigned)total_tiles, 256, 0, st>>>(it, count, *fmt);
    return launch_status();
}

int qt_fake_quant_multi_bf16(const qt_fq_item *items_dev, int count, unsigned long long total_tiles, const qt_format *fmt,
                             const uint16_t *lut_dev, void *stream) {
    if (count == 0 || total_tiles == 0) return QT_OK;
    if (!items_dev || !fmt || count < 0 || total_tiles > 0x7FFFFFFFull) return QT_ERR_BAD_ARG;
    if ((uintptr_t)items_dev & 7u) return QT_ERR_UNALIGNED;
    static_assert(sizeof(qt_fq_item) == sizeof(qt_fq_item_dev), "layout of qt_fq_item");
    const qt_fq_item_dev *it = (const qt_fq_item_dev *)items_dev;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)total_tiles;
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut_dev || !(fmt->p1 & 1)) return QT_ERR_BAD_ARG;          // table formats: the row form behind the map only
            fq_multi_kernel<kFmtRows><<<grid, kMultiBlock, 0, st>>>(it, count, *fmt, lut_dev);
            break;
        case QT_FMT_FP_SAT: fq_multi_kernel<QT_FMT_FP_SAT><<<grid, kMultiBlock, 0, st>>>(it, count, *fmt, lut_dev); break;
        case QT_FMT_INT: fq_multi_kernel<QT_FMT_INT><<<grid, kMultiBlock, 0, st>>>(it, count, *fmt, lut_dev); break;
        default: return QT_ERR_BAD_ARG;
    }
    return launch_status();
}

int qt_fake_quant_rows_bf16_fp8(const uint16_t *x, uint16_t *y, uint8_t *y8, long d0, long d1, long d2, long inner, long s0,
                                long s1, long s2, const qt_format *fmt, void *stream) {
    if (d0 * d1 * d2 * inner == 0) return QT_OK;
    if (!x || !y8 || !fmt || d0 < 0 || d1 < 0 || d2 < 0 || inner < 0 || fmt->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if ((inner & 7) || ((s0 | s1 | s2) & 7) || (((uintptr_t)x | (uintptr_t)y) & 15u) || ((uintptr_t)y8 & 7u)) return QT_ERR_UNALIGNED;
    RowsArgs a{x, y, d1, d2, inner / 8, s0, s1, s2, (size_t)(d0 * d1 * d2 * (inner / 8))};          // y may be NULL: FP8 codes only
    const unsigned grid = grid_for(a.nvec, 256, 32);
    hipStream_t st = (hipStream_t)stream;
    if (e5m2) fq_rows_kernel<QT_FMT_FP_SAT, false, 2><<<grid, 256, 0, st>>>(a, *fmt, nullptr, nullptr, nullptr, (uint2 *)y8);
    else fq_rows_kernel<QT_FMT_FP_SAT, false, 1><<<grid, 256, 0, st>>>(a, *fmt, nullptr, nullptr, nullptr, (uint2 *)y8);
    return launch_status();
}

int qt_fake_quant_mx_bf16(const uint16_t *x, uint16_t *y, uint16_t *sf, size_t rows, size_t cols, int block_size,
                          const qt_format *fmt, const uint16_t *lut, float quant_max, const uint16_t *scale_lut,
                          void *stream) {
    return launch_mx<kIoBf16>(x, y, sf, rows, cols, block_size, fmt, lut, quant_max, scale_lut, stream);
}
int qt_fake_quant_mx_f32(const float *x, float *y, float *sf, size_t rows, size_t cols, int block_size,
                         const qt_format *fmt, const uint16_t *lut, float quant_max, const uint16_t *scale_lut,
                         void *stream) {
    return launch_mx<kIoF32>(x, y, sf, rows, cols, block_size, fmt, lut, quant_max, scale_lut, stream);
}

// vmap == fake-quant with scale 1 and no observer (x/1 and r*1 are exact)
int qt_vmap_bf16(const uint16_t *x, uint16_t *y, size_t n, const qt_format *fmt, const uint16_t *lut, void *stream) {
    if (n && !y) return QT_ERR_BAD_ARG;
    return launch_fq<kIoBf16>(x, y, n, fmt, lut, nullptr, nullptr, stream);
}
int qt_vmap_f32(const float *x, float *y, size_t n, const qt_format *fmt, const uint16_t *lut, void *stream) {
    if (n && !y) return QT_ERR_BAD_ARG;
    return launch_fq<kIoF32>(x, y, n, fmt, lut, nullptr, nullptr, stream);
}
int qt_vmap_f16(const uint16_t *x, uint16_t *y, size_t n, const qt_format *fmt, const uint16_t *lut, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !y || !fmt || (fmt->kind == QT_FMT_LUT && !lut)) return QT_ERR_BAD_ARG;
    unsigned grid = grid_for(n, 256 * 4, 8);
    vmap_f16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)x, (_Float16 *)y, n, *fmt,
                                                          fmt->kind == QT_FMT_LUT ? lut : nullptr);
    return launch_status();
}

int qt_quantize_bf16(const uint16_t *x, uint16_t *y, size_t n, const qt_format *fmt, const uint16_t *lut,
                     const uint16_t *scale, const uint16_t *zp, void *stream) {
    if (!fmt || (fmt->kind == QT_FMT_LUT && !lut)) return QT_ERR_BAD_ARG;
    QdqArgs a{nullptr, fmt->kind == QT_FMT_LUT ? lut : nullptr, *fmt, 1, 0};
    return launch_qdq<kIoBf16>(x, y, n, a, scale, zp, stream);
}
int qt_quantize_f32(const float *x, float *y, size_t n, const qt_format *fmt, const uint16_t *lut, const float *scale,
                    const float *zp, void *stream) {
    if (!fmt || (fmt->kind == QT_FMT_LUT && !lut)) return QT_ERR_BAD_ARG;
    QdqArgs a{nullptr, fmt->kind == QT_FMT_LUT ? lut : nullptr, *fmt, 1, 0};
    return launch_qdq<kIoF32>(x, y, n, a, scale, zp, stream);
}
int qt_dequantize_bf16(const uint16_t *x, uint16_t *y, size_t n, const uint16_t *scale, const uint16_t *zp,
                       const uint16_t *in_lut, const uint16_t *out_lut, void *stream) {
    QdqArgs a{in_lut, out_lut, {QT_FMT_LUT, 0, 0, 0.f, 0.f}, out_lut ? 1 : 0, 1};
    return launch_qdq<kIoBf16>(x, y, n, a, scale, zp, stream);
}
int qt_dequantize_f32(const float *x, float *y, size_t n, const float *scale, const float *zp, const uint16_t *in_lut,
                      const uint16_t *out_lut, void *stream) {
    QdqArgs a{in_lut, out_lut, {QT_FMT_LUT, 0, 0, 0.f, 0.f}, out_lut ? 1 : 0, 1};
    return launch_qdq<kIoF32>(x, y, n, a, scale, zp, stream);
}

int qt_round_fp8_f32(const float *x, float *y, size_t n, int mbits, float fp8_max, float fp8_min, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !y || mbits < 1 || mbits > 22 || !(fp8_min > 0.0f)) return QT_ERR_BAD_ARG;
    unsigned grid = grid_for(n, 256 * 4, 8);
    round_fp8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, mbits, qt_internal_fp8_emin(fp8_min), fp8_max);
    return launch_status();
}
int qt_round_posit_f32(const float *x, float *y, size_t n, int nbits, int es, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !y || nbits < 3 || nbits > 24 || es < 0 || es > 4 || ((nbits - 2) << es) > 126) return QT_ERR_BAD_ARG;
    unsigned grid = grid_for(n, 256 * 4, 8);
    round_posit_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, nbits, es, qt_internal_posit_threshold(nbits, es));
    return launch_status();
}

int qt_posit_quantize_f32(const float *x, float *y, int32_t *pbits, size_t n, int nbits, int es, int round_to_even, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || !y || nbits < 3 || nbits > 24 || es < 0 || es > 4 || ((nbits - 2) << es) > 126) return QT_ERR_BAD_ARG;
    unsigned grid = grid_for(n, 256 * 4, 8);
    posit_bits_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, pbits, n, nbits, es,
                                                          round_to_even ? qt_internal_posit_threshold(nbits, es) : 0.0f);
    return launch_status();
}

int qt_bench_fake_quant_bf16(const uint16_t *x, uint16_t *y, size_t n, const qt_format *fmt, const uint16_t *lut,
                             const float *scale, uint32_t *amax, int iters, size_t pool_stride, int pool_count,
                             void *stream, float *ms_out) {
    if (!ms_out || iters < 1 || pool_count < 1) return QT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
    int rc = QT_OK;
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters && rc == QT_OK; ++i) {
        const size_t off = (size_t)(i % pool_count) * pool_stride;
        rc = launch_fq<kIoBf16>(x + off, y ? y + off : nullptr, n, fmt, lut, scale, amax, stream);
    }
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != QT_OK) return rc;
    if (e != hipSuccess) return (int)e;
    *ms_out = ms / (float)iters;
    return QT_OK;
}

int qt_bench_fake_quant_bf16_fp8(const uint16_t *x, uint16_t *y, uint8_t *y8, size_t n, const qt_format *fmt,
                                 const float *scale, uint32_t *amax, int iters, size_t pool_stride, int pool_count,
                                 void *stream, float *ms_out) {
    if (!ms_out || iters < 1 || pool_count < 1) return QT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
    int rc = QT_OK;
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters && rc == QT_OK; ++i) {
        const size_t off = (size_t)(i % pool_count) * pool_stride;
        rc = qt_fake_quant_bf16_fp8(x + off, y ? y + off : nullptr, y8 + off, n, fmt, scale, amax, stream);
    }
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != QT_OK) return rc;
    if (e != hipSuccess) return (int)e;
    *ms_out = ms / (float)iters;
    return QT_OK;
}

}  // extern "C"
