// qt_chain.h -- one 16-byte vector through a fake-quantizer (shared by the elementwise, chain and training kernels), and the stage
// description of qt_fake_quant_chain_bf16: several fake-quantizer calls of a training step applied to values a kernel holds in registers.
#pragma once
#include "qt_device.h"

namespace {

constexpr int kIoBf16 = 0;
constexpr int kIoF32 = 1;

// ---- cross-workgroup column sums in 64-bit fixed point (bias gradients: the chain kernel's bands, the attention backward's batch) -----
// `arrivals` workgroups each add one partial sum per column to a zeroed accumulator; the one that draws the last ticket reads it.
// One unit = 2^(E - shift), E = exponent of the largest value an element can take.  A partial sum that is NaN / Inf adds `poison`
// = 2^p instead, p = 62 - ceil(log2 arrivals): at most `arrivals` of them cannot wrap (sum < 2^63).  shift = min(42, p - 2 -
// ceil(log2 rows)) keeps the sum of ALL finite terms (|.| <= rows 2^shift) at or below 2^(p-2), so a total >= 2^(p-1) says "some
// partial was not finite" whatever else was added, and a clean total can never reach it.  (Round 5 added 2^62 per flagged partial:
// 4, 8, 16 or 32 of them wrapped to exactly zero and a fully-NaN gradient came out as a finite bias gradient.)
struct FxPlan {
    int shift;
    long long poison;
};
__device__ __forceinline__ FxPlan fx_plan(long rows, int arrivals) {
    const int lr = rows > 1 ? 64 - __clzll((unsigned long long)(rows - 1)) : 0;
    const int lk = arrivals > 1 ? 32 - __clz((unsigned)(arrivals - 1)) : 0;
    const int p = 62 - lk;
    const int sh = p - 2 - lr;
    return FxPlan{sh < 42 ? sh : 42, 1ll << p};
}
__device__ __forceinline__ long long fx_encode(float part, int E, const FxPlan &f) {
    const bool finite = part == part && fabsf(part) < 3.0e38f;
    return finite ? (long long)rintf(ldexpf(part, f.shift - E)) : f.poison;
}
__device__ __forceinline__ float fx_decode(long long fx, int E, const FxPlan &f) {
    return fx >= (f.poison >> 1) ? qt_u2f(0x7FC00000u) : ldexpf((float)fx, E - f.shift);
}

#ifndef QT_BF_LO_HI
#define QT_BF_LO_HI
__device__ __forceinline__ float bf_lo(uint32_t w) { return qt_u2f(w << 16); }          // the two bf16 halves of a packed word
__device__ __forceinline__ float bf_hi(uint32_t w) { return qt_u2f(w & 0xFFFF0000u); }
#endif

template <int IO, int KIND, int DIV, bool OBS>
__device__ __forceinline__ uint4 fq_vec_d(uint4 v, const UniformDiv &dv, const Rounder<KIND> &rnd, uint32_t &amax, bool &bad) {
    if constexpr (IO == kIoBf16 && KIND == kFmtRows) {
        // the row form on all eight values at once (csrc/qt_device.h, fq_rows_words): eight row gathers in flight and one rare branch per
        // vector instead of a gather, a wait and a branch per value -- what a short pass (one vector per lane) spends its time on
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t q[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = w[i] << 16, hi = w[i] & 0xFFFF0000u;
            if constexpr (OBS) {
                const uint32_t a0 = lo & 0x7FFFFFFFu, a1 = hi & 0x7FFFFFFFu;
                amax = amax > a0 ? amax : a0;     // integer order == float order on |x|; NaN patterns win -> propagate
                amax = amax > a1 ? amax : a1;
            }
            if constexpr (DIV == kDivFast) q[i] = pack_bf16x2(dv.fast16(qt_u2f(lo), bad), dv.fast16(qt_u2f(hi), bad));
            else if constexpr (DIV == kDivExact) q[i] = pack_bf16x2(dv.exact(qt_u2f(lo)), dv.exact(qt_u2f(hi)));
            else q[i] = w[i];
        }
        fq_rows_words<4, false>(q, r, rnd);
        if constexpr (DIV != kDivUnit) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(qt_u2f(r[i] << 16) * dv.s, qt_u2f(r[i] & 0xFFFF0000u) * dv.s);
        }
        v = uint4{r[0], r[1], r[2], r[3]};
    } else if constexpr (IO == kIoBf16) {
        v.x = fq_word_bf16_d<KIND, DIV, OBS>(v.x, dv, rnd, amax, bad);
        v.y = fq_word_bf16_d<KIND, DIV, OBS>(v.y, dv, rnd, amax, bad);
        v.z = fq_word_bf16_d<KIND, DIV, OBS>(v.z, dv, rnd, amax, bad);
        v.w = fq_word_bf16_d<KIND, DIV, OBS>(v.w, dv, rnd, amax, bad);
    } else {
        v.x = fq_word_f32_d<KIND, DIV, OBS>(v.x, dv, rnd, amax, bad);
        v.y = fq_word_f32_d<KIND, DIV, OBS>(v.y, dv, rnd, amax, bad);
        v.z = fq_word_f32_d<KIND, DIV, OBS>(v.z, dv, rnd, amax, bad);
        v.w = fq_word_f32_d<KIND, DIV, OBS>(v.w, dv, rnd, amax, bad);
    }
    return v;
}

// One 16-B vector.  With a non-unit scale the fast quotient is tried first and the whole vector is redone
// with the full division only if one of its elements left the range where the fast form is exact.
template <int IO, int KIND, int DIV, bool OBS>
__device__ __forceinline__ uint4 fq_vec(uint4 v, const UniformDiv &dv, const Rounder<KIND> &rnd, uint32_t &amax) {
    bool bad = false;
    uint4 r = fq_vec_d<IO, KIND, DIV, OBS>(v, dv, rnd, amax, bad);
    if constexpr (DIV == kDivFast) {
        if (__builtin_expect(bad, 0)) {
            uint32_t unused = 0;
            r = fq_vec_d<IO, KIND, kDivExact, false>(v, dv, rnd, unused, bad);
        }
    }
    return r;
}

// ---- stages of a chain (include/qt_hip.h, qt_chain_stage): stage i reads the producer's value (src -1) or the result of stage src < i
constexpr int kChainMax = 4;
struct ChainStageDev {
    const float *scale;
    uint32_t *amax;
    uint4 *out;
    int src;                  // -1: x; else the stage whose result this one reads
};

// One vector through one stage: amax of the input, the quotient x / s (fast form; a vector with an element the fast form cannot decide
// takes the IEEE division for all eight -- the same values fq_vec's redo computes), the format's rounding ONCE, the product with s.
// Written out here instead of through fq_vec: that instantiates the rounding code per division variant and again for its redo path --
// three to four copies per call site, and these kernels have up to seven call sites and run once through their code per launch.
template <int KIND>
__device__ __forceinline__ uint4 chain_apply(uint4 v, float s, const UniformDiv &dv, const Rounder<KIND> &rnd, uint32_t &amax) {
    (void)s;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t q[4], r[4];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = w[i] << 16, hi = w[i] & 0xFFFF0000u;
        const uint32_t a0 = lo & 0x7FFFFFFFu, a1 = hi & 0x7FFFFFFFu;
        amax = amax > a0 ? amax : a0;                              // integer order == float order on |x|; NaN patterns win -> propagate
        amax = amax > a1 ? amax : a1;
        q[i] = pack_bf16x2(dv.fast16(qt_u2f(lo), bad), dv.fast16(qt_u2f(hi), bad));
    }
    if (__builtin_expect(bad || !dv.safe, 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = pack_bf16x2(dv.exact(qt_u2f(w[i] << 16)), dv.exact(qt_u2f(w[i] & 0xFFFF0000u)));
    }
    if constexpr (KIND == kFmtRows) {
        fq_rows_words<4, false>(q, r, rnd);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t r0 = rnd(q[i] << 16), r1 = rnd(q[i] & 0xFFFF0000u);       // (images on the format's grid: exact in bf16)
            r[i] = (r0 >> 16) | (r1 & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(bf_lo(r[i]) * dv.s, bf_hi(r[i]) * dv.s);
    return uint4{r[0], r[1], r[2], r[3]};
}

// A table format's row words (4 - 8 KiB behind the map's 65 536 entries) into LDS: a chain applies several fake-quantizers per vector, and
// eight row gathers per stage from global memory made its kernels latency-bound (an LDS gather returns in ~100 cycles, an L1 hit in ~500).
// s_rows: __shared__ uint4[512].  Other kinds: nothing to stage.
template <int KIND>
__device__ __forceinline__ Rounder<KIND> chain_rounder(const qt_format &fmt, const uint16_t *lut, uint4 *s_rows, int block, bool in_lds = true) {
    Rounder<KIND> rnd{fmt, nullptr, lut};
    if constexpr (KIND == kFmtRows) {
        if (!in_lds) {                                            // (tuning: the table where it lies, as fq_rows_direct_kernel reads it)
            rnd.lds = lut + QT_MAP_ENTRIES;
            return rnd;
        }
        const uint4 *g = (const uint4 *)(lut + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += block) s_rows[i] = g[i];
        rnd.lds = (const uint16_t *)s_rows;
        __syncthreads();
    }
    return rnd;
}

// All stages of a chain on one vector `v` (the producer's bf16 values): results written where a stage has an output, amax of every
// stage's input accumulated, res[] left for the caller (column sums).  Stage indices are compile-time (a run-time index into the
// kernel-argument struct would send it to scratch memory).
template <int KIND, int NS>
__device__ __forceinline__ void chain_stages(const ChainStageDev (&st)[kChainMax], const float (&sc)[NS], const Rounder<KIND> &rnd, uint4 v,
                                             size_t index, uint32_t (&amax)[NS], uint4 (&res)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        uint4 in = v;
#pragma unroll
        for (int j = 0; j < i; ++j)
            if (st[i].src == j) in = res[j];
        const UniformDiv dv(sc[i]);
        res[i] = chain_apply<KIND>(in, sc[i], dv, rnd, amax[i]);
        if (st[i].out) st[i].out[index] = res[i];
    }
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence over ALL address spaces: it waits for every global store
// the wave has in flight (s_waitcnt vmcnt(0)) -- behind a kernel's result stores that is a full store round trip (1-2 us) in front of
// every LDS exchange.  Nothing that meets in LDS below (amax slots, column partials) depends on another wave's GLOBAL accesses.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// amax of every stage: wave, workgroup (LDS [NS][BLOCK / 64]), then at most one atomic per stage and workgroup
template <int NS, int BLOCK>
__device__ __forceinline__ void chain_amax_commit(const ChainStageDev (&st)[kChainMax], const uint32_t (&amax)[NS], uint32_t (*s_amax)[BLOCK / 64]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const uint32_t m = wave_max_u32(amax[i]);
        if (lane == 0) s_amax[i][wave] = m;
    }
    lds_only_barrier();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (t == i * 64 && st[i].amax) {                         // one lane of wave i
            uint32_t m = s_amax[i][0];
#pragma unroll
            for (int k = 1; k < BLOCK / 64; ++k) m = m > s_amax[i][k] ? m : s_amax[i][k];
            if (m != 0u && m > __hip_atomic_load(st[i].amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st[i].amax, m);
        }
    }
}

}  // namespace
