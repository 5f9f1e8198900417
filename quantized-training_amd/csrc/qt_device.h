// qt_device.h -- device-side building blocks shared by the elementwise and GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_formats.h"

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// two floats -> packed bf16x2 (RNE, NaN stays NaN): v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    float2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// four on-grid values -> four OCP FP8 bytes (exact for values the format holds)
template <bool E5M2>
__device__ __forceinline__ uint32_t qt_pack_fp8x4(float a, float b, float c, float d) {
    int w = 0;
    if constexpr (E5M2) {
        w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
    } else {
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    }
    return (uint32_t)w;
}

// x / s for a wave-uniform divisor, bit-identical to what torch computes.
//
// The quotient is only ever consumed through a 16-bit view: rounded to bf16 (bf16 tensors) or folded
// to hi16 | (lo16 != 0) (fp32 tensors, decomposed.py:151-153).  q = x * RN(1/s) is within 2 fp32 ULPs of
// the correctly rounded x / s, so it yields the same 16-bit view unless its low half lies within 3 ULPs
// of the decision point (0x8000 for the bf16 rounding, 0x0000 for the fold).  Those elements (4e-5 of
// random bf16 data; validated on 2e8 cases incl. subnormals / Inf) set `bad`, and the caller redoes the
// 16-B vector with the full IEEE division -- one rarely taken branch per vector, none per element.
// Power-of-two scales make x * (1/s) exact, so nothing is flagged; scales whose reciprocal is not a
// normal number use the full division throughout.
struct UniformDiv {
    float s, r;
    bool safe, pow2;
    __device__ __forceinline__ explicit UniformDiv(float scale) : s(scale), r(1.0f / scale) {
        const uint32_t as = qt_f2u(scale) & 0x7FFFFFFFu;
        safe = as >= ((127u - 100u) << 23) && as <= ((127u + 100u) << 23);
        pow2 = (as & 0x007FFFFFu) == 0u;
    }
    __device__ __forceinline__ float fast16(float x, bool &bad) const {    // result feeds a bf16 rounding
        const float q = x * r;
        bad |= (((qt_f2u(q) + 3u - 0x8000u) & 0xFFFFu) <= 6u) & !pow2;
        return q;
    }
    __device__ __forceinline__ float fast32(float x, bool &bad) const {    // result feeds the fp32 -> index fold
        const float q = x * r;
        bad |= (((qt_f2u(q) + 3u) & 0xFFFFu) <= 6u) & !pow2;
        return q;
    }
    __device__ __forceinline__ float exact(float x) const { return x / s; }
    __device__ __forceinline__ float operator()(float x) const { return x / s; }
};

template <int KIND>
struct Rounder {
    qt_format fmt;
    const uint16_t *lds;   // KIND == QT_FMT_LUT: table in LDS (or global for the gather kernels)
    // image -> image
    __device__ __forceinline__ uint32_t operator()(uint32_t img) const {
        if constexpr (KIND == QT_FMT_LUT) {
            return (uint32_t)lds[img >> 16] << 16;
        } else if constexpr (KIND == QT_FMT_FP_SAT) {
            return qt_fp_sat_u32(img, fmt.p0, fmt.p1, fmt.fhi);
        } else if constexpr (KIND == QT_FMT_INT) {
            return qt_int_img(img, fmt.flo, fmt.fhi);
        } else {
            return img;   // identity keeps the payload; x*s propagates NaN
        }
    }
};

// One element pair of a bf16 tensor held in one 32-bit word.
template <int KIND, bool UNIT, bool OBS>
__device__ __forceinline__ uint32_t fq_word_bf16(uint32_t w, float s, const Rounder<KIND> &rnd, uint32_t &amax) {
    uint32_t lo = w << 16, hi = w & 0xFFFF0000u;
    if constexpr (OBS) {
        uint32_t a0 = lo & 0x7FFFFFFFu, a1 = hi & 0x7FFFFFFFu;
        amax = amax > a0 ? amax : a0;     // integer order == float order on |x|; NaN patterns win -> propagate
        amax = amax > a1 ? amax : a1;
    }
    if constexpr (!UNIT) {
        const UniformDiv dv(s);   // loop-invariant: hoisted out of the streaming loop
        uint32_t q = pack_bf16x2(dv(qt_u2f(lo)), dv(qt_u2f(hi)));
        lo = q << 16;
        hi = q & 0xFFFF0000u;
    }
    uint32_t r0 = rnd(lo), r1 = rnd(hi);
    if constexpr (!UNIT) return pack_bf16x2(qt_u2f(r0) * s, qt_u2f(r1) * s);
    return (r0 >> 16) | (r1 & 0xFFFF0000u);
}

// Vector-level variants.  DIV: 0 = scale 1 (no division / multiply), 1 = fast quotient (sets `bad`),
// 2 = full IEEE division.
constexpr int kDivUnit = 0, kDivFast = 1, kDivExact = 2;

template <int KIND, int DIV, bool OBS>
__device__ __forceinline__ uint32_t fq_word_bf16_d(uint32_t w, const UniformDiv &dv, const Rounder<KIND> &rnd,
                                                   uint32_t &amax, bool &bad) {
    uint32_t lo = w << 16, hi = w & 0xFFFF0000u;
    if constexpr (OBS) {
        uint32_t a0 = lo & 0x7FFFFFFFu, a1 = hi & 0x7FFFFFFFu;
        amax = amax > a0 ? amax : a0;
        amax = amax > a1 ? amax : a1;
    }
    if constexpr (DIV != kDivUnit) {
        uint32_t q;
        if constexpr (DIV == kDivFast) q = pack_bf16x2(dv.fast16(qt_u2f(lo), bad), dv.fast16(qt_u2f(hi), bad));
        else q = pack_bf16x2(dv.exact(qt_u2f(lo)), dv.exact(qt_u2f(hi)));
        lo = q << 16;
        hi = q & 0xFFFF0000u;
    }
    uint32_t r0 = rnd(lo), r1 = rnd(hi);
    if constexpr (DIV != kDivUnit) return pack_bf16x2(qt_u2f(r0) * dv.s, qt_u2f(r1) * dv.s);
    return (r0 >> 16) | (r1 & 0xFFFF0000u);
}

template <int KIND, int DIV, bool OBS>
__device__ __forceinline__ uint32_t fq_word_f32_d(uint32_t w, const UniformDiv &dv, const Rounder<KIND> &rnd,
                                                  uint32_t &amax, bool &bad) {
    if constexpr (OBS) {
        uint32_t a = w & 0x7FFFFFFFu;
        amax = amax > a ? amax : a;
    }
    float q = qt_u2f(w);
    if constexpr (DIV == kDivFast) q = dv.fast32(q, bad);
    else if constexpr (DIV == kDivExact) q = dv.exact(q);
    float r = qt_u2f(rnd(qt_fold_img(qt_f2u(q))));
    if constexpr (DIV != kDivUnit) r = r * dv.s;
    return qt_f2u(r);
}

template <int KIND, bool UNIT, bool OBS>
__device__ __forceinline__ uint32_t fq_word_f32(uint32_t w, float s, const Rounder<KIND> &rnd, uint32_t &amax) {
    if constexpr (OBS) {
        uint32_t a = w & 0x7FFFFFFFu;
        amax = amax > a ? amax : a;
    }
    float q = qt_u2f(w);
    if constexpr (!UNIT) q = UniformDiv(s)(q);
    float r = qt_u2f(rnd(qt_fold_img(qt_f2u(q))));
    if constexpr (!UNIT) r = r * s;
    return qt_f2u(r);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = v > o ? v : o;
    }
    return v;
}


}  // namespace
