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

// x / s for a wave-uniform divisor, bit-identical to the IEEE fp32 division torch performs.
// Fast path: r = RN(1/s) once per thread, then q = x*r refined by two FMA residual steps (the same
// correction chain as the hardware v_div_* sequence, started from an exactly rounded reciprocal).
// It is exact when no intermediate can underflow or overflow: 2^-25 <= |s| <= 2^20 and
// (x == 0 or 2^-102 <= |x| <= 2^100); everything else (incl. Inf / NaN) takes the full division.
struct UniformDiv {
    float s, r;
    bool safe;
    __device__ __forceinline__ explicit UniformDiv(float scale) : s(scale), r(1.0f / scale) {
        const uint32_t as = qt_f2u(scale) & 0x7FFFFFFFu;
        safe = as >= ((127u - 25u) << 23) && as <= ((127u + 20u) << 23);
    }
    __device__ __forceinline__ float operator()(float x) const {
        const uint32_t ax = qt_f2u(x) & 0x7FFFFFFFu;
        constexpr uint32_t lo = (127u - 102u) << 23, hi = (127u + 100u) << 23;
        const bool ok = safe && (((ax - lo) <= (hi - lo)) || ax == 0u);
        if (__builtin_expect(ok, 1)) {
            const float q0 = x * r;
            float e = __builtin_fmaf(-q0, s, x);
            float q = __builtin_fmaf(e, r, q0);
            e = __builtin_fmaf(-q, s, x);
            q = __builtin_fmaf(e, r, q);
            return ax == 0u ? q0 : q;            // the residual steps would turn -0 into +0
        }
        return x / s;
    }
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
