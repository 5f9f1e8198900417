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
        uint32_t q = pack_bf16x2(qt_u2f(lo) / s, qt_u2f(hi) / s);
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
    if constexpr (!UNIT) q = q / s;
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
