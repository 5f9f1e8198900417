// qt_device.h -- device-side building blocks shared by the elementwise and GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Written for gfx950 only: the scaled matrix instructions (v_mfma_scale_f32_16x16x128_f8f6f4), v_mfma_f32_16x16x32_bf16, the
// v_cvt_scalef32_pk_* conversions, 16-byte LDS-DMA and the 160 KiB LDS budgets do not exist on any other target.  ARCH is a Makefile
// variable, so say so at compile time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libqt_hip is written for gfx950 (MI355X) only"
#endif

// hipFuncSetAttribute is per DEVICE, a `static bool` is per process: the package runs models on several devices of one process
// (_native.note_device), and the second device would never get the large-LDS attribute.  One bit per device ordinal.  The ordinal
// needed() looked up travels to done() in a thread_local: the object itself is a function-local static shared by every host thread
// (ctypes releases the GIL), and a member would let thread B's device overwrite thread A's between its needed() and done().
struct QtOncePerDevice {
    unsigned long long bits = 0;
    static int &current() {
        static thread_local int dev = 0;
        return dev;
    }
    bool needed() {
        int &dev = current();
        dev = 0;
        (void)hipGetDevice(&dev);
        return dev < 0 || dev > 63 || !((__atomic_load_n(&bits, __ATOMIC_ACQUIRE) >> dev) & 1ull);
    }
    void done() {
        const int dev = current();
        if (dev >= 0 && dev <= 63) __atomic_fetch_or(&bits, 1ull << dev, __ATOMIC_RELEASE);
    }
};

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

// ---- hardware FP8 conversion for the unit-scale E4M3 / E5M2 fake-quantizers (see fq8_fast16 in qt_elementwise.hip) ------
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

template <bool E5M2>
__device__ __forceinline__ uint32_t fq8_fast_word4(uint32_t w0, uint32_t w1) {          // four bf16 -> four FP8 bytes
    constexpr float F = E5M2 ? 57344.0f : 448.0f;
    const float a = __builtin_amdgcn_fmed3f(qt_u2f(w0 << 16), -F, F), b = __builtin_amdgcn_fmed3f(qt_u2f(w0 & 0xFFFF0000u), -F, F);
    const float c = __builtin_amdgcn_fmed3f(qt_u2f(w1 << 16), -F, F), d = __builtin_amdgcn_fmed3f(qt_u2f(w1 & 0xFFFF0000u), -F, F);
    const uint32_t q = qt_pack_fp8x4<E5M2>(a, b, c, d);
    const uint32_t t = q & 0x7F7F7F7Fu;                                   // zero results are +0: clear the sign of 0x80 bytes
    const uint32_t nz = ((t + 0x7F7F7F7Fu) | t) & 0x80808080u;            // 0x80 in every byte whose magnitude is not zero
    return q & (nz | 0x7F7F7F7Fu);
}

// Eight bf16 values (four packed words, IN: the unquantized values, OUT: fq(values) as bf16) -> their eight FP8 bytes.
// Finite inputs take the hardware conversion (decode of the code gives the quantized value back exactly); a vector
// holding a non-finite value takes the closed form (qt_fp_sat_u32), bit for bit what every producer kernel did before.
template <bool E5M2>
__device__ __forceinline__ uint2 fq8_hw_vec8(uint32_t (&o)[4], const qt_format &fmt) {
    const uint32_t m = pk_max_u16(pk_max_u16(o[0] & 0x7FFF7FFFu, o[1] & 0x7FFF7FFFu), pk_max_u16(o[2] & 0x7FFF7FFFu, o[3] & 0x7FFF7FFFu));
    if (__builtin_expect(((m & 0xFFFFu) >= 0x7F80u) | ((m >> 16) >= 0x7F80u), 0)) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t r0 = qt_fp_sat_u32(o[j] << 16, fmt.p0, fmt.p1, fmt.fhi), r1 = qt_fp_sat_u32(o[j] & 0xFFFF0000u, fmt.p0, fmt.p1, fmt.fhi);
            o[j] = (r0 >> 16) | (r1 & 0xFFFF0000u);
            f[2 * j] = qt_u2f(r0);
            f[2 * j + 1] = qt_u2f(r1);
        }
        return uint2{qt_pack_fp8x4<E5M2>(f[0], f[1], f[2], f[3]), qt_pack_fp8x4<E5M2>(f[4], f[5], f[6], f[7])};
    }
    const uint32_t c0 = fq8_fast_word4<E5M2>(o[0], o[1]), c1 = fq8_fast_word4<E5M2>(o[2], o[3]);
    float2_t a, b, c, d;
    if constexpr (E5M2) {
        a = __builtin_amdgcn_cvt_pk_f32_bf8((int)c0, false);
        b = __builtin_amdgcn_cvt_pk_f32_bf8((int)c0, true);
        c = __builtin_amdgcn_cvt_pk_f32_bf8((int)c1, false);
        d = __builtin_amdgcn_cvt_pk_f32_bf8((int)c1, true);
    } else {
        a = __builtin_amdgcn_cvt_pk_f32_fp8((int)c0, false);
        b = __builtin_amdgcn_cvt_pk_f32_fp8((int)c0, true);
        c = __builtin_amdgcn_cvt_pk_f32_fp8((int)c1, false);
        d = __builtin_amdgcn_cvt_pk_f32_fp8((int)c1, true);
    }
    o[0] = pack_bf16x2(a.x, a.y);
    o[1] = pack_bf16x2(b.x, b.y);
    o[2] = pack_bf16x2(c.x, c.y);
    o[3] = pack_bf16x2(d.x, d.y);
    return uint2{c0, c1};
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

// Internal kind (not part of the C ABI): a table format whose map is odd-symmetric (qt_format.p0 == 1, see qt_format_for);
// only entries 0 .. 0x7FFF are staged and the sign goes back on finite non-zero and infinite results.
constexpr int kFmtLutHalf = 4;
constexpr int kFmtRows = 5;      // QT_FMT_LUT with the row words behind the map (qt_format.p1 bit 0): the row form, table of rows in LDS

template <int KIND>
struct Rounder {
    qt_format fmt;
    const uint16_t *lds;   // KIND == QT_FMT_LUT: table in LDS (or global for the gather kernels); kFmtRows: the row words in LDS
    const uint16_t *glut;  // kFmtRows: the map in global memory, for the rows the row form does not cover
    // image -> image
    __device__ __forceinline__ uint32_t operator()(uint32_t img) const {
        if constexpr (KIND == kFmtRows) {
            // csrc/qt_host.cpp, qt_build_rowparams: t = f32(|x| bits + D); y = med3((t + C) - C, lo, hi), exact on every input of a
            // row that is not flagged (bit 0 of C)
            const uint32_t rowi = (img >> 23) & ((fmt.p1 & 2) ? 0x1FFu : 0xFFu);
            const uint4 p = ((const uint4 *)lds)[rowi];
            const float t = qt_u2f((img & 0x7FFFFFFFu) + p.x), c = qt_u2f(p.y);
            const uint32_t z = qt_f2u(__builtin_amdgcn_fmed3f((t + c) - c, qt_u2f(p.z), qt_u2f(p.w)));
            uint32_t sign = (fmt.p1 & 4) ? (img & 0x80000000u) : 0u;
            if (!(fmt.p1 & 8)) sign = z ? sign : 0u;                                 // zero results are +0 in this map
            uint32_t r = z | sign;
            if ((fmt.p1 & 16) && img == 0x80000000u) r = qt_f2u(fmt.fhi);
            if (__builtin_expect(p.y & 1u, 0)) r = (uint32_t)glut[img >> 16] << 16;
            return r;
        } else if constexpr (KIND == kFmtLutHalf) {
            const uint32_t t = lds[(img >> 16) & 0x7FFFu];
            const uint32_t sign = ((t - 1u) < 0x7F80u) ? (img & 0x80000000u) : 0u;        // zero and NaN results carry no sign
            return (t << 16) | sign;
        } else if constexpr (KIND == QT_FMT_LUT) {
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

// N words (2 N values) through the row form at once: every row gather is issued before the first result is needed and the flagged rows
// of all of them share ONE branch, so a wave pays the LDS latency once per batch instead of once per value (fq_word_bf16<kFmtRows>
// serialises: gather, wait, arithmetic, branch, for each value).  NONNEG: the caller's values carry no sign except on a NaN (softmax
// probabilities); a set sign bit then takes the same rare branch as a flagged row, and the common path drops the sign handling.
// There the first input of a flagged row (mantissa 0 -- exact zero in row 0, which posit maps flag because every bf16 subnormal
// rounds up to minpos) does not branch either: qt_build_rowparams leaves its result in the row's lo = hi.  The branch matters
// beyond its own cost: it reads the map from global memory, and the wait for that read also waits for every LDS-DMA request the
// caller has in flight.
template <int N, bool NONNEG>
__device__ __forceinline__ void fq_rows_words(const uint32_t (&w)[N], uint32_t (&o)[N], const Rounder<kFmtRows> &rnd) {
    const uint32_t rowmask = (rnd.fmt.p1 & 2) ? 0x1FFu : 0xFFu;
    const uint4 *rows = (const uint4 *)rnd.lds;
    uint32_t img[2 * N], r[2 * N], rare = 0;
    uint4 p[2 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        img[2 * i] = w[i] << 16;
        img[2 * i + 1] = w[i] & 0xFFFF0000u;
    }
#pragma unroll
    for (int j = 0; j < 2 * N; ++j) p[j] = rows[(img[j] >> 23) & rowmask];
#pragma unroll
    for (int j = 0; j < 2 * N; ++j) {
        const float t = qt_u2f((img[j] & 0x7FFFFFFFu) + p[j].x), c = qt_u2f(p[j].y);
        const uint32_t z = qt_f2u(__builtin_amdgcn_fmed3f((t + c) - c, qt_u2f(p[j].z), qt_u2f(p[j].w)));
        if constexpr (NONNEG) {
            r[j] = z;
            const uint32_t later = min((img[j] & 0x007F0000u) | p[j].x, 1u);   // flagged rows: 0 = the row's first input, covered by lo = hi
            rare |= (p[j].y & later) | (img[j] >> 31);
        } else {
            uint32_t sign = (rnd.fmt.p1 & 4) ? (img[j] & 0x80000000u) : 0u;
            if (!(rnd.fmt.p1 & 8)) sign = z ? sign : 0u;
            r[j] = z | sign;
            if ((rnd.fmt.p1 & 16) && img[j] == 0x80000000u) r[j] = qt_f2u(rnd.fmt.fhi);
            rare |= p[j].y;
        }
    }
    if (__builtin_expect(rare & 1u, 0)) {
#pragma unroll
        for (int j = 0; j < 2 * N; ++j) r[j] = rnd(img[j]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = (r[2 * i] >> 16) | (r[2 * i + 1] & 0xFFFF0000u);
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
