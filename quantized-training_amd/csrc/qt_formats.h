// qt_formats.h -- closed-form rounding shared by the host map builder and the gfx950 kernels.
//
// A bf16 value is handled as its fp32 "image": the 32-bit pattern with the bf16 bits in the
// high half and a zero low half.  Every closed form below maps image -> image and is proven
// equal to the reference's 65 536-entry value map for its dtype on ALL inputs
// (tests/test_capi_host.py for the host build; tests/test_gpu_parity.py for the device build).
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/qt_hip.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define QT_HD __host__ __device__ __forceinline__
#else
#define QT_HD static inline
#endif

#define QT_NAN16 ((uint16_t)0x7FC0)
#define QT_NAN32 (0x7FC00000u)

QT_HD float qt_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
QT_HD uint32_t qt_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
QT_HD float qt_bf2f(uint16_t b) { return qt_u2f((uint32_t)b << 16); }

// float -> bf16 bits, round-to-nearest-even, NaN -> 0x7FC0 (torch's float->bfloat16 cast;
// the NaN payload is not part of the contract).
QT_HD uint16_t qt_f2bf(float f) {
    uint32_t u = qt_f2u(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return QT_NAN16;
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// vmap index of an fp32 value as an image: (hi16 | (lo16 != 0)) << 16     (decomposed.py:151-153)
QT_HD uint32_t qt_fold_img(uint32_t u) { return (u & 0xFFFF0000u) | ((u & 0xFFFFu) != 0u ? 0x10000u : 0u); }

// ---- e4m3 / e5m2 "NVIDIA-style" saturating RNE cast (fp8.py:10-67) ---------------------------
//   mbits : mantissa bits kept (3 / 2);  emin : min normal exponent (-6 / -14);  fmax : 448 / 57344
// Works on ANY fp32 pattern (the exported quantize_to_fp8_* take fp32 tensors), images included.
// Branch-free: the grid spacing at |x| in [2^e, 2^(e+1)) is 2^(max(e, emin) - mbits); adding and
// subtracting 2^(max(e, emin) - mbits + 23) makes the FP adder perform the round-to-nearest-even,
// in the normal and the subnormal range alike; this also flushes |x| <= 2^(emin-mbits-1) to zero
// (fp8.py:33).
QT_HD uint32_t qt_fp_sat_u32(uint32_t u, int mbits, int emin, float fmax) {
    const uint32_t a = u & 0x7FFFFFFFu;
    const uint32_t fmax_bits = qt_f2u(fmax);
    // Pre-clamp: rounding is monotone, so round(min(a, fmax)) then min(.., fmax) == min(round(a), fmax)
    // (fp8.py:32); it also keeps the magic constant below finite.
    const uint32_t ac = a < fmax_bits ? a : fmax_bits;
    uint32_t eb = ac & 0x7F800000u;
    const uint32_t emin_b = (uint32_t)(emin + 127) << 23;
    eb = eb > emin_b ? eb : emin_b;
    const float c = qt_u2f(eb + ((uint32_t)(23 - mbits) << 23));   // 2^(max(e, emin) - mbits + 23)
    float r = (qt_u2f(ac) + c) - c;                                // the FP adder rounds to nearest even at 2^(max(e,emin)-mbits)
    r = r > fmax ? fmax : r;
    uint32_t ru = qt_f2u(r);
    ru = ru == 0u ? 0u : (ru | (u & 0x80000000u));                 // fp8.py:33-35: zero results are +0
    return a >= 0x7F800000u ? QT_NAN32 : ru;                       // fp8.py:36: non-finite -> NaN
}

// ---- intN / uintN: clamp(round_half_even(v), lo, hi) on the bf16 value (fake_quantize.py:43-52)
// torch.clamp is min(max(v, lo), hi) with std::max/min operand order: -0.0 and NaN pass through.
QT_HD uint32_t qt_int_img(uint32_t u, float lo, float hi) {
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return QT_NAN32;
    float r = __builtin_rintf(qt_u2f(u));                        // to nearest even (default mode)
    r = r < lo ? lo : r;
    r = r > hi ? hi : r;
    return qt_f2u(r);                                            // integer <= 256 or already bf16: exact image
}

// ---- posit<nbits,es>, round to nearest even (posit.py:6-67) on any fp32 pattern ---------------
// thr = 2^floor(-(nbits-1)*2^es + 2^(es-1)) is passed in (posit.py:52).
QT_HD uint32_t qt_posit_u32(uint32_t raw, int nbits, int es, float thr) {
    uint32_t a = raw & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return QT_NAN32;                       // :57
    if (a == 0u || qt_u2f(a) < thr) return 0u;                   // :53, :56
    int scale = (int)(a >> 23) - 127;                            // :14
    uint32_t frac = a & 0x7FFFFFu;                               // :15
    bool r = scale >= 0;                                         // :16
    int max_scale = (nbits - 2) << es;                           // :18
    bool dominated = r ? (scale > max_scale) : (scale < -max_scale);   // :19
    int k = scale >> es;                                         // arithmetic shift = floor
    int run = r ? 1 + k : -k;                                    // :21
    uint32_t rb = 0;
    if (!dominated) {
        int sh = 2 + run + es + 23 - nbits;                      // :27-28 (len - nbits), > 0 for nbits <= 24
        uint64_t regime = r ? ((((uint64_t)1 << (run + 1)) - 1) ^ 1) : 1;   // :22
        uint64_t pt = (regime << (23 + es)) | ((uint64_t)(scale & ((1 << es) - 1)) << 23) | frac;   // :23-24
        uint32_t lb = (uint32_t)(pt >> sh) & 1u, gb = (uint32_t)(pt >> (sh - 1)) & 1u;
        uint32_t sb = (pt & (((uint64_t)1 << (sh - 1)) - 1)) != 0;
        rb = (lb & gb) | (gb & sb);                              // :32-35
    }
    int ne = 2 + run + es - nbits;                               // :38
    ne = ne < 0 ? 0 : (ne > es ? es : ne);
    int sc = scale & ~((1 << ne) - 1);                           // :39
    sc = sc < -max_scale ? -max_scale : (sc > max_scale ? max_scale : sc);   // :40
    int nf = 2 + run + es + 23 - nbits;                          // :43
    nf = nf < 0 ? 0 : (nf > 23 ? 23 : nf);
    uint32_t out = ((uint32_t)(sc + 127) << 23) | (frac & ~((1u << nf) - 1u));   // :44-46
    if (rb) out += 1u << (nf + ne);                              // :47
    return out | (raw & 0x80000000u);                            // :48
}

// Same rounding with the posit bit pattern next to the value (posit.py:60-65, `return_pbits=True`): the nbits-1 magnitude
// bits of the encoding (regime | exponent | fraction after rounding) times the sign of the input, as an integer.  Upstream
// builds the pattern in int32, which overflows once 2 + run + es + 23 > 33 and is shifted by platform-dependent amounts for
// regime-dominated inputs; here the pattern is computed in 64 bits, regime-dominated magnitudes give the pattern of the
// value they are clamped to (maxpos = 2^(nbits-1) - 1, minpos = 1, or 0 when flushed), +-Inf the maxpos pattern, NaN 0.
// thr = 0 reproduces round_to_even=False (no flush of values below the smallest representable step, posit.py:50-53).
QT_HD uint32_t qt_posit_bits_u32(uint32_t raw, int nbits, int es, float thr, int32_t *pbits) {
    const uint32_t a = raw & 0x7FFFFFFFu;
    const int32_t sgn = (raw >> 31) ? -1 : 1;
    const int32_t maxpat = (int32_t)((1u << (nbits - 1)) - 1u);
    if (a > 0x7F800000u) { *pbits = 0; return QT_NAN32; }
    if (a == 0x7F800000u) { *pbits = sgn * maxpat; return QT_NAN32; }
    if (a == 0u) { *pbits = 0; return 0u; }
    const uint32_t out = qt_posit_u32(raw, nbits, es, thr);
    const int scale = (int)(a >> 23) - 127;
    const int max_scale = (nbits - 2) << es;
    if (scale > max_scale || scale < -max_scale) {
        *pbits = (out << 1) == 0u ? 0 : sgn * (scale > 0 ? maxpat : 1);
        return out;
    }
    const bool r = scale >= 0;
    const int k = scale >> es;
    const int run = r ? 1 + k : -k;
    const int sh = 2 + run + es + 23 - nbits;
    const uint64_t regime = r ? ((((uint64_t)1 << (run + 1)) - 1) ^ 1) : 1;
    const uint64_t pt = (regime << (23 + es)) | ((uint64_t)(scale & ((1 << es) - 1)) << 23) | (a & 0x7FFFFFu);
    const uint32_t lb = (uint32_t)(pt >> sh) & 1u, gb = (uint32_t)(pt >> (sh - 1)) & 1u;
    const uint32_t sb = (pt & (((uint64_t)1 << (sh - 1)) - 1)) != 0;
    const int32_t body = (int32_t)((pt >> sh) & (uint64_t)maxpat) + (int32_t)((lb & gb) | (gb & sb));
    *pbits = sgn * body;
    return out;
}

QT_HD uint32_t qt_apply_format_img(const qt_format &f, uint32_t u) {
    switch (f.kind) {
        case QT_FMT_FP_SAT: return qt_fp_sat_u32(u, f.p0, f.p1, f.fhi);
        case QT_FMT_INT: return qt_int_img(u, f.flo, f.fhi);
        default: return ((u & 0x7FFFFFFFu) > 0x7F800000u) ? QT_NAN32 : u;
    }
}
