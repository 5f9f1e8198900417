// Fused quantize_mx for blocks along the last axis: one read of x, one write of the element values, the block
// scales, and -- when the element format is one the matrix instruction takes -- the packed operand for qt_mx_gemm.
//
// Replaces   quantize_mx(input, qmap, axes=[-1], block_size, quant_max, force_scale_power_of_two, scale_qmap)
//            decomposed.py:365-448  (calculate_mx_qparam: _reshape_to_blocks, _shared_exponents / amax, 2 ** e or
//            amax / quant_max [+ scale map], where(scale > 0, scale, 1);  quantize: input / expand(scale) -> vmap)
// which runs as a dozen elementwise / reduction passes in the reference, every op in the tensor's dtype.
//
// Shared exponent (mx_utils.py:19-55 with ebits = 0): floor(log2(amax + 2^-126 * (amax == 0))) with log2 evaluated
// in the tensor's dtype -- for bf16 that is bf16(log2f(amax)), whose rounding can reach the next integer (e.g.
// amax = 255 -> log2 = 7.994 -> bf16 8.0), so the floor is taken of the ROUNDED logarithm here as well.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_formats.h"
#include "qt_mx.h"
#include "qt_mx_log2_tables.h"

namespace {

constexpr int kIoBf16 = 0, kIoF32 = 1;

struct MxQuantArgs {
    const uint4 *x;
    uint4 *q;                 // element values, tensor dtype (NULL = not wanted)
    void *sf;                 // block scales, tensor dtype, [rows][cols / bs]
    uint8_t *codes, *e8m0;    // packed operand (NULL = not wanted)
    size_t nvec;
    int group;                // lanes per block
    int lanes32;              // lanes per 32 elements (4 for bf16, 8 for fp32)
    qt_format fmt;
    const uint16_t *lut, *scale_lut;
    float quant_max;
    int qmax_exp;             // floor(log2(quant_max))
    int pow2, pack_fmt;
};

// floor of log2(a) as the reference computes it in dtype IO (a > 0 finite, given as its fp32 bit image): the exponent,
// plus one when the mantissa is high enough for the dtype-rounded logarithm to reach the next integer
// (thresholds: qt_mx_log2_tables.h, generated from torch's own CPU log2 by tools/gen_mx_log2_tables.py).
template <int IO>
__device__ __forceinline__ int floor_log2_dtype(uint32_t au) {
    const uint32_t E = au >> 23, M = au & 0x7FFFFFu;
    int e;
    uint32_t mn;
    if (E) { e = (int)E - 127; mn = M; }
    else { const int p = 31 - __clz((int)M); e = p - 149; mn = (M << (23 - p)) & 0x7FFFFFu; }
    const uint32_t thr = (IO == kIoBf16 ? kMxLog2ThrBf16 : kMxLog2ThrF32)[e + 1 - kMxLog2Lo];
    return e + (mn >= thr ? 1 : 0);
}

// LDS_TABLE: the 128 KiB value map is staged in LDS once per workgroup (one 1024-thread workgroup per CU), so the
// per-element lookups are LDS reads instead of L2 gathers; worth it from a few million elements up.
// value (exactly representable in the target format) -> element code; mbits / bias / ebits of the target at run time
__device__ __forceinline__ uint32_t encode_exact(float v, int ebits, int mbits, int bias) {
    const uint32_t u = qt_f2u(v), au = u & 0x7FFFFFFFu;
    const int E = (int)(au >> 23) - 127;
    const uint32_t normal = ((uint32_t)(E + bias) << mbits) | ((au >> (23 - mbits)) & ((1u << mbits) - 1u));
    const uint32_t sub = (uint32_t)(qt_u2f(au) * qt_u2f((uint32_t)(127 + bias - 1 + mbits) << 23));     // multiples of 2^(1 - bias - mbits)
    return (E >= 1 - bias ? normal : sub) | ((u >> 31) << (ebits + mbits));
}

// PACK_BITS: 0 = no packed operand, else the element width (8 / 6 / 4) -- compile-time so the packing is straight-line.
// ROWS: the value map in its row form (qt_format.p1 bit 0: the row words of qt_build_rowparams behind the map's 65 536 entries; csrc/qt_device.h
// Rounder<kFmtRows>): arithmetic against a 4 - 8 KiB row table in LDS instead of table gathers, at 256-thread occupancy.
template <int IO, bool LDS_TABLE, int PACK_BITS, bool ROWS = false>
__global__ __launch_bounds__(LDS_TABLE ? 1024 : 256) void quantize_mx_kernel(MxQuantArgs a) {
    constexpr int kThreads = LDS_TABLE ? 1024 : 256;
    static_assert(!(LDS_TABLE && ROWS), "one or the other");
    Rounder<kFmtRows> rnd{a.fmt, nullptr, a.lut};
    if constexpr (ROWS) {
        __shared__ uint4 s_rows[512];
        const uint4 *gr = (const uint4 *)(a.lut + QT_MAP_ENTRIES);
        const int nrows = (a.fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += kThreads) s_rows[i] = gr[i];
        rnd.lds = (const uint16_t *)s_rows;
        __syncthreads();
    }
    auto look = [&](uint32_t img) __attribute__((always_inline)) -> uint32_t {      // image of map[x]
        if constexpr (ROWS) return rnd(img);
        else return a.lut ? (uint32_t)a.lut[img >> 16] << 16 : qt_apply_format_img(a.fmt, img);
    };
    if constexpr (LDS_TABLE) {
        extern __shared__ __attribute__((aligned(16))) uint16_t s_table[];
        const uint4 *src = (const uint4 *)a.lut;
        uint4 *dst = (uint4 *)s_table;
#pragma unroll
        for (int i = 0; i < 65536 * 2 / 16 / kThreads; ++i) dst[threadIdx.x + i * kThreads] = src[threadIdx.x + i * kThreads];
        __syncthreads();
        a.lut = s_table;
    }
    const size_t stride = (size_t)gridDim.x * kThreads;
    const size_t iters = (a.nvec + stride - 1) / stride;
    for (size_t it = 0; it < iters; ++it) {
        const size_t v = it * stride + (size_t)blockIdx.x * kThreads + threadIdx.x;
        const bool live = v < a.nvec;
        uint4 in = {0u, 0u, 0u, 0u};
        if (live) in = a.x[v];
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        uint32_t am = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (IO == kIoBf16) {
                const uint32_t a0 = (w[j] << 16) & 0x7FFFFFFFu, a1 = w[j] & 0x7FFF0000u;
                am = am > a0 ? am : a0;
                am = am > a1 ? am : a1;
            } else {
                const uint32_t a0 = w[j] & 0x7FFFFFFFu;
                am = am > a0 ? am : a0;
            }
        }
        for (int off = 1; off < a.group; off <<= 1) {       // block amax over `group` adjacent lanes; NaN patterns win
            const uint32_t o = (uint32_t)__shfl_xor((int)am, off, 64);
            am = am > o ? am : o;
        }
        float s;
        int se = 0;                                          // scale = 2^se when pow2 and representable
        bool s_is_pow2 = false;
        if (a.pow2) {
            if (am > 0x7F800000u) s = 1.0f;                  // NaN: 2 ** NaN = NaN, where(NaN > 0) -> 1
            else if (am == 0x7F800000u) s = qt_u2f(0x7F800000u);   // +Inf stays +Inf
            else {
                se = floor_log2_dtype<IO>(am ? am : 0x00800000u) - a.qmax_exp;   // + FP32_MIN_NORMAL * (amax == 0)
                const int lowest = IO == kIoBf16 ? -133 : -149;                 // smallest power of two of the dtype
                if (se < lowest) s = 1.0f;                                      // 2 ** e underflows to 0 -> 1
                else if (se > 127) s = qt_u2f(0x7F800000u);
                else {
                    s = se >= -126 ? qt_u2f((uint32_t)(se + 127) << 23) : qt_u2f(1u << (se + 149));
                    s_is_pow2 = true;
                }
            }
        } else {
            s = qt_u2f(am) / a.quant_max;
            if constexpr (IO == kIoBf16) s = qt_u2f(pack_bf16x2(s, 0.0f) << 16);
            if (a.scale_lut) {
                const uint32_t img = IO == kIoBf16 ? qt_f2u(s) : qt_fold_img(qt_f2u(s));
                s = qt_bf2f(a.scale_lut[img >> 16]);
            }
            s = s > 0.0f ? s : 1.0f;
        }
        if (live && (threadIdx.x & (a.group - 1)) == 0) {
            const size_t blk = v / (size_t)a.group;
            if constexpr (IO == kIoBf16) ((uint16_t *)a.sf)[blk] = (uint16_t)(qt_f2u(s) >> 16);
            else ((float *)a.sf)[blk] = s;
        }
        // element values: map[x / s], every op in the tensor dtype
        const bool recip_ok = s_is_pow2 && se >= -126 && se <= 126;
        const float r = recip_ok ? qt_u2f((uint32_t)(127 - se) << 23) : 0.0f;
        constexpr int kPer = IO == kIoBf16 ? 8 : 4;
        float qv[kPer];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (IO == kIoBf16) {
                const float x0 = qt_u2f(w[j] << 16), x1 = qt_u2f(w[j] & 0xFFFF0000u);
                const uint32_t p = recip_ok ? pack_bf16x2(x0 * r, x1 * r) : pack_bf16x2(x0 / s, x1 / s);
                const uint32_t i0 = p << 16, i1 = p & 0xFFFF0000u;
                qv[2 * j] = qt_u2f(look(i0));
                qv[2 * j + 1] = qt_u2f(look(i1));
            } else {
                const float x0 = qt_u2f(w[j]);
                const uint32_t img = qt_fold_img(qt_f2u(recip_ok ? x0 * r : x0 / s));
                qv[j] = qt_u2f(look(img));
            }
        }
        if (!live) continue;
        if (a.q) {
            uint4 o;
            if constexpr (IO == kIoBf16) {
                o.x = (qt_f2u(qv[0]) >> 16) | (qt_f2u(qv[1]) & 0xFFFF0000u);
                o.y = (qt_f2u(qv[2]) >> 16) | (qt_f2u(qv[3]) & 0xFFFF0000u);
                o.z = (qt_f2u(qv[4]) >> 16) | (qt_f2u(qv[5]) & 0xFFFF0000u);
                o.w = (qt_f2u(qv[6]) >> 16) | (qt_f2u(qv[7]) & 0xFFFF0000u);
            } else {
                o.x = qt_f2u(qv[0]); o.y = qt_f2u(qv[1]); o.z = qt_f2u(qv[2]); o.w = qt_f2u(qv[3]);
            }
            a.q[v] = o;
        }
        if constexpr (PACK_BITS != 0) {
            // E8M0 holds 2^-127 .. 2^127: a block whose scale is lower (|x| below ~1e-36) is flushed to zeros,
            // an absolute error under 2^-119 per element; scale 1 from the where() guard is byte 127.
            const bool flush = s_is_pow2 && se < -127;
            const int f = a.pack_fmt;
            const int ebits = PACK_BITS == 8 ? (f == 0 ? 4 : 5) : (PACK_BITS == 6 ? (f == 2 ? 2 : 3) : 2);
            const int mbits = PACK_BITS - 1 - ebits;
            const int bias = (1 << (ebits - 1)) - 1;
            constexpr int nbytes = kPer * PACK_BITS / 8;         // 8 / 6 / 4 (bf16) or 4 / 3 / 2 (fp32)
            uint8_t *dst = a.codes + v * (size_t)nbytes;
            // 8- and 4-bit elements: the values are on the format's grid, so the hardware conversions give their codes
            // exactly (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32, v_cvt_scalef32_pk_fp4_f32 with scale 1) -- half an
            // instruction per element instead of the ~8 of encode_exact, on a kernel that is issue-limited
            if constexpr (PACK_BITS == 8 && kPer == 8) {
                uint32_t lo, hi;
                if (f == 0) { lo = qt_pack_fp8x4<false>(qv[0], qv[1], qv[2], qv[3]); hi = qt_pack_fp8x4<false>(qv[4], qv[5], qv[6], qv[7]); }
                else { lo = qt_pack_fp8x4<true>(qv[0], qv[1], qv[2], qv[3]); hi = qt_pack_fp8x4<true>(qv[4], qv[5], qv[6], qv[7]); }
                *(uint2 *)dst = flush ? uint2{0u, 0u} : uint2{lo, hi};
            } else if constexpr (PACK_BITS == 4 && kPer == 8) {
                uint32_t w4 = 0;
                w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w4, qv[0], qv[1], 1.0f, 0);
                w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w4, qv[2], qv[3], 1.0f, 1);
                w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w4, qv[4], qv[5], 1.0f, 2);
                w4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w4, qv[6], qv[7], 1.0f, 3);
                *(uint32_t *)dst = flush ? 0u : w4;
            } else {
            uint32_t c[kPer];
#pragma unroll
            for (int e = 0; e < kPer; ++e) c[e] = flush ? 0u : encode_exact(qv[e], ebits, mbits, bias);
            if constexpr (PACK_BITS == 8) {
                const uint32_t lo = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
                if constexpr (kPer == 8) *(uint2 *)dst = uint2{lo, c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24)};
                else *(uint32_t *)dst = lo;
            } else if constexpr (PACK_BITS == 4) {
                const uint32_t lo = c[0] | (c[1] << 4) | (c[2] << 8) | (c[3] << 12);
                if constexpr (kPer == 8) *(uint32_t *)dst = lo | (c[4] << 16) | (c[5] << 20) | (c[6] << 24) | (c[7] << 28);
                else *(uint16_t *)dst = (uint16_t)lo;
            } else {
                const uint32_t lo = c[0] | (c[1] << 6) | (c[2] << 12) | (c[3] << 18);       // 24 bits
                if constexpr (kPer == 8) {
                    const uint32_t hi = c[4] | (c[5] << 6) | (c[6] << 12) | (c[7] << 18);
                    ((uint16_t *)dst)[0] = (uint16_t)lo;
                    ((uint16_t *)dst)[1] = (uint16_t)((lo >> 16) | (hi << 8));
                    ((uint16_t *)dst)[2] = (uint16_t)(hi >> 8);
                } else {
                    dst[0] = (uint8_t)lo; dst[1] = (uint8_t)(lo >> 8); dst[2] = (uint8_t)(lo >> 16);
                }
            }
            }
            if ((threadIdx.x & (a.lanes32 - 1)) == 0) {
                int eb = s_is_pow2 ? se + 127 : 127;
                eb = eb < 0 ? 0 : eb;
                a.e8m0[v / (size_t)a.lanes32] = (uint8_t)eb;
            }
        }
    }
}

int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int IO>
int launch(const void *x, void *q, void *sf, uint8_t *codes, uint8_t *e8m0, size_t rows, size_t cols, int bs,
           const qt_format *fmt, const uint16_t *lut, float quant_max, int pow2, const uint16_t *scale_lut, int pack_fmt,
           void *stream) {
    if (rows * cols == 0) return QT_OK;
    constexpr int kPer = IO == kIoBf16 ? 8 : 4;
    if (!x || !sf || !fmt || bs < kPer || (bs & (bs - 1)) || bs > 64 * kPer || !(quant_max > 0.0f)) return QT_ERR_BAD_ARG;
    if (fmt->kind == QT_FMT_LUT && !lut) return QT_ERR_BAD_ARG;
    if ((codes != nullptr) != (e8m0 != nullptr)) return QT_ERR_BAD_ARG;
    if (codes && (pack_fmt < 0 || pack_fmt > 4 || bs % 32 != 0 || !pow2)) return QT_ERR_BAD_ARG;
    if ((cols % (size_t)bs) || (((uintptr_t)x | (uintptr_t)q) & 15u)) return QT_ERR_UNALIGNED;
    int qe = 0;
    (void)frexpf(quant_max, &qe);                      // quant_max = m * 2^qe, m in [0.5, 1)  ->  floor(log2) = qe - 1
    MxQuantArgs a{(const uint4 *)x, (uint4 *)q, sf, codes, e8m0, rows * cols / kPer, bs / kPer, 32 / kPer, *fmt,
                  fmt->kind == QT_FMT_LUT ? lut : nullptr, scale_lut, quant_max, qe - 1, pow2, pack_fmt};
    const int pb = codes ? qt_mx::elem_bits(pack_fmt) : 0;
    const bool rowform = a.lut && (fmt->p1 & 1) && (((uintptr_t)a.lut) & 15u) == 0;
    const bool lds = !rowform && a.lut && rows * cols >= ((size_t)1 << 22) && (((uintptr_t)a.lut) & 15u) == 0;
    hipStream_t st = (hipStream_t)stream;
#define QT_MXQ(PB)                                                                                                 \
    if (pb == PB) {                                                                                                \
        if (rowform) {                               /* (8 workgroups per CU measured slower: 3.6 against 4.4 TB/s) */     \
            size_t want = (a.nvec + 255) / 256, cap = (size_t)num_cus() * 32;                                      \
            quantize_mx_kernel<IO, false, PB, true><<<(unsigned)(want < cap ? want : cap), 256, 0, st>>>(a);       \
        } else if (lds) {                                                                                              \
            static QtOncePerDevice configured;                                                                              \
            if (configured.needed()) {                                                                                     \
                const hipError_t e = hipFuncSetAttribute((const void *)quantize_mx_kernel<IO, true, PB>,           \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2);   \
                if (e != hipSuccess) return (int)e;                                                                \
                configured.done();                                                                                  \
            }                                                                                                      \
            quantize_mx_kernel<IO, true, PB><<<(unsigned)num_cus(), 1024, 65536 * 2, st>>>(a);                     \
        } else {                                                                                                   \
            size_t want = (a.nvec + 255) / 256, cap = (size_t)num_cus() * 32;                                      \
            quantize_mx_kernel<IO, false, PB><<<(unsigned)(want < cap ? want : cap), 256, 0, st>>>(a);             \
        }                                                                                                          \
        return launch_status();                                                                                    \
    }
    QT_MXQ(0) QT_MXQ(8) QT_MXQ(6) QT_MXQ(4)
#undef QT_MXQ
    return QT_ERR_BAD_ARG;
}

}  // namespace

extern "C" {

int qt_quantize_mx_bf16(const uint16_t *x, uint16_t *q, uint16_t *scales, uint8_t *codes, uint8_t *e8m0, size_t rows,
                        size_t cols, int block_size, const qt_format *fmt, const uint16_t *lut, float quant_max,
                        int force_pow2, const uint16_t *scale_lut, int pack_format, void *stream) {
    return launch<kIoBf16>(x, q, scales, codes, e8m0, rows, cols, block_size, fmt, lut, quant_max, force_pow2, scale_lut,
                           pack_format, stream);
}

int qt_quantize_mx_f32(const float *x, float *q, float *scales, uint8_t *codes, uint8_t *e8m0, size_t rows, size_t cols,
                       int block_size, const qt_format *fmt, const uint16_t *lut, float quant_max, int force_pow2,
                       const uint16_t *scale_lut, int pack_format, void *stream) {
    return launch<kIoF32>(x, q, scales, codes, e8m0, rows, cols, block_size, fmt, lut, quant_max, force_pow2, scale_lut,
                          pack_format, stream);
}

}  // extern "C"
