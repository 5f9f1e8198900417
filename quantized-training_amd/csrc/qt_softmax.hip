// qt_softmax.hip -- fused attention-score path: scale -> (+ mask) -> softmax -> fake-quantize.
//
// In the reference's quantizable attention blocks (modules/quantizable/modeling_bert.py:142-158,
// modeling_llama.py:228-246) the S x S score tensor makes six full trips through memory between the
// two attention GEMMs: attn_scaling (MulFunctional), mask add, fp32 up-cast, softmax, down-cast, and
// the fake-quant of the probabilities in av_matmul's input hook.  When only the GEMM inputs are
// quantized (`--quantize_forward gemm`, the reference default) nothing observes the intermediate
// tensors, so this kernel reads the raw QK^T scores once and writes the fake-quantized
// probabilities once (4 B/element instead of ~32 B/element):
//     t = bf16(bf16(score * scaling) + mask)            MulFunctional + mask add, each rounded to bf16
//     p = bf16(exp(t - max_row) / sum_row)              softmax evaluated in fp32 (what torch does for
//                                                       bf16 inputs and what HF LLaMA asks for explicitly)
//     y = fq(p)                                         same per-element function as qt_fake_quant_bf16,
//                                                       amax(p) max-accumulated for the observer
// One wavefront owns one row: 16-B loads, lane-local + __shfl_xor reductions, no LDS.
// Numerics: every step except exp, the row-sum order and e * (1/sum) vs e / sum is bit-defined; p can differ
// from torch's by at most one bf16 ULP on a small fraction of elements (tests/test_gpu_parity.py::test_softmax_fq states the tolerance).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_device.h"
#include "qt_chain.h"

namespace {

constexpr int kMaxVec = 8;   // 16-B vectors per lane -> rows up to 64 * 8 * 8 = 4096 columns

struct SoftmaxArgs {
    const uint16_t *scores;
    const uint16_t *mask;     // additive, bf16, or NULL
    uint16_t *out;
    long rows, cols;
    int heads, q_len;         // row = (b * heads + h) * q_len + q
    long mask_sb, mask_sh, mask_sq;   // element strides of the mask for (b, h, q); columns are contiguous
    float scaling;
    qt_format fmt;
    const uint16_t *lut;
    const float *scale;
    uint32_t *amax;
    uint8_t *out8;            // optional FP8 code of the quantized probabilities (E4M3 / E5M2 spec, unit scale)
    int out8_e5m2;
    uint16_t *probs;          // optional: the UNQUANTIZED bf16 probabilities (what torch's softmax returns; the backward of a training step needs them)
    const int *row_live;      // optional, per mask row (same (b, h, q) strides / 8-column granularity as the mask rows, in ROWS): one past
                              // the last column whose mask entry is above -1e30 (qt_mask_row_live); columns from there on are masked
    long live_sb, live_sh, live_sq;
};

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// G lanes per row: 64, or 16 / 32 for short rows (128 / 256 columns: the [16, 12, 128, 128] scores of a training step would leave three
// lanes in four idle with a wave per row), NV vectors per lane
template <int KIND, int NV, int G = 64>
__global__ __launch_bounds__(256) void softmax_fq_kernel(SoftmaxArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % G, sub = lane / G;
    constexpr int RPW = 64 / G;
    Rounder<KIND> rnd{a.fmt, a.lut};
    if constexpr (KIND == kFmtRows) {
        // table formats whose map came with its row form (qt_format.p1 bit 0): the row words in LDS instead of a gather from the
        // 128 KiB map in L2 per probability (csrc/qt_device.h, Rounder<kFmtRows>)
        __shared__ uint4 s_rows[512];
        const uint4 *g = (const uint4 *)(a.lut + QT_MAP_ENTRIES);
        const int nrows = (a.fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rows[i] = g[i];
        rnd.lds = (const uint16_t *)s_rows;
        rnd.glut = a.lut;
        __syncthreads();
    }
    const float s = a.scale ? qt_bf2f(qt_f2bf(*a.scale)) : 1.0f;
    const bool unit = s == 1.0f;
    const UniformDiv dv(s);
    const int nvec_row = (int)(a.cols / 8);
    uint32_t amax = 0;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * RPW; row0 < a.rows; row0 += (long)gridDim.x * 4 * RPW) {
        const long row = row0 + sub < a.rows ? row0 + sub : a.rows - 1;       // (a group past the last row recomputes it and stores nothing)
        const bool row_ok = row0 + sub < a.rows;
        const uint4 *src = (const uint4 *)(a.scores + row * a.cols);
        const uint4 *msk = nullptr;
        if (a.mask) {
            const long q = row % a.q_len, bh = row / a.q_len;
            const long h = bh % a.heads, b = bh / a.heads;
            msk = (const uint4 *)(a.mask + b * a.mask_sb + h * a.mask_sh + q * a.mask_sq);
        }
        // Columns at or beyond `live` are masked (mask <= -1e30): whatever the score, bf16(bf16(score * scaling) + mask) is
        // about -3.4e38 and its exponential 0 -- as long as the row has one unmasked column to carry the maximum (live > 0).
        // A 512-column piece that lies entirely beyond `live` is then not even loaded (under a causal mask: the second piece
        // of the first 512 rows).
        long live = 0x7FFFFFFF;
        if (a.row_live) {
            const long q = row % a.q_len, bh = row / a.q_len;
            live = a.row_live[(bh / a.heads) * a.live_sb + (bh % a.heads) * a.live_sh + q * a.live_sq];
            if (live <= 0) live = 0x7FFFFFFF;                  // a fully masked row is a uniform distribution over ALL columns
        }
        float t[NV][8];
        float mx = -INFINITY;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int iv = v * G + li;
            if (iv < nvec_row && (long)v * G * 8 < live) {
                const uint4 x = src[iv];
                uint4 m = {0u, 0u, 0u, 0u};
                if (msk) m = msk[iv];
                const uint32_t xw[4] = {x.x, x.y, x.z, x.w}, mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // bf16(score * scaling), then bf16(. + mask)
                    uint32_t p = pack_bf16x2(qt_u2f(xw[j] << 16) * a.scaling, qt_u2f(xw[j] & 0xFFFF0000u) * a.scaling);
                    float lo = qt_u2f(p << 16), hi = qt_u2f(p & 0xFFFF0000u);
                    if (msk) {
                        p = pack_bf16x2(lo + qt_u2f(mw[j] << 16), hi + qt_u2f(mw[j] & 0xFFFF0000u));
                        lo = qt_u2f(p << 16);
                        hi = qt_u2f(p & 0xFFFF0000u);
                    }
                    t[v][2 * j] = lo;
                    t[v][2 * j + 1] = hi;
                    mx = fmaxf(mx, fmaxf(lo, hi));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[v][j] = -INFINITY;
            }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.0f;
        // A 16-byte vector whose eight logits all lie more than 110 below the row maximum (a causal mask's -3.4e38)
        // contributes exp(.) == 0 exactly -- v_exp_f32 of anything below -158 is 0 -- so its exponentials, roundings and
        // fake-quant are skipped and zeros are stored: half of all vectors under a causal mask.  The comparison is false
        // for NaN, so NaN logits still take the full path.
        bool dead[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float cut = mx - 110.0f;
            bool d = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) d = d && (t[v][j] < cut);
            dead[v] = d;
            if (!d) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t[v][j] = __expf(t[v][j] - mx);     // v_exp_f32 path; within the 1-bf16-ULP budget stated above
                    sum += t[v][j];
                }
            }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float inv = 1.0f / sum;                 // one division per row; p = e * inv (<= 1 fp32 ULP from e / sum)
        uint4 *dst = (a.out && row_ok) ? (uint4 *)(a.out + row * a.cols) : nullptr;
        uint2 *dst8 = (a.out8 && row_ok) ? (uint2 *)(a.out8 + row * a.cols) : nullptr;
        uint4 *dstp = (a.probs && row_ok) ? (uint4 *)(a.probs + row * a.cols) : nullptr;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int iv = v * G + li;
            if (iv < nvec_row && dead[v]) {                  // fq(0) = +0, FP8 code 0; amax unaffected
                if (dst) dst[iv] = uint4{0u, 0u, 0u, 0u};
                if (dst8) dst8[iv] = uint2{0u, 0u};
                if (dstp) dstp[iv] = uint4{0u, 0u, 0u, 0u};
            } else if (iv < nvec_row) {
                uint32_t w[4], pw[4];
                float f8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t p = pack_bf16x2(t[v][2 * j] * inv, t[v][2 * j + 1] * inv);   // probabilities, bf16
                    pw[j] = p;
                    if (a.amax && row_ok) {
                        uint32_t a0 = (p << 16) & 0x7FFFFFFFu, a1 = p & 0x7FFF0000u;
                        amax = amax > a0 ? amax : a0;
                        amax = amax > a1 ? amax : a1;
                    }
                    uint32_t lo = p << 16, hi = p & 0xFFFF0000u;
                    if constexpr (KIND == QT_FMT_FP_SAT) {
                        if (!dst && dst8) {
                            // FP8 code only: for a probability (finite, 0 <= p <= 1: no saturation, no sign) the format's
                            // round-to-nearest-even IS the hardware conversion of the bf16 value, so the 12-operation
                            // closed form is not needed (checked against it in tests/test_gpu_parity.py)
                            f8[2 * j] = qt_u2f(lo);
                            f8[2 * j + 1] = qt_u2f(hi);
                            continue;
                        }
                    }
                    if (!unit) {
                        uint32_t qd = pack_bf16x2(dv.exact(qt_u2f(lo)), dv.exact(qt_u2f(hi)));
                        lo = qd << 16;
                        hi = qd & 0xFFFF0000u;
                    }
                    const uint32_t r0 = rnd(lo), r1 = rnd(hi);
                    w[j] = unit ? ((r0 >> 16) | (r1 & 0xFFFF0000u)) : pack_bf16x2(qt_u2f(r0) * s, qt_u2f(r1) * s);
                    f8[2 * j] = qt_u2f(r0);
                    f8[2 * j + 1] = qt_u2f(r1);
                }
                if (dst) dst[iv] = uint4{w[0], w[1], w[2], w[3]};
                if (dstp) dstp[iv] = uint4{pw[0], pw[1], pw[2], pw[3]};
                if (dst8) {
                    if (a.out8_e5m2) dst8[iv] = uint2{qt_pack_fp8x4<true>(f8[0], f8[1], f8[2], f8[3]), qt_pack_fp8x4<true>(f8[4], f8[5], f8[6], f8[7])};
                    else dst8[iv] = uint2{qt_pack_fp8x4<false>(f8[0], f8[1], f8[2], f8[3]), qt_pack_fp8x4<false>(f8[4], f8[5], f8[6], f8[7])};
                }
            }
        }
    }
    if (a.amax) {
        // one atomic per workgroup at most (the slot starts a training step at zero: every wave could raise it, and same-address atomics
        // serialise)
        __shared__ uint32_t s_am[4];
        amax = wave_max_u32(amax);
        if (lane == 0) s_am[wave] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = s_am[0];
            m = m > s_am[1] ? m : s_am[1];
            m = m > s_am[2] ? m : s_am[2];
            m = m > s_am[3] ? m : s_am[3];
            if (m != 0u && m > __hip_atomic_load(a.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.amax, m);
        }
    }
}

// ---- backward of scale -> (+ mask) -> softmax in a training step, with the fake-quantizer chain behind it ---------------------------
// torch runs _softmax_backward_data (fp32: (dP - sum dP P) P, one rounding to bf16) and the scaling's backward (x scaling, another
// rounding); the result is the grad_output of qk_matmul, which its backward-pre hook fake-quantizes (quantize.py:116-179).  One wave
// per row, rows up to 64 * 8 * kMaxVec columns; P: the unquantized probabilities the forward kept.
struct SoftmaxBwdArgs {
    const uint16_t *dp, *p;
    uint16_t *ds;
    long rows, cols;
    float scaling;
    ChainStageDev st[kChainMax];
};

// G lanes per row (16 / 32 / 64: rows of 128 / 256 / more columns keep every lane busy), NV vectors per lane
template <int KIND, int NS, int NV, int G>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(SoftmaxBwdArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, 256);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G, li = lane % G;
    constexpr int RPW = 64 / G;                                  // rows per wave
    float sc[NS];
    uint32_t amax[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        sc[i] = a.st[i].scale ? qt_bf2f(qt_f2bf(*a.st[i].scale)) : 1.0f;
        amax[i] = 0u;
    }
    const int nvec_row = (int)(a.cols / 8);
    for (long row0 = ((long)blockIdx.x * 4 + wave) * RPW; row0 < a.rows; row0 += (long)gridDim.x * 4 * RPW) {
        const long row = row0 + sub;
        const bool live = row < a.rows;
        const uint4 *gp = (const uint4 *)(a.dp + row * a.cols), *pp = (const uint4 *)(a.p + row * a.cols);
        float g[NV][8], pr[NV][8];
        float dot = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int iv = v * G + li;
            if (live && iv < nvec_row) {
                const uint4 gv = gp[iv], pv = pp[iv];
                const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, pw[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    g[v][2 * j] = bf_lo(gw[j]); g[v][2 * j + 1] = bf_hi(gw[j]);
                    pr[v][2 * j] = bf_lo(pw[j]); pr[v][2 * j + 1] = bf_hi(pw[j]);
                    dot += g[v][2 * j] * pr[v][2 * j];
                    dot += g[v][2 * j + 1] * pr[v][2 * j + 1];
                }
            }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
        uint4 *dst = (uint4 *)(a.ds + row * a.cols);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int iv = v * G + li;
            if (live && iv < nvec_row) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t d1 = pack_bf16x2((g[v][2 * j] - dot) * pr[v][2 * j], (g[v][2 * j + 1] - dot) * pr[v][2 * j + 1]);     // softmax backward, bf16
                    o[j] = pack_bf16x2(bf_lo(d1) * a.scaling, bf_hi(d1) * a.scaling);                                                  // x scaling, bf16
                }
                const uint4 dsv = uint4{o[0], o[1], o[2], o[3]};
                const size_t idx = (size_t)(row * nvec_row + iv);
                dst[iv] = dsv;
                uint4 res[NS];
                chain_stages<KIND, NS>(a.st, sc, rnd, dsv, idx, amax, res);
            }
        }
    }
    __shared__ uint32_t s_amax[NS][4];
    chain_amax_commit<NS, 256>(a.st, amax, s_amax);
}

template <int KIND>
int launch_softmax(const SoftmaxArgs &a, hipStream_t st) {
    const long nvec_row = a.cols / 8;
    const int nv = (int)((nvec_row + 63) / 64);
    static int cus = 0;      // queried once (never inside a stream capture after the first call)
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
                  ? p.multiProcessorCount : 256;
    }
    long want = (a.rows + 3) / 4;
    long cap = (long)cus * (a.amax ? 3 : 16);       // observed (a training step): one amax atomic per workgroup, so fewer, longer workgroups
    unsigned grid = (unsigned)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (nvec_row <= 32 && !a.row_live) {                        // short rows: several rows per wave
        const int g = nvec_row <= 16 ? 16 : 32;
        const long per_wg = 4L * (64 / g);
        want = (a.rows + per_wg - 1) / per_wg;
        grid = (unsigned)(want < cap ? want : cap);
        if (grid < 1) grid = 1;
        if (g == 16) softmax_fq_kernel<KIND, 1, 16><<<grid, 256, 0, st>>>(a);
        else softmax_fq_kernel<KIND, 1, 32><<<grid, 256, 0, st>>>(a);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? QT_OK : (int)e;
    }
    switch (nv) {
        case 1: softmax_fq_kernel<KIND, 1><<<grid, 256, 0, st>>>(a); break;
        case 2: softmax_fq_kernel<KIND, 2><<<grid, 256, 0, st>>>(a); break;
        case 3: case 4: softmax_fq_kernel<KIND, 4><<<grid, 256, 0, st>>>(a); break;
        default: softmax_fq_kernel<KIND, kMaxVec><<<grid, 256, 0, st>>>(a); break;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

}  // namespace

extern "C" int qt_softmax_fq_bf16(const uint16_t *scores, const uint16_t *mask, uint16_t *out, long batch, int heads,
                                  int q_len, long cols, long mask_sb, long mask_sh, long mask_sq, float scaling,
                                  const qt_format *fmt, const uint16_t *lut, const float *scale, uint32_t *amax,
                                  void *stream) {
    const long rows = batch * heads * q_len;
    if (rows == 0 || cols == 0) return QT_OK;
    if (!scores || !out || !fmt || batch < 0 || heads < 1 || q_len < 1 || cols < 0) return QT_ERR_BAD_ARG;
    if (fmt->kind == QT_FMT_LUT && !lut) return QT_ERR_BAD_ARG;
    if (cols > 64L * 8 * kMaxVec) return QT_ERR_BAD_ARG;
    if ((cols & 7) || (((uintptr_t)scores | (uintptr_t)out | (uintptr_t)mask) & 15u) ||
        (mask && ((mask_sb | mask_sh | mask_sq) & 7)))
        return QT_ERR_UNALIGNED;
    SoftmaxArgs a{scores, mask, out, rows, cols, heads, q_len, mask_sb, mask_sh, mask_sq, scaling, *fmt, lut, scale, amax, nullptr, 0, nullptr};
    hipStream_t st = (hipStream_t)stream;
    switch (fmt->kind) {
        case QT_FMT_LUT: return (fmt->p1 & 1) ? launch_softmax<kFmtRows>(a, st) : launch_softmax<QT_FMT_LUT>(a, st);
        case QT_FMT_FP_SAT: return launch_softmax<QT_FMT_FP_SAT>(a, st);
        case QT_FMT_INT: return launch_softmax<QT_FMT_INT>(a, st);
        case QT_FMT_IDENTITY: return launch_softmax<QT_FMT_IDENTITY>(a, st);
        default: return QT_ERR_BAD_ARG;
    }
}

namespace {
// one wave per mask row: one past the last column whose entry is above -1e30.  With `irregular`: rows that are not exactly "zeros up
// to that column, the bf16 minimum from there on" (causal masks and right padding are) set it to 1
__global__ __launch_bounds__(256) void mask_row_live_kernel(const uint16_t *mask, long rows, long cols, long row_stride, int *out, int *irregular) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t *m = mask + row * row_stride;
    int last = 0;
    for (long c = lane; c < cols; c += 64)
        if (qt_bf2f(m[c]) > -1e30f) last = (int)c + 1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
    if (lane == 0) out[row] = last;
    if (irregular) {
        bool bad = false;
        for (long c = lane; c < cols; c += 64) bad |= c < last ? (m[c] & 0x7FFFu) != 0 : m[c] != 0xFF7Fu;
        if (__any(bad) && lane == 0) atomicOr(irregular, 1);
    }
}
}  // namespace

extern "C" int qt_mask_row_live(const uint16_t *mask, long rows, long cols, long row_stride, int *out, void *stream) {
    if (rows == 0) return QT_OK;
    if (!mask || !out || rows < 0 || cols < 0 || row_stride < cols) return QT_ERR_BAD_ARG;
    mask_row_live_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(mask, rows, cols, row_stride, out, nullptr);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

extern "C" int qt_mask_row_live_checked(const uint16_t *mask, long rows, long cols, long row_stride, int *out, int *irregular_dev, void *stream) {
    if (!irregular_dev) return QT_ERR_BAD_ARG;
    if (hipMemsetAsync(irregular_dev, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return QT_ERR_BAD_ARG;
    if (rows == 0) return QT_OK;
    if (!mask || !out || rows < 0 || cols < 0 || row_stride < cols) return QT_ERR_BAD_ARG;
    mask_row_live_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(mask, rows, cols, row_stride, out, irregular_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

extern "C" int qt_softmax_fq_bf16_fp8_live(const uint16_t *scores, const uint16_t *mask, uint8_t *out8, long batch, int heads, int q_len,
                                           long cols, long mask_sb, long mask_sh, long mask_sq, float scaling, const qt_format *fmt,
                                           const int *row_live, long live_sb, long live_sh, long live_sq, void *stream) {
    const long rows = batch * heads * q_len;
    if (rows == 0 || cols == 0) return QT_OK;
    if (!scores || !out8 || !fmt || !mask || !row_live || batch < 0 || heads < 1 || q_len < 1 || cols < 0 || fmt->kind != QT_FMT_FP_SAT)
        return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if (cols > 64L * 8 * kMaxVec) return QT_ERR_BAD_ARG;
    if ((cols & 7) || (((uintptr_t)scores | (uintptr_t)mask) & 15u) || ((uintptr_t)out8 & 7u) || ((mask_sb | mask_sh | mask_sq) & 7))
        return QT_ERR_UNALIGNED;
    SoftmaxArgs a{scores, mask, nullptr, rows, cols, heads, q_len, mask_sb, mask_sh, mask_sq, scaling, *fmt, nullptr, nullptr, nullptr,
                  out8, e5m2 ? 1 : 0, nullptr, row_live, live_sb, live_sh, live_sq};
    return launch_softmax<QT_FMT_FP_SAT>(a, (hipStream_t)stream);
}

extern "C" int qt_softmax_fq_bf16_fp8(const uint16_t *scores, const uint16_t *mask, uint16_t *out, uint8_t *out8, long batch,
                                      int heads, int q_len, long cols, long mask_sb, long mask_sh, long mask_sq, float scaling,
                                      const qt_format *fmt, void *stream) {
    const long rows = batch * heads * q_len;
    if (rows == 0 || cols == 0) return QT_OK;
    if (!scores || !out8 || !fmt || batch < 0 || heads < 1 || q_len < 1 || cols < 0 || fmt->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if (cols > 64L * 8 * kMaxVec) return QT_ERR_BAD_ARG;
    if ((cols & 7) || (((uintptr_t)scores | (uintptr_t)out | (uintptr_t)mask) & 15u) || ((uintptr_t)out8 & 7u) ||
        (mask && ((mask_sb | mask_sh | mask_sq) & 7)))
        return QT_ERR_UNALIGNED;
    SoftmaxArgs a{scores, mask, out, rows, cols, heads, q_len, mask_sb, mask_sh, mask_sq, scaling, *fmt, nullptr, nullptr, nullptr,
                  out8, e5m2 ? 1 : 0, nullptr};
    return launch_softmax<QT_FMT_FP_SAT>(a, (hipStream_t)stream);
}

// The same pass for a training step: also writes the unquantized probabilities (probs_dev, nullable) the backward needs.
extern "C" int qt_softmax_fq_probs_bf16(const uint16_t *scores, const uint16_t *mask, uint16_t *out, uint16_t *probs_dev, long batch, int heads,
                                        int q_len, long cols, long mask_sb, long mask_sh, long mask_sq, float scaling, const qt_format *fmt,
                                        const uint16_t *lut, const float *scale, uint32_t *amax, void *stream) {
    const long rows = batch * heads * q_len;
    if (rows == 0 || cols == 0) return QT_OK;
    if (!scores || !out || !fmt || batch < 0 || heads < 1 || q_len < 1 || cols < 0) return QT_ERR_BAD_ARG;
    if (fmt->kind == QT_FMT_LUT && !lut) return QT_ERR_BAD_ARG;
    if (cols > 64L * 8 * kMaxVec) return QT_ERR_BAD_ARG;
    if ((cols & 7) || (((uintptr_t)scores | (uintptr_t)out | (uintptr_t)mask | (uintptr_t)probs_dev) & 15u) ||
        (mask && ((mask_sb | mask_sh | mask_sq) & 7)))
        return QT_ERR_UNALIGNED;
    SoftmaxArgs a{scores, mask, out, rows, cols, heads, q_len, mask_sb, mask_sh, mask_sq, scaling, *fmt, lut, scale, amax, nullptr, 0, probs_dev};
    hipStream_t st = (hipStream_t)stream;
    switch (fmt->kind) {
        case QT_FMT_LUT: return (fmt->p1 & 1) ? launch_softmax<kFmtRows>(a, st) : launch_softmax<QT_FMT_LUT>(a, st);
        case QT_FMT_FP_SAT: return launch_softmax<QT_FMT_FP_SAT>(a, st);
        case QT_FMT_INT: return launch_softmax<QT_FMT_INT>(a, st);
        case QT_FMT_IDENTITY: return launch_softmax<QT_FMT_IDENTITY>(a, st);
        default: return QT_ERR_BAD_ARG;
    }
}

namespace {
template <int KIND, int NS>
int launch_softmax_bwd(const SoftmaxBwdArgs &a, const qt_format &fmt, const uint16_t *lut, hipStream_t st) {
    const int nvec = (int)(a.cols / 8);
    const int g = nvec <= 16 ? 16 : (nvec <= 32 ? 32 : 64);
    const long per_wg = 4L * (64 / g);
    long want = (a.rows + per_wg - 1) / per_wg;
    if (want > 768) want = 768;                                 // (one amax atomic per stage and workgroup)
    const unsigned grid = (unsigned)(want < 1 ? 1 : want);
    if (g == 16) softmax_bwd_kernel<KIND, NS, 1, 16><<<grid, 256, 0, st>>>(a, fmt, lut);
    else if (g == 32) softmax_bwd_kernel<KIND, NS, 1, 32><<<grid, 256, 0, st>>>(a, fmt, lut);
    else if (nvec <= 64) softmax_bwd_kernel<KIND, NS, 1, 64><<<grid, 256, 0, st>>>(a, fmt, lut);
    else if (nvec <= 128) softmax_bwd_kernel<KIND, NS, 2, 64><<<grid, 256, 0, st>>>(a, fmt, lut);
    else softmax_bwd_kernel<KIND, NS, 4, 64><<<grid, 256, 0, st>>>(a, fmt, lut);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}
}  // namespace

extern "C" int qt_softmax_backward_chain_bf16(const uint16_t *grad_probs_dev, const uint16_t *probs_dev, uint16_t *grad_scores_dev, long rows, long cols,
                                              float scaling, const qt_chain_stage *stages, int nstage, const qt_format *fmt, const uint16_t *lut_dev,
                                              void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!grad_probs_dev || !probs_dev || !grad_scores_dev || !stages || !fmt || rows < 0 || cols < 8 || cols % 8 || cols > 64L * 8 * 4 || nstage < 1 ||
        nstage > 2)
        return QT_ERR_BAD_ARG;
    if (((uintptr_t)grad_probs_dev | (uintptr_t)probs_dev | (uintptr_t)grad_scores_dev) & 15u) return QT_ERR_UNALIGNED;
    SoftmaxBwdArgs a{};
    a.dp = grad_probs_dev; a.p = probs_dev; a.ds = grad_scores_dev; a.rows = rows; a.cols = cols; a.scaling = scaling;
    for (int i = 0; i < nstage; ++i) {
        if (stages[i].src >= i || stages[i].src < -1) return QT_ERR_BAD_ARG;
        if ((uintptr_t)stages[i].out_dev & 15u) return QT_ERR_UNALIGNED;
        a.st[i] = ChainStageDev{stages[i].scale_f32_dev, stages[i].amax_bits_dev, (uint4 *)stages[i].out_dev, stages[i].src};
    }
    hipStream_t st = (hipStream_t)stream;
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut_dev || !(fmt->p1 & 1)) return QT_ERR_BAD_DTYPE;
            return nstage == 1 ? launch_softmax_bwd<kFmtRows, 1>(a, *fmt, lut_dev, st) : launch_softmax_bwd<kFmtRows, 2>(a, *fmt, lut_dev, st);
        case QT_FMT_FP_SAT: return nstage == 1 ? launch_softmax_bwd<QT_FMT_FP_SAT, 1>(a, *fmt, lut_dev, st) : launch_softmax_bwd<QT_FMT_FP_SAT, 2>(a, *fmt, lut_dev, st);
        case QT_FMT_INT: return nstage == 1 ? launch_softmax_bwd<QT_FMT_INT, 1>(a, *fmt, lut_dev, st) : launch_softmax_bwd<QT_FMT_INT, 2>(a, *fmt, lut_dev, st);
        default: return QT_ERR_BAD_DTYPE;
    }
}
