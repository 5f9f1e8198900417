// The model's own elementwise chains between the fake-quantized GEMMs of a LLaMA block, one launch each.
// Hugging Face runs them as 3-8 torch kernels apiece (~9 of the 24 ms of a LLaMA-2-7B window once everything on the
// fake-quant path was fused); the arithmetic below repeats their operation order and every bf16 rounding point:
//   RMSNorm      x32 = float(x); v = mean(x32^2); h = bf16(x32 * rsqrt(v + eps)); y = bf16(w * h)
//   SiLU * up    a = bf16(g32 / (1 + exp(-g32)));  y = bf16(a * u)
//   rotary       y = bf16( bf16(x * cos) + bf16(rotate_half(x) * sin) ),  rotate_half(x) = cat(-x[D/2:], x[:D/2])
// SiLU*up and rotary are bit-identical to the torch chains; RMSNorm differs only through the summation order of the
// mean (last-bit differences of v can move isolated outputs by one bf16 ulp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_formats.h"
#include "qt_value_codes.h"
#include "qt_value_rows.h"

namespace {

__device__ __forceinline__ float bf_lo(uint32_t w) { return qt_u2f(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return qt_u2f(w & 0xFFFF0000u); }
__device__ __forceinline__ float rbf(float f) { return qt_u2f(pack_bf16x2(f, 0.0f) << 16); }      // round to bf16, keep as float

constexpr int kNormThreads = 256;
constexpr int kNormMaxVec = 8;        // 16-byte vectors per thread: rows up to 256 * 8 * 8 = 16384 elements

// FQ: 0 plain; 1 / 2 = the first consumer's stateless E4M3 / E5M2 fake-quantizer applied to the result, which is
// written as bf16 plus its FP8 code (producer-fused fake-quant, model_fusions.py)
// ADD: the row is bf16(x + res) (the residual add in front of the norm, torch's rounding), also written to `sum`
// EXTRA: 1 or 2 further consumers of the result (k, v beside q; up beside gate): each one's own stateless E4M3 / E5M2 fake-quantizer is
// evaluated on the unquantized result too and its FP8 codes written to its own buffer -- the calls their hooks would otherwise make
// as separate launches over the tensor this kernel has in registers
struct NormExtra {
    uint2 *y8[2];
    qt_format fmt[2];
    int e5m2[2];
};

// NV: 16-byte vectors per thread the loops are unrolled for (2, 4 or 8, picked by the row length: the headline's 4096-element rows
// need 2 -- unrolled for 8 the kernel was 35 KB of code, most of it jumped over, fetched with a cold instruction cache on every one of
// its 63 launches per window)
template <int FQ, bool ADD = false, int EXTRA = 0, int NV = kNormMaxVec>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ w,
                                                               uint4 *__restrict__ y, int nvec, float inv_cols, float eps,
                                                               uint2 *__restrict__ y8, qt_format fmt,
                                                               const uint4 *__restrict__ res = nullptr, uint4 *__restrict__ sum = nullptr,
                                                               NormExtra extra = NormExtra{}, int sum_fq = 0, qt_format sum_fmt = qt_format{},
                                                               const uint16_t *__restrict__ map = nullptr) {
    __shared__ float s_part[kNormThreads / 64];
    // FQ == 3: the consumers' stateless TABLE-format fake-quantizer in its row form (qt_format.p1 bit 0; `map` = the 65 536 entries with
    // the row words behind them, csrc/qt_device.h Rounder<kFmtRows>); bf16 values only
    // (the 4 - 8 KiB row table is read where it lies, behind the map in global memory: a row of the tensor is two or three vectors per
    // lane, and staging the table into LDS put a load, a store and a barrier in front of them)
    Rounder<kFmtRows> rnd{fmt, FQ == 3 ? map + QT_MAP_ENTRIES : nullptr, map};
    const size_t row = blockIdx.x;
    const uint4 *xr = x + row * (size_t)nvec;
    uint4 v[NV];
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = threadIdx.x + i * kNormThreads;
        if (c < nvec) {
            v[i] = xr[c];
            if constexpr (ADD) {
                const uint4 r = res[row * (size_t)nvec + c];
                v[i].x = pack_bf16x2(bf_lo(v[i].x) + bf_lo(r.x), bf_hi(v[i].x) + bf_hi(r.x));
                v[i].y = pack_bf16x2(bf_lo(v[i].y) + bf_lo(r.y), bf_hi(v[i].y) + bf_hi(r.y));
                v[i].z = pack_bf16x2(bf_lo(v[i].z) + bf_lo(r.z), bf_hi(v[i].z) + bf_hi(r.z));
                v[i].w = pack_bf16x2(bf_lo(v[i].w) + bf_lo(r.w), bf_hi(v[i].w) + bf_hi(r.w));
                if (sum_fq) {            // the NEXT residual add reads fq(sum) (PT2E graphs quantize an add's earlier operand): written quantized,
                    uint32_t t[4] = {v[i].x, v[i].y, v[i].z, v[i].w};       // normalised unquantized
                    if constexpr (FQ == 3) {                                 // (sum_fq == 3: the same table format as the result's, row form)
                        const uint32_t tin[4] = {t[0], t[1], t[2], t[3]};
                        fq_rows_words<4, false>(tin, t, rnd);
                    } else if (sum_fq == 2) fq8_hw_vec8<true>(t, sum_fmt);
                    else fq8_hw_vec8<false>(t, sum_fmt);
                    sum[row * (size_t)nvec + c] = uint4{t[0], t[1], t[2], t[3]};
                } else {
                    sum[row * (size_t)nvec + c] = v[i];
                }
            }
            const uint32_t q[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(q[j]), b = bf_hi(q[j]);
                ss += a * a;
                ss += b * b;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < kNormThreads / 64; ++i) tot += s_part[i];
    const float r = rsqrtf(tot * inv_cols + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = threadIdx.x + i * kNormThreads;
        if (c < nvec) {
            const uint4 ww = w[c];
            const uint32_t q[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, g[4] = {ww.x, ww.y, ww.z, ww.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h0 = rbf(bf_lo(q[j]) * r), h1 = rbf(bf_hi(q[j]) * r);
                o[j] = pack_bf16x2(bf_lo(g[j]) * h0, bf_hi(g[j]) * h1);
            }
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) {
                uint32_t t[4] = {o[0], o[1], o[2], o[3]};
                extra.y8[e][row * (size_t)nvec + c] = extra.e5m2[e] ? fq8_hw_vec8<true>(t, extra.fmt[e]) : fq8_hw_vec8<false>(t, extra.fmt[e]);
            }
            if constexpr (FQ == 1 || FQ == 2) y8[row * (size_t)nvec + c] = fq8_hw_vec8<FQ == 2>(o, fmt);
            if constexpr (FQ == 3) {
                const uint32_t oin[4] = {o[0], o[1], o[2], o[3]};
                fq_rows_words<4, false>(oin, o, rnd);
            }
            if ((FQ != 1 && FQ != 2) || y) y[row * (size_t)nvec + c] = uint4{o[0], o[1], o[2], o[3]};      // NULL with FP8 codes: codes only
        }
    }
}

template <int FQ, bool ADD, int EXTRA, typename... A>
void launch_rms(unsigned rows, int nvec, hipStream_t st, A... args) {
    if (nvec <= 2 * kNormThreads) rmsnorm_kernel<FQ, ADD, EXTRA, 2><<<rows, kNormThreads, 0, st>>>(args...);
    else if (nvec <= 4 * kNormThreads) rmsnorm_kernel<FQ, ADD, EXTRA, 4><<<rows, kNormThreads, 0, st>>>(args...);
    else rmsnorm_kernel<FQ, ADD, EXTRA, 8><<<rows, kNormThreads, 0, st>>>(args...);
}

// gate / up rows may be column slices of one wider GEMM output: row r of an operand starts at r * rs vectors (rs = cv,
// the row length in vectors, when contiguous); the result is always contiguous
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ u,
                                                       uint4 *__restrict__ y, size_t nvec, size_t cv, size_t rs_g, size_t rs_u) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const size_t row = i / cv, col = i - row * cv;
        const uint4 a = g[row * rs_g + col], b = u[row * rs_u + col];
        const uint32_t p[4] = {a.x, a.y, a.z, a.w}, q[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g0 = bf_lo(p[j]), g1 = bf_hi(p[j]);
            const float s0 = rbf(g0 / (1.0f + expf(-g0))), s1 = rbf(g1 / (1.0f + expf(-g1)));
            o[j] = pack_bf16x2(s0 * bf_lo(q[j]), s1 * bf_hi(q[j]));
        }
        y[i] = uint4{o[0], o[1], o[2], o[3]};
    }
}

// SiLU * up with the consumer's stateless TABLE-format fake-quantizer in its row form applied to the product (bf16 values out)
__global__ __launch_bounds__(256) void silu_mul_map_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ u, uint4 *__restrict__ y,
                                                           size_t nvec, qt_format fmt, const uint16_t *__restrict__ map, size_t cv, size_t rs_g,
                                                           size_t rs_u) {
    __shared__ uint4 s_rows[512];
    {
        const uint4 *gr = (const uint4 *)(map + QT_MAP_ENTRIES);
        const int nrows = (fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rows[i] = gr[i];
        __syncthreads();
    }
    const Rounder<kFmtRows> rnd{fmt, (const uint16_t *)s_rows, map};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const size_t row = i / cv, col = i - row * cv;
        const uint4 a = g[row * rs_g + col], b = u[row * rs_u + col];
        const uint32_t p[4] = {a.x, a.y, a.z, a.w}, q[4] = {b.x, b.y, b.z, b.w};
        uint32_t pr[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g0 = bf_lo(p[j]), g1 = bf_hi(p[j]);
            const float s0 = rbf(g0 / (1.0f + expf(-g0))), s1 = rbf(g1 / (1.0f + expf(-g1)));
            pr[j] = pack_bf16x2(s0 * bf_lo(q[j]), s1 * bf_hi(q[j]));
        }
        fq_rows_words<4, false>(pr, o, rnd);
        y[i] = uint4{o[0], o[1], o[2], o[3]};
    }
}

// SiLU * up with the consumer's fake-quantizer (stateless E4M3 / E5M2, unit scale) applied to the product on the way
// out: writes the quantized bf16 tensor and its FP8 code, i.e. silu_mul_kernel + fq8_kernel without the round trip of
// the intermediate through HBM.  Same values as the two-kernel sequence, bit for bit.
template <bool E5M2>
__global__ __launch_bounds__(256) void silu_mul_fq8_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ u,
                                                           uint4 *__restrict__ y, uint2 *__restrict__ y8, size_t nvec,
                                                           qt_format fmt, size_t cv, size_t rs_g, size_t rs_u) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const size_t row = i / cv, col = i - row * cv;
        const uint4 a = g[row * rs_g + col], b = u[row * rs_u + col];
        const uint32_t p[4] = {a.x, a.y, a.z, a.w}, q[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g0 = bf_lo(p[j]), g1 = bf_hi(p[j]);
            const float s0 = rbf(g0 / (1.0f + expf(-g0))), s1 = rbf(g1 / (1.0f + expf(-g1)));
            o[j] = pack_bf16x2(s0 * bf_lo(q[j]), s1 * bf_hi(q[j]));                        // the unquantized product, bf16
        }
        const uint2 codes = fq8_hw_vec8<E5M2>(o, fmt);
        y[i] = uint4{o[0], o[1], o[2], o[3]};
        y8[i] = codes;
    }
}

struct RopeArgs {
    const uint16_t *x;       // [B][S][H][D] order; a (b, s) row starts every `rsv` vectors (>= H * D / 8: the projections
                             // may be column slices of one wider GEMM output)
    uint16_t *y;
    const uint16_t *cos, *sin;   // [B][S][D]
    long B, S, H, D;
    size_t nvec;             // B * S * H * D / 8
    long rsv;
};

__device__ __forceinline__ void rope_one(const RopeArgs &a, size_t i) {
    const long dv = a.D / 8;                             // vectors per head row
    const long d8 = (long)(i % dv);
    const size_t bsh = i / dv;
    const size_t bs = bsh / (size_t)a.H;
    const long half = dv / 2;
    const bool low = d8 < half;
    const size_t in_row = bs * (size_t)a.rsv + (bsh % (size_t)a.H) * dv;      // vector index of x[b][s][h][0]
    const uint4 xv = *(const uint4 *)(a.x + (in_row + d8) * 8);
    const uint4 pv = *(const uint4 *)(a.x + (in_row + (low ? d8 + half : d8 - half)) * 8);     // rotate_half partner
    const uint4 cv = *(const uint4 *)(a.cos + (bs * dv + d8) * 8);
    const uint4 sv = *(const uint4 *)(a.sin + (bs * dv + d8) * 8);
    const uint32_t X[4] = {xv.x, xv.y, xv.z, xv.w}, P[4] = {pv.x, pv.y, pv.z, pv.w};
    const uint32_t C[4] = {cv.x, cv.y, cv.z, cv.w}, S[4] = {sv.x, sv.y, sv.z, sv.w};
    const float sgn = low ? -1.0f : 1.0f;
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = rbf(bf_lo(X[j]) * bf_lo(C[j])), a1 = rbf(bf_hi(X[j]) * bf_hi(C[j]));
        const float b0 = rbf(sgn * bf_lo(P[j]) * bf_lo(S[j])), b1 = rbf(sgn * bf_hi(P[j]) * bf_hi(S[j]));
        o[j] = pack_bf16x2(a0 + b0, a1 + b1);
    }
    *(uint4 *)(a.y + i * 8) = uint4{o[0], o[1], o[2], o[3]};
}

__global__ __launch_bounds__(256) void rope_kernel(RopeArgs q, RopeArgs k) {
    const size_t total = q.nvec + k.nvec;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (i < q.nvec) rope_one(q, i);
        else rope_one(k, i - q.nvec);
    }
}

// rotary + the two qk_matmul input fake-quantizers (stateless E4M3 / E5M2, unit scale): output in [B][H][S][D]
// order, contiguous -- the layout the fake-quant pass of a permuted view produces (csrc/qt_elementwise.hip,
// fq_rows_kernel) -- so rope_kernel + two fq_rows_kernel launches become one.
struct RopeFqArgs {
    RopeArgs r;              // r.y is the [B][H][S][D] output
    qt_format fmt;
    uint8_t *y8;             // optional FP8 code of the output, same order (NULL = not wanted)
    int e5m2;
    int inner;               // 1: x * cos goes through `inner_fmt` (stateless closed-form FP) before the sum -- PT2E graphs fake-quantize the
    qt_format inner_fmt;     // earlier-defined operand of the rotary's add (xnnpack_quantizer_utils.py:232-282)
    const uint16_t *map;     // fmt.kind == QT_FMT_LUT: the map with its row words behind it (row form; rows staged in LDS by the kernel)
    const uint16_t *rows_lds;
};

// One workgroup per token (b, s): its vectors are the H * D / 8 of a row of x, so the only divisions left are one per token
// (scalar) and one 32-bit division per vector (64-bit index arithmetic per vector cost more than the rotation itself).
// A token with fewer than 256 vectors (BERT-base: 96) would leave most of the workgroup idle: a workgroup then takes `tpb` consecutive
// tokens and spreads their vectors over its threads (two more 32-bit divisions per vector).
// CODES: the launch's one mode is "FP8 codes by the hardware conversion, no inner quantizer, no table" (the headline's): the other modes'
// code is not even in the kernel then -- it is all jumped over at run time anyway, but a launch walks its code once with a cold
// instruction cache, and every jump over a few KB is another miss (the norm kernels' lesson)
template <bool CODES = false>
__device__ __forceinline__ void rope_fq_token(const RopeFqArgs &a, size_t bs0, uint32_t tpb, size_t tokens) {
    const uint32_t dv = (uint32_t)a.r.D / 8, half = dv / 2, nv = (uint32_t)a.r.H * dv;
    const uint32_t S = (uint32_t)a.r.S;
    for (uint32_t i = threadIdx.x; i < tpb * nv; i += 256) {
        const uint32_t lt = tpb == 1 ? 0u : i / nv, v = i - lt * nv;
        const size_t bs = bs0 + lt;
        if (bs >= tokens) break;
        const uint32_t b = (uint32_t)(bs / S), s = (uint32_t)(bs - (size_t)b * S);
        const uint16_t *xrow = a.r.x + bs * (size_t)a.r.rsv * 8;
        const uint16_t *crow = a.r.cos + bs * dv * 8, *srow = a.r.sin + bs * dv * 8;
        const uint32_t h = v / dv, d8 = v - h * dv;
        const bool low = d8 < half;
        const uint4 xv = *(const uint4 *)(xrow + (size_t)v * 8);
        uint32_t out[4] = {xv.x, xv.y, xv.z, xv.w};
        if (a.r.cos) {                                       // NULL: no rotation (BERT's transpose_for_scores + the two fake-quantizers)
            const uint4 pv = *(const uint4 *)(xrow + (size_t)(low ? v + half : v - half) * 8);      // rotate_half partner
            const uint4 cv = *(const uint4 *)(crow + d8 * 8);
            const uint4 sv = *(const uint4 *)(srow + d8 * 8);
            const uint32_t X[4] = {xv.x, xv.y, xv.z, xv.w}, P[4] = {pv.x, pv.y, pv.z, pv.w};
            const uint32_t C[4] = {cv.x, cv.y, cv.z, cv.w}, S[4] = {sv.x, sv.y, sv.z, sv.w};
            const float sgn = low ? -1.0f : 1.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a0 = rbf(bf_lo(X[j]) * bf_lo(C[j])), a1 = rbf(bf_hi(X[j]) * bf_hi(C[j]));
                if (!CODES && a.inner) {
                    if (a.map) {                                         // table format: the same map as the result's (host-checked)
                        const Rounder<kFmtRows> rin{a.fmt, a.rows_lds, a.map};
                        a0 = qt_u2f(rin(qt_f2u(a0)));
                        a1 = qt_u2f(rin(qt_f2u(a1)));
                    } else {
                        a0 = qt_u2f(qt_fp_sat_u32(qt_f2u(a0), a.inner_fmt.p0, a.inner_fmt.p1, a.inner_fmt.fhi));
                        a1 = qt_u2f(qt_fp_sat_u32(qt_f2u(a1), a.inner_fmt.p0, a.inner_fmt.p1, a.inner_fmt.fhi));
                    }
                }
                const float b0 = rbf(sgn * bf_lo(P[j]) * bf_lo(S[j])), b1 = rbf(sgn * bf_hi(P[j]) * bf_hi(S[j]));
                out[j] = pack_bf16x2(a0 + b0, a1 + b1);                  // the rotary output, bf16
            }
        }
        const size_t o = (((size_t)b * (size_t)a.r.H + h) * (size_t)a.r.S + s) * dv + d8;     // [B][H][S][D] order
        if (CODES || a.y8) {                                 // exact E4M3 / E5M2 (checked on the host): hardware conversion
            const uint2 codes = a.e5m2 ? fq8_hw_vec8<true>(out, a.fmt) : fq8_hw_vec8<false>(out, a.fmt);
            *(uint2 *)(a.y8 + o * 8) = codes;
        } else if (a.map) {                                  // table format, row form
            const Rounder<kFmtRows> rnd{a.fmt, a.rows_lds, a.map};
            const uint32_t oin[4] = {out[0], out[1], out[2], out[3]};
            fq_rows_words<4, false>(oin, out, rnd);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t r0 = qt_fp_sat_u32(out[j] << 16, a.fmt.p0, a.fmt.p1, a.fmt.fhi);
                const uint32_t r1 = qt_fp_sat_u32(out[j] & 0xFFFF0000u, a.fmt.p0, a.fmt.p1, a.fmt.fhi);
                out[j] = (r0 >> 16) | (r1 & 0xFFFF0000u);
            }
        }
        if (a.r.y) *(uint4 *)(a.r.y + o * 8) = uint4{out[0], out[1], out[2], out[3]};     // NULL: codes only (they decode to these values)
    }
}

__global__ __launch_bounds__(256) void rope_fq_kernel(RopeFqArgs q, RopeFqArgs k, unsigned tpb) {
    __shared__ uint4 s_rope_rows[512];
    if (q.map) {                                             // one table format for q and k (checked on the host)
        const uint4 *gr = (const uint4 *)(q.map + QT_MAP_ENTRIES);
        const int nrows = (q.fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rope_rows[i] = gr[i];
        __syncthreads();
        q.rows_lds = k.rows_lds = (const uint16_t *)s_rope_rows;
    }
    const size_t tokens = (size_t)q.r.B * (size_t)q.r.S;                 // q and k hold the same tokens
    for (size_t bs = (size_t)blockIdx.x * tpb; bs < tokens; bs += (size_t)gridDim.x * tpb) {
        rope_fq_token(q, bs, tpb, tokens);
        rope_fq_token(k, bs, tpb, tokens);
    }
}

// The rotary kernel and the attention kernel's value-code pass (qt_value_codes.h) in ONE launch: both read slices of the q / k / v
// projections' product, neither depends on the other, and each alone is too small to fill the chip (11 us + 8 us back to back).
// Workgroups [0, rope_blocks) walk the tokens, the rest take one (batch * head, block of 128 keys) each.
struct ValueArgs {
    const uint16_t *v;
    uint8_t *vt8;
    long sb, sh, sk, Sk;
    int H, nkb;
    qt_format fmt;
};

template <bool VE5M2, int VD, bool CODES = false>
__global__ __launch_bounds__(256) void rope_fq_value_kernel(RopeFqArgs q, RopeFqArgs k, ValueArgs v, unsigned rope_blocks, unsigned tpb) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[VD * 128];
    if (blockIdx.x < rope_blocks) {
        const size_t tokens = (size_t)q.r.B * (size_t)q.r.S;
        for (size_t bs = (size_t)blockIdx.x * tpb; bs < tokens; bs += (size_t)rope_blocks * tpb) {
            rope_fq_token<CODES>(q, bs, tpb, tokens);
            rope_fq_token<CODES>(k, bs, tpb, tokens);
        }
    } else {
        const unsigned vb = blockIdx.x - rope_blocks;
        value_codes_block<VE5M2, VD>(tile, v.v, v.vt8, (int)(vb % (unsigned)v.nkb), (long)(vb / (unsigned)v.nkb), v.H, v.Sk, v.sb, v.sh, v.sk, v.fmt);
    }
}

// The same pairing for TABLE formats: the rotary kernel in its row form (rope_fq_kernel with a map) and the table-format attention
// core's value pass (csrc/qt_value_rows.h, 64 keys per workgroup here so that six workgroups share a CU's LDS) in one launch; one
// format -- one row table -- for q, k and v (checked on the host).
struct ValueRowsArgs {
    const uint16_t *v;
    uint16_t *vt;
    long sb, sh, sk, Sk;
    int H, nkb;
};
// A third independent job of the same format for the same launch: the fake-quant pass of a WEIGHT -- the output projection's, where that
// Linear runs as weight pass + library GEMM.  HBM-bound (26 M elements in and out for a 5120 x 5120 weight) next to two latency-bound
// jobs: its workgroups come first in the grid so that its stream is under way while the rotary workgroups wait on theirs.
struct WeightPassArgs {
    const uint4 *w;
    uint4 *wq;
    size_t nvec;
    unsigned blocks;
};

__global__ __launch_bounds__(256) void rope_map_value_kernel(RopeFqArgs q, RopeFqArgs k, ValueRowsArgs v, WeightPassArgs wp, unsigned rope_blocks,
                                                             unsigned tpb) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[value_rows_tile_elems<64>()];
    __shared__ uint4 s_rows[512];
    {
        const uint4 *gr = (const uint4 *)(q.map + QT_MAP_ENTRIES);
        const int nrows = (q.fmt.p1 & 2) ? 512 : 256;
        for (int i = threadIdx.x; i < nrows; i += 256) s_rows[i] = gr[i];
        __syncthreads();
        q.rows_lds = k.rows_lds = (const uint16_t *)s_rows;
    }
    if (blockIdx.x < wp.blocks) {
        const Rounder<kFmtRows> rnd{q.fmt, (const uint16_t *)s_rows, q.map};
        // two workgroups per CU with four loads in flight per lane: the bytes in flight of the stand-alone pass (eight workgroups of one
        // load), in a third of the workgroup slots -- the rest are the rotary and value jobs', which need them to cover their latency
        constexpr int kU = 4;
        const size_t stride = (size_t)wp.blocks * 256;
        for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < wp.nvec; i0 += stride * kU) {
            uint4 a[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (i0 + u * stride < wp.nvec) a[u] = wp.w[i0 + u * stride];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (i0 + u * stride < wp.nvec) {
                    const uint32_t in[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
                    uint32_t o[4];
                    fq_rows_words<4, false>(in, o, rnd);
                    wp.wq[i0 + u * stride] = uint4{o[0], o[1], o[2], o[3]};
                }
            }
        }
        return;
    }
    const unsigned bid = blockIdx.x - wp.blocks;
    if (bid < rope_blocks) {
        const size_t tokens = (size_t)q.r.B * (size_t)q.r.S;
        for (size_t bs = (size_t)bid * tpb; bs < tokens; bs += (size_t)rope_blocks * tpb) {
            rope_fq_token(q, bs, tpb, tokens);
            rope_fq_token(k, bs, tpb, tokens);
        }
    } else {
        const unsigned vb = bid - rope_blocks;
        const Rounder<kFmtRows> rnd{q.fmt, (const uint16_t *)s_rows, q.map};
        value_t_rows_block<64>(tile, rnd, v.v, v.vt, v.H, v.Sk, v.sb, v.sh, v.sk, (long)(vb / (unsigned)v.nkb), (int)(vb % (unsigned)v.nkb), (int)threadIdx.x);
    }
}

// ---- BERT-style blocks: LayerNorm(dense(x) + residual) and GELU between the fake-quantized GEMMs ----------------
//   add + LayerNorm   s = bf16(x + res);  y = bf16(w * (rstd * (s32 - mean)) + b),  mean / biased variance over the row
//                     in fp32 (two passes over registers), rstd = rsqrt(var + eps)   [torch: add, native_layer_norm]
//   GELU (erf form)   y = bf16((x32 * 0.5) * (1 + erf(x32 * sqrt(1/2))))                    [torch: GeluCUDAKernelImpl]
// With FQ the consumer's stateless E4M3 / E5M2 fake-quantizer is applied on the way out.  A LayerNorm result also feeds
// the next residual connection unquantized, so that kernel writes three tensors: y, fq(y) as bf16 and its FP8 code.
struct LnArgs {
    const uint4 *x, *res, *w, *b;
    uint4 *y, *yq;
    uint2 *y8;
    long rows;
    int nvec;
    float inv_cols, eps;
    qt_format fmt;
    int n_extra;             // further consumers of the result (k, v beside q): each one's own fake-quantizer is evaluated too,
    NormExtra extra;         // its FP8 codes written to its own buffer (as rmsnorm_kernel's EXTRA)
};

// G threads per row (64: one wavefront, rows up to 1024 elements keep <= 2 vectors per lane; 256: the whole workgroup)
// NV: vectors per lane the loops are unrolled for (as rmsnorm_kernel's)
template <int G, int FQ, bool ADD, int NV = kNormMaxVec>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
    constexpr int RPB = 256 / G;
    __shared__ float s_part[2][4];
    const int sub = threadIdx.x / G, lane = threadIdx.x % G;
    const long row = (long)blockIdx.x * RPB + sub;
    if (G == 64 && row >= a.rows) return;                    // whole wavefronts leave; no workgroup barrier on this path
    const size_t base = (size_t)row * (size_t)a.nvec;
    uint4 v[NV];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * G;
        if (c < a.nvec) {
            v[i] = a.x[base + c];
            if constexpr (ADD) {
                const uint4 r = a.res[base + c];
                v[i].x = pack_bf16x2(bf_lo(v[i].x) + bf_lo(r.x), bf_hi(v[i].x) + bf_hi(r.x));
                v[i].y = pack_bf16x2(bf_lo(v[i].y) + bf_lo(r.y), bf_hi(v[i].y) + bf_hi(r.y));
                v[i].z = pack_bf16x2(bf_lo(v[i].z) + bf_lo(r.z), bf_hi(v[i].z) + bf_hi(r.z));
                v[i].w = pack_bf16x2(bf_lo(v[i].w) + bf_lo(r.w), bf_hi(v[i].w) + bf_hi(r.w));
            }
            sum += (bf_lo(v[i].x) + bf_hi(v[i].x)) + (bf_lo(v[i].y) + bf_hi(v[i].y)) + (bf_lo(v[i].z) + bf_hi(v[i].z)) +
                   (bf_lo(v[i].w) + bf_hi(v[i].w));
        }
    }
    auto reduce = [&](float t, int slot) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        if constexpr (G == 256) {
            if ((threadIdx.x & 63) == 0) s_part[slot][threadIdx.x >> 6] = t;
            __syncthreads();
            t = (s_part[slot][0] + s_part[slot][1]) + (s_part[slot][2] + s_part[slot][3]);
        }
        return t;
    };
    const float mean = reduce(sum, 0) * a.inv_cols;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * G;
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
    const float rstd = rsqrtf(reduce(sq, 1) * a.inv_cols + a.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * G;
        if (c < a.nvec) {
            const uint4 ww = a.w[c], bb = a.b[c];
            const uint32_t q[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, g[4] = {ww.x, ww.y, ww.z, ww.w}, h[4] = {bb.x, bb.y, bb.z, bb.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(bf_lo(g[j]) * (rstd * (bf_lo(q[j]) - mean)) + bf_lo(h[j]),
                                   bf_hi(g[j]) * (rstd * (bf_hi(q[j]) - mean)) + bf_hi(h[j]));
            a.y[base + c] = uint4{o[0], o[1], o[2], o[3]};
            if constexpr (FQ != 0) {
                for (int e = 0; e < a.n_extra; ++e) {
                    uint32_t t[4] = {o[0], o[1], o[2], o[3]};
                    a.extra.y8[e][base + c] = a.extra.e5m2[e] ? fq8_hw_vec8<true>(t, a.extra.fmt[e]) : fq8_hw_vec8<false>(t, a.extra.fmt[e]);
                }
                const uint2 codes = fq8_hw_vec8<FQ == 2>(o, a.fmt);
                if (a.yq) a.yq[base + c] = uint4{o[0], o[1], o[2], o[3]};         // NULL: codes only (they decode to exactly these values)
                a.y8[base + c] = codes;
            }
        }
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return (x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int FQ>
__global__ __launch_bounds__(256) void gelu_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, uint2 *__restrict__ y8,
                                                   size_t nvec, qt_format fmt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const uint4 a = x[i];
        const uint32_t p[4] = {a.x, a.y, a.z, a.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(gelu_erf(bf_lo(p[j])), gelu_erf(bf_hi(p[j])));
        if constexpr (FQ != 0) y8[i] = fq8_hw_vec8<FQ == 2>(o, fmt);
        if (FQ == 0 || y) y[i] = uint4{o[0], o[1], o[2], o[3]};                  // NULL with FQ: codes only
    }
}

int fp8_code_of(const qt_format *f) {                       // 1 E4M3, 2 E5M2, 0 neither
    if (!f || f->kind != QT_FMT_FP_SAT) return 0;
    if (f->p0 == 2 && f->p1 == -14 && f->fhi == 57344.0f) return 2;
    if (f->p0 == 3 && f->p1 == -6 && f->fhi == 448.0f) return 1;
    return 0;
}

template <int G, bool ADD, int NV>
void launch_layernorm_nv(const LnArgs &a, int fq, unsigned blocks, hipStream_t st) {
    if (fq == 2) layernorm_kernel<G, 2, ADD, NV><<<blocks, 256, 0, st>>>(a);
    else if (fq == 1) layernorm_kernel<G, 1, ADD, NV><<<blocks, 256, 0, st>>>(a);
    else layernorm_kernel<G, 0, ADD, NV><<<blocks, 256, 0, st>>>(a);
}

template <int G, bool ADD>
void launch_layernorm(const LnArgs &a, int fq, unsigned blocks, hipStream_t st) {
    if (a.nvec <= 2 * G) launch_layernorm_nv<G, ADD, 2>(a, fq, blocks, st);
    else if (a.nvec <= 4 * G) launch_layernorm_nv<G, ADD, 4>(a, fq, blocks, st);
    else launch_layernorm_nv<G, ADD, 8>(a, fq, blocks, st);
}

int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int FQ, bool ADD, int EXTRA>
void launch_norm_consumers(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, uint16_t *sum, uint16_t *y, uint8_t *y8,
                                  long rows, int nvec, float inv, float eps, const qt_format &f, const NormExtra &ex, hipStream_t st) {
    launch_rms<FQ, ADD, EXTRA>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps,
                                                                           (uint2 *)y8, f, (const uint4 *)residual, (uint4 *)sum, ex);
}

// ---- causal-LM loss of a window: logits -> mean NLL -------------------------------------------------------------------------
// What the caller of the hot path computes from the lm head's bf16 logits (examples/language_modeling/wikitext.py:146-158 through
// transformers' ForCausalLMLoss): logits.float(); label of position s = labels[s + 1] (the last position has none); cross entropy in
// fp32 with ignore_index, mean over the scored positions.  torch runs that as a bf16 -> fp32 copy of the [S, V] logits, a softmax
// kernel and a reduction (60 + 35 + 12 us at S 1024, V 32000); here one workgroup per position reads its row once (online maximum /
// sum of exponentials per lane, combined across the workgroup), and a second one-workgroup kernel sums the rows in a fixed order --
// deterministic, no atomics.
__global__ __launch_bounds__(256) void nll_rows_kernel(const uint16_t *__restrict__ logits, const long long *__restrict__ labels, long S, long V,
                                                       long row_stride, long long ignore_index, float *__restrict__ row_loss) {
    __shared__ float s_m[4], s_s[4];
    const long row = blockIdx.x, s_pos = row % S;
    const long long lab = (s_pos + 1 < S) ? labels[row + 1] : ignore_index;
    if (lab == ignore_index || lab < 0 || lab >= V) {                    // (uniform per workgroup)
        if (threadIdx.x == 0) row_loss[row] = -1.0f;                     // marks "not scored": a loss is never negative
        return;
    }
    const uint16_t *x = logits + row * row_stride;
    float m = -INFINITY, sum = 0.0f;
    const long nvec = V / 8;
    for (long i = threadIdx.x; i < nvec; i += 256) {
        const uint4 v = *(const uint4 *)(x + i * 8);
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = bf_lo(q[j]);
            f[2 * j + 1] = bf_hi(q[j]);
        }
        float vm = f[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) vm = fmaxf(vm, f[j]);
        if (vm > m) {
            sum *= __expf(m - vm);
            m = vm;
        }
        if (m != -INFINITY) {                                            // (eight -inf logits in front of everything else add nothing)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += __expf(f[j] - m);
        }
    }
    for (long c = nvec * 8 + threadIdx.x; c < V; c += 256) {             // a vocabulary that is not a multiple of 8
        const float f = qt_bf2f(x[c]);
        if (f > m) {
            sum *= __expf(m - f);
            m = f;
        }
        if (m != -INFINITY) sum += __expf(f - m);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float om = __shfl_xor(m, off, 64), os = __shfl_xor(sum, off, 64);
        const float nm = fmaxf(m, om);
        sum = (m == -INFINITY ? 0.0f : sum * __expf(m - nm)) + (om == -INFINITY ? 0.0f : os * __expf(om - nm));
        m = nm;
    }
    if ((threadIdx.x & 63) == 0) {
        s_m[threadIdx.x >> 6] = m;
        s_s[threadIdx.x >> 6] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])), Ssum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) Ssum += s_m[i] == -INFINITY ? 0.0f : s_s[i] * __expf(s_m[i] - M);
        row_loss[row] = (M + logf(Ssum)) - qt_bf2f(x[lab]);             // -log softmax(x)[label]
    }
}

__global__ __launch_bounds__(256) void nll_mean_kernel(const float *__restrict__ row_loss, long rows, float *__restrict__ out) {
    __shared__ float s_sum[256];
    __shared__ int s_cnt[256];
    float acc = 0.0f;
    int cnt = 0;
    for (long r = threadIdx.x; r < rows; r += 256) {
        const float v = row_loss[r];
        if (v >= 0.0f || v != v) {                                        // scored (a NaN loss stays a NaN mean)
            acc += v;
            ++cnt;
        }
    }
    s_sum[threadIdx.x] = acc;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s_sum[0] / (float)s_cnt[0];             // 0 / 0 = NaN when nothing is scored, as torch
}

}  // namespace

// ---- column sums of a bf16 [rows, cols] matrix: the bias gradient of a Linear's backward ---------------------------------------
// autograd's linear backward computes grad_bias = grad_output.sum(0) (here: of the E5M2 fake-quantized gradient the backward-pre
// hook produced, quantize.py:116-179) with torch's generic reduction -- 12.6 us per [2048, 768] inside the replayed training step.
// A workgroup owns 32 columns (64 bytes: four 16-byte vectors); its 256 threads = 64 row lanes x 4 vectors walk the rows, fp32
// sums per lane, then a fixed-order tree over the 64 row lanes in LDS: deterministic, one rounding to bf16 at the end.
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ out, long rows, long cols) {
    const int t = threadIdx.x, v = t & 3, rl = t >> 2;                  // vector within the 64-byte column group, row lane
    const long c0 = (long)blockIdx.x * 32 + v * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < cols) {
        // eight rows in flight per lane (one load per iteration is latency-bound: 12 us for 2048 rows), added in row order
        for (long r0 = rl; r0 < rows; r0 += 64 * 8) {
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long r = r0 + (long)u * 64;
                q[u] = r < rows ? *(const uint4 *)(x + r * cols + c0) : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[0] += qt_u2f(q[u].x << 16); acc[1] += qt_u2f(q[u].x & 0xFFFF0000u);
                acc[2] += qt_u2f(q[u].y << 16); acc[3] += qt_u2f(q[u].y & 0xFFFF0000u);
                acc[4] += qt_u2f(q[u].z << 16); acc[5] += qt_u2f(q[u].z & 0xFFFF0000u);
                acc[6] += qt_u2f(q[u].w << 16); acc[7] += qt_u2f(q[u].w & 0xFFFF0000u);
            }
        }
    }
    __shared__ float part[64][33];
#pragma unroll
    for (int e = 0; e < 8; ++e) part[rl][v * 8 + e] = acc[e];
    __syncthreads();
    for (int half = 32; half >= 1; half >>= 1) {
        for (int i = t; i < half * 32; i += 256) {
            const int r = i / 32, c = i % 32;
            part[r][c] += part[r + half][c];
        }
        __syncthreads();
    }
    if (t < 32 && (long)blockIdx.x * 32 + t < cols) out[(long)blockIdx.x * 32 + t] = (uint16_t)(pack_bf16x2(part[0][t], 0.0f) & 0xFFFFu);
}

extern "C" {

int qt_rmsnorm_bf16(const uint16_t *x, const uint16_t *weight, uint16_t *y, long rows, long cols, float eps, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !weight || !y || rows < 0 || cols < 0) return QT_ERR_BAD_ARG;
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 || (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15u))
        return QT_ERR_UNALIGNED;
    launch_rms<0, false, 0>((unsigned)rows, (int)(cols / 8), (hipStream_t)stream, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y,
                                                                                (int)(cols / 8), 1.0f / (float)cols, eps, nullptr,
                                                                                qt_format{});
    return launch_status();
}

int qt_rmsnorm_fq8_bf16(const uint16_t *x, const uint16_t *weight, uint16_t *y, uint8_t *y8, long rows, long cols, float eps,
                        const qt_format *fmt, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !weight || !y8 || !fmt || rows < 0 || cols < 0 || fmt->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;       // y NULL: codes only
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 || (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15u) ||
        ((uintptr_t)y8 & 7u))
        return QT_ERR_UNALIGNED;
    if (e5m2)
        launch_rms<2, false, 0>((unsigned)rows, (int)(cols / 8), (hipStream_t)stream, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y,
                                                                                    (int)(cols / 8), 1.0f / (float)cols, eps, (uint2 *)y8, *fmt);
    else
        launch_rms<1, false, 0>((unsigned)rows, (int)(cols / 8), (hipStream_t)stream, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y,
                                                                                    (int)(cols / 8), 1.0f / (float)cols, eps, (uint2 *)y8, *fmt);
    return launch_status();
}

int qt_add_rmsnorm_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, uint16_t *sum, uint16_t *y, uint8_t *y8,
                        long rows, long cols, float eps, const qt_format *fmt, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !residual || !weight || !sum || (!y && !y8) || rows < 0 || cols < 0) return QT_ERR_BAD_ARG;              // y NULL with y8: codes only
    const int fq = y8 ? fp8_code_of(fmt) : 0;
    if (y8 && !fq) return QT_ERR_BAD_ARG;
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 ||
        (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)sum | (uintptr_t)y) & 15u) || ((uintptr_t)y8 & 7u))
        return QT_ERR_UNALIGNED;
    hipStream_t st = (hipStream_t)stream;
    const int nvec = (int)(cols / 8);
    const float inv = 1.0f / (float)cols;
    const qt_format f = fq ? *fmt : qt_format{};
    if (fq == 2) launch_rms<2, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, (uint2 *)y8, f, (const uint4 *)residual, (uint4 *)sum);
    else if (fq == 1) launch_rms<1, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, (uint2 *)y8, f, (const uint4 *)residual, (uint4 *)sum);
    else launch_rms<0, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, nullptr, f, (const uint4 *)residual, (uint4 *)sum);
    return launch_status();
}

int qt_add_rmsnorm_sumfq_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, uint16_t *sum, uint16_t *y, uint8_t *y8,
                              long rows, long cols, float eps, const qt_format *fmt, const qt_format *sum_fmt, void *stream) {
    if (!sum_fmt) return qt_add_rmsnorm_bf16(x, residual, weight, sum, y, y8, rows, cols, eps, fmt, stream);
    if (rows * cols == 0) return QT_OK;
    if (!x || !residual || !weight || !sum || !y || rows < 0 || cols < 0) return QT_ERR_BAD_ARG;
    const int fq = y8 ? fp8_code_of(fmt) : 0, sfq = fp8_code_of(sum_fmt);
    if ((y8 && !fq) || !sfq) return QT_ERR_BAD_ARG;
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 ||
        (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)sum | (uintptr_t)y) & 15u) || ((uintptr_t)y8 & 7u))
        return QT_ERR_UNALIGNED;
    hipStream_t st = (hipStream_t)stream;
    const int nvec = (int)(cols / 8);
    const float inv = 1.0f / (float)cols;
    const qt_format f = fq ? *fmt : qt_format{};
    if (fq == 2) launch_rms<2, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, (uint2 *)y8, f, (const uint4 *)residual, (uint4 *)sum, NormExtra{}, sfq, *sum_fmt);
    else if (fq == 1) launch_rms<1, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, (uint2 *)y8, f, (const uint4 *)residual, (uint4 *)sum, NormExtra{}, sfq, *sum_fmt);
    else launch_rms<0, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, nullptr, f, (const uint4 *)residual, (uint4 *)sum, NormExtra{}, sfq, *sum_fmt);
    return launch_status();
}

int qt_rmsnorm_consumers_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, uint16_t *sum, uint16_t *y, long rows,
                              long cols, float eps, int consumers, uint8_t *const *y8, const qt_format *const *fmt, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !weight || !y || rows < 0 || cols < 0 || consumers < 2 || consumers > 3 || !y8 || !fmt || (residual != nullptr) != (sum != nullptr))
        return QT_ERR_BAD_ARG;
    int code[3] = {0, 0, 0};
    for (int i = 0; i < consumers; ++i) {
        if (!y8[i] || !fmt[i] || ((uintptr_t)y8[i] & 7u)) return QT_ERR_BAD_ARG;
        code[i] = fp8_code_of(fmt[i]);
        if (!code[i]) return QT_ERR_BAD_ARG;
    }
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 ||
        (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)sum | (uintptr_t)y) & 15u))
        return QT_ERR_UNALIGNED;
    NormExtra ex{};
    for (int i = 1; i < consumers; ++i) {
        ex.y8[i - 1] = (uint2 *)y8[i];
        ex.fmt[i - 1] = *fmt[i];
        ex.e5m2[i - 1] = code[i] == 2;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nvec = (int)(cols / 8);
    const float inv = 1.0f / (float)cols;
    const int key = (code[0] == 2 ? 4 : 0) | (residual ? 2 : 0) | (consumers == 3 ? 1 : 0);
    switch (key) {
        case 0: launch_norm_consumers<1, false, 1>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        case 1: launch_norm_consumers<1, false, 2>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        case 2: launch_norm_consumers<1, true, 1>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        case 3: launch_norm_consumers<1, true, 2>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        case 4: launch_norm_consumers<2, false, 1>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        case 5: launch_norm_consumers<2, false, 2>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        case 6: launch_norm_consumers<2, true, 1>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
        default: launch_norm_consumers<2, true, 2>(x, residual, weight, sum, y, y8[0], rows, nvec, inv, eps, *fmt[0], ex, st); break;
    }
    return launch_status();
}

static int launch_layernorm_any(const LnArgs &a, int fq, bool with_residual, void *stream);

int qt_layernorm_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, const uint16_t *bias, uint16_t *y,
                      uint16_t *yq, uint8_t *y8, long rows, long cols, float eps, const qt_format *fmt, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !weight || !bias || !y || rows < 0 || cols < 0 || (yq && !y8)) return QT_ERR_BAD_ARG;
    const int fq = y8 ? fp8_code_of(fmt) : 0;
    if (y8 && !fq) return QT_ERR_BAD_ARG;
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 ||
        (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)yq) & 15u) || ((uintptr_t)y8 & 7u))
        return QT_ERR_UNALIGNED;
    LnArgs a{(const uint4 *)x, (const uint4 *)residual, (const uint4 *)weight, (const uint4 *)bias, (uint4 *)y, (uint4 *)yq, (uint2 *)y8,
             rows, (int)(cols / 8), 1.0f / (float)cols, eps, fmt && fq ? *fmt : qt_format{}, 0, NormExtra{}};
    return launch_layernorm_any(a, fq, residual != nullptr, stream);
}

int qt_layernorm_consumers_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, const uint16_t *bias, uint16_t *y,
                                uint16_t *yq, long rows, long cols, float eps, int consumers, uint8_t *const *y8,
                                const qt_format *const *fmt, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !weight || !bias || !y || rows < 0 || cols < 0 || consumers < 2 || consumers > 3 || !y8 || !fmt) return QT_ERR_BAD_ARG;
    int code[3] = {0, 0, 0};
    for (int i = 0; i < consumers; ++i) {
        if (!y8[i] || !fmt[i] || ((uintptr_t)y8[i] & 7u)) return QT_ERR_BAD_ARG;
        code[i] = fp8_code_of(fmt[i]);
        if (!code[i]) return QT_ERR_BAD_ARG;
    }
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 ||
        (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)yq) & 15u))
        return QT_ERR_UNALIGNED;
    LnArgs a{(const uint4 *)x, (const uint4 *)residual, (const uint4 *)weight, (const uint4 *)bias, (uint4 *)y, (uint4 *)yq, (uint2 *)y8[0],
             rows, (int)(cols / 8), 1.0f / (float)cols, eps, *fmt[0], consumers - 1, NormExtra{}};
    for (int i = 1; i < consumers; ++i) {
        a.extra.y8[i - 1] = (uint2 *)y8[i];
        a.extra.fmt[i - 1] = *fmt[i];
        a.extra.e5m2[i - 1] = code[i] == 2;
    }
    return launch_layernorm_any(a, code[0], residual != nullptr, stream);
}

static int launch_layernorm_any(const LnArgs &a, int fq, bool with_residual, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const long rows = a.rows;
    const bool residual = with_residual;
    if (a.nvec <= 128) {
        const unsigned blocks = (unsigned)((rows + 3) / 4);
        if (residual) launch_layernorm<64, true>(a, fq, blocks, st);
        else launch_layernorm<64, false>(a, fq, blocks, st);
    } else {
        if (residual) launch_layernorm<256, true>(a, fq, (unsigned)rows, st);
        else launch_layernorm<256, false>(a, fq, (unsigned)rows, st);
    }
    return launch_status();
}

int qt_gelu_bf16(const uint16_t *x, uint16_t *y, uint8_t *y8, size_t n, const qt_format *fmt, void *stream) {
    if (n == 0) return QT_OK;
    if (!x || (!y && !y8)) return QT_ERR_BAD_ARG;
    const int fq = y8 ? fp8_code_of(fmt) : 0;
    if (y8 && !fq) return QT_ERR_BAD_ARG;
    if ((n & 7) || (((uintptr_t)x | (uintptr_t)y) & 15u) || ((uintptr_t)y8 & 7u)) return QT_ERR_UNALIGNED;
    const size_t nvec = n / 8;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = (hipStream_t)stream;
    if (fq == 2) gelu_kernel<2><<<(unsigned)blocks, 256, 0, st>>>((const uint4 *)x, (uint4 *)y, (uint2 *)y8, nvec, *fmt);
    else if (fq == 1) gelu_kernel<1><<<(unsigned)blocks, 256, 0, st>>>((const uint4 *)x, (uint4 *)y, (uint2 *)y8, nvec, *fmt);
    else gelu_kernel<0><<<(unsigned)blocks, 256, 0, st>>>((const uint4 *)x, (uint4 *)y, nullptr, nvec, qt_format{});
    return launch_status();
}

int qt_silu_mul_bf16(const uint16_t *gate, const uint16_t *up, uint16_t *y, size_t rows, size_t cols, size_t gate_row_stride,
                     size_t up_row_stride, void *stream) {
    const size_t n = rows * cols;
    if (n == 0) return QT_OK;
    if (!gate || !up || !y || gate_row_stride < cols || up_row_stride < cols) return QT_ERR_BAD_ARG;
    if ((cols & 7) || ((gate_row_stride | up_row_stride) & 7) || (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)y) & 15u))
        return QT_ERR_UNALIGNED;
    const size_t nvec = n / 8;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    silu_mul_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint4 *)gate, (const uint4 *)up, (uint4 *)y, nvec,
                                                                       cols / 8, gate_row_stride / 8, up_row_stride / 8);
    return launch_status();
}

int qt_silu_mul_fq8_bf16(const uint16_t *gate, const uint16_t *up, uint16_t *y, uint8_t *y8, size_t rows, size_t cols,
                         size_t gate_row_stride, size_t up_row_stride, const qt_format *fmt, void *stream) {
    const size_t n = rows * cols;
    if (n == 0) return QT_OK;
    if (!gate || !up || !y || !y8 || !fmt || fmt->kind != QT_FMT_FP_SAT || gate_row_stride < cols || up_row_stride < cols)
        return QT_ERR_BAD_ARG;
    const bool e5m2 = fmt->p0 == 2 && fmt->p1 == -14 && fmt->fhi == 57344.0f;
    const bool e4m3 = fmt->p0 == 3 && fmt->p1 == -6 && fmt->fhi == 448.0f;
    if (!e5m2 && !e4m3) return QT_ERR_BAD_ARG;
    if ((cols & 7) || ((gate_row_stride | up_row_stride) & 7) || (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)y) & 15u) ||
        ((uintptr_t)y8 & 7u))
        return QT_ERR_UNALIGNED;
    const size_t nvec = n / 8;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (e5m2)
        silu_mul_fq8_kernel<true><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint4 *)gate, (const uint4 *)up, (uint4 *)y, (uint2 *)y8, nvec, *fmt,
                                                                                     cols / 8, gate_row_stride / 8, up_row_stride / 8);
    else
        silu_mul_fq8_kernel<false><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint4 *)gate, (const uint4 *)up, (uint4 *)y, (uint2 *)y8, nvec, *fmt,
                                                                                      cols / 8, gate_row_stride / 8, up_row_stride / 8);
    return launch_status();
}

int qt_rope_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out,
                 uint16_t *k_out, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride, void *stream) {
    if (B * S * D == 0) return QT_OK;
    if (!q || !k || !cos || !sin || !q_out || !k_out || B < 0 || S < 0 || Hq < 0 || Hk < 0) return QT_ERR_BAD_ARG;
    if (D % 16 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos | (uintptr_t)sin | (uintptr_t)q_out | (uintptr_t)k_out) & 15u))
        return QT_ERR_UNALIGNED;
    if (q_row_stride < Hq * D || k_row_stride < Hk * D || (q_row_stride | k_row_stride) % 8) return QT_ERR_BAD_ARG;
    RopeArgs aq{q, q_out, cos, sin, B, S, Hq, D, (size_t)(B * S * Hq * D / 8), q_row_stride / 8};
    RopeArgs ak{k, k_out, cos, sin, B, S, Hk, D, (size_t)(B * S * Hk * D / 8), k_row_stride / 8};
    size_t blocks = (aq.nvec + ak.nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    rope_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(aq, ak);
    return launch_status();
}

static int rope_fq_launch(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out,
                          uint8_t *q_out8, uint8_t *k_out8, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride,
                          const qt_format *fmt_q, const qt_format *fmt_k, bool need_values, const uint16_t *v, uint8_t *vt8, long v_sb, long v_sh,
                          long v_sk, const qt_format *fmt_v, void *stream, const qt_format *inner_q = nullptr, const qt_format *inner_k = nullptr) {
    if (B * S * D == 0) return QT_OK;
    if ((inner_q && inner_q->kind != QT_FMT_FP_SAT) || (inner_k && inner_k->kind != QT_FMT_FP_SAT) || ((inner_q || inner_k) && !cos)) return QT_ERR_BAD_ARG;
    if (!q || !k || (need_values && !cos) || (cos == nullptr) != (sin == nullptr) || !fmt_q || !fmt_k || B < 0 || S < 0 || Hq < 0 || Hk < 0)
        return QT_ERR_BAD_ARG;
    if (need_values ? (!q_out || !k_out) : (!q_out8 || !k_out8 || (q_out == nullptr) != (k_out == nullptr))) return QT_ERR_BAD_ARG;
    if (fmt_q->kind != QT_FMT_FP_SAT || fmt_k->kind != QT_FMT_FP_SAT) return QT_ERR_BAD_ARG;
    if (D % 16 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos | (uintptr_t)sin | (uintptr_t)q_out | (uintptr_t)k_out) & 15u))
        return QT_ERR_UNALIGNED;
    if (q_row_stride < Hq * D || k_row_stride < Hk * D || (q_row_stride | k_row_stride) % 8) return QT_ERR_BAD_ARG;
    if (((uintptr_t)q_out8 | (uintptr_t)k_out8) & 7u) return QT_ERR_UNALIGNED;
    auto is_e5m2 = [](const qt_format *f) { return f->p0 == 2 && f->p1 == -14 && f->fhi == 57344.0f; };
    auto is_e4m3 = [](const qt_format *f) { return f->p0 == 3 && f->p1 == -6 && f->fhi == 448.0f; };
    if ((q_out8 && !is_e5m2(fmt_q) && !is_e4m3(fmt_q)) || (k_out8 && !is_e5m2(fmt_k) && !is_e4m3(fmt_k))) return QT_ERR_BAD_ARG;
    RopeFqArgs aq{{q, q_out, cos, sin, B, S, Hq, D, (size_t)(B * S * Hq * D / 8), q_row_stride / 8}, *fmt_q, q_out8, is_e5m2(fmt_q) ? 1 : 0,
                  inner_q ? 1 : 0, inner_q ? *inner_q : qt_format{}, nullptr, nullptr};
    RopeFqArgs ak{{k, k_out, cos, sin, B, S, Hk, D, (size_t)(B * S * Hk * D / 8), k_row_stride / 8}, *fmt_k, k_out8, is_e5m2(fmt_k) ? 1 : 0,
                  inner_k ? 1 : 0, inner_k ? *inner_k : qt_format{}, nullptr, nullptr};
    if (Hq * D / 8 > 0xFFFFFFFFl || Hk * D / 8 > 0xFFFFFFFFl || B > 0x7FFFFFFFl || S > 0x7FFFFFFFl) return QT_ERR_BAD_ARG;
    const long nv_max = (Hq > Hk ? Hq : Hk) * D / 8;                      // vectors of a token: a workgroup takes 256 / that many tokens
    const unsigned tpb = nv_max >= 256 || nv_max < 1 ? 1u : (unsigned)(256 / nv_max);
    size_t blocks = ((size_t)B * (size_t)S + tpb - 1) / tpb;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = (hipStream_t)stream;
    if (!v) {
        rope_fq_kernel<<<(unsigned)blocks, 256, 0, st>>>(aq, ak, tpb);
        return launch_status();
    }
    // the value job: v is [B][Hk][S][D] by strides, D contiguous; vt8 [B][Hk][D][S]
    if (!vt8 || !fmt_v || fmt_v->kind != QT_FMT_FP_SAT || (!is_e5m2(fmt_v) && !is_e4m3(fmt_v)) || (D != 64 && D != 128) || S % 128 != 0 ||
        B * Hk > 65535)
        return QT_ERR_BAD_ARG;
    if ((((uintptr_t)v | (uintptr_t)vt8) & 15u) || ((v_sb | v_sh | v_sk) & 7)) return QT_ERR_UNALIGNED;
    ValueArgs av{v, vt8, v_sb, v_sh, v_sk, S, (int)Hk, (int)(S / 128), *fmt_v};
    const unsigned total = (unsigned)blocks + (unsigned)(B * Hk * (S / 128));
    const bool ve5 = is_e5m2(fmt_v);
    if (D == 128) {
        const bool codes = aq.y8 && ak.y8 && !aq.inner && !ak.inner && !aq.map && !ak.map;       // (one mode for the whole launch)
        if (codes && !ve5) rope_fq_value_kernel<false, 128, true><<<total, 256, 0, st>>>(aq, ak, av, (unsigned)blocks, tpb);
        else if (ve5) rope_fq_value_kernel<true, 128><<<total, 256, 0, st>>>(aq, ak, av, (unsigned)blocks, tpb);
        else rope_fq_value_kernel<false, 128><<<total, 256, 0, st>>>(aq, ak, av, (unsigned)blocks, tpb);
    } else {
        const bool codes = aq.y8 && ak.y8 && !aq.inner && !ak.inner && !aq.map && !ak.map;
        if (codes && !ve5) rope_fq_value_kernel<false, 64, true><<<total, 256, 0, st>>>(aq, ak, av, (unsigned)blocks, tpb);
        else if (ve5) rope_fq_value_kernel<true, 64><<<total, 256, 0, st>>>(aq, ak, av, (unsigned)blocks, tpb);
        else rope_fq_value_kernel<false, 64><<<total, 256, 0, st>>>(aq, ak, av, (unsigned)blocks, tpb);
    }
    return launch_status();
}

int qt_rope_fq_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out,
                    uint16_t *k_out, uint8_t *q_out8, uint8_t *k_out8, long B, long S, long Hq, long Hk, long D,
                    long q_row_stride, long k_row_stride, const qt_format *fmt_q, const qt_format *fmt_k, void *stream) {
    return rope_fq_launch(q, k, cos, sin, q_out, k_out, q_out8, k_out8, B, S, Hq, Hk, D, q_row_stride, k_row_stride, fmt_q, fmt_k, true, nullptr,
                          nullptr, 0, 0, 0, nullptr, stream);
}

int qt_rope_fq_value(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out,
                     uint8_t *q_out8, uint8_t *k_out8, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride,
                     const qt_format *fmt_q, const qt_format *fmt_k, const uint16_t *v, uint8_t *vt8, long v_stride_b, long v_stride_h,
                     long v_stride_k, const qt_format *fmt_v, void *stream) {
    return rope_fq_launch(q, k, cos, sin, q_out, k_out, q_out8, k_out8, B, S, Hq, Hk, D, q_row_stride, k_row_stride, fmt_q, fmt_k, false, v, vt8,
                          v_stride_b, v_stride_h, v_stride_k, fmt_v, stream);
}

static bool map_format_ok(const qt_format *fmt, const uint16_t *map) {
    return fmt && map && fmt->kind == QT_FMT_LUT && (fmt->p1 & 1) && (((uintptr_t)map) & 15u) == 0;
}

int qt_rmsnorm_map_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *weight, uint16_t *sum, uint16_t *y, long rows, long cols,
                        float eps, const qt_format *fmt, const uint16_t *map, int quantize_sum, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!x || !weight || !y || rows < 0 || cols < 0 || (residual != nullptr) != (sum != nullptr) || !map_format_ok(fmt, map)) return QT_ERR_BAD_ARG;
    if (cols % 8 || cols > (long)kNormThreads * kNormMaxVec * 8 ||
        (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)sum | (uintptr_t)y) & 15u))
        return QT_ERR_UNALIGNED;
    hipStream_t st = (hipStream_t)stream;
    const int nvec = (int)(cols / 8);
    const float inv = 1.0f / (float)cols;
    if (residual) launch_rms<3, true, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, nullptr, *fmt, (const uint4 *)residual, (uint4 *)sum, NormExtra{}, quantize_sum ? 3 : 0, qt_format{}, map);
    else launch_rms<3, false, 0>((unsigned)rows, nvec, st, (const uint4 *)x, (const uint4 *)weight, (uint4 *)y, nvec, inv, eps, nullptr, *fmt, nullptr, nullptr, NormExtra{}, 0, qt_format{}, map);
    return launch_status();
}

int qt_silu_mul_map_bf16(const uint16_t *gate, const uint16_t *up, uint16_t *y, size_t rows, size_t cols, size_t gate_row_stride,
                         size_t up_row_stride, const qt_format *fmt, const uint16_t *map, void *stream) {
    if (rows * cols == 0) return QT_OK;
    if (!gate || !up || !y || !map_format_ok(fmt, map)) return QT_ERR_BAD_ARG;
    if (cols % 8 || gate_row_stride % 8 || up_row_stride % 8 || gate_row_stride < cols || up_row_stride < cols ||
        (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)y) & 15u))
        return QT_ERR_UNALIGNED;
    const size_t nvec = rows * cols / 8;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    silu_mul_map_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint4 *)gate, (const uint4 *)up, (uint4 *)y, nvec, *fmt, map, cols / 8,
                                                                          gate_row_stride / 8, up_row_stride / 8);
    return launch_status();
}

int qt_rope_map_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out, long B,
                     long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride, const qt_format *fmt, const uint16_t *map,
                     int inner_q, int inner_k, void *stream) {
    if (B * S * D == 0) return QT_OK;
    if (!q || !k || !cos || !sin || !q_out || !k_out || B < 0 || S < 0 || Hq < 0 || Hk < 0 || !map_format_ok(fmt, map)) return QT_ERR_BAD_ARG;
    if (D % 16 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos | (uintptr_t)sin | (uintptr_t)q_out | (uintptr_t)k_out) & 15u))
        return QT_ERR_UNALIGNED;
    if (q_row_stride < Hq * D || k_row_stride < Hk * D || (q_row_stride | k_row_stride) % 8) return QT_ERR_BAD_ARG;
    if (Hq * D / 8 > 0xFFFFFFFFl || Hk * D / 8 > 0xFFFFFFFFl || B > 0x7FFFFFFFl || S > 0x7FFFFFFFl) return QT_ERR_BAD_ARG;
    RopeFqArgs aq{{q, q_out, cos, sin, B, S, Hq, D, (size_t)(B * S * Hq * D / 8), q_row_stride / 8}, *fmt, nullptr, 0, inner_q ? 1 : 0, qt_format{}, map, nullptr};
    RopeFqArgs ak{{k, k_out, cos, sin, B, S, Hk, D, (size_t)(B * S * Hk * D / 8), k_row_stride / 8}, *fmt, nullptr, 0, inner_k ? 1 : 0, qt_format{}, map, nullptr};
    const long nv_max = (Hq > Hk ? Hq : Hk) * D / 8;
    const unsigned tpb = nv_max >= 256 || nv_max < 1 ? 1u : (unsigned)(256 / nv_max);
    size_t blocks = ((size_t)B * (size_t)S + tpb - 1) / tpb;
    if (blocks > 256 * 64) blocks = 256 * 64;
    rope_fq_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(aq, ak, tpb);
    return launch_status();
}

static int rope_map_value_launch(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out,
                                 long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride, const qt_format *fmt,
                                 const uint16_t *map, int inner_q, int inner_k, const uint16_t *v, uint16_t *vt, long v_stride_b, long v_stride_h,
                                 long v_stride_k, const uint16_t *w, uint16_t *wq, size_t w_elems, void *stream) {
    if (B * S * D == 0) return QT_OK;
    if (!q || !k || !cos || !sin || !q_out || !k_out || !v || !vt || B < 0 || S < 0 || Hq < 0 || Hk < 1 || !map_format_ok(fmt, map)) return QT_ERR_BAD_ARG;
    if (D != kValueRowsD || S % 128 != 0 || B * Hk > 65535 || (w_elems != 0 && (!w || !wq))) return QT_ERR_BAD_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos | (uintptr_t)sin | (uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)v | (uintptr_t)vt) & 15u) ||
        ((v_stride_b | v_stride_h | v_stride_k) & 7) || (w_elems != 0 && ((((uintptr_t)w | (uintptr_t)wq) & 15u) || (w_elems & 7))))
        return QT_ERR_UNALIGNED;
    if (q_row_stride < Hq * D || k_row_stride < Hk * D || (q_row_stride | k_row_stride) % 8) return QT_ERR_BAD_ARG;
    if (Hq * D / 8 > 0xFFFFFFFFl || Hk * D / 8 > 0xFFFFFFFFl || B > 0x7FFFFFFFl || S > 0x7FFFFFFFl) return QT_ERR_BAD_ARG;
    RopeFqArgs aq{{q, q_out, cos, sin, B, S, Hq, D, (size_t)(B * S * Hq * D / 8), q_row_stride / 8}, *fmt, nullptr, 0, inner_q ? 1 : 0, qt_format{}, map, nullptr};
    RopeFqArgs ak{{k, k_out, cos, sin, B, S, Hk, D, (size_t)(B * S * Hk * D / 8), k_row_stride / 8}, *fmt, nullptr, 0, inner_k ? 1 : 0, qt_format{}, map, nullptr};
    const long nv_max = (Hq > Hk ? Hq : Hk) * D / 8;
    const unsigned tpb = nv_max >= 256 || nv_max < 1 ? 1u : (unsigned)(256 / nv_max);
    size_t blocks = ((size_t)B * (size_t)S + tpb - 1) / tpb;
    if (blocks > 256 * 64) blocks = 256 * 64;
    ValueRowsArgs av{v, vt, v_stride_b, v_stride_h, v_stride_k, S, (int)Hk, (int)(S / 64)};
    WeightPassArgs aw{(const uint4 *)w, (uint4 *)wq, w_elems / 8, 0};
    if (aw.nvec) {
        size_t wb = (aw.nvec + 256 * 4 - 1) / (256 * 4);                  // four vectors per lane and trip, at most two workgroups per CU
        if (wb > 256 * 2) wb = 256 * 2;
        aw.blocks = (unsigned)wb;
    }
    const unsigned total = aw.blocks + (unsigned)blocks + (unsigned)(B * Hk * (S / 64));
    rope_map_value_kernel<<<total, 256, 0, (hipStream_t)stream>>>(aq, ak, av, aw, (unsigned)blocks, tpb);
    return launch_status();
}

int qt_rope_map_value(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out, long B,
                      long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride, const qt_format *fmt, const uint16_t *map,
                      int inner_q, int inner_k, const uint16_t *v, uint16_t *vt, long v_stride_b, long v_stride_h, long v_stride_k, void *stream) {
    return rope_map_value_launch(q, k, cos, sin, q_out, k_out, B, S, Hq, Hk, D, q_row_stride, k_row_stride, fmt, map, inner_q, inner_k, v, vt,
                                 v_stride_b, v_stride_h, v_stride_k, nullptr, nullptr, 0, stream);
}

int qt_rope_map_value_weight(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out, long B,
                             long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride, const qt_format *fmt, const uint16_t *map,
                             int inner_q, int inner_k, const uint16_t *v, uint16_t *vt, long v_stride_b, long v_stride_h, long v_stride_k,
                             const uint16_t *w, uint16_t *wq, size_t w_elems, void *stream) {
    return rope_map_value_launch(q, k, cos, sin, q_out, k_out, B, S, Hq, Hk, D, q_row_stride, k_row_stride, fmt, map, inner_q, inner_k, v, vt,
                                 v_stride_b, v_stride_h, v_stride_k, w, wq, w_elems, stream);
}

int qt_rope_fq_inner_value(const uint16_t *q, const uint16_t *k, const uint16_t *cos, const uint16_t *sin, uint16_t *q_out, uint16_t *k_out,
                           uint8_t *q_out8, uint8_t *k_out8, long B, long S, long Hq, long Hk, long D, long q_row_stride, long k_row_stride,
                           const qt_format *fmt_q, const qt_format *fmt_k, const qt_format *inner_q, const qt_format *inner_k, const uint16_t *v,
                           uint8_t *vt8, long v_stride_b, long v_stride_h, long v_stride_k, const qt_format *fmt_v, void *stream) {
    return rope_fq_launch(q, k, cos, sin, q_out, k_out, q_out8, k_out8, B, S, Hq, Hk, D, q_row_stride, k_row_stride, fmt_q, fmt_k, false, v, vt8,
                          v_stride_b, v_stride_h, v_stride_k, fmt_v, stream, inner_q, inner_k);
}

int qt_causal_lm_loss_bf16(const uint16_t *logits, const long long *labels, long batch, long seq_len, long vocab, long row_stride,
                           long long ignore_index, float *row_loss_scratch, float *loss_out, void *stream) {
    if (!logits || !labels || !row_loss_scratch || !loss_out || batch < 1 || seq_len < 1 || vocab < 1 || row_stride < vocab) return QT_ERR_BAD_ARG;
    if (((uintptr_t)logits & 15u) || (row_stride & 7)) return QT_ERR_UNALIGNED;
    const long rows = batch * seq_len;
    if (rows > 0x7FFFFFFFl) return QT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    nll_rows_kernel<<<(unsigned)rows, 256, 0, st>>>(logits, labels, seq_len, vocab, row_stride, ignore_index, row_loss_scratch);
    nll_mean_kernel<<<1, 256, 0, st>>>(row_loss_scratch, rows, loss_out);
    return launch_status();
}

int qt_colsum_bf16(const uint16_t *x_dev, uint16_t *out_dev, long rows, long cols, void *stream) {
    if (rows < 0 || cols < 0) return QT_ERR_BAD_ARG;
    if (cols == 0) return QT_OK;
    if (!x_dev || !out_dev || cols % 8 != 0) return QT_ERR_BAD_ARG;
    if ((uintptr_t)x_dev & 15u) return QT_ERR_UNALIGNED;
    colsum_kernel<<<(unsigned)((cols + 31) / 32), 256, 0, (hipStream_t)stream>>>(x_dev, out_dev, rows, cols);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

}  // extern "C"
