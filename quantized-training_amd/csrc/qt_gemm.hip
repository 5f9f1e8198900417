// qt_gemm.hip -- fake-quant GEMMs on the gfx950 matrix cores.
//
// Replaces   F.linear(fq(x), weight_fake_quant(W), b)      modules/qat/linear.py:40-41
// and        torch.matmul(fq(a), fq(b))                    modules/quantizable/functional_modules.py:22-26
// where the reference first materialises both fake-quantized operands in memory
// (fake_quantize.py:245-246) and then calls a library GEMM.
//
// Here the operand tiles are fake-quantized WHILE they are staged global -> registers -> LDS
// (x/s -> bf16 -> value map / closed form -> *s -> bf16, the same per-element function as the
// elementwise pass, qt_device.h), so a quantized weight is never written to HBM: W is read once
// per row-block of the output instead of read + written + read again.  Products accumulate in
// fp32 on v_mfma_f32_16x16x32_bf16; bias is added in fp32; one rounding to bf16 at the end.
//
// Tile: 128 x 128 x 64, 256 threads = 4 waves (2 x 2), each wave a 64 x 64 sub-tile = 4 x 4 MFMA
// tiles (32 MFMAs per K-step).  LDS: two buffers x (A 128 rows + B 128 rows) x 144-B padded rows =
// 72 KiB -> 2 workgroups per CU, so one workgroup's staging/quantisation (VALU) overlaps the other's
// MFMAs.  Staging is register-prefetched one K-tile ahead: the 16-B global loads of tile k+1 are
// issued before the MFMAs of tile k and are quantised + written to the other LDS buffer after them;
// one barrier per K-step.  Rows past M / N are clamped on load (their results are never stored) so the
// main loop has no predicated loads; a K tail (K % 64 != 0) zero-fills.  The output tile goes through
// LDS so that global stores are 16-B, row-contiguous.
// B operand layouts: "NT" (B[k][n] at b + n*ldb_n + k, k contiguous: nn.Linear weights, K^T) is
// staged with 16-B loads; "NN" (n contiguous: attention's V) is transposed on the way into LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_device.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kRowBytes = BK * 2 + 16;          // padded LDS row (one ds_read_b128 width): conflict-free fragment reads
constexpr int kTileBytes = BM * kRowBytes;      // one operand tile
constexpr int kLdsBytes = 2 * 2 * kTileBytes;   // {A,B} x double buffer = 73 728 B
constexpr int kOutRowBytes = BN * 2 + 16;       // padded output staging row

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct OperandQ {
    qt_format fmt;
    const uint16_t *lut;
    const float *scale;
    uint32_t *amax;
};

struct GemmArgs {
    const uint16_t *a, *b, *bias;
    uint16_t *y;
    int M, N, K;
    long lda, sa;          // A[m][k] at a + batch*sa + m*lda + k
    long ldb_k, ldb_n, sb; // B[k][n] at b + batch*sb + k*ldb_k + n*ldb_n
    long ldy, sy;
    OperandQ qa, qb;
};

// Fake-quantise the 8 bf16 of one staged 16-B vector (DIV: kDivUnit / kDivFast / kDivExact, qt_device.h).
template <int KIND, int DIV, bool OBS>
__device__ __forceinline__ uint4 quant_vec(uint4 v, const UniformDiv &dv, const Rounder<KIND> &rnd, uint32_t &amax) {
    if constexpr (KIND == QT_FMT_IDENTITY && DIV == kDivUnit && !OBS) return v;
    bool bad = false;
    uint4 r;
    r.x = fq_word_bf16_d<KIND, DIV, OBS>(v.x, dv, rnd, amax, bad);
    r.y = fq_word_bf16_d<KIND, DIV, OBS>(v.y, dv, rnd, amax, bad);
    r.z = fq_word_bf16_d<KIND, DIV, OBS>(v.z, dv, rnd, amax, bad);
    r.w = fq_word_bf16_d<KIND, DIV, OBS>(v.w, dv, rnd, amax, bad);
    if constexpr (DIV == kDivFast) {
        if (__builtin_expect(bad, 0)) {
            uint32_t unused = 0;
            r.x = fq_word_bf16_d<KIND, kDivExact, false>(v.x, dv, rnd, unused, bad);
            r.y = fq_word_bf16_d<KIND, kDivExact, false>(v.y, dv, rnd, unused, bad);
            r.z = fq_word_bf16_d<KIND, kDivExact, false>(v.z, dv, rnd, unused, bad);
            r.w = fq_word_bf16_d<KIND, kDivExact, false>(v.w, dv, rnd, unused, bad);
        }
    }
    return r;
}

// Cold, out-of-line path for everything the specialised kernels do not cover (table formats, scales
// outside the fast-division range, in-kernel quantisation of the A operand): one copy of the code per
// kernel instead of one per staged vector, so the hot loop stays small.
struct QuantOut {
    uint4 v;
    uint32_t amax;
};

__device__ __noinline__ QuantOut quant_vec_generic(uint4 v, qt_format fmt, const uint16_t *lut, float s, int obs,
                                                   uint32_t amax) {
    const UniformDiv dv(s);
    const bool unit = s == 1.0f;
    uint4 r = v;
    if (obs) {
        const Rounder<QT_FMT_IDENTITY> id{fmt, nullptr};
        r = quant_vec<QT_FMT_IDENTITY, kDivUnit, true>(v, dv, id, amax);    // observe only
    }
    switch (fmt.kind) {
        case QT_FMT_LUT: {
            const Rounder<QT_FMT_LUT> rnd{fmt, lut};
            uint32_t un = 0;
            r = unit ? quant_vec<QT_FMT_LUT, kDivUnit, false>(v, dv, rnd, un) : quant_vec<QT_FMT_LUT, kDivExact, false>(v, dv, rnd, un);
            break;
        }
        case QT_FMT_FP_SAT: {
            const Rounder<QT_FMT_FP_SAT> rnd{fmt, nullptr};
            uint32_t un = 0;
            r = unit ? quant_vec<QT_FMT_FP_SAT, kDivUnit, false>(v, dv, rnd, un) : quant_vec<QT_FMT_FP_SAT, kDivExact, false>(v, dv, rnd, un);
            break;
        }
        case QT_FMT_INT: {
            const Rounder<QT_FMT_INT> rnd{fmt, nullptr};
            uint32_t un = 0;
            r = unit ? quant_vec<QT_FMT_INT, kDivUnit, false>(v, dv, rnd, un) : quant_vec<QT_FMT_INT, kDivExact, false>(v, dv, rnd, un);
            break;
        }
        default: {
            if (!unit) {
                const Rounder<QT_FMT_IDENTITY> rnd{fmt, nullptr};
                uint32_t un = 0;
                r = quant_vec<QT_FMT_IDENTITY, kDivExact, false>(v, dv, rnd, un);
            } else {
                r = v;
            }
        }
    }
    return QuantOut{r, amax};
}

// B-operand specialisations compiled inline into the K loop.
constexpr int kBPass = 0;      // identity, scale 1, no observer: plain bf16 GEMM operand
constexpr int kBFpUnit = 1;    // e4m3 / e5m2 closed form, scale 1
constexpr int kBFpFast = 2;    // e4m3 / e5m2 closed form, scale in the fast-division range
constexpr int kBIntFast = 3;   // intN / uintN closed form, scaled (fast division)
constexpr int kBGeneric = 4;   // everything else -> quant_vec_generic

template <int BMODE, bool OBS_B>
__device__ __forceinline__ uint4 stage_b(uint4 v, const OperandQ &q, float s, const UniformDiv &dv, bool obs_rt,
                                         uint32_t &amax) {
    if constexpr (BMODE == kBPass) {
        return v;
    } else if constexpr (BMODE == kBFpUnit) {
        const Rounder<QT_FMT_FP_SAT> rnd{q.fmt, nullptr};
        return quant_vec<QT_FMT_FP_SAT, kDivUnit, OBS_B>(v, dv, rnd, amax);
    } else if constexpr (BMODE == kBFpFast || BMODE == kBIntFast) {
        if (__builtin_expect(!dv.safe, 0)) {       // wave-uniform: reciprocal not a normal number -> full division
            const QuantOut o = quant_vec_generic(v, q.fmt, q.lut, s, OBS_B ? 1 : 0, amax);
            amax = o.amax;
            return o.v;
        }
        if constexpr (BMODE == kBFpFast) {
            const Rounder<QT_FMT_FP_SAT> rnd{q.fmt, nullptr};
            return quant_vec<QT_FMT_FP_SAT, kDivFast, OBS_B>(v, dv, rnd, amax);
        } else {
            const Rounder<QT_FMT_INT> rnd{q.fmt, nullptr};
            return quant_vec<QT_FMT_INT, kDivFast, OBS_B>(v, dv, rnd, amax);
        }
    } else {
        const QuantOut o = quant_vec_generic(v, q.fmt, q.lut, s, obs_rt ? 1 : 0, amax);
        amax = o.amax;
        return o.v;
    }
}

// BMODE: how the B operand is fake-quantised while staged (above).  OBS_B: B's amax observer (inline
// modes only; generic mode takes it at run time).  B_NN: B is n-contiguous (transposed while staging).
// The A operand normally arrives already fake-quantised (identity); anything else goes through the
// out-of-line generic path.
template <int BMODE, bool OBS_B, bool B_NN>
__global__ __launch_bounds__(kThreads, 2) void gemm_fq_kernel(GemmArgs g) {
    const bool KTAIL = (g.K % BK) != 0;   // wave-uniform: loads past K are zero-filled
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const uint16_t *A = g.a + (long)batch * g.sa;
    const uint16_t *B = g.b + (long)batch * g.sb;

    // each operand element is observed by exactly one block column / row (and one batch when B is shared)
    const bool obsA = g.qa.amax != nullptr && blockIdx.x == 0;
    const bool obsB = g.qb.amax != nullptr && blockIdx.y == 0 && (g.sb != 0 || batch == 0);
    const float sA = g.qa.scale ? qt_bf2f(qt_f2bf(*g.qa.scale)) : 1.0f;
    const float sB = g.qb.scale ? qt_bf2f(qt_f2bf(*g.qb.scale)) : 1.0f;
    const bool a_active = g.qa.fmt.kind != QT_FMT_IDENTITY || sA != 1.0f || g.qa.amax != nullptr;
    const UniformDiv dvB(sB);
    uint32_t amaxA = 0, amaxB = 0;
    auto qa = [&](uint4 v, uint32_t &am) -> uint4 {
        if (!a_active) return v;
        const QuantOut o = quant_vec_generic(v, g.qa.fmt, g.qa.lut, sA, obsA ? 1 : 0, am);
        am = o.amax;
        return o.v;
    };
    auto qb = [&](uint4 v, uint32_t &am) -> uint4 {
        if constexpr (OBS_B) {
            // blocks that do not own the observation still quantise; they just discard their amax
            uint32_t local = am;
            uint4 r = stage_b<BMODE, true>(v, g.qb, sB, dvB, obsB, local);
            am = obsB ? local : am;
            return r;
        } else {
            return stage_b<BMODE, false>(v, g.qb, sB, dvB, obsB, am);
        }
    };

    // staging map: 4 vectors per operand per thread; vector v covers tile row (tid>>3) + 32*v, k-chunk tid&7.
    // Rows past the matrix edge are clamped (valid memory, results discarded).
    const int srow = tid >> 3, schunk = tid & 7;
    const uint16_t *pa0, *pa1, *pa2, *pa3, *pb0, *pb1, *pb2, *pb3;
    {
        auto arow = [&](int v) { int r = m0 + srow + 32 * v; return r < g.M ? r : g.M - 1; };
        pa0 = A + (long)arow(0) * g.lda + schunk * 8;
        pa1 = A + (long)arow(1) * g.lda + schunk * 8;
        pa2 = A + (long)arow(2) * g.lda + schunk * 8;
        pa3 = A + (long)arow(3) * g.lda + schunk * 8;
        if constexpr (!B_NN) {
            auto brow = [&](int v) { int r = n0 + srow + 32 * v; return r < g.N ? r : g.N - 1; };
            pb0 = B + (long)brow(0) * g.ldb_n + schunk * 8;
            pb1 = B + (long)brow(1) * g.ldb_n + schunk * 8;
            pb2 = B + (long)brow(2) * g.ldb_n + schunk * 8;
            pb3 = B + (long)brow(3) * g.ldb_n + schunk * 8;
        } else {
            // NN: vector v covers k-row (tid>>4) + 16*v, n-chunk tid&15 (8 n values, clamped as a group)
            int n = n0 + (tid & 15) * 8;
            n = n + 8 <= g.N ? n : (g.N >= 8 ? g.N - 8 : 0);
            const uint16_t *base = B + n + (long)(tid >> 4) * g.ldb_k;
            pb0 = base;
            pb1 = base + 16 * g.ldb_k;
            pb2 = base + 32 * g.ldb_k;
            pb3 = base + 48 * g.ldb_k;
        }
    }
    const long bstep = B_NN ? (long)BK * g.ldb_k : (long)BK;
    uint4 a0, a1, a2, a3, b0, b1, b2, b3;
    const uint4 zero4 = {0u, 0u, 0u, 0u};

#define QT_LOAD_TILES(k0)                                                                          \
    do {                                                                                           \
        if (KTAIL) {                                                                               \
            const bool ka = (k0) + schunk * 8 < g.K;                                               \
            a0 = ka ? *(const uint4 *)pa0 : zero4; a1 = ka ? *(const uint4 *)pa1 : zero4;          \
            a2 = ka ? *(const uint4 *)pa2 : zero4; a3 = ka ? *(const uint4 *)pa3 : zero4;          \
            if constexpr (!B_NN) {                                                                 \
                b0 = ka ? *(const uint4 *)pb0 : zero4; b1 = ka ? *(const uint4 *)pb1 : zero4;      \
                b2 = ka ? *(const uint4 *)pb2 : zero4; b3 = ka ? *(const uint4 *)pb3 : zero4;      \
            } else {                                                                               \
                const int kr = (k0) + (tid >> 4);                                                  \
                b0 = kr < g.K ? *(const uint4 *)pb0 : zero4;                                       \
                b1 = kr + 16 < g.K ? *(const uint4 *)pb1 : zero4;                                  \
                b2 = kr + 32 < g.K ? *(const uint4 *)pb2 : zero4;                                  \
                b3 = kr + 48 < g.K ? *(const uint4 *)pb3 : zero4;                                  \
            }                                                                                      \
        } else {                                                                                   \
            a0 = *(const uint4 *)pa0; a1 = *(const uint4 *)pa1;                                    \
            a2 = *(const uint4 *)pa2; a3 = *(const uint4 *)pa3;                                    \
            b0 = *(const uint4 *)pb0; b1 = *(const uint4 *)pb1;                                    \
            b2 = *(const uint4 *)pb2; b3 = *(const uint4 *)pb3;                                    \
        }                                                                                          \
        pa0 += BK; pa1 += BK; pa2 += BK; pa3 += BK;                                                \
        pb0 += bstep; pb1 += bstep; pb2 += bstep; pb3 += bstep;                                    \
    } while (0)

    const int a_off = srow * kRowBytes + schunk * 16;
#define QT_STORE_TILES(buf)                                                                        \
    do {                                                                                           \
        unsigned char *la = lds + (buf) * 2 * kTileBytes + a_off;                                  \
        *(uint4 *)(la) = qa(a0, amaxA);                                                            \
        *(uint4 *)(la + 32 * kRowBytes) = qa(a1, amaxA);                                           \
        *(uint4 *)(la + 64 * kRowBytes) = qa(a2, amaxA);                                           \
        *(uint4 *)(la + 96 * kRowBytes) = qa(a3, amaxA);                                           \
        unsigned char *lb = lds + (buf) * 2 * kTileBytes + kTileBytes;                             \
        if constexpr (!B_NN) {                                                                     \
            *(uint4 *)(lb + a_off) = qb(b0, amaxB);                                                \
            *(uint4 *)(lb + a_off + 32 * kRowBytes) = qb(b1, amaxB);                               \
            *(uint4 *)(lb + a_off + 64 * kRowBytes) = qb(b2, amaxB);                               \
            *(uint4 *)(lb + a_off + 96 * kRowBytes) = qb(b3, amaxB);                               \
        } else {                                                                                   \
            const uint4 q4[4] = {qb(b0, amaxB), qb(b1, amaxB), qb(b2, amaxB), qb(b3, amaxB)};      \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                        \
                uint16_t *p = (uint16_t *)(lb + ((tid >> 4) + 16 * v) * 2) + (tid & 15) * 8 * (kRowBytes / 2); \
                p[0 * (kRowBytes / 2)] = (uint16_t)q4[v].x; p[1 * (kRowBytes / 2)] = (uint16_t)(q4[v].x >> 16); \
                p[2 * (kRowBytes / 2)] = (uint16_t)q4[v].y; p[3 * (kRowBytes / 2)] = (uint16_t)(q4[v].y >> 16); \
                p[4 * (kRowBytes / 2)] = (uint16_t)q4[v].z; p[5 * (kRowBytes / 2)] = (uint16_t)(q4[v].z >> 16); \
                p[6 * (kRowBytes / 2)] = (uint16_t)q4[v].w; p[7 * (kRowBytes / 2)] = (uint16_t)(q4[v].w >> 16); \
            }                                                                                      \
        }                                                                                          \
    } while (0)

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (g.K + BK - 1) / BK;
    QT_LOAD_TILES(0);
    QT_STORE_TILES(0);
    __syncthreads();

    const int frow = lane & 15, fk = (lane >> 4) * 16;   // fragment row and byte offset of its 8 k-values
    const int fa_off = (wm * 64 + frow) * kRowBytes + fk;
    const int fb_off = kTileBytes + (wn * 64 + frow) * kRowBytes + fk;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) QT_LOAD_TILES((kt + 1) * BK);
        const unsigned char *la = lds + buf * 2 * kTileBytes + fa_off;
        const unsigned char *lb = lds + buf * 2 * kTileBytes + fb_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8_t *)(la + i * 16 * kRowBytes + ks * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8_t *)(lb + j * 16 * kRowBytes + ks * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) QT_STORE_TILES(buf ^ 1);
        __syncthreads();
    }
#undef QT_LOAD_TILES
#undef QT_STORE_TILES

    // ---- epilogue: acc (+bias) -> bf16 -> LDS (row-major, padded) -> 16-B row-contiguous global stores
    // C fragment of a 16x16 tile: row = (lane>>4)*4 + r, col = lane&15.
    {
        const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + ccol;
            const int n = n0 + nl;
            const float bv = (g.bias && n < g.N) ? qt_bf2f(g.bias[n]) : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = wm * 64 + i * 16 + crow;
                const uint32_t p01 = pack_bf16x2(acc[i][j][0] + bv, acc[i][j][1] + bv);
                const uint32_t p23 = pack_bf16x2(acc[i][j][2] + bv, acc[i][j][3] + bv);
                uint16_t *o = (uint16_t *)(lds + ml * kOutRowBytes) + nl;
                o[0] = (uint16_t)p01;
                o[kOutRowBytes / 2] = (uint16_t)(p01 >> 16);
                o[2 * (kOutRowBytes / 2)] = (uint16_t)p23;
                o[3 * (kOutRowBytes / 2)] = (uint16_t)(p23 >> 16);
            }
        }
        __syncthreads();
        uint16_t *Y = g.y + (long)batch * g.sy;
        const bool vec_ok = ((g.ldy & 7) == 0) && ((((uintptr_t)Y) & 15u) == 0);
        // 128 rows x 16 vectors of 8 columns; thread t handles vector (t & 15) of rows (t >> 4) + 16*p
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ml = (tid >> 4) + 16 * p, c8 = (tid & 15) * 8;
            const int m = m0 + ml, n = n0 + c8;
            if (m >= g.M || n >= g.N) continue;
            const uint4 v = *(const uint4 *)(lds + ml * kOutRowBytes + c8 * 2);
            uint16_t *dst = Y + (long)m * g.ldy + n;
            if (vec_ok && n + 8 <= g.N) {
                *(uint4 *)dst = v;
            } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < g.N) dst[e] = (uint16_t)(w4[e >> 1] >> ((e & 1) * 16));
            }
        }
    }

    if (g.qa.amax != nullptr || g.qb.amax != nullptr) {
        __syncthreads();
        uint32_t(*s_red)[kThreads / 64] = (uint32_t(*)[kThreads / 64])lds;   // tiles are dead by now
        amaxA = wave_max_u32(amaxA);
        amaxB = wave_max_u32(amaxB);
        if (lane == 0) {
            s_red[0][wave] = amaxA;
            s_red[1][wave] = amaxB;
        }
        __syncthreads();
        if (tid < 2) {
            uint32_t m = 0;
            for (int w = 0; w < kThreads / 64; ++w) m = m > s_red[tid][w] ? m : s_red[tid][w];
            uint32_t *dst = tid == 0 ? g.qa.amax : g.qb.amax;
            if (dst && m && m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
        }
    }
}

template <int BMODE, bool OBS_B, bool B_NN>
int launch_one(const GemmArgs &g, int batches, hipStream_t st) {
    static bool attr_set = false;   // per instantiation; idempotent, so a race is harmless
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)gemm_fq_kernel<BMODE, OBS_B, B_NN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batches);
    gemm_fq_kernel<BMODE, OBS_B, B_NN><<<grid, kThreads, kLdsBytes, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int BMODE, bool OBS_B>
int launch_layout(const GemmArgs &g, int batches, bool b_nn, hipStream_t st) {
    return b_nn ? launch_one<BMODE, OBS_B, true>(g, batches, st) : launch_one<BMODE, OBS_B, false>(g, batches, st);
}

// Picks the inline specialisation for the B operand.  The scale lives on the device, so a non-NULL
// scale pointer selects the scaled (fast-division) kernel; that kernel multiplies by 1 exactly when the
// scale turns out to be 1, and falls back per vector when the scale leaves the fast range.
int launch_gemm(const GemmArgs &g, int batches, bool b_nn, hipStream_t st) {
    const OperandQ &q = g.qb;
    const bool obs = q.amax != nullptr;
    if (q.fmt.kind == QT_FMT_IDENTITY && !q.scale && !obs) return launch_layout<kBPass, false>(g, batches, b_nn, st);
    if (q.fmt.kind == QT_FMT_FP_SAT && !q.scale)
        return obs ? launch_layout<kBFpUnit, true>(g, batches, b_nn, st) : launch_layout<kBFpUnit, false>(g, batches, b_nn, st);
    if (q.fmt.kind == QT_FMT_FP_SAT)
        return obs ? launch_layout<kBFpFast, true>(g, batches, b_nn, st) : launch_layout<kBFpFast, false>(g, batches, b_nn, st);
    if (q.fmt.kind == QT_FMT_INT && q.scale)
        return obs ? launch_layout<kBIntFast, true>(g, batches, b_nn, st) : launch_layout<kBIntFast, false>(g, batches, b_nn, st);
    return launch_layout<kBGeneric, false>(g, batches, b_nn, st);
}

OperandQ make_opq(const qt_operand_q *q) {
    OperandQ o;
    if (q) {
        o.fmt = q->fmt;
        o.lut = q->lut_dev;
        o.scale = q->scale_f32_dev;
        o.amax = q->amax_bits_dev;
    } else {
        o.fmt = qt_format{QT_FMT_IDENTITY, 0, 0, 0.f, 0.f};
        o.lut = nullptr;
        o.scale = nullptr;
        o.amax = nullptr;
    }
    return o;
}

bool opq_ok(const OperandQ &o) { return o.fmt.kind != QT_FMT_LUT || o.lut != nullptr; }

}  // namespace

extern "C" {

int qt_linear_fq_bf16(const uint16_t *x, const uint16_t *w, const uint16_t *bias, uint16_t *y, int M, int N, int K,
                      const qt_operand_q *qx, const qt_operand_q *qw, void *stream) {
    if (M == 0 || N == 0) return QT_OK;
    if (!x || !w || !y || M < 0 || N < 0 || K <= 0) return QT_ERR_BAD_ARG;
    if ((K & 7) || (((uintptr_t)x | (uintptr_t)w) & 15u)) return QT_ERR_UNALIGNED;
    GemmArgs g{x, w, bias, y, M, N, K, (long)K, 0, 1, (long)K, 0, (long)N, 0, make_opq(qx), make_opq(qw)};
    if (!opq_ok(g.qa) || !opq_ok(g.qb)) return QT_ERR_BAD_ARG;
    return launch_gemm(g, 1, false, (hipStream_t)stream);
}

int qt_bmm_fq_bf16(const uint16_t *a, const uint16_t *b, uint16_t *y, int B, int M, int N, int K, long lda, long sa,
                   long ldb_k, long ldb_n, long sb, const qt_operand_q *qa, const qt_operand_q *qb, void *stream) {
    if (B == 0 || M == 0 || N == 0) return QT_OK;
    if (!a || !b || !y || B < 0 || M < 0 || N < 0 || K <= 0 || B > 65535) return QT_ERR_BAD_ARG;
    const bool nt = (ldb_k == 1);     // k contiguous
    const bool nn = (ldb_n == 1);     // n contiguous
    if (!nt && !nn) return QT_ERR_BAD_ARG;
    if ((((uintptr_t)a | (uintptr_t)b) & 15u) || (lda & 7) || (sa & 7) || (sb & 7)) return QT_ERR_UNALIGNED;
    if (nt && ((K & 7) || (ldb_n & 7))) return QT_ERR_UNALIGNED;
    if (!nt && ((N & 7) || (ldb_k & 7) || (K & 7))) return QT_ERR_UNALIGNED;
    GemmArgs g{a, b, nullptr, y, M, N, K, lda, sa, ldb_k, ldb_n, sb, (long)N, (long)M * N, make_opq(qa), make_opq(qb)};
    if (!opq_ok(g.qa) || !opq_ok(g.qb)) return QT_ERR_BAD_ARG;
    return launch_gemm(g, B, !nt, (hipStream_t)stream);
}

}  // extern "C"
