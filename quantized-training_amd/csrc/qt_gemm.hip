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
// tiles.  LDS: two buffers x (A 128 rows + B 128 rows) x 144-B padded rows = 72 KiB -> 2 blocks/CU.
// Staging is register-prefetched one K-tile ahead (loads of tile k+1 are issued before the MFMAs
// of tile k, quantised and written to the other LDS buffer after them): one barrier per K-step.
// B operand layouts: "NT" (B[k][n] at b + n*ldb_n + k, k contiguous: nn.Linear weights, K^T) is
// staged with 16-B loads; "NN" (n contiguous: attention's V) is transposed on the way into LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_device.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kRowBytes = BK * 2 + 16;          // padded LDS row (one ds_read_b128 width) -> no power-of-2 stride
constexpr int kTileBytes = BM * kRowBytes;      // one operand tile
constexpr int kLdsBytes = 2 * 2 * kTileBytes;   // {A,B} x double buffer = 73 728 B

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

// Quantise the 8 bf16 of one staged 16-B vector.
template <int KIND>
__device__ __forceinline__ uint4 quant_vec(uint4 v, float s, bool unit, bool obs, const Rounder<KIND> &rnd, uint32_t &amax) {
    if constexpr (KIND == QT_FMT_IDENTITY) {
        if (!obs) return v;
    }
    if (unit) {
        if (obs) {
            v.x = fq_word_bf16<KIND, true, true>(v.x, s, rnd, amax);
            v.y = fq_word_bf16<KIND, true, true>(v.y, s, rnd, amax);
            v.z = fq_word_bf16<KIND, true, true>(v.z, s, rnd, amax);
            v.w = fq_word_bf16<KIND, true, true>(v.w, s, rnd, amax);
        } else {
            v.x = fq_word_bf16<KIND, true, false>(v.x, s, rnd, amax);
            v.y = fq_word_bf16<KIND, true, false>(v.y, s, rnd, amax);
            v.z = fq_word_bf16<KIND, true, false>(v.z, s, rnd, amax);
            v.w = fq_word_bf16<KIND, true, false>(v.w, s, rnd, amax);
        }
    } else {
        if (obs) {
            v.x = fq_word_bf16<KIND, false, true>(v.x, s, rnd, amax);
            v.y = fq_word_bf16<KIND, false, true>(v.y, s, rnd, amax);
            v.z = fq_word_bf16<KIND, false, true>(v.z, s, rnd, amax);
            v.w = fq_word_bf16<KIND, false, true>(v.w, s, rnd, amax);
        } else {
            v.x = fq_word_bf16<KIND, false, false>(v.x, s, rnd, amax);
            v.y = fq_word_bf16<KIND, false, false>(v.y, s, rnd, amax);
            v.z = fq_word_bf16<KIND, false, false>(v.z, s, rnd, amax);
            v.w = fq_word_bf16<KIND, false, false>(v.w, s, rnd, amax);
        }
    }
    return v;
}

// KA / KB: rounding kind of each operand.  B_NN: B is n-contiguous (transposed while staging).
template <int KA, int KB, bool B_NN>
__global__ __launch_bounds__(kThreads, 2) void gemm_fq_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const uint16_t *A = g.a + (long)batch * g.sa;
    const uint16_t *B = g.b + (long)batch * g.sb;

    Rounder<KA> ra{g.qa.fmt, g.qa.lut};
    Rounder<KB> rb{g.qb.fmt, g.qb.lut};
    float sA = g.qa.scale ? qt_bf2f(qt_f2bf(*g.qa.scale)) : 1.0f;
    float sB = g.qb.scale ? qt_bf2f(qt_f2bf(*g.qb.scale)) : 1.0f;
    const bool unitA = sA == 1.0f, unitB = sB == 1.0f;
    // each operand element is observed by exactly one block column / row / batch owner
    const bool obsA = g.qa.amax != nullptr && blockIdx.x == 0;
    const bool obsB = g.qb.amax != nullptr && blockIdx.y == 0 && (g.sb != 0 || batch == 0);
    uint32_t amaxA = 0, amaxB = 0;

    // staging map: 4 vectors per operand per thread; vector v covers row (tid>>3) + 32*v, k-chunk tid&7
    const int srow = tid >> 3, schunk = tid & 7;
    uint4 ra_reg[4], rb_reg[4];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = srow + 32 * v;
            const int k = k0 + schunk * 8;
            uint4 z = {0u, 0u, 0u, 0u};
            ra_reg[v] = (m0 + row < g.M && k < g.K) ? *(const uint4 *)(A + (long)(m0 + row) * g.lda + k) : z;
            if constexpr (!B_NN) {
                rb_reg[v] = (n0 + row < g.N && k < g.K) ? *(const uint4 *)(B + (long)(n0 + row) * g.ldb_n + k) : z;
            } else {
                // NN: vector v covers k-row (tid>>4) + 16*v, n-chunk tid&15 (8 n values)
                const int kr = k0 + (tid >> 4) + 16 * v;
                const int n = n0 + (tid & 15) * 8;
                rb_reg[v] = (kr < g.K && n < g.N) ? *(const uint4 *)(B + (long)kr * g.ldb_k + n) : z;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char *la = lds + buf * 2 * kTileBytes;
        unsigned char *lb = la + kTileBytes;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = srow + 32 * v;
            uint4 qa = quant_vec<KA>(ra_reg[v], sA, unitA, obsA, ra, amaxA);
            *(uint4 *)(la + row * kRowBytes + schunk * 16) = qa;
            uint4 qb = quant_vec<KB>(rb_reg[v], sB, unitB, obsB, rb, amaxB);
            if constexpr (!B_NN) {
                *(uint4 *)(lb + row * kRowBytes + schunk * 16) = qb;
            } else {
                const int kk = (tid >> 4) + 16 * v;       // k within the tile
                const int nb = (tid & 15) * 8;            // first of 8 n rows
                uint16_t *p = (uint16_t *)(lb + kk * 2);
                p[(nb + 0) * (kRowBytes / 2)] = (uint16_t)qb.x;
                p[(nb + 1) * (kRowBytes / 2)] = (uint16_t)(qb.x >> 16);
                p[(nb + 2) * (kRowBytes / 2)] = (uint16_t)qb.y;
                p[(nb + 3) * (kRowBytes / 2)] = (uint16_t)(qb.y >> 16);
                p[(nb + 4) * (kRowBytes / 2)] = (uint16_t)qb.z;
                p[(nb + 5) * (kRowBytes / 2)] = (uint16_t)(qb.z >> 16);
                p[(nb + 6) * (kRowBytes / 2)] = (uint16_t)qb.w;
                p[(nb + 7) * (kRowBytes / 2)] = (uint16_t)(qb.w >> 16);
            }
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (g.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 15, fk = (lane >> 4) * 16;   // fragment row and byte offset of its 8 k-values
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const unsigned char *la = lds + buf * 2 * kTileBytes + (wm * 64 + frow) * kRowBytes + fk;
        const unsigned char *lb = lds + buf * 2 * kTileBytes + kTileBytes + (wn * 64 + frow) * kRowBytes + fk;
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
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C[row = (lane>>4)*4 + r][col = lane&15] per 16x16 tile
    uint16_t *Y = g.y + (long)batch * g.sy;
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + ccol;
        if (n >= g.N) continue;
        const float bv = g.bias ? qt_bf2f(g.bias[n]) : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + crow + r;
                if (m < g.M) Y[(long)m * g.ldy + n] = (uint16_t)pack_bf16x2(acc[i][j][r] + bv, 0.0f);
            }
        }
    }

    if (g.qa.amax != nullptr || g.qb.amax != nullptr) {
        uint32_t(*s_red)[kThreads / 64] = (uint32_t(*)[kThreads / 64])lds;   // tiles are dead after the last barrier
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
            if (dst && m) atomicMax(dst, m);
        }
    }
}

template <int KA, int KB>
int launch_gemm_kk(const GemmArgs &g, int batches, bool b_nn, hipStream_t st) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batches);
    hipError_t e;
    static bool attr_set[2] = {false, false};   // per <KA,KB> instantiation; idempotent, so a race is harmless
    if (b_nn) {
        if (!attr_set[1]) {
            e = hipFuncSetAttribute((const void *)gemm_fq_kernel<KA, KB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_set[1] = true;
        }
        gemm_fq_kernel<KA, KB, true><<<grid, kThreads, kLdsBytes, st>>>(g);
    } else {
        if (!attr_set[0]) {
            e = hipFuncSetAttribute((const void *)gemm_fq_kernel<KA, KB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_set[0] = true;
        }
        gemm_fq_kernel<KA, KB, false><<<grid, kThreads, kLdsBytes, st>>>(g);
    }
    e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int KA>
int launch_gemm_k(const GemmArgs &g, int batches, bool b_nn, hipStream_t st) {
    switch (g.qb.fmt.kind) {
        case QT_FMT_LUT: return launch_gemm_kk<KA, QT_FMT_LUT>(g, batches, b_nn, st);
        case QT_FMT_FP_SAT: return launch_gemm_kk<KA, QT_FMT_FP_SAT>(g, batches, b_nn, st);
        case QT_FMT_INT: return launch_gemm_kk<KA, QT_FMT_INT>(g, batches, b_nn, st);
        case QT_FMT_IDENTITY: return launch_gemm_kk<KA, QT_FMT_IDENTITY>(g, batches, b_nn, st);
        default: return QT_ERR_BAD_ARG;
    }
}

int launch_gemm(const GemmArgs &g, int batches, bool b_nn, hipStream_t st) {
    switch (g.qa.fmt.kind) {
        case QT_FMT_LUT: return launch_gemm_k<QT_FMT_LUT>(g, batches, b_nn, st);
        case QT_FMT_FP_SAT: return launch_gemm_k<QT_FMT_FP_SAT>(g, batches, b_nn, st);
        case QT_FMT_INT: return launch_gemm_k<QT_FMT_INT>(g, batches, b_nn, st);
        case QT_FMT_IDENTITY: return launch_gemm_k<QT_FMT_IDENTITY>(g, batches, b_nn, st);
        default: return QT_ERR_BAD_ARG;
    }
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
