// qt_train_gemm.hip -- the bf16 GEMMs of a TRAINING step's Linear layers (BASELINE configs[4]: RoBERTa-base, [16, 128] batches), in-tree.
//
// The reference's QAT Linear is F.linear(input, weight_fake_quant(weight), bias) under autograd (modules/qat/linear.py:40-41, hooks of
// quantize.py:116-179): per layer and step one forward product and, in the backward, the input gradient and the weight gradient -- on
// operands the fake-quantizers have already rounded (bf16 VALUES of int8 / E5M2 codes times a scale).  Rounds 1-5 ran those 18 products
// per encoder layer on hipBLASLt: 216 launches of 12-28 us for 2.4-9.7 GFLOP each (0.08-0.2 of the bf16 peak: launch- and tile-
// quantisation-bound), with q / k / v as three launches per direction.  Here:
//     C[M][N] (bf16) = op(A) . op(B) (+ bias[N]),   fp32 accumulation on v_mfma_f32_16x16x32_bf16, ONE rounding to bf16
//         forward   y  = x  . Wq^T + b     A = x  [M][K]  (k contiguous)            B = Wq [N][K]  (k contiguous)
//         dgrad     gx = gy . Wq           A = gy [M][K]  (k contiguous)            B = Wq [K][N]  (n contiguous: trans_b)
//         wgrad     gW = gy^T . x          A = gy [K][M]  (m contiguous: trans_a)   B = x  [K][N]  (n contiguous: trans_b)
// for up to four problems of ONE shape per launch (query / key / value: 16.9 us for the three against 3 x 7.7; the package still issues them
// one by one -- under autograd the three calls arrive at different times).  The summation order of an output element is fixed by the
// tile walk (k tiles in ascending order, 32 products per matrix instruction): the same bits on every launch.
//
// Work decomposition.  A workgroup of 512 threads (8 waves, 4 x 2) owns a BM x BN tile -- 128 x 64, 128 x 128 or 64 x 64 by a fixed rule
// of the layout and the problem size (pick_tile) --; k tiles of 64 through an LDS ring of 3-4 stages filled by
// LDS-DMA (global_load_lds_dwordx4, hand-counted vmcnt: two or three k tiles in flight per workgroup), one barrier per k tile.
// Operands whose contraction index is contiguous are staged as [rows][64 k] images of 128-byte rows (16-byte chunks XOR-swizzled by
// (row >> 1) & 7, one ds_read_b128 per fragment).  Operands stored with the contraction index as the ROW index (trans_a / trans_b) are
// staged as they come, [64 k][BM or BN columns], and read through ds_read_b64_tr_b16, gfx950's transposing LDS read (two per
// fragment): a 16-lane group reads a 4 (k) x 16 (columns) block and every lane receives its column's four k values -- exactly the
// matrix instruction's operand layout.  The 32-byte column groups of a row are XOR-swizzled so that the 8 rows a 32-lane half touches
// per read fall on distinct banks (s_tr below).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/qt_hip.h"
#include "qt_device.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 64;                 // k tile
constexpr int kThreads = 512;          // 8 waves: 4 (rows) x 2 (columns); two per SIMD, so one wave's LDS latencies hide under the other's matrix instructions
constexpr int kMaxProblems = 4;

struct Problem {
    const uint16_t *a, *b, *bias;
    uint16_t *c;
};
struct Args {
    Problem p[kMaxProblems];
    int count, M, N, K;
    long lda, ldb, ldc;
    int tiles_m, tiles_n;
#ifdef QT_TUNING_BUILD
    unsigned long long *dbg;
#endif
};

// [rows][64 k] image, 128-byte rows: byte offset of 16-byte chunk `ch` (0..7) of row `row`
__device__ __forceinline__ int off_rows(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// [64 k][C columns] image read by ds_read_b64_tr_b16.  RB = 2 C bytes per row (128 or 256).  A 32-lane half reads rows r0 + q and r0 + 8 + q
// (q = 0..3), 32 bytes of the same column group each: eight 32-byte pieces that must fall on eight different bank groups (64 banks x 4 B =
// eight groups of 32 B).  Row r starts at bank group (r RB / 32) mod 8 -- 0 for RB = 256, 0 or 4 for RB = 128 -- so the column group index
// is XORed with a value that is distinct over {q, 8 + q} (RB = 256) resp. over the rows of equal parity among them (RB = 128).
template <int RB>
__device__ __forceinline__ int s_tr(int row) {
    static_assert(RB == 128 || RB == 256, "row lengths of 64 and 128 columns (192: 384-byte rows start at bank group 4 r mod 8 like 128-byte ones and take their formula -- built, measured, not used)");
    if constexpr (RB == 256) return (row & 3) | (((row >> 3) & 1) << 2);
    else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}
template <int RB>
__device__ __forceinline__ int off_tr(int row, int ch) {       // 16-byte chunk `ch` of k row `row`
    return row * RB + ((((ch >> 1) ^ s_tr<RB>(row))) << 5) + ((ch & 1) << 4);
}

__device__ __forceinline__ bf16x8 ld_frag_rows(const unsigned char *img, int row, int ch) {
    return *(const bf16x8 *)(img + off_rows(row, ch));
}
// the fragment of 16 columns c0 .. c0 + 15 (c0 a multiple of 16) for k = k0 .. k0 + 31 of a transposed image: lane l receives column
// c0 + l % 16, k = k0 + 8 (l / 16) .. + 7
template <int RB>
__device__ __forceinline__ bf16x8 ld_frag_tr(const unsigned char *img, int c0, int k0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r_lo = k0 + 8 * g + q, r_hi = r_lo + 4;
    const int cp = c0 >> 4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(img + r_lo * RB + ((cp ^ s_tr<RB>(r_lo)) << 5) + p * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(img + r_hi * RB + ((cp ^ s_tr<RB>(r_hi)) << 5) + p * 8));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// one LDS-DMA piece: the wave's 64 lanes bring 16 bytes each from `src` (per lane) to LDS bytes dst .. dst + 1023 (wave-uniform dst).
// Inline asm: the compiler must not know about these entries of the vector-memory queue (it would wait for all of them in front of
// every other access); the waits are counted by hand below.
__device__ __forceinline__ void dma16(const void *src, uint32_t dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_and_barrier() {       // at most N of this wave's DMA pieces still in flight; LDS reads drained
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// The LDS ring: S - 1 k tiles in flight per workgroup.  Small tiles leave room for two workgroups per CU (four waves per SIMD); a ring
// over the whole CU's LDS for a single workgroup measured SLOWER on every shape (profiles/r06_train_gemm.txt): what these launches lack
// is waves to overlap, not bytes in flight.
template <int BM, int BN>
struct Ring {
    static constexpr int kStage = (BM + BN) * kBK * 2;                  // 16 / 24 / 32 KiB
    // (128 x 192: 3 x 40 KiB; a fourth stage -- the whole LDS of a CU -- changes nothing: with 24 multiplications and 16 fragment reads per
    // wave and k tile that shape is paced by the LDS reads, 0.76 us per k tile, not by the operands in flight)
    static constexpr int kStages = (BM + BN) <= 128 ? 4 : ((BM + BN) <= 192 ? 3 : ((BM + BN) <= 256 ? 4 : 3));
    static constexpr int kBytes = kStage * kStages;
};

// the last S - 1 k tiles: nothing left to request, the queue drains (D k tiles still in flight behind the one being multiplied)
template <int D, int PIECES, class F>
__device__ __forceinline__ void drain(int &kt, F &&compute) {
    wait_and_barrier<D * PIECES>();
    compute(kt++);
    if constexpr (D > 0) drain<D - 1, PIECES>(kt, compute);
}

// One output tile: `tile` of problem P (column tile tile / tiles_m, row tile tile % tiles_m); `lds`: the workgroup's dynamic LDS
// (Ring<BM, BN>::kBytes of it).
// the problem of workgroup `block` (compile-time indices only into the kernel arguments: a run-time index, or a reference to the
// argument struct, copies it to scratch memory)
// Workgroups go to the eight XCDs round robin (workgroup b to XCD b % 8), each with its own L2.  Where the row-tile count is a multiple
// of 8 the walk "row tiles of one column tile first" already gives an XCD two or three row tiles for ALL column tiles (it fetches an
// eighth of A and shares every B tile between its workgroups).  Where it is not (768 / 64 = 12 row tiles: the small weight gradients) an
// XCD's tiles are scattered over the whole product and every L2 fetches both operands whole; there the tiles are dealt out so that XCD x
// takes a contiguous run of the walk -- one and a half column tiles with all their row tiles.
__device__ __forceinline__ int xcd_run(int b, int total) {
    const int x = b & 7, j = b >> 3, q = total >> 3, r = total & 7;
    return x * q + (x < r ? x : r) + j;
}
#define QT_TG_PICK(a, block, P, tile_)                                     \
    const int tiles_##tile_ = (a).tiles_m * (a).tiles_n;                   \
    const int tile_ = ((a).tiles_m & 7) ? xcd_run((block) % tiles_##tile_, tiles_##tile_) : (block) % tiles_##tile_; \
    Problem P = (a).p[0];                                                  \
    {                                                                      \
        const int prob_ = (block) / ((a).tiles_m * (a).tiles_n);           \
        if (prob_ == 1) P = (a).p[1];                                      \
        if (prob_ == 2) P = (a).p[2];                                      \
        if (prob_ == 3) P = (a).p[3];                                      \
    }
struct Dims {
    int M, N, K;
    long lda, ldb, ldc;
    int tiles_m;
    unsigned long long *dbg;       // tuning build (QT_TG_STAMPS = device address): s_memrealtime of wave 0 at four points, per workgroup
};
#ifdef QT_TUNING_BUILD
#define QT_TG_DIMS(a) Dims{(a).M, (a).N, (a).K, (a).lda, (a).ldb, (a).ldc, (a).tiles_m, (a).dbg}
#define QT_TG_STAMP(a, slot)                                                                                     \
    do {                                                                                                         \
        if ((a).dbg && threadIdx.x == 0) (a).dbg[(size_t)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define QT_TG_DIMS(a) Dims{(a).M, (a).N, (a).K, (a).lda, (a).ldb, (a).ldc, (a).tiles_m, nullptr}
#define QT_TG_STAMP(a, slot)
#endif

template <bool TA, bool TB, int BM, int BN, int SS = 0>
__device__ __forceinline__ void gemm_tile(const Problem P, const Dims a, const int tile, unsigned char *lds) {
    constexpr int kAImg = BM * kBK * 2;                                 // bytes of the A image inside a stage
    constexpr int kStage = Ring<BM, BN>::kStage, S = SS ? SS : Ring<BM, BN>::kStages;      // (SS: a deeper ring for a lone workgroup per CU)
    constexpr int kAV = BM / 64, kBV = BN / 64;                         // DMA pieces (1 KiB) per wave and k tile
    constexpr int kPieces = kAV + kBV;
    constexpr int WM = BM / 64, WN = BN / 32;                           // 16 x 16 output tiles per wave: the wave owns (BM / 4) x (BN / 2)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wr = wave >> 1, wc = wave & 1;      // wr 0..3, wc 0..1
    // consecutive workgroups walk the row tiles of one column tile: they share its B tile while it is hot in L2
    const int tn = tile / a.tiles_m, tm = tile % a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = a.K / kBK;

    // ---- DMA sources of k tile 0 (per lane) and what a k tile adds to them; piece v of this wave is image piece v * 8 + wave.
    // The XOR swizzle of the image goes on the SOURCE address (the DMA writes a wave's 1 KiB linearly): position `pos` of a row receives
    // the chunk that off_rows / off_tr place there.
    const unsigned char *sa[kAV], *sb[kBV];
    uint32_t da[kAV], db[kBV];                                          // piece offsets inside a stage
#pragma unroll
    for (int v = 0; v < kAV; ++v) {
        const int piece = v * 8 + wave;
        da[v] = piece * 1024;
        if constexpr (!TA) {
            const int row = piece * 8 + (lane >> 3), pos = lane & 7, ch = pos ^ ((row >> 1) & 7);
            sa[v] = (const unsigned char *)(P.a + (long)min(m0 + row, a.M - 1) * a.lda + ch * 8);
        } else {
            constexpr int RB = BM * 2, kPos = RB / 16;
            const int row = (piece * 64 + lane) / kPos, pos = (piece * 64 + lane) % kPos;       // (the image's 16-byte chunks in order)
            const int ch = (((pos >> 1) ^ s_tr<RB>(row)) << 1) | (pos & 1);
            sa[v] = (const unsigned char *)(P.a + (long)row * a.lda + min(m0 + ch * 8, a.M - 8));
        }
    }
#pragma unroll
    for (int v = 0; v < kBV; ++v) {
        const int piece = v * 8 + wave;
        db[v] = kAImg + piece * 1024;
        if constexpr (!TB) {
            const int row = piece * 8 + (lane >> 3), pos = lane & 7, ch = pos ^ ((row >> 1) & 7);
            sb[v] = (const unsigned char *)(P.b + (long)min(n0 + row, a.N - 1) * a.ldb + ch * 8);
        } else {
            constexpr int RB = BN * 2, kPos = RB / 16;
            const int row = (piece * 64 + lane) / kPos, pos = (piece * 64 + lane) % kPos;
            const int ch = (((pos >> 1) ^ s_tr<RB>(row)) << 1) | (pos & 1);
            sb[v] = (const unsigned char *)(P.b + (long)row * a.ldb + min(n0 + ch * 8, a.N - 8));
        }
    }
    const long step_a = TA ? (long)kBK * a.lda * 2 : (long)kBK * 2, step_b = TB ? (long)kBK * a.ldb * 2 : (long)kBK * 2;   // bytes per k tile
    const uint32_t l0 = lds_addr(lds);
    int rq_stage = 0;
    auto request = [&]() __attribute__((always_inline)) {              // the next k tile -> the next stage; the sources move on by one k tile
        const uint32_t st = l0 + rq_stage * kStage;
#pragma unroll
        for (int v = 0; v < kAV; ++v) {
            dma16(sa[v], st + da[v]);
            sa[v] += step_a;
        }
#pragma unroll
        for (int v = 0; v < kBV; ++v) {
            dma16(sb[v], st + db[v]);
            sb[v] += step_b;
        }
        rq_stage = rq_stage + 1 == S ? 0 : rq_stage + 1;
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, g = lane >> 4;
    int cp_stage = 0;
    auto compute = [&](int) __attribute__((always_inline)) {
        const unsigned char *ia = lds + cp_stage * kStage, *ib = ia + kAImg;
        cp_stage = cp_stage + 1 == S ? 0 : cp_stage + 1;
#pragma unroll
        for (int ks = 0; ks < kBK / 32; ++ks) {
            bf16x8 fa[WM], fb[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row0 = wr * (BM / 4) + i * 16;
                if constexpr (!TA) fa[i] = ld_frag_rows(ia, row0 + r, ks * 4 + g);
                else fa[i] = ld_frag_tr<BM * 2>(ia, row0, ks * 32, lane);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col0 = wc * (BN / 2) + j * 16;
                if constexpr (!TB) fb[j] = ld_frag_rows(ib, col0 + r, ks * 4 + g);
                else fb[j] = ld_frag_tr<BN * 2>(ib, col0, ks * 32, lane);
            }
            // operands swapped: a lane then owns four consecutive output COLUMNS of one row (an 8-byte store)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };

    // ---- k loop: S - 1 k tiles in flight.  Iteration kt: this wave's pieces of tile kt have landed (all but the (S - 2) kPieces youngest
    // entries of its queue are done), barrier (every wave's have, and every wave is done reading the stage tile kt + S - 1 goes to -- the
    // one tile kt - 1 was read from), request tile kt + S - 1, multiply tile kt.  The host guarantees nk >= S - 1.
    QT_TG_STAMP(a, 0);
#pragma unroll
    for (int kt = 0; kt < S - 1; ++kt) request();
    int kt = 0;
    for (; kt + S - 1 < nk; ++kt) {
        wait_and_barrier<(S - 2) * kPieces>();
        if (kt == 0) QT_TG_STAMP(a, 1);                                 // the first k tile has landed
        request();
        compute(kt);
    }
    drain<S - 2, kPieces>(kt, compute);
    QT_TG_STAMP(a, 2);

    // ---- epilogue: lane (r, g) of tile (i, j) holds C[row0 + r][col0 + 4 g .. + 3]
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 16 + 4 * g;
        if (col >= a.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (P.bias) {
            const uint2 b = *(const uint2 *)(P.bias + col);
            bv[0] = qt_u2f(b.x << 16); bv[1] = qt_u2f(b.x & 0xFFFF0000u); bv[2] = qt_u2f(b.y << 16); bv[3] = qt_u2f(b.y & 0xFFFF0000u);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int row = m0 + wr * (BM / 4) + i * 16 + r;
            if (row < a.M)
                *(uint2 *)(P.c + (long)row * a.ldc + col) = uint2{pack_bf16x2(acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]),
                                                                 pack_bf16x2(acc[i][j][2] + bv[2], acc[i][j][3] + bv[3])};
        }
    }
    QT_TG_STAMP(a, 3);
}

template <bool TA, bool TB, int BM, int BN, int SS = 0>
__global__ __launch_bounds__(kThreads) void train_gemm_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    QT_TG_PICK(a, (int)blockIdx.x, P, tile)
    gemm_tile<TA, TB, BM, BN, SS>(P, QT_TG_DIMS(a), tile, lds);
}

// The two backward products of a Linear (or of query / key / value together) in ONE launch: the weight gradient's tiles first (their
// k range is the token count, the longest of the step), the input gradient's 128 x 64 tiles behind them.  Alone, each of the two leaves
// a quarter to a half of the CUs idle (144-192 workgroups of one partial wave) and pays its own launch, ring fill and drain; together the
// second product's workgroups start on the CUs the first one leaves free.  Same tiles, same k order as the separate launches: same bits.
template <int BMW, int BNW>
__global__ __launch_bounds__(kThreads) void train_gemm_backward_kernel(Args w, Args d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nw = w.count * w.tiles_m * w.tiles_n;
    if ((int)blockIdx.x < nw) {
        QT_TG_PICK(w, (int)blockIdx.x, P, tile)
        gemm_tile<true, true, BMW, BNW>(P, QT_TG_DIMS(w), tile, lds);
    } else {
        QT_TG_PICK(d, (int)blockIdx.x - nw, P, tile)
        gemm_tile<false, true, 128, 64>(P, QT_TG_DIMS(d), tile, lds);
    }
}

// ---- skinny forward products (a classifier head: [16, 768] -> [16, 2]).  hipBLASLt runs such a shape as a split-K kernel whose partial
// sums meet through atomics: the order of the additions, and now and then the last bit of a logit, changes from launch to launch -- the
// one GEMM of the training step that is not bit-reproducible (profiles/r06_graph_eager_determinism.txt).  One wave per output element,
// lanes stride over k, a fixed-order butterfly: deterministic, and at M N <= 4096 a few microseconds.
struct SkinnyArgs {
    const uint16_t *a, *b, *bias;
    uint16_t *c;
    int M, N, K;
    long lda, ldb, ldc;
};
__global__ __launch_bounds__(256) void train_gemm_skinny_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x & 63;
    const long out = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (out >= (long)a.M * a.N) return;                                 // (wave-uniform)
    const int m = (int)(out / a.N), n = (int)(out % a.N);
    const uint16_t *x = a.a + (long)m * a.lda, *w = a.b + (long)n * a.ldb;
    float acc = 0.0f;
    for (int k = lane * 8; k < a.K; k += 64 * 8) {
        const uint4 xv = *(const uint4 *)(x + k), wv = *(const uint4 *)(w + k);
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = fmaf(qt_u2f(xw[j] << 16), qt_u2f(ww[j] << 16), acc);
            acc = fmaf(qt_u2f(xw[j] & 0xFFFF0000u), qt_u2f(ww[j] & 0xFFFF0000u), acc);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) {
        if (a.bias) acc += qt_u2f((uint32_t)a.bias[n] << 16);
        a.c[(long)m * a.ldc + n] = (uint16_t)(pack_bf16x2(acc, 0.0f) & 0xFFFFu);
    }
}

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <bool TA, bool TB, int BM, int BN, int SS = 0>
int launch_tile(Args &a, hipStream_t st) {
    constexpr int kStages = SS ? SS : Ring<BM, BN>::kStages;
    constexpr int kLds = Ring<BM, BN>::kStage * kStages;
    static_assert(kLds <= 160 * 1024, "the ring does not fit a CU's LDS");
    if (a.K / kBK < kStages - 1) return QT_ERR_BAD_ARG;
    static QtOncePerDevice configured;
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)train_gemm_kernel<TA, TB, BM, BN, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    train_gemm_kernel<TA, TB, BM, BN, SS><<<a.count * a.tiles_m * a.tiles_n, kThreads, kLds, st>>>(a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

// Tile choice, a fixed rule of the layout, the problem sizes and the CU count (no timing: every process and rank cuts a shape the same
// way), from the sweep in profiles/r06_train_gemm.txt: every tile shape lands within 15 % of the others -- what paces these launches is
// the rate at which a CU's LDS-DMA requests are served (about 30-60 GB/s per CU), not the tile -- and the best of them is
//   k-contiguous A (forward, dgrad):  128 x 64 tiles (two workgroups per CU) where they give at least half a workgroup per CU, else 64 x 64;
//   transposed A (wgrad):             128 x 128 tiles under the same condition (3072 x 768: 144 tiles), else 64 x 64 (768 x 768: 144 tiles).
// A 256 x 128 tile and whole-CU rings of 6-8 stages were built and measured too: no faster (19.4 against 19.0 us at 2048 x 3072 x 768).
void pick_tile(const Args &a, bool trans_a, int &bm, int &bn) {
    const int cus = cu_count();
    const int big_m = 128, big_n = trans_a ? 128 : 64;
    const long tiles = (long)a.count * ((a.M + big_m - 1) / big_m) * ((a.N + big_n - 1) / big_n);
    if (tiles * 2 >= cus) { bm = big_m; bn = big_n; }
    else { bm = 64; bn = 64; }
}

template <bool TA, bool TB>
int launch(Args &a, hipStream_t st, int force_bm, int force_bn) {
    int bm, bn;
    pick_tile(a, TA, bm, bn);
    if (force_bm) bm = force_bm;
    if (force_bn) bn = force_bn;
    if constexpr (!TA && !TB) {
        // Forward products wide enough for it: 128 x 192 tiles.  The timeline of a launch (tools/exp_train_gemm_stamps.py) shows what paces
        // these kernels: a CU moves its operands at 85-100 GB/s whatever the tile, and 768 tiles of 128 x 64 are three per CU -- two
        // together, then one alone at a third of the rate.  128 x 192 tiles are one per CU (2048 x 3072: 256; q / k / v together: 192)
        // and need 491 KB per CU instead of 885.
        const int cus = cu_count();
        const long tm = (a.M + 127) / 128;
        const long t64 = (long)a.count * tm * ((a.N + 63) / 64), t192 = (long)a.count * tm * (a.N / 192);
        if (!force_bm && !force_bn && a.N % 192 == 0 && t64 > 2L * cus && t192 * 10 >= 7L * cus) return launch_tile<false, false, 128, 192>(a, st);
        if (force_bm == 128 && force_bn == 192) return launch_tile<false, false, 128, 192>(a, st);
        // One 128 x 64 tile per CU or fewer and a long contraction (2048 x 768 x 3072: 192 tiles, 48 k tiles): the workgroup is alone on
        // its CU and the LDS holds six stages.  Measured (profiles/r06_train_gemm_stamps.txt): 23.5 -> 22.5 us at K = 3072, 7.1 -> 6.9 at K = 768
        // -- five k tiles in flight instead of two shorten the k tile from 0.45 to 0.41 us only: what paces a lone workgroup is not the
        // requests in flight (nor the order of request and multiplications in a k tile, nor the LDS latency: both tried).
        int deep = (bm == 128 && bn == 64 && (long)a.count * ((a.M + 127) / 128) * ((a.N + 63) / 64) <= cus && a.K / kBK >= 16) ? 1 : 0;
#ifdef QT_TUNING_BUILD
        if (const char *e = getenv("QT_TRAIN_GEMM_DEEP")) deep = atoi(e);
#endif
        if (deep && bm == 128 && bn == 64) return launch_tile<false, false, 128, 64, 6>(a, st);
    }
    if (bm == 128 && bn == 128) return launch_tile<TA, TB, 128, 128>(a, st);
    if (bm == 128) return launch_tile<TA, TB, 128, 64>(a, st);
    if (bn == 128) return launch_tile<TA, TB, 64, 128>(a, st);
    return launch_tile<TA, TB, 64, 64>(a, st);
}

template <int BMW, int BNW>
int launch_backward(Args &w, Args &d, hipStream_t st) {
    constexpr int kLds = Ring<BMW, BNW>::kBytes > Ring<128, 64>::kBytes ? Ring<BMW, BNW>::kBytes : Ring<128, 64>::kBytes;
    if (w.K / kBK < Ring<BMW, BNW>::kStages - 1 || d.K / kBK < Ring<128, 64>::kStages - 1) return QT_ERR_BAD_ARG;
    static QtOncePerDevice configured;
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)train_gemm_backward_kernel<BMW, BNW>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    w.tiles_m = (w.M + BMW - 1) / BMW;
    w.tiles_n = (w.N + BNW - 1) / BNW;
    d.tiles_m = (d.M + 127) / 128;
    d.tiles_n = (d.N + 63) / 64;
    const long grid = (long)w.count * w.tiles_m * w.tiles_n + (long)d.count * d.tiles_m * d.tiles_n;
    train_gemm_backward_kernel<BMW, BNW><<<(unsigned)grid, kThreads, kLds, st>>>(w, d);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

}  // namespace

extern "C" {

int qt_train_gemm_backward_bf16(const qt_linear_backward *items, int count, int T, int O, int I, long ld_gy, long ld_w, long ld_x, long ld_gx,
                                long ld_gw, void *stream) {
    if (!items || count < 1 || count > kMaxProblems || T < 0 || O < 0 || I < 0) return QT_ERR_BAD_ARG;
    if ((long)T * O * I == 0) return QT_ERR_BAD_ARG;                   // (an empty product: the caller's single launches handle it)
    // dgrad: gx [T][I] = gy [T][O] . Wq [O][I] -- k = O;   wgrad: gW [O][I] = gy^T [O][T] . x [T][I] -- k = T
    if (O < 4 * kBK || O % kBK != 0 || T < 4 * kBK || T % kBK != 0 || I % 8 != 0 || I < 8 || T % 8 != 0 || O % 8 != 0) return QT_ERR_BAD_ARG;
    if (ld_gy % 8 != 0 || ld_w % 8 != 0 || ld_x % 8 != 0 || ld_gx % 4 != 0 || ld_gw % 4 != 0) return QT_ERR_BAD_ARG;
    Args w{}, d{};
    for (int i = 0; i < count; ++i) {
        const qt_linear_backward &p = items[i];
        if (!p.gy || !p.wq || !p.x || !p.gx || !p.gw) return QT_ERR_BAD_ARG;
        if (((uintptr_t)p.gy | (uintptr_t)p.wq | (uintptr_t)p.x) & 15u) return QT_ERR_UNALIGNED;
        if (((uintptr_t)p.gx | (uintptr_t)p.gw) & 7u) return QT_ERR_UNALIGNED;
        d.p[i] = Problem{p.gy, p.wq, nullptr, p.gx};
        w.p[i] = Problem{p.gy, p.x, nullptr, p.gw};
    }
    d.count = w.count = count;
    d.M = T; d.N = I; d.K = O; d.lda = ld_gy; d.ldb = ld_w; d.ldc = ld_gx;
    w.M = O; w.N = I; w.K = T; w.lda = ld_gy; w.ldb = ld_x; w.ldc = ld_gw;
    int bm, bn;
    pick_tile(w, true, bm, bn);                                        // the weight gradient's tile by the rule of the single launches
    hipStream_t st = (hipStream_t)stream;
    if (bm != 128) return launch_backward<64, 64>(w, d, st);
    // Beside the input gradient's 128 x 64 tiles the weight gradient takes 128 x 64 tiles too: their ring leaves room for a second
    // workgroup on the CU.  A 128 x 128 tile's ring fills the LDS -- the CUs that hold one take no other tile until it is done: gW [3072][768]
    // + gx [2048][768] (k = 3072) 43.2 -> 34.9 us, the step 5.66 -> 5.57 ms.  Also measured for gW [768][3072] + gx [2048][3072]: 128 x 128
    // beside 128 x 192 input-gradient tiles (256 instead of 768 of them) 5.564-5.570 ms against 5.536-5.560 with everything 64 wide.
    return launch_backward<128, 64>(w, d, st);
}

int qt_train_gemm_bf16(const qt_gemm_problem *problems, int count, int trans_a, int trans_b, int M, int N, int K, long lda, long ldb, long ldc,
                       void *stream) {
    if (!problems || count < 1 || count > kMaxProblems || M < 0 || N < 0 || K < 0) return QT_ERR_BAD_ARG;
    if ((long)M * N == 0) return QT_OK;
    if (!trans_a && !trans_b && count == 1 && (N < 8 || N % 8 != 0) && (long)M * N <= 4096) {
        // a skinny forward product (classifier heads): the deterministic one-wave-per-output kernel
        const qt_gemm_problem &p = problems[0];
        if (!p.a || !p.b || !p.c || K < 8 || K % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0) return QT_ERR_BAD_ARG;
        if (((uintptr_t)p.a | (uintptr_t)p.b) & 15u) return QT_ERR_UNALIGNED;
        if (((uintptr_t)p.c | (uintptr_t)p.bias) & 1u) return QT_ERR_UNALIGNED;
        const SkinnyArgs sa{p.a, p.b, p.bias, p.c, M, N, K, lda, ldb, ldc};
        train_gemm_skinny_kernel<<<(unsigned)(((long)M * N + 3) / 4), 256, 0, (hipStream_t)stream>>>(sa);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? QT_OK : (int)e;
    }
    // (what the package's callers check before they come here; anything else keeps the library GEMM)
    if (K < 4 * kBK || K % kBK != 0 || M % 8 != 0 || N % 8 != 0 || M < 8 || N < 8 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0) return QT_ERR_BAD_ARG;
    Args a{};
    for (int i = 0; i < count; ++i) {
        const qt_gemm_problem &p = problems[i];
        if (!p.a || !p.b || !p.c) return QT_ERR_BAD_ARG;
        if (((uintptr_t)p.a | (uintptr_t)p.b) & 15u) return QT_ERR_UNALIGNED;
        if (((uintptr_t)p.c | (uintptr_t)p.bias) & 7u) return QT_ERR_UNALIGNED;
        a.p[i] = Problem{p.a, p.b, p.bias, p.c};
    }
    a.count = count; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    int fbm = 0, fbn = 0;
#ifdef QT_TUNING_BUILD
    if (const char *e = getenv("QT_TG_STAMPS")) a.dbg = (unsigned long long *)strtoull(e, nullptr, 0);   // tools/ only
    if (const char *e = getenv("QT_TRAIN_GEMM_TILE")) {             // tools/ only: "128x64"
        if (sscanf(e, "%dx%d", &fbm, &fbn) != 2 || (fbm != 64 && fbm != 128) || (fbn != 64 && fbn != 128 && fbn != 192)) fbm = fbn = 0;
    }
#endif
    hipStream_t st = (hipStream_t)stream;
    if (!trans_a && !trans_b) return launch<false, false>(a, st, fbm, fbn);
    if (!trans_a && trans_b) return launch<false, true>(a, st, fbm, fbn);
    if (trans_a && trans_b) return launch<true, true>(a, st, fbm, fbn);
    return launch<true, false>(a, st, fbm, fbn);
}

}  // extern "C"
