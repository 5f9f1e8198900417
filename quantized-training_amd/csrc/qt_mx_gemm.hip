// Block-scaled (microscaling) GEMMs on gfx950's scaled matrix instruction.
//
// Replaces, for operands whose element format and scales the hardware takes natively,
//     linear_mx / matmul_mx          decomposed.py:304-363
//         x = x * expand(input_scale, x.shape, block_size); w = w * expand(weight_scale, ...); F.linear / matmul
// i.e. two dequantize passes over the operands plus a bf16 GEMM, by v_mfma_scale_f32_16x16x128_f8f6f4 reading the
// 8 / 6 / 4-bit element codes and one E8M0 scale per 32 elements directly.
//
// Operand memory layout ("packed MX operand"): codes[rows][K * bits / 8] row-major, K contiguous, element i of a row
// in bits [i*bits, (i+1)*bits) little-endian; scales[rows][K / 32] E8M0 bytes (2^(byte - 127)).
// Fragment layout of the instruction (measured, tools/probe_mx_mfma*.hip): lane l = (row r = l & 15, group g = l >> 4)
//   fp6 / fp4: k = 32 g + [0, 32) of the 128-deep step;   fp8: bytes 0-15 <- k = 16 g + [0, 16), bytes 16-31 <- k = 64 + 16 g + [0, 16)
//   scale register byte 0 of lane (r, g) = scale of row r, k in [32 g, 32 g + 32);   D: col = l & 15, row = 4 (l >> 4) + i.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/qt_hip.h"
#include "qt_device.h"
#include "qt_formats.h"
#include "qt_mx.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kBM = 128, kBN = 128, kBK = 128;      // block tile (elements)

__host__ __device__ constexpr int tile_row_bytes(int f) { return f < 2 ? 128 : (f < 4 ? 96 : 64); }
using qt_mx::elem_bits;
using qt_mx::encode_elem;

struct MxGemmArgs {
    const uint8_t *A, *B;        // packed codes [batch][M][K*bitsA/8], [batch][N][K*bitsB/8]
    const uint8_t *sA, *sB;      // E8M0 [batch][M][K/32], [batch][N][K/32]
    void *C;                     // [batch][M][N] bf16 or f32
    const void *bias;            // [N] in C's dtype, or NULL
    int M, N, K;
    long bA, bB, bsA, bsB, bC;   // batch strides: bytes for A / B / scales, elements for C
    int out_f32;
    const void *out_scale;       // I8 kernels: dequantize factor(s) in C's dtype applied after the rounding of (acc + bias), or NULL
    int out_scale_per_col;       // 0: one factor, 1: one per output column
    int out_fold_f32;            // fp32 output: pass (acc + bias) through the identity value map first (hi16 | sticky, decomposed.py:151-153)
};

typedef int v4i __attribute__((ext_vector_type(4)));

// One 128-deep step of a 16 x 16 output tile.  I8: both operands are int8 codes (same tile layout as the 8-bit float
// formats: a lane's bytes 0-15 are k = 16 g + [0, 16) of the first 64-deep half, bytes 16-31 the same of the second), two
// v_mfma_i32_16x16x64_i8 with exact int32 accumulation; otherwise the block-scaled float instruction.
template <int FA, int FB, bool I8>
struct Mma {
    using acc_t = std::conditional_t<I8, v4i, v4f>;
    static __device__ __forceinline__ acc_t zero() {
        if constexpr (I8) return v4i{0, 0, 0, 0};
        else return v4f{0.f, 0.f, 0.f, 0.f};
    }
    static __device__ __forceinline__ acc_t run(const v8i &a, const v8i &b, acc_t c, int sa, int sb) {
        if constexpr (I8) {
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{a[0], a[1], a[2], a[3]}, v4i{b[0], b[1], b[2], b[3]}, c, 0, 0, 0);
            return __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{a[4], a[5], a[6], a[7]}, v4i{b[4], b[5], b[6], b[7]}, c, 0, 0, 0);
        } else {
            return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, sa, 0, sb);
        }
    }
};

// (acc + bias) rounded to the output dtype, then -- converted per-tensor graphs, quantize_pt2e.py:323-446: the
// dequantize(s_x * s_w) node behind the GEMM -- multiplied by the dequantize factor and rounded again.
__device__ __forceinline__ void emit_out(const MxGemmArgs &a, float accv, float bv, float sv, long idx) {
    float v = accv + bv;
    if (a.out_f32) {
        if (a.out_fold_f32) v = qt_u2f(qt_fold_img(qt_f2u(v)));
        if (a.out_scale) v = v * sv;
        ((float *)a.C)[idx] = v;
    } else {
        uint16_t b = qt_f2bf(v);
        if (a.out_scale) b = qt_f2bf(qt_bf2f(b) * sv);
        ((uint16_t *)a.C)[idx] = b;
    }
}
__device__ __forceinline__ float out_scale_of(const MxGemmArgs &a, int col) {
    if (!a.out_scale) return 1.0f;
    const long i = a.out_scale_per_col ? col : 0;
    return a.out_f32 ? ((const float *)a.out_scale)[i] : qt_bf2f(((const uint16_t *)a.out_scale)[i]);
}

// LDS image of a 128-row operand tile.  8- and 4-bit tiles are unpadded with the 16-byte chunk index XOR-swizzled by
// the row so that the four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ... : sixteen different rows, half of
// them one chunk further along k) each touch sixteen different 16-byte slots of the 256-byte bank row; 6-bit tiles
// (96-byte rows, read with three ds_read_b64) are stored in groups of 8 rows (768 bytes) followed by 16 bytes of
// padding: a fragment's rows r and r + 8 then sit 2 eight-byte slots apart modulo the 32-slot bank row, which makes the
// two 32-lane halves of ds_read_b64 conflict-free (unpadded 96-byte rows collide pairwise: 12 r mod 32 has period 8),
// and one 48-lane LDS-DMA instruction fills exactly one group.
template <int F>
struct Tile {
    static constexpr int kRow = tile_row_bytes(F);                  // bytes of one row of a 128-deep tile
    static constexpr bool kSix = (F == 2 || F == 3);
    static constexpr int kGroup = 8 * kRow + 16;                    // 6-bit tiles: 8 rows + pad
    static constexpr int kChunks = kRow / 16;
    static constexpr int kBytes = kSix ? 16 * kGroup : 128 * kRow;
    static __device__ __forceinline__ int row_off(int row) {
        if constexpr (kSix) return (row >> 3) * kGroup + (row & 7) * kRow;
        else return row * kRow;
    }
    static __device__ __forceinline__ int chunk_off(int row, int chunk) {
        if constexpr (F < 2) return row * kRow + ((chunk ^ ((row >> 1) & 7)) << 4);
        else if constexpr (F < 4) return row_off(row) + (chunk << 4);
        else return row * kRow + ((chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) << 4);     // f = [0,2,3,1], see below
    }
};
// fp4 swizzle: rows r and r' of one lane group with r & 3 == r' & 3 collide when f(r >> 2) ^ f(r' >> 2) == 1 for a
// row pair that sits on different k chunks; those pairs are (0|3, 1|2), so f must pair 0 with 3 and 1 with 2:
// f = [0, 2, 3, 1] (packed two bits each, low first, in 0x78).

template <int F>
__device__ __forceinline__ v8i read_frag(const uint8_t *tile, int row, int g) {
    v8i f = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (F < 2) {
        const uint4 lo = *(const uint4 *)(tile + Tile<F>::chunk_off(row, g));
        const uint4 hi = *(const uint4 *)(tile + Tile<F>::chunk_off(row, 4 + g));
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    } else if constexpr (F < 4) {
        const uint8_t *p = tile + Tile<F>::row_off(row) + 24 * g;
        const uint2 a = *(const uint2 *)(p), b = *(const uint2 *)(p + 8), c = *(const uint2 *)(p + 16);
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y;
    } else {
        const uint4 lo = *(const uint4 *)(tile + Tile<F>::chunk_off(row, g));
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
    }
    return f;
}

template <int FA, int FB, bool I8 = false>
__global__ __launch_bounds__(256) void mx_gemm_kernel(MxGemmArgs a) {
    using M_ = Mma<FA, FB, I8>;
    using TA = Tile<FA>;
    using TB = Tile<FB>;
    constexpr int RA = TA::kRow, RB = TB::kRow;
    constexpr int CA = TA::kChunks, CB = TB::kChunks;            // 16-byte chunks per tile row
    constexpr int NA = kBM * CA / 256, NB = kBN * CB / 256;       // chunks per thread
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    auto s_a = [&](int buf) __attribute__((always_inline)) { return lds + buf * TA::kBytes; };
    auto s_b = [&](int buf) __attribute__((always_inline)) { return lds + 2 * TA::kBytes + buf * TB::kBytes; };

    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int r = l & 15, g = l >> 4;
    const int wm = w >> 1, wn = w & 1;
    // Workgroup ids go round-robin over the 8 XCDs; give each XCD a contiguous run of tiles, and walk the tiles in
    // groups of 8 M-tiles per N-tile column so that a run re-uses its A and B panels out of that XCD's L2.
    const int tiles_m = (a.M + kBM - 1) / kBM, tiles_n = (a.N + kBN - 1) / kBN;
    const int ntiles = tiles_m * tiles_n;
    int id = blockIdx.x;
    {
        const int per = ntiles / 8, rem = ntiles % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;                  // XCD x owns [x*per + min(x, rem), ...)
    }
    constexpr int kGroup = 8;
    const int width = kGroup * tiles_n, grp = id / width, first_m = grp * kGroup;
    const int gsize = min(tiles_m - first_m, kGroup);
    const int m0 = (first_m + (id % width) % gsize) * kBM, n0 = ((id % width) / gsize) * kBN;
    const long bz = blockIdx.y;
    const long kbA = (long)a.K * elem_bits(FA) / 8, kbB = (long)a.K * elem_bits(FB) / 8;   // bytes per row
    const int nblk = a.K / 32;
    const uint8_t *Ab = a.A + bz * a.bA, *Bb = a.B + bz * a.bB;
    const uint8_t *sAb = a.sA + bz * a.bsA, *sBb = a.sB + bz * a.bsB;
    const int nk = (a.K + kBK - 1) / kBK;

    // per-thread staging geometry (fixed over the k loop); rows beyond M / N re-read the last row (never stored)
    const uint8_t *ga[NA], *gb[NB];
    int la[NA], lb[NB], ca[NA], cb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = t + 256 * i, row = c / CA, cc = c % CA;
        ga[i] = Ab + (long)min(m0 + row, a.M - 1) * kbA + cc * 16;
        la[i] = TA::chunk_off(row, cc);
        ca[i] = cc * 16;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int c = t + 256 * i, row = c / CB, cc = c % CB;
        gb[i] = Bb + (long)min(n0 + row, a.N - 1) * kbB + cc * 16;
        lb[i] = TB::chunk_off(row, cc);
        cb[i] = cc * 16;
    }
    const uint8_t *psa[4], *psb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        psa[i] = sAb + (long)min(m0 + wm * 64 + i * 16 + r, a.M - 1) * nblk;
        psb[i] = sBb + (long)min(n0 + wn * 64 + i * 16 + r, a.N - 1) * nblk;
    }

    uint4 ra[NA], rb[NB];
    int sca[4], scb[4];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        // the k tail (K % 128 != 0) re-reads chunk 0 of the row and is zeroed when it is stored
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const long off = (long)kt * RA + ca[i];
            ra[i] = *(const uint4 *)(ga[i] + (off < kbA ? (long)kt * RA : -(long)ca[i]));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const long off = (long)kt * RB + cb[i];
            rb[i] = *(const uint4 *)(gb[i] + (off < kbB ? (long)kt * RB : -(long)cb[i]));
        }
        const int kb = min(kt * 4 + g, nblk - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sca[i] = psa[i][kb];
            scb[i] = psb[i][kb];
        }
    };
    auto store_tile = [&](int buf, int kt) __attribute__((always_inline)) {
        // branch-free so that the k loop stays one basic block (the scheduling fences below only act inside one)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool live = (long)kt * RA + ca[i] < kbA;
            uint4 v = ra[i];
            v.x = live ? v.x : 0u; v.y = live ? v.y : 0u; v.z = live ? v.z : 0u; v.w = live ? v.w : 0u;
            *(uint4 *)(s_a(buf) + la[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool live = (long)kt * RB + cb[i] < kbB;
            uint4 v = rb[i];
            v.x = live ? v.x : 0u; v.y = live ? v.y : 0u; v.z = live ? v.z : 0u; v.w = live ? v.w : 0u;
            *(uint4 *)(s_b(buf) + lb[i]) = v;
        }
    };

    typename M_::acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = M_::zero();

    load_tile(0);
    store_tile(0, 0);
    int cs_a[4], cs_b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { cs_a[i] = sca[i]; cs_b[i] = scb[i]; }
    __syncthreads();
    auto compute = [&](int cur) __attribute__((always_inline)) {
        v8i fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag<FA>(s_a(cur), wm * 64 + i * 16 + r, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<FB>(s_b(cur), wn * 64 + j * 16 + r, g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = M_::run(fa[i], fb[j], acc[i][j], cs_a[i], cs_b[j]);
    };
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
        load_tile(kt + 1);                                // in flight during this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);                // keep the LDS refill (and its wait on the loads) behind the MFMAs
        store_tile(cur ^ 1, kt + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { cs_a[i] = sca[i]; cs_b[i] = scb[i]; }
        __syncthreads();
    }
    compute((nk - 1) & 1);

    // epilogue: D col = l & 15, row = 4 g + e
    const long cbase = bz * a.bC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + r;
        if (col >= a.N) continue;
        float bv = 0.f;
        if (a.bias) bv = a.out_f32 ? ((const float *)a.bias)[col] : qt_bf2f(((const uint16_t *)a.bias)[col]);
        const float sv = out_scale_of(a, col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 64 + i * 16 + 4 * g + e;
                if (row >= a.M) continue;
                emit_out(a, (float)acc[i][j][e], bv, sv, cbase + (long)row * a.N + col);
            }
        }
    }
}

// ---- LDS-DMA variant: 8- and 4-bit operands, K % 128 == 0 -------------------------------------------------
// Tiles and their scale bytes go global -> LDS with global_load_lds (no staging registers, so three workgroups fit
// per CU and cover each other's load latency); the XOR swizzle is applied on the per-lane SOURCE address because
// the DMA writes a wave's 64 x 16 bytes linearly.  One 128-deep tile per step, two barriers per step.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int F>
struct DmaTile {                                    // one wave-instruction = kRows rows of the tile
    static constexpr int kRow = tile_row_bytes(F), kChunks = kRow / 16;
    static constexpr int kRows = Tile<F>::kSix ? 8 : 64 / kChunks;     // fp8: 8 rows, fp6: 8 rows (48 lanes), fp4: 16 rows
    static constexpr int kLanes = kRows * kChunks;                      // active lanes of the instruction
    static constexpr int kStride = Tile<F>::kSix ? Tile<F>::kGroup : 1024;   // LDS bytes from one instruction's block to the next
    static constexpr int kInstr = 128 / kRows / 4;  // wave-instructions per wave per tile (4 for fp8 / fp6, 2 for fp4)
};

// Fragment reads for the prefetch ring, as inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every ds_read it can
// see while an LDS-DMA is in flight (the DMA's LDS store carries no alias scope, so nothing can be proven about it),
// which would drain the ring on every step.  An asm ds_read has no memory operand for that rule to act on; the price is
// that the lgkmcnt wait before the first use is ours to place (one s_waitcnt lgkmcnt(0) after the batch of reads).
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ uint4 asm_ds_read_b128(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ uint2 asm_ds_read_b64(uint32_t addr) {
    uint2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ int asm_ds_read_u8(uint32_t addr) {
    int v;
    asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// Immediate-offset forms: within a wave's column of 16-row fragment tiles the swizzle depends on the lane only, so
// tile i is the lane's base address + i * (16 rows), a compile-time offset -- one address register per operand
// instead of one per read.
template <int OFF>
__device__ __forceinline__ uint4 asm_ds_read_b128_off(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));      // native vector: usable as a "+v" asm operand
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x4w ds_read128w(uint32_t addr) {
    u32x4w v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ int asm_ds_read_u8_off(uint32_t addr) {
    int v;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ uint2 asm_ds_read_b64_off(uint32_t addr) {
    uint2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// bytes from one 16-row fragment tile to the next in the LDS image of format F (6-bit: two 8-row groups)
template <int F>
constexpr int frag_step() { return Tile<F>::kSix ? 2 * Tile<F>::kGroup : 16 * Tile<F>::kRow; }
template <int F, int I>
__device__ __forceinline__ v8i read_frag_imm(uint32_t lane_lo, uint32_t lane_hi) {
    v8i f = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (Tile<F>::kSix) {                       // lane_lo = row offset + 24 g; three 8-byte reads
        const uint2 a = asm_ds_read_b64_off<I * frag_step<F>()>(lane_lo);
        const uint2 b = asm_ds_read_b64_off<I * frag_step<F>() + 8>(lane_lo);
        const uint2 c = asm_ds_read_b64_off<I * frag_step<F>() + 16>(lane_lo);
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y;
        return f;
    }
    const uint4 lo = asm_ds_read_b128_off<I * frag_step<F>()>(lane_lo);
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
    if constexpr (F < 2) {
        const uint4 hi = asm_ds_read_b128_off<I * frag_step<F>()>(lane_hi);
        f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    }
    return f;
}

template <int F>
__device__ __forceinline__ v8i read_frag_asm(uint32_t tile, int row, int g) {
    v8i f = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (F < 2) {
        const uint4 lo = asm_ds_read_b128(tile + Tile<F>::chunk_off(row, g));
        const uint4 hi = asm_ds_read_b128(tile + Tile<F>::chunk_off(row, 4 + g));
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    } else if constexpr (F < 4) {
        const uint32_t p = tile + Tile<F>::row_off(row) + 24 * g;
        const uint2 a = asm_ds_read_b64(p), b = asm_ds_read_b64(p + 8), c = asm_ds_read_b64(p + 16);
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y;
    } else {
        const uint4 lo = asm_ds_read_b128(tile + Tile<F>::chunk_off(row, g));
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
    }
    return f;
}

// which source chunk of `row` belongs in LDS slot `cs` of that row (the XOR swizzles are involutions)
template <int F>
__device__ __forceinline__ int swz_chunk(int row, int cs) {
    return (Tile<F>::chunk_off(row, cs) - Tile<F>::row_off(row)) >> 4;
}

// One __shared__ object per ring stage: hipcc orders a ds_read after an in-flight LDS-DMA with s_waitcnt vmcnt(0) unless it
// can prove they touch different objects (distinct LDS globals carry distinct alias scopes; offsets into one array do
// not), which would drain the prefetch ring on every step.
template <int S, int BYTES>
__device__ __forceinline__ uint8_t *ring_stage() {
    __shared__ __attribute__((aligned(16))) uint8_t buf[BYTES];
    return buf;
}
template <int BYTES>
__device__ __forceinline__ uint8_t *ring_stage_of(int s) {
    switch (s) {                     // s is a compile-time constant wherever this is called (unrolled ring)
        case 0: return ring_stage<0, BYTES>();
        case 1: return ring_stage<1, BYTES>();
        case 2: return ring_stage<2, BYTES>();
        case 3: return ring_stage<3, BYTES>();
        case 4: return ring_stage<4, BYTES>();
        case 5: return ring_stage<5, BYTES>();
        case 6: return ring_stage<6, BYTES>();
        default: return ring_stage<7, BYTES>();
    }
}

// STAGES == 1: one LDS stage, two barriers per step -- relies on >= 3 resident workgroups per CU covering each other's
// load latency (large grids).  STAGES > 1: a ring of LDS stages with the DMA of the next STAGES - 1 tiles in flight
// while a tile is multiplied; the wait is a counted s_waitcnt vmcnt(pieces still allowed in flight) followed by a raw
// s_barrier (a __syncthreads() would drain the prefetch) -- for grids of about one workgroup per CU, where a tile's
// ~1.5 us load latency is otherwise fully exposed.
template <int FA, int FB, int STAGES, bool I8 = false>
__global__ __launch_bounds__(256, STAGES == 1 ? 3 : (STAGES == 2 ? 2 : 1)) void mx_gemm_dma_kernel(MxGemmArgs a) {
    using M_ = Mma<FA, FB, I8>;
    using TA = Tile<FA>;
    using TB = Tile<FB>;
    using DA = DmaTile<FA>;
    using DB = DmaTile<FB>;
    constexpr int kStage = TA::kBytes + TB::kBytes + 1024;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // STAGES == 1 only (dynamic); rings use ring_stage<>

    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int r = l & 15, g = l >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_m = (a.M + kBM - 1) / kBM, tiles_n = (a.N + kBN - 1) / kBN;
    const int ntiles = tiles_m * tiles_n;
    int id = blockIdx.x;
    {
        const int per = ntiles / 8, rem = ntiles % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;
    }
    constexpr int kGroup = 8;
    const int width = kGroup * tiles_n, grp = id / width, first_m = grp * kGroup;
    const int gsize = min(tiles_m - first_m, kGroup);
    const int m0 = (first_m + (id % width) % gsize) * kBM, n0 = ((id % width) / gsize) * kBN;
    const long bz = blockIdx.y;
    const long kbA = (long)a.K * elem_bits(FA) / 8, kbB = (long)a.K * elem_bits(FB) / 8;
    const int nblk = a.K / 32, nk = a.K / kBK;

    // source addresses of this lane's DMA pieces (k tile 0); LDS destinations are wave-uniform
    const uint8_t *ga[DA::kInstr], *gb[DB::kInstr];
#pragma unroll
    for (int i = 0; i < DA::kInstr; ++i) {
        const int ll = min(l, DA::kLanes - 1);
        const int row = (w * DA::kInstr + i) * DA::kRows + ll / DA::kChunks, cs = ll % DA::kChunks;
        ga[i] = a.A + bz * a.bA + (long)min(m0 + row, a.M - 1) * kbA + swz_chunk<FA>(row, cs) * 16;
    }
#pragma unroll
    for (int i = 0; i < DB::kInstr; ++i) {
        const int ll = min(l, DB::kLanes - 1);
        const int row = (w * DB::kInstr + i) * DB::kRows + ll / DB::kChunks, cs = ll % DB::kChunks;
        gb[i] = a.B + bz * a.bB + (long)min(n0 + row, a.N - 1) * kbB + swz_chunk<FB>(row, cs) * 16;
    }
    // scale bytes of the tile: 128 rows x 4 bytes per operand; waves 0,1 fetch A's, waves 2,3 fetch B's
    const int srow = (w & 1) * 64 + l;
    const uint8_t *gs = w < 2 ? a.sA + bz * a.bsA + (long)min(m0 + srow, a.M - 1) * nblk
                              : a.sB + bz * a.bsB + (long)min(n0 + srow, a.N - 1) * nblk;

    typename M_::acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = M_::zero();

    auto issue = [&](int kt, uint8_t *stage) __attribute__((always_inline)) {
        uint8_t *const s_a = stage, *const s_b = s_a + TA::kBytes, *const s_s = s_b + TB::kBytes;
#pragma unroll
        for (int i = 0; i < DA::kInstr; ++i)
            if (DA::kLanes == 64 || l < DA::kLanes)
                __builtin_amdgcn_global_load_lds((glb_void *)(ga[i] + (long)kt * TA::kRow),
                                                 (lds_void *)(s_a + (w * DA::kInstr + i) * DA::kStride), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < DB::kInstr; ++i)
            if (DB::kLanes == 64 || l < DB::kLanes)
                __builtin_amdgcn_global_load_lds((glb_void *)(gb[i] + (long)kt * TB::kRow),
                                                 (lds_void *)(s_b + (w * DB::kInstr + i) * DB::kStride), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(gs + kt * 4), (lds_void *)(s_s + (w < 2 ? 0 : 512) + (w & 1) * 256), 4, 0, 0);
    };
    auto compute = [&](const uint8_t *stage) __attribute__((always_inline)) {
        const uint8_t *const s_a = stage, *const s_b = s_a + TA::kBytes, *const s_sa = s_b + TB::kBytes,
                      *const s_sb = s_sa + 512;
        int sa[4], sb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = s_sa[(wm * 64 + i * 16 + r) * 4 + g];
            sb[i] = s_sb[(wn * 64 + i * 16 + r) * 4 + g];
        }
        v8i fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag<FA>(s_a, wm * 64 + i * 16 + r, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<FB>(s_b, wn * 64 + j * 16 + r, g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = M_::run(fa[i], fb[j], acc[i][j], sa[i], sb[j]);
    };
    auto compute_ring = [&](const uint8_t *stage) __attribute__((always_inline)) {
        const uint32_t s_a = lds_addr(stage), s_b = s_a + TA::kBytes, s_sa = s_b + TB::kBytes, s_sb = s_sa + 512;
        int sa[4], sb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = asm_ds_read_u8(s_sa + (wm * 64 + i * 16 + r) * 4 + g);
            sb[i] = asm_ds_read_u8(s_sb + (wn * 64 + i * 16 + r) * 4 + g);
        }
        v8i fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag_asm<FA>(s_a, wm * 64 + i * 16 + r, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag_asm<FB>(s_b, wn * 64 + j * 16 + r, g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);               // no MFMA above the wait: the asm results are not tracked
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = M_::run(fa[i], fb[j], acc[i][j], sa[i], sb[j]);
    };
    if constexpr (STAGES == 1) {
        for (int kt = 0; kt < nk; ++kt) {
            issue(kt, lds);
            __syncthreads();
            compute(lds);
            __syncthreads();
        }
    } else {
        // ring of STAGES tiles: tiles kt+1 .. kt+STAGES-1 are in flight while tile kt is multiplied
        constexpr int kPieces = DA::kInstr + DB::kInstr + 1;         // DMA instructions one wave issues per stage
#pragma unroll
        for (int p = 0; p < STAGES - 1; ++p)
            if (p < nk) issue(p, ring_stage_of<kStage>(p));
        for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
            for (int sidx = 0; sidx < STAGES; ++sidx) {              // unrolled: every stage is a compile-time object
                const int kt = kt0 + sidx;
                if (kt < nk) {
                    const int ahead = kt + STAGES - 1;
                    if (ahead < nk) {
                        issue(ahead, ring_stage_of<kStage>((sidx + STAGES - 1) % STAGES));
                        // tile kt's pieces are the oldest: wait until only the (STAGES - 1) younger tiles' pieces remain
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * kPieces) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // pipeline drain: the last STAGES - 1 tiles
                    }
                    __builtin_amdgcn_s_barrier();                    // every wave's pieces of tile kt have landed
                    compute_ring(ring_stage_of<kStage>(sidx));
                    __builtin_amdgcn_s_barrier();                    // this stage may be refilled by the next issue
                }
            }
        }
    }

    const long cbase = bz * a.bC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + r;
        if (col >= a.N) continue;
        float bv = 0.f;
        if (a.bias) bv = a.out_f32 ? ((const float *)a.bias)[col] : qt_bf2f(((const uint16_t *)a.bias)[col]);
        const float sv = out_scale_of(a, col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 64 + i * 16 + 4 * g + e;
                if (row >= a.M) continue;
                emit_out(a, (float)acc[i][j][e], bv, sv, cbase + (long)row * a.N + col);
            }
        }
    }
}

// ---- 256 x 256 tiles for large GEMMs --------------------------------------------------------------------------
// At 128 x 128 the 8192^3 GEMMs are bound by operand re-reads through L2 (~12.5 TB/s); a 256 x 256 tile halves them.
// 8 waves x (128 x 64): 32 accumulator tiles per wave (128 accumulator registers) leave room for two waves per SIMD,
// one workgroup per CU, two-stage LDS-DMA ring (2 x 66 KiB), fragment reads as immediate-offset inline asm.
template <int FA, int FB, bool I8 = false>
__global__ __launch_bounds__(512, 1) void mx_gemm_big_kernel(MxGemmArgs a) {
    using M_ = Mma<FA, FB, I8>;
    using TA = Tile<FA>;
    using TB = Tile<FB>;
    using DA = DmaTile<FA>;
    using DB = DmaTile<FB>;
    constexpr int kTM = 256, kTN = 256, kWaves = 8;
    constexpr int kABytes = 2 * TA::kBytes, kBBytes = 2 * TB::kBytes;          // 256 rows each
    constexpr int kStage = kABytes + kBBytes + 2048;                           // + 256 x 4 scale bytes per operand
    constexpr int NA = kTM / DA::kRows / kWaves, NB = kTN / DB::kRows / kWaves; // DMA pieces per wave
    constexpr int kPieces = NA + NB + 1;

    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int r = l & 15, g = l >> 4;
    const int wm = w >> 2, wn = w & 3;
    const int tiles_m = (a.M + kTM - 1) / kTM, tiles_n = (a.N + kTN - 1) / kTN;
    const int ntiles = tiles_m * tiles_n;
    int id = blockIdx.x;
    {
        const int per = ntiles / 8, rem = ntiles % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;
    }
    constexpr int kGroup = 4;
    const int width = kGroup * tiles_n, grp = id / width, first_m = grp * kGroup;
    const int gsize = min(tiles_m - first_m, kGroup);
    const int m0 = (first_m + (id % width) % gsize) * kTM, n0 = ((id % width) / gsize) * kTN;
    const long bz = blockIdx.y;
    const long kbA = (long)a.K * elem_bits(FA) / 8, kbB = (long)a.K * elem_bits(FB) / 8;
    const int nblk = a.K / 32, nk = a.K / kBK;

    const uint8_t *ga[NA], *gb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int ll = min(l, DA::kLanes - 1);
        const int row = (w * NA + i) * DA::kRows + ll / DA::kChunks, cs = ll % DA::kChunks;
        ga[i] = a.A + bz * a.bA + (long)min(m0 + row, a.M - 1) * kbA + swz_chunk<FA>(row, cs) * 16;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int ll = min(l, DB::kLanes - 1);
        const int row = (w * NB + i) * DB::kRows + ll / DB::kChunks, cs = ll % DB::kChunks;
        gb[i] = a.B + bz * a.bB + (long)min(n0 + row, a.N - 1) * kbB + swz_chunk<FB>(row, cs) * 16;
    }
    // scale bytes: 256 rows x 4 bytes per operand = 8 pieces of 64 rows; waves 0-3 fetch A's, waves 4-7 fetch B's
    const int srow = (w & 3) * 64 + l;
    const uint8_t *gs = w < 4 ? a.sA + bz * a.bsA + (long)min(m0 + srow, a.M - 1) * nblk
                              : a.sB + bz * a.bsB + (long)min(n0 + srow, a.N - 1) * nblk;

    typename M_::acc_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = M_::zero();

    auto issue = [&](int kt, uint8_t *stage) __attribute__((always_inline)) {
        uint8_t *const s_a = stage, *const s_b = s_a + kABytes, *const s_s = s_b + kBBytes;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (DA::kLanes == 64 || l < DA::kLanes)
                __builtin_amdgcn_global_load_lds((glb_void *)(ga[i] + (long)kt * TA::kRow), (lds_void *)(s_a + (w * NA + i) * DA::kStride), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (DB::kLanes == 64 || l < DB::kLanes)
                __builtin_amdgcn_global_load_lds((glb_void *)(gb[i] + (long)kt * TB::kRow), (lds_void *)(s_b + (w * NB + i) * DB::kStride), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(gs + kt * 4), (lds_void *)(s_s + w * 256), 4, 0, 0);
    };
    // lane-constant parts of the fragment addresses (row r of fragment tile 0 of this wave's band, chunks g and 4 + g)
    const uint32_t a_lo = TA::kSix ? TA::row_off(wm * 128 + r) + 24 * g : TA::chunk_off(wm * 128 + r, g);
    const uint32_t a_hi = TA::chunk_off(wm * 128 + r, (FA < 2 ? 4 : 0) + g);
    const uint32_t b_lo = TB::kSix ? TB::row_off(wn * 64 + r) + 24 * g : TB::chunk_off(wn * 64 + r, g);
    const uint32_t b_hi = TB::chunk_off(wn * 64 + r, (FB < 2 ? 4 : 0) + g);
    const uint32_t sa_off = (wm * 128 + r) * 4 + g, sb_off = (wn * 64 + r) * 4 + g;
    // A fragments are read in two halves of four row tiles so that at most 8 fragments (64 registers for 8-bit
    // operands) are live next to the 128 accumulator registers: with all 12 live the 8-bit kernels spilled, and scratch
    // traffic shares the vmcnt queue with the LDS-DMA ring (every spill reload would drain it).
    int sa[4], sb[4];
    v8i fa[4], fb[4];
    auto read_a = [&](uint32_t base, uint32_t sbase, auto idx, auto half) __attribute__((always_inline)) {
        constexpr int I = decltype(idx)::value, H = decltype(half)::value;
        sa[I] = asm_ds_read_u8_off<(4 * H + I) * 64>(sbase + sa_off);
        fa[I] = read_frag_imm<FA, 4 * H + I>(base + a_lo, base + a_hi);
    };
    auto read_b = [&](uint32_t base, uint32_t sbase, auto idx) __attribute__((always_inline)) {
        constexpr int I = decltype(idx)::value;
        sb[I] = asm_ds_read_u8_off<I * 64>(sbase + sb_off);
        fb[I] = read_frag_imm<FB, I>(base + b_lo, base + b_hi);
    };
#define QT_I(n) std::integral_constant<int, n>{}
    auto compute = [&](const uint8_t *stage) __attribute__((always_inline)) {
        const uint32_t s_a = lds_addr(stage), s_b = s_a + kABytes, s_sa = s_b + kBBytes, s_sb = s_sa + 1024;
        read_b(s_b, s_sb, QT_I(0)); read_b(s_b, s_sb, QT_I(1)); read_b(s_b, s_sb, QT_I(2)); read_b(s_b, s_sb, QT_I(3));
        read_a(s_a, s_sa, QT_I(0), QT_I(0)); read_a(s_a, s_sa, QT_I(1), QT_I(0));
        read_a(s_a, s_sa, QT_I(2), QT_I(0)); read_a(s_a, s_sa, QT_I(3), QT_I(0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = M_::run(fa[i], fb[j], acc[i][j], sa[i], sb[j]);
        __builtin_amdgcn_sched_barrier(0);
        read_a(s_a, s_sa, QT_I(0), QT_I(1)); read_a(s_a, s_sa, QT_I(1), QT_I(1));
        read_a(s_a, s_sa, QT_I(2), QT_I(1)); read_a(s_a, s_sa, QT_I(3), QT_I(1));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[4 + i][j] = M_::run(fa[i], fb[j], acc[4 + i][j], sa[i], sb[j]);
    };
#undef QT_I

    issue(0, ring_stage_of<kStage>(0));
    for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const int kt = kt0 + sidx;
            if (kt < nk) {
                if (kt + 1 < nk) {
                    issue(kt + 1, ring_stage_of<kStage>(sidx ^ 1));
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                compute(ring_stage_of<kStage>(sidx));
                __builtin_amdgcn_s_barrier();
            }
        }
    }

    const long cbase = bz * a.bC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + r;
        if (col >= a.N) continue;
        float bv = 0.f;
        if (a.bias) bv = a.out_f32 ? ((const float *)a.bias)[col] : qt_bf2f(((const uint16_t *)a.bias)[col]);
        const float sv = out_scale_of(a, col);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 128 + i * 16 + 4 * g + e;
                if (row >= a.M) continue;
                emit_out(a, (float)acc[i][j][e], bv, sv, cbase + (long)row * a.N + col);
            }
        }
    }
}

// ---- TM x (16 nt) tiles, TM in {256, 128} and nt chosen per launch: M = 1024-class GEMMs --------------------------------
// The 256 x 256 kernel needs >= 192 tiles, and the 128 x 128 kernels run one 4-wave workgroup per CU on these shapes (256
// tiles), so every LDS and DMA latency of a step is exposed (1024 x 4096 x 4096: 1.1 us per step).  Here a workgroup is 8
// waves owning TM rows x 16 nt columns, and the host picks TM and the column widths so that the grid is a whole number of
// rounds over the CUs with the fewest operand bytes per CU, K (TM + 16 nt): 1024 x 11008 -> 4 x 64 tiles of 256 x
// (176 | 160), 1024 x 4096 -> 8 x 32 tiles of 128 x 128.  Waves: TM / 64 row bands (64 rows) x 8 / bands column parts
// (nt split as evenly as 16-column groups allow); the two waves of a SIMD share their A fragments.  8-bit formats only.
//
// LDS, all filled by LDS-DMA: rings of A tiles (TM x 128 bytes) and B tiles (nt x 2 KiB, rounded up to the wave count), as
// deep as 160 KiB allow (a tile's issue-to-landed time under load is about a step and a half), and two buffers of scale
// bytes.  Scales are fetched per QUAD of four k steps (16 bytes per row): one DMA instruction = 16 rows x 4 dwords, so it
// touches 16 lines of the [rows][K / 32] scale array -- fetched per step with one row per lane (64 lines for 256 bytes)
// the scale pieces alone cost as much as the operand tiles (1024 x 4096 x 4096, DMA only: 30 us with them, 18 without).
// A wave issues its tile pieces for the step (depth - 1) ahead spread over its multiplications of the current step (a burst
// in front of them stalls the wave until the memory pipe has drained it), with no branch in between -- hipcc sinks the
// multiplications of a step into its last basic block otherwise; one counted s_waitcnt vmcnt + ONE raw barrier per step.
// The operands are multiplied swapped (B fragment first), so a lane ends up with four consecutive output columns of one
// row: 8-byte (bf16) or 16-byte (fp32) stores.
struct WideGeom {
    int tm;                     // 256 or 128
    int tiles_m, tiles_n;
    int gbase, gextra;          // column tile j covers gbase + (j < gextra) groups of 16 columns
    int dbg;                    // QT_MX_WIDE_DEBUG: 2 = no multiplications (DMA only), 32 = no barriers
};

template <int FA, int FB, int TM, int NBP>         // NBP: B DMA pieces (8 rows x 128 bytes) per wave and stage
struct MxWide {
    // ring depths: 128 rows: four (nt <= 8) or three stages of A and B; 256 rows: three of each up to nt = 8, beyond that two of
    // A and three of B (A runs one step ahead, B two)
    static constexpr int kADepth = TM == 128 ? (NBP <= 2 ? 4 : 3) : (NBP <= 2 ? 3 : 2);
    static constexpr int kBDepth = kADepth == 2 ? 3 : kADepth;
    static constexpr int kBands = TM / 64, kParts = 8 / kBands;
    static constexpr int kAStage = TM * 128;
    static constexpr int kBStage = NBP * 8 * 1024;
    static constexpr int kQRows = TM + NBP * 64;                       // rows of one scale buffer: the A tile's, then the B tile's
    static constexpr int kQBytes = kQRows * 16;
    static constexpr int kQBase = kADepth * kAStage + kBDepth * kBStage;
    static constexpr int kLds = kQBase + 2 * kQBytes;
    static constexpr int kItems = kBands + NBP;                        // tile DMA instructions of one wave per step
    static constexpr int kMaxNTW = TM == 256 ? 6 : 4;

    static __device__ __forceinline__ int chunk_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

    template <int NTW>
    static __device__ __forceinline__ void run(const MxGemmArgs &a, const WideGeom &geo, uint8_t *lds, int m0, int tg0, int nt, int jbase,
                                               int w, int l, long bz) {
        const int r = l & 15, g = l >> 4, wm = w % kBands;
        const int nk = a.K / kBK, nblk = a.K / 32;
        // ---- DMA sources at k tile 0
        const uint8_t *ga[kBands], *gb[NBP];
        int pb[NBP];
#pragma unroll
        for (int i = 0; i < kBands; ++i) {
            const int row = (w * kBands + i) * 8 + (l >> 3), slot = l & 7;
            ga[i] = a.A + bz * a.bA + (long)min(m0 + row, a.M - 1) * a.K + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        const int npieces = nt * 2;
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int p = w + 8 * i;
            pb[i] = p < npieces ? p : p % npieces;                    // surplus pieces repeat one (same bytes, same place)
            const int row = pb[i] * 8 + (l >> 3), slot = l & 7;
            gb[i] = a.B + bz * a.bB + (long)(tg0 * 16 + row) * a.K + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        // Scale instructions of a quad: instruction c covers rows 16 c .. 16 c + 15 of the buffer (A rows first), lane = (row, dword
        // = k step within the quad); wave w issues instructions w, w + 8, ...  Lanes past the last k tile are masked off.
        constexpr int kQInstrMax = (kQRows / 16 + 7) / 8;
        const int qinstr = TM / 16 + nt;                              // instructions per quad
        const uint8_t *gq[kQInstrMax];
#pragma unroll
        for (int i = 0; i < kQInstrMax; ++i) {
            const int c = w + 8 * i, row = (c < qinstr ? c : 0) * 16 + (l >> 2);
            gq[i] = row < TM ? a.sA + bz * a.bsA + (long)min(m0 + row, a.M - 1) * nblk + (l & 3) * 4
                             : a.sB + bz * a.bsB + (long)(tg0 * 16 + row - TM) * nblk + (l & 3) * 4;
        }
        auto issue_quad = [&](int q, uint8_t *qb) __attribute__((always_inline)) {
            const bool live = 4 * q + (l & 3) < nk;
#pragma unroll
            for (int i = 0; i < kQInstrMax; ++i) {
                const int c = w + 8 * i;
                if (c < qinstr && live) __builtin_amdgcn_global_load_lds((glb_void *)(gq[i] + q * 16), (lds_void *)(qb + c * 256), 4, 0, 0);
            }
        };

        // item 0 .. kBands-1: A pieces of step ka, then the B pieces of step kb.  No conditions in here: a branch would split the
        // step into basic blocks, and hipcc then sinks every multiplication of the step into the last one.  Past the last k tile the
        // callers repeat it instead (into a slot nobody reads any more).
        auto issue_item = [&](auto ic, int ka, uint8_t *as, int kb, uint8_t *bs) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < kBands) {
                __builtin_amdgcn_global_load_lds((glb_void *)(ga[I] + (long)ka * kBK), (lds_void *)(as + (w * kBands + I) * 1024), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((glb_void *)(gb[I - kBands] + (long)kb * kBK), (lds_void *)(bs + pb[I - kBands] * 1024), 16, 0, 0);
            }
        };
        auto issue_range = [&](auto lo, auto hi, int ka, uint8_t *as, int kb, uint8_t *bs) __attribute__((always_inline)) {
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            static_assert(HI - LO <= 8, "issue_range covers at most eight items");
            if constexpr (LO + 0 < HI) issue_item(std::integral_constant<int, LO + 0>{}, ka, as, kb, bs);
            if constexpr (LO + 1 < HI) issue_item(std::integral_constant<int, LO + 1>{}, ka, as, kb, bs);
            if constexpr (LO + 2 < HI) issue_item(std::integral_constant<int, LO + 2>{}, ka, as, kb, bs);
            if constexpr (LO + 3 < HI) issue_item(std::integral_constant<int, LO + 3>{}, ka, as, kb, bs);
            if constexpr (LO + 4 < HI) issue_item(std::integral_constant<int, LO + 4>{}, ka, as, kb, bs);
            if constexpr (LO + 5 < HI) issue_item(std::integral_constant<int, LO + 5>{}, ka, as, kb, bs);
            if constexpr (LO + 6 < HI) issue_item(std::integral_constant<int, LO + 6>{}, ka, as, kb, bs);
            if constexpr (LO + 7 < HI) issue_item(std::integral_constant<int, LO + 7>{}, ka, as, kb, bs);
        };
        constexpr auto kI0 = std::integral_constant<int, 0>{};
        constexpr auto kIA = std::integral_constant<int, kBands>{};
        constexpr auto kIN = std::integral_constant<int, kItems>{};

        v4f acc[4][NTW > 0 ? NTW : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < (NTW > 0 ? NTW : 1); ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // lane-constant parts of the fragment addresses: tiles relative to their ring slot, scale bytes to (buffer + 4 (k & 3))
        const uint32_t a_lo = chunk_off(wm * 64 + r, g), a_hi = chunk_off(wm * 64 + r, 4 + g);
        const uint32_t b_lo = chunk_off(jbase * 16 + r, g), b_hi = chunk_off(jbase * 16 + r, 4 + g);
        const uint32_t qa_off = (wm * 64 + r) * 16 + g, qb_off = (TM + jbase * 16 + r) * 16 + g;

        // multiplications of one step (A at LDS address sa_, B at sb_, scale bytes at sq_) with the DMA of (ka -> as, kb -> bs)
        // spread over them
        auto compute = [&](uint32_t sa_, uint32_t sb_, uint32_t sq_, int ka, uint8_t *as, int kb, uint8_t *bs) __attribute__((always_inline)) {
            if constexpr (NTW > 0) {
                u32x4w fa_lo[4], fa_hi[4], fb_lo[3], fb_hi[3];
                int sa[4], sb[3];
#define QT_RA(i)                                                                                                  \
    fa_lo[i] = ds_read128w<i * 2048>(sa_ + a_lo); fa_hi[i] = ds_read128w<i * 2048>(sa_ + a_hi);                   \
    sa[i] = asm_ds_read_u8_off<i * 256>(sq_ + qa_off);
                QT_RA(0) QT_RA(1) QT_RA(2) QT_RA(3)
#undef QT_RA
                auto read_b = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    fb_lo[J % 3] = ds_read128w<J * 2048>(sb_ + b_lo);
                    fb_hi[J % 3] = ds_read128w<J * 2048>(sb_ + b_hi);
                    sb[J % 3] = asm_ds_read_u8_off<J * 256>(sq_ + qb_off);
                };
                v8i fa[4];
                auto step = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int P = J % 3;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (J + 2 < NTW) read_b(std::integral_constant<int, J + 2>{});
                    // LDS reads still allowed in flight: the fragments of the groups behind this one (three reads each)
                    constexpr int kAhead = (J + 2 < NTW ? 6 : (J + 1 < NTW ? 3 : 0));
                    if constexpr (J == 0) {
                        asm volatile("s_waitcnt lgkmcnt(%15)"
                                     : "+v"(fa_lo[0]), "+v"(fa_hi[0]), "+v"(fa_lo[1]), "+v"(fa_hi[1]), "+v"(fa_lo[2]), "+v"(fa_hi[2]), "+v"(fa_lo[3]),
                                       "+v"(fa_hi[3]), "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(sa[3]), "+v"(fb_lo[0]), "+v"(fb_hi[0]), "+v"(sb[0])
                                     : "n"(kAhead));
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            fa[i] = v8i{(int)fa_lo[i].x, (int)fa_lo[i].y, (int)fa_lo[i].z, (int)fa_lo[i].w,
                                        (int)fa_hi[i].x, (int)fa_hi[i].y, (int)fa_hi[i].z, (int)fa_hi[i].w};
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fb_lo[P]), "+v"(fb_hi[P]), "+v"(sb[P]) : "n"(kAhead));
                    }
                    const v8i fb = v8i{(int)fb_lo[P].x, (int)fb_lo[P].y, (int)fb_lo[P].z, (int)fb_lo[P].w,
                                       (int)fb_hi[P].x, (int)fb_hi[P].y, (int)fb_hi[P].z, (int)fb_hi[P].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[i], acc[i][J], FB, FA, 0, sb[P], 0, sa[i]);
                    // this group's share of the wave's DMA instructions
                    issue_range(std::integral_constant<int, J * kItems / NTW>{}, std::integral_constant<int, (J + 1) * kItems / NTW>{}, ka, as, kb, bs);
                    __builtin_amdgcn_sched_barrier(0);
                };
                read_b(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) read_b(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 0>{});
                if constexpr (NTW > 1) step(std::integral_constant<int, 1>{});
                if constexpr (NTW > 2) step(std::integral_constant<int, 2>{});
                if constexpr (NTW > 3) step(std::integral_constant<int, 3>{});
                if constexpr (NTW > 4) step(std::integral_constant<int, 4>{});
                if constexpr (NTW > 5) step(std::integral_constant<int, 5>{});
            } else {
                issue_range(kI0, kIN, ka, as, kb, bs);
            }
        };

        uint8_t *const bs0 = lds + kADepth * kAStage, *const q0 = lds + kQBase;
        const uint32_t l0 = lds_addr(lds);
        const int klast = nk - 1;
        // Ring positions: step kt is multiplied from slot kt % depth while step kt + depth - 1 is requested into the slot step
        // kt - 1 left.  Queue of a wave, oldest first -- quad 0 of the scales, then for equal depths: group(0) .. group(depth - 2);
        // A two deep: B(0) | A(0) | B(1).
        int a_slot = 0, a_tgt = kADepth - 1, b_slot = 0, b_tgt = kBDepth - 1;
        issue_quad(0, q0);
        if constexpr (kADepth == kBDepth) {
#pragma unroll
            for (int d = 0; d < kADepth - 1; ++d) issue_range(kI0, kIN, min(d, klast), lds + d * kAStage, min(d, klast), bs0 + d * kBStage);
        } else {
            issue_range(kIA, kIN, 0, lds, 0, bs0);
            issue_range(kI0, kIA, 0, lds, 0, bs0);
            issue_range(kIA, kIN, 0, lds, min(1, klast), bs0 + kBStage);
        }
        for (int kt = 0; kt < nk; ++kt) {
            // still allowed in flight: what was requested for the steps behind kt; everything older must have landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kADepth == kBDepth ? (kADepth - 2) * kItems : NBP) : "memory");
            if (!(geo.dbg & 32)) __builtin_amdgcn_s_barrier();       // ... for every wave; and every wave is done with step kt - 1: its slots may be refilled
            // the next quad of scale bytes, into the buffer quad (kt / 4) - 1 was read from: older than everything this wave will
            // request before the quad is needed, so the counted waits above cover it (they may wait for one tile piece more)
            if ((kt & 3) == 0 && kt + 4 < nk) issue_quad((kt >> 2) + 1, q0 + (((kt >> 2) + 1) & 1) * kQBytes);
            const int ka = min(kt + kADepth - 1, klast), kb = min(kt + kBDepth - 1, klast);
            uint8_t *const a_dst = lds + a_tgt * kAStage, *const b_dst = bs0 + b_tgt * kBStage;
            const uint32_t sa_ = l0 + a_slot * kAStage, sb_ = l0 + kADepth * kAStage + b_slot * kBStage;
            const uint32_t sq_ = l0 + kQBase + ((kt >> 2) & 1) * kQBytes + (kt & 3) * 4;
            if (!(geo.dbg & 2)) compute(sa_, sb_, sq_, ka, a_dst, kb, b_dst);
            else issue_range(kI0, kIN, ka, a_dst, kb, b_dst);
            a_tgt = a_slot; a_slot = a_slot + 1 == kADepth ? 0 : a_slot + 1;
            b_tgt = b_slot; b_slot = b_slot + 1 == kBDepth ? 0 : b_slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the repeats of the last k tile

        // ---- epilogue: lane (r, g) of tile (i, j) holds C[row wm*64 + i*16 + r][column group j, columns 4g .. 4g+3].  bf16 output: every
        // wave turns its 64 x 16 NTW tile around in its own 13 KiB of the (now dead) rings and stores whole rows, 32 NTW contiguous
        // bytes each instead of 32; fp32 output: 16-byte stores straight from the registers.
        if constexpr (NTW > 0) {
            const long cbase = bz * a.bC;
            float bvs[NTW][4];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int col = (tg0 + jbase + j) * 16 + 4 * g;
                bvs[j][0] = bvs[j][1] = bvs[j][2] = bvs[j][3] = 0.f;
                if (a.bias) {
                    if (a.out_f32) {
                        const float4 b = *(const float4 *)((const float *)a.bias + col);
                        bvs[j][0] = b.x; bvs[j][1] = b.y; bvs[j][2] = b.z; bvs[j][3] = b.w;
                    } else {
                        const uint2 b = *(const uint2 *)((const uint16_t *)a.bias + col);
                        bvs[j][0] = qt_u2f(b.x << 16); bvs[j][1] = qt_u2f(b.x & 0xFFFF0000u);
                        bvs[j][2] = qt_u2f(b.y << 16); bvs[j][3] = qt_u2f(b.y & 0xFFFF0000u);
                    }
                }
            }
            if (a.out_f32) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int col = (tg0 + jbase + j) * 16 + 4 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = m0 + wm * 64 + i * 16 + r;
                        if (row >= a.M) continue;
                        *(float4 *)((float *)a.C + cbase + (long)row * a.N + col) =
                            float4{acc[i][j][0] + bvs[j][0], acc[i][j][1] + bvs[j][1], acc[i][j][2] + bvs[j][2], acc[i][j][3] + bvs[j][3]};
                    }
                }
            } else {
                constexpr int kRowB = NTW * 32 + 8;                    // + 8: rows 16 apart would otherwise share banks
                const uint32_t tbase = l0 + w * (64 * (6 * 32 + 8));
                __syncthreads();                                       // every wave is out of the k loop: the rings are free
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t p0 = (uint32_t)qt_f2bf(acc[i][j][0] + bvs[j][0]) | ((uint32_t)qt_f2bf(acc[i][j][1] + bvs[j][1]) << 16);
                        const uint32_t p1 = (uint32_t)qt_f2bf(acc[i][j][2] + bvs[j][2]) | ((uint32_t)qt_f2bf(acc[i][j][3] + bvs[j][3]) << 16);
                        asm volatile("ds_write_b64 %0, %1" ::"v"(tbase + (i * 16 + r) * kRowB + j * 32 + g * 8), "v"(u32x2w{p0, p1}) : "memory");
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                constexpr int kChunksPerRow = NTW * 2, kChunks = 64 * kChunksPerRow;
                const long col0 = (long)(tg0 + jbase) * 16;
#pragma unroll
                for (int it = 0; it < (kChunks + 63) / 64; ++it) {
                    const int c = it * 64 + l, row = c / kChunksPerRow, ch = c % kChunksPerRow;
                    uint2 lo_, hi_;
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(lo_), "=&v"(hi_) : "v"(tbase + row * kRowB + ch * 16) : "memory");
                    const int grow = m0 + wm * 64 + row;
                    if (c < kChunks && grow < a.M) *(uint4 *)((uint16_t *)a.C + cbase + (long)grow * a.N + col0 + ch * 8) = uint4{lo_.x, lo_.y, hi_.x, hi_.y};
                }
            }
        } else {
            if (!a.out_f32) __syncthreads();                           // the barrier of the bf16 epilogue above
        }
    }
};

template <int FA, int FB, int TM, int NBP>
__global__ __launch_bounds__(512, 1) void mx_gemm_wide_kernel(MxGemmArgs a, WideGeom geo) {
    extern __shared__ __attribute__((aligned(16))) uint8_t wide_lds[];
    using W_ = MxWide<FA, FB, TM, NBP>;
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    // Workgroup ids go round-robin over the 8 XCDs: XCD x gets a contiguous run of tiles; column tile = id / tiles_m, so the
    // row tiles of one column tile (one B tile) are neighbours on one XCD and B leaves HBM once.
    const int ntiles = geo.tiles_m * geo.tiles_n;
    int id = blockIdx.x;
    {
        const int per = ntiles / 8, rem = ntiles % 8, x = id % 8, q = id / 8;
        id = x * per + (x < rem ? x : rem) + q;
    }
    const int tn = id / geo.tiles_m, tm = id % geo.tiles_m;
    const int nt = geo.gbase + (tn < geo.gextra ? 1 : 0);
    const int tg0 = tn * geo.gbase + min(tn, geo.gextra);
    const int m0 = tm * TM;
    const int part = w / W_::kBands, pbase = nt / W_::kParts, pextra = nt % W_::kParts;
    const int ntw = pbase + (part < pextra ? 1 : 0), jbase = part * pbase + min(part, pextra);
    const long bz = blockIdx.y;
    switch (ntw) {                                          // wave-uniform
        case 0: W_::template run<0>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
        case 1: W_::template run<1>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
        case 2: W_::template run<2>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
        case 3: W_::template run<3>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
        case 4: W_::template run<4>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
        case 5: if constexpr (W_::kMaxNTW >= 5) W_::template run<5>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
        default: if constexpr (W_::kMaxNTW >= 6) W_::template run<6>(a, geo, wide_lds, m0, tg0, nt, jbase, w, l, bz); break;
    }
}

int wide_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// Tiling of the wide kernel.  For each row-tile height: as many column tiles as make whole rounds over the CUs (each at most
// 12 / 16 groups of 16 columns); the height with fewer operand bytes per CU, rounds x (TM + widest tile), wins.
bool wide_geometry(int M, int N, long batch, WideGeom &geo) {
    if (N % 16 != 0 || M < 1) return false;
    const long groups = N / 16;
    const int cus = wide_cu_count();
    int force_tn = 0, dbg = 0;
#ifdef QT_TUNING_BUILD
    static const int e_force_tn = getenv("QT_MX_WIDE_TILES_N") ? atoi(getenv("QT_MX_WIDE_TILES_N")) : 0;     // tools/ only
    static const int e_dbg = getenv("QT_MX_WIDE_DEBUG") ? atoi(getenv("QT_MX_WIDE_DEBUG")) : 0;
    force_tn = e_force_tn; dbg = e_dbg;
#endif
    const char *e_tm = getenv("QT_MX_WIDE_TM");                     // test hook, read per call: the parity tests drive both tile heights
    const int force_tm = e_tm ? atoi(e_tm) : 0;
    long best = -1;
    for (int tm : {256, 128}) {
        if (force_tm && tm != force_tm) continue;
        const int max_nt = tm == 256 ? 12 : 16;
        const long tiles_m = (M + tm - 1) / tm;
        const long tn_min = (groups + max_nt - 1) / max_nt;
        const long per_round = tiles_m * batch;
        const long rounds = (per_round * tn_min + cus - 1) / cus;
        long tn = rounds * cus / per_round;
        if (force_tn > 0) tn = force_tn;
        if (tn < tn_min) tn = tn_min;
        if (tn > groups) tn = groups;
        const long gbase = groups / tn, gextra = groups % tn, worst = gbase + (gextra ? 1 : 0);
        if (worst > max_nt) continue;
        const long real_rounds = (per_round * tn + cus - 1) / cus;
        const long cost = real_rounds * (tm + 16 * worst);
        if (best < 0 || cost < best) {
            best = cost;
            geo.tm = tm; geo.tiles_m = (int)tiles_m; geo.tiles_n = (int)tn; geo.gbase = (int)gbase; geo.gextra = (int)gextra;
        }
    }
    geo.dbg = dbg;
    return best >= 0;
}

template <int FA, int FB, int TM, int NBP>
int launch_wide_nbp(const MxGemmArgs &g, const WideGeom &geo, long batch, hipStream_t st) {
    constexpr int kLds = MxWide<FA, FB, TM, NBP>::kLds;
    static_assert(kLds <= 160 * 1024, "LDS rings of the wide kernel exceed a CU's 160 KiB");
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)mx_gemm_wide_kernel<FA, FB, TM, NBP>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    mx_gemm_wide_kernel<FA, FB, TM, NBP><<<dim3((unsigned)(geo.tiles_m * geo.tiles_n), (unsigned)batch), 512, kLds, st>>>(g, geo);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <int FA, int FB>
int launch_wide(const MxGemmArgs &g, const WideGeom &geo, long batch, hipStream_t st) {
    const int worst = geo.gbase + (geo.gextra ? 1 : 0);
    const int nbp = (worst * 2 + 7) / 8;
    if (geo.tm == 256) {
        if (nbp <= 1) return launch_wide_nbp<FA, FB, 256, 1>(g, geo, batch, st);
        if (nbp == 2) return launch_wide_nbp<FA, FB, 256, 2>(g, geo, batch, st);
        return launch_wide_nbp<FA, FB, 256, 3>(g, geo, batch, st);
    }
    if (nbp <= 1) return launch_wide_nbp<FA, FB, 128, 1>(g, geo, batch, st);
    if (nbp == 2) return launch_wide_nbp<FA, FB, 128, 2>(g, geo, batch, st);
    if (nbp == 3) return launch_wide_nbp<FA, FB, 128, 3>(g, geo, batch, st);
    return launch_wide_nbp<FA, FB, 128, 4>(g, geo, batch, st);
}

// ---- packing: (values, block scales) -> element codes + E8M0 ------------------------------------------------
// A value that the format holds exactly converts exactly; anything else (and a scale that is not a power of
// two) raises the `bad` flag so the caller can fall back to the dequantize + GEMM path.
struct PackArgs {
    const void *x, *scale;       // values and block scales, bf16 or f32
    uint8_t *codes, *e8m0;
    long rows, K;                // logical [rows][K]
    long x_rs, x_ks;             // element strides of x
    long s_rs, s_ks;             // element strides of scale ([rows][K / block_size])
    long batch, x_bs, s_bs;      // leading batch dimension (strides in elements)
    int block_size, fmt, f32;
    int *bad;
};

// one thread per (row, 32-element block)
__global__ __launch_bounds__(256) void mx_pack_kernel(PackArgs p) {
    const long nb = p.K / 32;
    const long total = p.batch * p.rows * nb;
    const long rows_fast = p.x_rs == 1 ? 1 : 0;          // adjacent threads walk the contiguous dimension
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long b, row, blk;
        if (rows_fast) { row = i % p.rows; blk = (i / p.rows) % nb; b = i / (p.rows * nb); }
        else { blk = i % nb; row = (i / nb) % p.rows; b = i / (p.rows * nb); }
        bool bad = false;
        const long sidx = b * p.s_bs + row * p.s_rs + (blk * 32 / p.block_size) * p.s_ks;
        const float s = p.f32 ? ((const float *)p.scale)[sidx] : qt_bf2f(((const uint16_t *)p.scale)[sidx]);
        const uint32_t su = qt_f2u(s);
        const uint32_t eb = (su >> 23) & 0xFFu;
        const bool pow2 = (su >> 31) == 0 && eb != 0xFFu && eb != 0 && (su & 0x7FFFFFu) == 0;
        const bool tiny = (su >> 31) == 0 && eb == 0 && su != 0;      // below 2^-126: fine only in front of an all-zero block
        if (!pow2 && !tiny) bad = true;
        const int bits = elem_bits(p.fmt);
        uint32_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool allzero = true;
        for (int e = 0; e < 32; ++e) {
            const long xi = b * p.x_bs + row * p.x_rs + (blk * 32 + e) * p.x_ks;
            const float v = p.f32 ? ((const float *)p.x)[xi] : qt_bf2f(((const uint16_t *)p.x)[xi]);
            const uint32_t code = encode_elem(p.fmt, v, bad);
            allzero = allzero && ((code & ((1u << (bits - 1)) - 1u)) == 0);
            const int pos = e * bits;
            out[pos >> 5] |= code << (pos & 31);
            if ((pos & 31) + bits > 32) out[(pos >> 5) + 1] |= code >> (32 - (pos & 31));
        }
        if (tiny && !allzero) bad = true;
        const long obytes = (long)bits * 4;               // bytes per 32 elements
        uint32_t *dst = (uint32_t *)(p.codes + ((b * p.rows + row) * nb + blk) * obytes);
        for (int q = 0; q < bits; ++q) dst[q] = out[q];
        p.e8m0[(b * p.rows + row) * nb + blk] = (uint8_t)eb;
        if (bad && p.bad) atomicOr(p.bad, 1);
    }
}

int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

}  // namespace

extern "C" {

int qt_mx_pack(const void *x_dev, const void *scale_dev, int is_f32, uint8_t *codes_dev, uint8_t *e8m0_dev, long batch,
               long rows, long K, long x_batch_stride, long x_row_stride, long x_k_stride, long s_batch_stride,
               long s_row_stride, long s_k_stride, int block_size, int elem_format, int *bad_dev, void *stream) {
    if (batch * rows * K == 0) return QT_OK;
    if (!x_dev || !scale_dev || !codes_dev || !e8m0_dev || batch < 0 || rows < 0 || K < 0) return QT_ERR_BAD_ARG;
    if (elem_format < 0 || elem_format > 4 || block_size < 32 || block_size % 32 != 0 || K % 32 != 0) return QT_ERR_BAD_ARG;
    PackArgs p{x_dev, scale_dev, codes_dev, e8m0_dev, rows, K, x_row_stride, x_k_stride, s_row_stride, s_k_stride,
               batch, x_batch_stride, s_batch_stride, block_size, elem_format, is_f32, bad_dev};
    const long total = batch * rows * (K / 32);
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 65535u * 4u) grid = 65535u * 4u;
    mx_pack_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
    return launch_status();
}

int qt_mx_gemm(const uint8_t *a_codes, const uint8_t *a_e8m0, int a_format, const uint8_t *b_codes, const uint8_t *b_e8m0,
               int b_format, void *c_dev, int c_is_f32, const void *bias_dev, long batch, int M, int N, int K,
               long a_batch_stride_rows, long b_batch_stride_rows, void *stream) {
    if (batch * M * N == 0) return QT_OK;
    if (!a_codes || !a_e8m0 || !b_codes || !b_e8m0 || !c_dev || batch < 0 || batch > 65535 || M < 0 || N < 0 || K < 32 || K % 32)
        return QT_ERR_BAD_ARG;
    if (a_format < 0 || a_format > 4 || b_format < 0 || b_format > 4) return QT_ERR_BAD_ARG;
    const long kbA = (long)K * elem_bits(a_format) / 8, kbB = (long)K * elem_bits(b_format) / 8;
    if ((kbA & 15) || (kbB & 15) || (((uintptr_t)a_codes | (uintptr_t)b_codes) & 15u)) return QT_ERR_UNALIGNED;
    MxGemmArgs g{a_codes, b_codes, a_e8m0, b_e8m0, c_dev, bias_dev, M, N, K,
                 a_batch_stride_rows * kbA, b_batch_stride_rows * kbB, a_batch_stride_rows * (K / 32), b_batch_stride_rows * (K / 32),
                 (long)M * N, c_is_f32, nullptr, 0, 0};
    const dim3 grid(((N + kBN - 1) / kBN) * ((M + kBM - 1) / kBM), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    bool no_dma = false;
    int force_stages = 0, force_big = -1;
#ifdef QT_TUNING_BUILD
    static const bool e_no_dma = getenv("QT_MX_NO_DMA") != nullptr;      // tools/ only: A-B switches of tools/exp_mx_gemm.py
    static const int e_stages = getenv("QT_MX_STAGES") ? atoi(getenv("QT_MX_STAGES")) : 0;
    static const int e_big = getenv("QT_MX_BIG") ? atoi(getenv("QT_MX_BIG")) : -1;
    no_dma = e_no_dma; force_stages = e_stages; force_big = e_big;
#endif
    const bool dma_ok = !no_dma && K % kBK == 0 && (((uintptr_t)a_e8m0 | (uintptr_t)b_e8m0) & 3u) == 0;
    const long nblocks = (long)grid.x * grid.y;
    // Up to two workgroups per CU: the two-stage prefetch ring (a tile's load latency would otherwise be exposed on
    // every step).  Larger grids: one stage, three resident workgroups per CU cover each other.  Deeper rings
    // (4-8 stages at one workgroup per CU) measured no better than two stages at two workgroups per CU.
    const bool ring = force_stages ? force_stages >= 2 : nblocks <= 2L * 256;
    const long big_tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    const bool big = dma_ok && (force_big >= 0 ? force_big == 1 : (M >= 512 && N >= 512 && big_tiles >= 192));
    // 8-bit formats, at least two 256-row tiles, fewer tiles than the 256 x 256 kernel wants: column widths fitted to the chip
    const char *e_wide = getenv("QT_MX_WIDE");                              // test hook, read per call: 0 / 1 force the 128 x 128 / the wide kernel
    const int force_wide = e_wide ? atoi(e_wide) : -1;
    WideGeom geo{};
    if (dma_ok && a_format < 2 && b_format < 2 && force_wide != 0 && (force_wide == 1 || (!big && M >= 512 && N >= 512))
        && wide_geometry(M, N, batch, geo)) {
        if (a_format == 0 && b_format == 0) return launch_wide<0, 0>(g, geo, batch, st);
        if (a_format == 0 && b_format == 1) return launch_wide<0, 1>(g, geo, batch, st);
        if (a_format == 1 && b_format == 0) return launch_wide<1, 0>(g, geo, batch, st);
        return launch_wide<1, 1>(g, geo, batch, st);
    }
#define QT_MX_BIG(FA, FB)                                                                                          \
    if (big && a_format == FA && b_format == FB) {                                                                 \
        const dim3 bgrid((unsigned)big_tiles, (unsigned)batch);                                                    \
        mx_gemm_big_kernel<FA, FB><<<bgrid, 512, 0, st>>>(g);                                                      \
        return launch_status();                                                                                    \
    }
    QT_MX_BIG(0, 0) QT_MX_BIG(0, 1) QT_MX_BIG(1, 0) QT_MX_BIG(1, 1) QT_MX_BIG(4, 4) QT_MX_BIG(0, 4)
    QT_MX_BIG(2, 2) QT_MX_BIG(3, 3) QT_MX_BIG(2, 4) QT_MX_BIG(3, 4)
#undef QT_MX_BIG
#define QT_MX_DMA(FA, FB)                                                                                          \
    if (dma_ok && a_format == FA && b_format == FB) {                                                              \
        constexpr int kLds = Tile<FA>::kBytes + Tile<FB>::kBytes + 1024;                                           \
        if (ring) {                                                                                                \
            mx_gemm_dma_kernel<FA, FB, 2><<<grid, 256, 0, st>>>(g);              /* static LDS, one object per stage */ \
        } else {                                                                                                   \
            mx_gemm_dma_kernel<FA, FB, 1><<<grid, 256, kLds, st>>>(g);                                             \
        }                                                                                                          \
        return launch_status();                                                                                    \
    }
    QT_MX_DMA(0, 0) QT_MX_DMA(0, 1) QT_MX_DMA(1, 0) QT_MX_DMA(1, 1) QT_MX_DMA(4, 4) QT_MX_DMA(0, 4)
    QT_MX_DMA(2, 2) QT_MX_DMA(3, 3) QT_MX_DMA(2, 4) QT_MX_DMA(3, 4)
#undef QT_MX_DMA
#define QT_MX(FA, FB)                                                                                              \
    if (a_format == FA && b_format == FB) {                                                                        \
        constexpr int kLds = 2 * Tile<FA>::kBytes + 2 * Tile<FB>::kBytes;                                          \
        static QtOncePerDevice configured;                                                                                  \
        if (configured.needed()) {                                                                                         \
            const hipError_t e = hipFuncSetAttribute((const void *)mx_gemm_kernel<FA, FB>,                         \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);            \
            if (e != hipSuccess) return (int)e;                                                                    \
            configured.done();                                                                                      \
        }                                                                                                          \
        mx_gemm_kernel<FA, FB><<<grid, 256, kLds, st>>>(g);                                                        \
        return launch_status();                                                                                    \
    }
    QT_MX(0, 0) QT_MX(0, 1) QT_MX(1, 0) QT_MX(1, 1) QT_MX(2, 2) QT_MX(3, 3) QT_MX(4, 4) QT_MX(0, 4) QT_MX(2, 4) QT_MX(3, 4)
#undef QT_MX
    return QT_ERR_BAD_DTYPE;          // element-format pair without a kernel: caller dequantizes
}


int qt_q8_gemm(const int8_t *a_codes, const int8_t *b_codes, void *c_dev, int c_is_f32, const void *bias_dev, const void *out_scale_dev,
               int out_scale_per_col, int fold_f32, long batch, int M, int N, int K, long a_batch_stride_rows, long b_batch_stride_rows, void *stream) {
    if (batch * M * N == 0) return QT_OK;
    if (!a_codes || !b_codes || !c_dev || batch < 0 || batch > 65535 || M < 0 || N < 0 || K < 16 || K % 16) return QT_ERR_BAD_ARG;
    if (((uintptr_t)a_codes | (uintptr_t)b_codes) & 15u) return QT_ERR_UNALIGNED;
    const uint8_t *A = (const uint8_t *)a_codes, *B = (const uint8_t *)b_codes;
    // the scale-byte pointers of the block-scaled kernels are never used by the int8 instruction; the code buffers stand in
    // for them (every address those loads form lies inside the buffers)
    MxGemmArgs g{A, B, A, B, c_dev, bias_dev, M, N, K, a_batch_stride_rows * (long)K, b_batch_stride_rows * (long)K,
                 a_batch_stride_rows * (long)(K / 32), b_batch_stride_rows * (long)(K / 32), (long)M * N, c_is_f32, out_scale_dev,
                 out_scale_per_col, fold_f32};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(((N + kBN - 1) / kBN) * ((M + kBM - 1) / kBM), (unsigned)batch);
    const bool dma_ok = K % kBK == 0;
    const long big_tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    if (dma_ok && M >= 512 && N >= 512 && big_tiles >= 192) {
        mx_gemm_big_kernel<0, 0, true><<<dim3((unsigned)big_tiles, (unsigned)batch), 512, 0, st>>>(g);
        return launch_status();
    }
    if (dma_ok) {
        if ((long)grid.x * grid.y <= 2L * 256) {
            mx_gemm_dma_kernel<0, 0, 2, true><<<grid, 256, 0, st>>>(g);
        } else {
            constexpr int kLds = Tile<0>::kBytes + Tile<0>::kBytes + 1024;
            mx_gemm_dma_kernel<0, 0, 1, true><<<grid, 256, kLds, st>>>(g);
        }
        return launch_status();
    }
    constexpr int kLds = 4 * Tile<0>::kBytes;
    static QtOncePerDevice configured;      
    if (configured.needed()) {
        const hipError_t e = hipFuncSetAttribute((const void *)mx_gemm_kernel<0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        configured.done();
    }
    mx_gemm_kernel<0, 0, true><<<grid, 256, kLds, st>>>(g);
    return launch_status();
}

}  // extern "C"
