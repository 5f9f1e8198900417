// qt_attention_train.hip -- the attention core of a TRAINING step (BASELINE configs[4]) as one launch forward, one backward.
//
// Between the query / key / value projections and the output projection the reference's quantizable attention block
// (modules/quantizable/modeling_bert.py:118-158; functional_modules.py:22-26; hooks: quantize.py:116-179) runs, per layer and step,
//     forward   q' = fq(q), k' = fq(k)                  qk_matmul's forward-pre hook            2 launches
//               S  = bf16(q' k'^T)                      torch.matmul                            1
//               P  = softmax(bf16(bf16(S * scaling) + mask)),  P' = fq(P), v' = fq(v)           scaling, mask, softmax, av_matmul's hook
//               O  = bf16(P' v'), permuted to [B, S, H * D] (a copy), fq of the output projection's input
//     backward  g  = fq_e(dO)                           av_matmul's backward-pre hook
//               dP = bf16(g v'^T), dV = bf16(P'^T g)    two GEMMs
//               dS = bf16(bf16((dP - sum dP P) P) * scaling), dS' = fq_e(dS)                    softmax backward, qk_matmul's backward-pre hook
//               dQ = bf16(dS' k'), dK = bf16(dS'^T q')  two GEMMs, three permute copies
// -- 9 + 11 launches of 5 - 17 us on tensors of 1.5 - 6 MB ([16, 12, 128, 64] and [16, 12, 128, 128]).  Here one workgroup owns one
// (batch, head): every tensor of that head fits in its LDS (<= 128 positions, head dimension 64), the four products run on the matrix
// cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation), every fake-quantizer is the per-element function of qt_fake_quant_bf16 with its
// own scale and amax slot, and every rounding point listed above is kept.  What the backward needs is written once (q', k', v', P, P');
// the result and the three gradients are written in the [B, S, H, D] layout their consumers read, so no permute copy remains.
//
// Riding along: attention-probability dropout with a caller-drawn keep mask (torch's masked-scale arithmetic, both ways); forward, the
// output projection's input quantizer on the result; backward, the q / k / v projections' own backward-pre quantizers on dQ / dK / dV
// and the column sums of their results = the three bias gradients (64-bit fixed-point atomics across the batch: deterministic).
//
// Bit-defined against the launches it replaces: the fake-quantizers (exactly), the softmax forward and backward (the row code of
// qt_softmax.hip: same lanes per row, same order of the row sums).  Not bit-defined: the order in which a matrix instruction adds
// the products of a dot (the library GEMM's is not defined either) -- with power-of-two scales every dot of int8 values is exact in
// fp32, and tests/test_gpu_parity.py then demands bit equality with the launches replaced; else it states the tolerance.
//
// Shape of a launch: 512 threads, one workgroup per CU (LDS).  Every global read is issued before anything waits (named registers:
// arrays of loaded vectors went through scratch memory), barriers order LDS only (`s_waitcnt lgkmcnt(0); s_barrier` -- __syncthreads()
// would drain every global store in flight), every global store leaves in the last phase.  profiles/r05_attention_train_stamps.txt has
// the cycles per phase; a 1024-thread variant halved the vector phases and lost it all in the barriers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "qt_device.h"
#include "qt_chain.h"

namespace {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kD = 64;                 // head dimension
constexpr int kSMax = 128;             // positions per (batch, head)
constexpr int kRowD = kD * 2 + 16;     // bytes of a padded row of an [S][64] tile
constexpr int kRowS = kSMax * 2 + 16;  // ... of an [S][S] tile
constexpr int kTileD = kSMax * kRowD;
constexpr int kTileS = kSMax * kRowS;

struct FqDev {
    const float *scale;
    uint32_t *amax;
};

// matrix-core operands (v_mfma_f32_16x16x32_bf16: lane l supplies A[l % 16][8 (l / 16) ...+7] and B[8 (l / 16) ...+7][l % 16],
// and receives D[4 (l / 16) + i][l % 16]).  frag_rows: the tile holds the operand with its contraction index contiguous
// (16 rows x 32 k: one 16-byte LDS read); frag_cols: the tile holds [k][x] (eight 2-byte reads, one per k).
__device__ __forceinline__ bf16x8_t frag_rows(const unsigned char *tile, int row_bytes, int r0, int k0, int lane) {
    return *(const bf16x8_t *)(tile + (r0 + (lane & 15)) * row_bytes + (k0 + (lane >> 4) * 8) * 2);
}
__device__ __forceinline__ bf16x8_t frag_cols(const unsigned char *tile, int row_bytes, int x0, int k0, int lane) {
    const unsigned char *p = tile + (k0 + (lane >> 4) * 8) * row_bytes + (x0 + (lane & 15)) * 2;
    bf16x8_t f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = *(const short *)(p + e * row_bytes);
    return f;
}
__device__ __forceinline__ uint16_t bf16_bits(float f) { return (uint16_t)pack_bf16x2(f, 0.0f); }

// accumulators of one 16 x 16 tile -> bf16 into an LDS tile
__device__ __forceinline__ void store_tile(unsigned char *tile, int row_bytes, int r0, int c0, const f32x4_t &acc, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *(uint16_t *)(tile + (r0 + 4 * (lane >> 4) + i) * row_bytes + (c0 + (lane & 15)) * 2) = bf16_bits(acc[i]);
}

// A workgroup barrier that orders LDS only.  __syncthreads() is a workgroup-scope fence over ALL address spaces: on this target it waits
// for every outstanding global store and load of the wave (s_waitcnt vmcnt(0)) -- here that would put the round trip of the q' / k' / v' /
// P / P' stores in front of every phase.  Nothing in these kernels reads global memory another wave of the launch has written.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N, int WAVES>
__device__ __forceinline__ void amax_commit_w(const FqDev (&fq)[N], const uint32_t (&amax)[N], uint32_t (*s_amax)[WAVES]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = wave_max_u32(amax[i]);
        if (lane == 0) s_amax[i][wave] = m;
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (t == i * 32 && fq[i].amax) {
            uint32_t m = s_amax[i][0];
#pragma unroll
            for (int k = 1; k < WAVES; ++k) m = m > s_amax[i][k] ? m : s_amax[i][k];
            if (m != 0u && m > __hip_atomic_load(fq[i].amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(fq[i].amax, m);
        }
    }
}

struct AttnTrainFwdArgs {
    const uint16_t *q, *k, *v;
    uint16_t *qq, *kq, *vq;
    long sb, ss, sh;                   // element strides of batch, position, head (the 64 values of a head are contiguous)
    const uint16_t *mask;
    long msb, msh, msq;
    uint16_t *probs, *pq;              // [B, H, S, S]
    uint16_t *out, *oq;                // [B, S, H, 64]
    int H, S;
    float scaling;
    FqDev fq[5];                       // q, k, v, probabilities, result
    const uint8_t *drop;               // optional dropout on the probabilities: keep mask [B, H, S, S] (1 = keep), P_d = bf16(P * keep * drop_scale)
    float drop_scale;                  // 1 / (1 - p)
    unsigned long long *dbg;           // tuning build, QT_AT_STAMPS = device address: s_memtime stamps of workgroup 0, waves 0 and 7 (16 slots each)
};

constexpr int kThreads = 512;                       // 8 waves: wave w owns the 16 query (or key) rows of tile w
constexpr int kVecIters = kSMax * 8 / kThreads;     // 16-byte vectors of an [S][64] tile per thread
constexpr int kRowIters = kSMax / (kThreads / 16);  // rows of the softmax per 16-lane group

template <int KIND>
__global__ __launch_bounds__(kThreads) void attn_train_fwd_kernel(AttnTrainFwdArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    // q', k', v', the scores (then P'), P and the result each keep a tile of their own until the end: every global store of the launch is
    // issued in the last phase, where nothing waits behind it (a store in flight stalls the next write to its data registers)
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kTileD + 2 * kTileS];
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    __shared__ uint32_t s_amax[5][kThreads / 64];
    unsigned char *Qs = lds, *Ks = lds + kTileD, *Vs = lds + 2 * kTileD, *Os = lds + 3 * kTileD, *Ss = lds + 4 * kTileD, *Pr = Ss + kTileS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, S = a.S, mt = S / 16;
#ifdef QT_TUNING_BUILD
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (a.dbg && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) a.dbg[0 + (wave ? 16 : 0) + slot] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [](int) __attribute__((always_inline)) {};
#endif
    stamp(0);
    // every global read of the launch is issued before anything waits (one round trip, not one per loop iteration)
    const long base = (long)b * a.sb + (long)h * a.sh;
    uint4 xq[kVecIters], xk[kVecIters], xv[kVecIters];
#pragma unroll
    for (int i = 0; i < kVecIters; ++i) {
        const int id = tid + i * kThreads;
        if (id < S * 8) {
            const long off = base + (long)(id >> 3) * a.ss + (id & 7) * 8;
            xq[i] = *(const uint4 *)(a.q + off);
            xk[i] = *(const uint4 *)(a.k + off);
            xv[i] = *(const uint4 *)(a.v + off);
        }
    }
    const int gi = tid >> 4, li = tid & 15, nvec_row = S / 8;
    const uint16_t *mrow = a.mask ? a.mask + (long)b * a.msb + (long)h * a.msh : nullptr;
    uint4 m0 = {0u, 0u, 0u, 0u};
    if (mrow && a.msq == 0 && li < nvec_row) m0 = ((const uint4 *)mrow)[li];      // one mask row for the whole (batch, head): the usual padding mask
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, kThreads);
    float sc[5];
    uint32_t amax[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        sc[i] = a.fq[i].scale ? qt_bf2f(qt_f2bf(*a.fq[i].scale)) : 1.0f;
        amax[i] = 0u;
    }
    // ---- q' = fq(q), k' = fq(k), v' = fq(v) into LDS
    {
        const UniformDiv dq(sc[0]), dk(sc[1]), dv(sc[2]);
#pragma unroll
        for (int i = 0; i < kVecIters; ++i) {
            const int id = tid + i * kThreads;
            if (id < S * 8) {
                const int r = id >> 3, c = id & 7;
                *(uint4 *)(Qs + r * kRowD + c * 16) = chain_apply<KIND>(xq[i], sc[0], dq, rnd, amax[0]);
                *(uint4 *)(Ks + r * kRowD + c * 16) = chain_apply<KIND>(xk[i], sc[1], dk, rnd, amax[1]);
                *(uint4 *)(Vs + r * kRowD + c * 16) = chain_apply<KIND>(xv[i], sc[2], dv, rnd, amax[2]);
            }
        }
    }
    stamp(1);
    lds_barrier();
    stamp(2);
    // ---- S = bf16(q' k'^T)
    if (wave < mt) {
        f32x4_t acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t af = frag_rows(Qs, kRowD, wave * 16, ks * 32, lane);
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (n < mt) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag_rows(Ks, kRowD, n * 16, ks * 32, lane), acc[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < mt) store_tile(Ss, kRowS, wave * 16, n * 16, acc[n], lane);
    }
    stamp(3);
    lds_barrier();
    stamp(4);
    // ---- P = softmax(bf16(bf16(S * scaling) + mask)), P' = fq(P): 16 lanes per row, eight columns per lane -- the row code of
    //      softmax_fq_kernel<KIND, 1, 16> (qt_softmax.hip) on the scores in LDS
    {
        const float s = sc[3];
        const UniformDiv dv(s);
        // (loops of these kernels are NOT unrolled: a launch runs once through its code with a cold instruction cache, and the code of
        // an unrolled fake-quantizer -- three division variants per call site -- was most of the 66 KiB the first version fetched)
#pragma unroll 1
        for (int row = gi; row < S; row += kThreads / 16) {
            const bool act = li < nvec_row;
            float t[8];
            float mx = -INFINITY;
            if (act) {
                const uint4 x = *(const uint4 *)(Ss + row * kRowS + li * 16);
                uint4 m = m0;
                if (mrow && a.msq != 0) m = ((const uint4 *)(mrow + (long)row * a.msq))[li];
                const uint32_t xw[4] = {x.x, x.y, x.z, x.w}, mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t p = pack_bf16x2(bf_lo(xw[j]) * a.scaling, bf_hi(xw[j]) * a.scaling);
                    float lo = bf_lo(p), hi = bf_hi(p);
                    if (mrow) {
                        p = pack_bf16x2(lo + bf_lo(mw[j]), hi + bf_hi(mw[j]));
                        lo = bf_lo(p);
                        hi = bf_hi(p);
                    }
                    t[2 * j] = lo;
                    t[2 * j + 1] = hi;
                    mx = fmaxf(mx, fmaxf(lo, hi));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = -INFINITY;
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            float sum = 0.0f;
            const float cut = mx - 110.0f;
            bool dead = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) dead = dead && (t[j] < cut);
            if (!dead) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t[j] = __expf(t[j] - mx);
                    sum += t[j];
                }
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            const float inv = 1.0f / sum;
            if (act) {
                uint4 pv = {0u, 0u, 0u, 0u}, qv = {0u, 0u, 0u, 0u};
                if (!dead) {
                    pv = uint4{pack_bf16x2(t[0] * inv, t[1] * inv), pack_bf16x2(t[2] * inv, t[3] * inv), pack_bf16x2(t[4] * inv, t[5] * inv),
                               pack_bf16x2(t[6] * inv, t[7] * inv)};
                    uint4 dv8 = pv;
                    if (a.drop) {                              // torch's dropout: (P * keep) * scale in fp32, one rounding
                        const uint2 mk = *(const uint2 *)(a.drop + ((long)bh * S + row) * S + li * 8);
                        const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
                        uint32_t r[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t m = j < 2 ? mk.x >> (16 * j) : mk.y >> (16 * (j - 2));
                            r[j] = pack_bf16x2((m & 0xFFu) ? bf_lo(pw[j]) * a.drop_scale : 0.0f, (m & 0xFF00u) ? bf_hi(pw[j]) * a.drop_scale : 0.0f);
                        }
                        dv8 = uint4{r[0], r[1], r[2], r[3]};
                    }
                    qv = chain_apply<KIND>(dv8, s, dv, rnd, amax[3]);
                }
                *(uint4 *)(Pr + row * kRowS + li * 16) = pv;
                *(uint4 *)(Ss + row * kRowS + li * 16) = qv;
            }
        }
    }
    stamp(5);
    lds_barrier();
    stamp(6);
    // ---- O = bf16(P' v'): four column tiles, k over the positions
    if (wave < mt) {
        f32x4_t o[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < S / 32; ++ks) {
            const bf16x8_t af = frag_rows(Ss, kRowS, wave * 16, ks * 32, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag_cols(Vs, kRowD, n * 16, ks * 32, lane), o[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) store_tile(Os, kRowD, wave * 16, n * 16, o[n], lane);
    }
    stamp(7);
    lds_barrier();
    stamp(8);
    // ---- everything leaves: q', k', v' (what the backward reads), P, P', the result in [B, S, H, 64] and its quantized form
    {
        const UniformDiv dv(sc[4]);
#pragma unroll 1
        for (int id = tid; id < S * 8; id += kThreads) {
            {
                const int r = id >> 3, c = id & 7;
                const long off = base + (long)r * a.ss + c * 8, ooff = (((long)b * S + r) * a.H + h) * kD + c * 8;
                *(uint4 *)(a.qq + off) = *(const uint4 *)(Qs + r * kRowD + c * 16);
                *(uint4 *)(a.kq + off) = *(const uint4 *)(Ks + r * kRowD + c * 16);
                *(uint4 *)(a.vq + off) = *(const uint4 *)(Vs + r * kRowD + c * 16);
                const uint4 y = *(const uint4 *)(Os + r * kRowD + c * 16);
                *(uint4 *)(a.out + ooff) = y;
                if (a.oq) *(uint4 *)(a.oq + ooff) = chain_apply<KIND>(y, sc[4], dv, rnd, amax[4]);
            }
        }
#pragma unroll 1
        for (int row = gi; row < S; row += kThreads / 16) {
            if (li < nvec_row) {
                const long o = ((long)bh * S + row) * S + li * 8;
                *(uint4 *)(a.probs + o) = *(const uint4 *)(Pr + row * kRowS + li * 16);
                *(uint4 *)(a.pq + o) = *(const uint4 *)(Ss + row * kRowS + li * 16);
            }
        }
    }
    stamp(9);
    amax_commit_w<5, kThreads / 64>(a.fq, amax, s_amax);
    stamp(10);
}

struct GradStage {
    const float *scale;
    uint32_t *amax;
    uint16_t *out;                     // fq(gradient), [B, S, H, 64]; NULL: this gradient is not quantized here
    uint16_t *colsum;                  // [H * 64] bf16 column sums of `out` over batch and position; NULL: not wanted
};

struct AttnTrainBwdArgs {
    const uint16_t *gy;                // [B, S, H, 64]
    const uint16_t *qq, *kq, *vq;
    long sb, ss, sh;
    const uint16_t *probs, *pq;        // [B, H, S, S]
    uint16_t *dq, *dk, *dv;            // [B, S, H, 64]
    uint16_t *g_out;                   // optional: g = e0(dO), [B, S, H, 64]
    uint16_t *ds_out, *dsq_out;        // optional: dS and dS' = e1(dS), [B, H, S, S]
    int H, S;
    float scaling;
    FqDev fq[2];                       // grad of the result (av_matmul's backward-pre quantizer), grad of the scores (qk_matmul's)
    // the projections' own backward-pre quantizers on dQ / dK / dV (each optional), and the column sums of their results = the bias
    // gradients: 64-bit fixed-point accumulators per (tensor, column) and a ticket per (tensor, head), zero between launches
    GradStage gs[3];
    float colsum_max;
    long long *acc;                    // [3][H * 64]
    unsigned int *ticket;              // [3][H]
    int B;
    const uint8_t *drop;               // the forward's keep mask: dP := bf16(dP * keep * drop_scale) before the softmax backward
    float drop_scale;
    unsigned long long *dbg;           // as in the forward, slots 32..
};

template <int KIND>
__global__ __launch_bounds__(kThreads) void attn_train_bwd_kernel(AttnTrainBwdArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[5 * kTileD + kTileS];
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    __shared__ uint32_t s_amax[5][kThreads / 64];
    __shared__ unsigned int s_old[3];
    unsigned char *Gs = lds, *Vs = lds + kTileD, *Ks = lds + 2 * kTileD, *Qs = lds + 3 * kTileD, *Ps = lds + 4 * kTileD, *Ds = Ps + kTileS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, S = a.S, mt = S / 16, nvec_row = S / 8;
    const int gi = tid >> 4, li = tid & 15;
#ifdef QT_TUNING_BUILD
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (a.dbg && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) a.dbg[32 + (wave ? 16 : 0) + slot] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [](int) __attribute__((always_inline)) {};
#endif
    stamp(0);
    // every global read of the launch is issued before anything waits
    const long base = (long)b * a.sb + (long)h * a.sh;
    // (named registers, not arrays: arrays of loaded vectors went through scratch memory or serialised the loads with copies)
    static_assert(kVecIters == 2 && kRowIters == 4, "the staging registers below are written out for these counts");
    const uint4 z4 = {0u, 0u, 0u, 0u};
    uint4 xg0 = z4, xg1 = z4, xq0 = z4, xq1 = z4, xk0 = z4, xk1 = z4, xv0 = z4, xv1 = z4;
    uint4 xp0 = z4, xp1 = z4, xp2 = z4, xp3 = z4, pr0 = z4, pr1 = z4, pr2 = z4, pr3 = z4;
    uint2 mk0 = {0u, 0u}, mk1 = mk0, mk2 = mk0, mk3 = mk0;
    auto vec_load = [&](int i, uint4 &g, uint4 &q, uint4 &k, uint4 &v) __attribute__((always_inline)) {
        const int id = tid + i * kThreads;
        if (id < S * 8) {
            const int r = id >> 3, c = id & 7;
            const long off = base + (long)r * a.ss + c * 8;
            g = *(const uint4 *)(a.gy + (((long)b * S + r) * a.H + h) * kD + c * 8);
            q = *(const uint4 *)(a.qq + off);
            k = *(const uint4 *)(a.kq + off);
            v = *(const uint4 *)(a.vq + off);
        }
    };
    auto row_load = [&](int it, uint4 &xp, uint4 &pr, uint2 &mk) __attribute__((always_inline)) {
        const int row = gi + it * (kThreads / 16);
        if (row < S && li < nvec_row) {
            const long o = ((long)bh * S + row) * S + li * 8;
            xp = *(const uint4 *)(a.pq + o);
            pr = *(const uint4 *)(a.probs + o);
            if (a.drop) mk = *(const uint2 *)(a.drop + o);
        }
    };
    vec_load(0, xg0, xq0, xk0, xv0);
    vec_load(1, xg1, xq1, xk1, xv1);
    row_load(0, xp0, pr0, mk0);
    row_load(1, xp1, pr1, mk1);
    row_load(2, xp2, pr2, mk2);
    row_load(3, xp3, pr3, mk3);
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, kThreads);
    float sc[2], gsc[3];
    uint32_t amax[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        sc[i] = a.fq[i].scale ? qt_bf2f(qt_f2bf(*a.fq[i].scale)) : 1.0f;
        amax[i] = 0u;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) gsc[t] = a.gs[t].scale ? qt_bf2f(qt_f2bf(*a.gs[t].scale)) : 1.0f;      // (read here: a dependent global load each)
    // ---- g = fq_e(dO), q', k', v', P' into LDS
    {
        const UniformDiv dg(sc[0]);
        auto vec_stage = [&](int i, const uint4 &g, const uint4 &q, const uint4 &k, const uint4 &v) __attribute__((always_inline)) {
            const int id = tid + i * kThreads;
            if (id < S * 8) {
                const int r = id >> 3, c = id & 7;
                const uint4 yg = chain_apply<KIND>(g, sc[0], dg, rnd, amax[0]);
                *(uint4 *)(Gs + r * kRowD + c * 16) = yg;
                if (a.g_out) *(uint4 *)(a.g_out + (((long)b * S + r) * a.H + h) * kD + c * 8) = yg;
                *(uint4 *)(Qs + r * kRowD + c * 16) = q;
                *(uint4 *)(Ks + r * kRowD + c * 16) = k;
                *(uint4 *)(Vs + r * kRowD + c * 16) = v;
            }
        };
        auto row_stage = [&](int it, const uint4 &xp) __attribute__((always_inline)) {
            const int row = gi + it * (kThreads / 16);
            if (row < S && li < nvec_row) *(uint4 *)(Ps + row * kRowS + li * 16) = xp;
        };
        vec_stage(0, xg0, xq0, xk0, xv0);
        vec_stage(1, xg1, xq1, xk1, xv1);
        row_stage(0, xp0);
        row_stage(1, xp1);
        row_stage(2, xp2);
        row_stage(3, xp3);
    }
    stamp(1);
    lds_barrier();
    stamp(2);
    // ---- dV = bf16(P'^T g) (rows: key positions), dP = bf16(g v'^T) (rows: query positions); wave w owns tile w of either
    const int m0 = (wave < mt ? wave : 0) * 16;
    f32x4_t accv[4], accp[8];
#pragma unroll
    for (int n = 0; n < 4; ++n) accv[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) accp[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (wave < mt) {
        for (int ks = 0; ks < S / 32; ++ks) {
            const bf16x8_t af = frag_cols(Ps, kRowS, m0, ks * 32, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) accv[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag_cols(Gs, kRowD, n * 16, ks * 32, lane), accv[n], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t af = frag_rows(Gs, kRowD, m0, ks * 32, lane);
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (n < mt) accp[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag_rows(Vs, kRowD, n * 16, ks * 32, lane), accp[n], 0, 0, 0);
        }
    }
    lds_barrier();                   // every wave has read P', g and v': their tiles take dP and dV
    if (wave < mt) {
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < mt) store_tile(Ps, kRowS, m0, n * 16, accp[n], lane);
#pragma unroll
        for (int n = 0; n < 4; ++n) store_tile(Ds, kRowD, m0, n * 16, accv[n], lane);
    }
    stamp(3);
    lds_barrier();
    stamp(4);
    // ---- dS = bf16(bf16((dP - sum dP P) P) * scaling), dS' = fq_e(dS): the row code of softmax_bwd_kernel<KIND, 1, 1, 16>
    {
        const UniformDiv dv(sc[1]);
        auto row_body = [&](int it, const uint4 &pv, const uint2 &mk) __attribute__((always_inline)) {
            const int row = gi + it * (kThreads / 16);
            if (row >= S) return;                              // (uniform per workgroup: S is a multiple of 32)
            const bool act = li < nvec_row;
            float g[8], p[8];
            float dot = 0.0f;
            if (act) {
                const uint4 gv = *(const uint4 *)(Ps + row * kRowS + li * 16);
                uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
                const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
                if (a.drop) {                                  // the dropout's backward: (dP * keep) * scale, one rounding
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t m = j < 2 ? mk.x >> (16 * j) : mk.y >> (16 * (j - 2));
                        gw[j] = pack_bf16x2((m & 0xFFu) ? bf_lo(gw[j]) * a.drop_scale : 0.0f, (m & 0xFF00u) ? bf_hi(gw[j]) * a.drop_scale : 0.0f);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    g[2 * j] = bf_lo(gw[j]); g[2 * j + 1] = bf_hi(gw[j]);
                    p[2 * j] = bf_lo(pw[j]); p[2 * j + 1] = bf_hi(pw[j]);
                    dot += g[2 * j] * p[2 * j];
                    dot += g[2 * j + 1] * p[2 * j + 1];
                }
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
            if (act) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t d1 = pack_bf16x2((g[2 * j] - dot) * p[2 * j], (g[2 * j + 1] - dot) * p[2 * j + 1]);
                    o[j] = pack_bf16x2(bf_lo(d1) * a.scaling, bf_hi(d1) * a.scaling);
                }
                const uint4 dsv = uint4{o[0], o[1], o[2], o[3]};
                const uint4 dsq = chain_apply<KIND>(dsv, sc[1], dv, rnd, amax[1]);
                *(uint4 *)(Ps + row * kRowS + li * 16) = dsq;
                if (a.ds_out) *(uint4 *)(a.ds_out + ((long)bh * S + row) * S + li * 8) = dsv;
                if (a.dsq_out) *(uint4 *)(a.dsq_out + ((long)bh * S + row) * S + li * 8) = dsq;
            }
        };
        row_body(0, pr0, mk0);
        row_body(1, pr1, mk1);
        row_body(2, pr2, mk2);
        row_body(3, pr3, mk3);
    }
    stamp(5);
    lds_barrier();
    stamp(6);
    // ---- dQ = bf16(dS' k') (rows: query positions), dK = bf16(dS'^T q') (rows: key positions)
    if (wave < mt) {
        f32x4_t accq[4], acck[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) accq[n] = acck[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < S / 32; ++ks) {
            const bf16x8_t aq = frag_rows(Ps, kRowS, m0, ks * 32, lane), ak = frag_cols(Ps, kRowS, m0, ks * 32, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                accq[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, frag_cols(Ks, kRowD, n * 16, ks * 32, lane), accq[n], 0, 0, 0);
                acck[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, frag_cols(Qs, kRowD, n * 16, ks * 32, lane), acck[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            store_tile(Gs, kRowD, m0, n * 16, accq[n], lane);      // (g and v' have left these tiles)
            store_tile(Vs, kRowD, m0, n * 16, acck[n], lane);
        }
    }
    stamp(7);
    lds_barrier();
    stamp(8);
    // ---- dQ, dK, dV through their projections' backward-pre quantizers where one rides along (results held in registers: every
    // global store of the launch is issued at the very end, where nothing waits behind it); per-thread column partials
    uint32_t gamax[3] = {0u, 0u, 0u};
    float col[3][8];
    uint4 zq[kVecIters][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) col[t][e] = 0.0f;
#pragma unroll
    for (int i = 0; i < kVecIters; ++i) {
        const int id = tid + i * kThreads;
#pragma unroll
        for (int t = 0; t < 3; ++t) zq[i][t] = uint4{0u, 0u, 0u, 0u};
        if (id < S * 8) {
            const int r = id >> 3, c = id & 7;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (!a.gs[t].out) continue;
                const unsigned char *tile = t == 0 ? Gs : (t == 1 ? Vs : Ds);
                const uint4 y = *(const uint4 *)(tile + r * kRowD + c * 16);
                const UniformDiv dv(gsc[t]);
                const uint4 z = chain_apply<KIND>(y, gsc[t], dv, rnd, gamax[t]);
                zq[i][t] = z;
                const uint32_t zw[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    col[t][2 * j] += bf_lo(zw[j]);
                    col[t][2 * j + 1] += bf_hi(zw[j]);
                }
            }
        }
    }
    stamp(9);
    // ---- the five amax slots and the bias gradients: column sums of the quantized gradients -- inside the workgroup the 64 row lanes of
    // a column in row order (fp32); across the batch 64-bit fixed-point atomic adds and a ticket, as qt_fake_quant_chain_bf16 does
    // (integer addition is associative: the result does not depend on the arrival order, and no cross-XCD fence is needed).  All the
    // launch's atomics are issued together, one wait, then the tickets: two round trips, not one per step
    const bool sums = a.acc && (a.gs[0].colsum || a.gs[1].colsum || a.gs[2].colsum);
    float *s_col = (float *)Ks;                                // [3][64 row lanes][65]: k', q' and the scores' tile are free now
    {
        const uint32_t am5[5] = {amax[0], amax[1], gamax[0], gamax[1], gamax[2]};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint32_t m = wave_max_u32(am5[i]);
            if (lane == 0) s_amax[i][wave] = m;
        }
        if (sums) {
            const int rl = tid >> 3, c8 = (tid & 7) * 8;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) s_col[(t * 64 + rl) * 65 + c8 + e] = col[t][e];
        }
    }
    lds_barrier();
    const int t = tid >> 6, cc = tid & 63;                     // waves 0..2: one thread per (tensor, column); wave 3: the amax slots
    uint16_t *cs = t == 0 ? a.gs[0].colsum : (t == 1 ? a.gs[1].colsum : (t == 2 ? a.gs[2].colsum : nullptr));
    const float tsc = t == 0 ? gsc[0] : (t == 1 ? gsc[1] : gsc[2]);
    int E = 0;
    (void)frexpf(a.colsum_max * tsc, &E);
    const FxPlan fxp = fx_plan((long)a.B * S, a.B);           // one arrival per batch element, S rows each
    if (sums && t < 3 && cs) {
        float part = 0.0f;
        for (int r = 0; r < 64; ++r) part += s_col[(t * 64 + r) * 65 + cc];
        const long long fx = fx_encode(part, E, fxp);          // NaN / Inf: the poison term (qt_chain.h)
        (void)__hip_atomic_fetch_add(a.acc + ((long)t * a.H + h) * 64 + cc, fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 3 && cc < 5) {
        uint32_t *slot = cc == 0 ? a.fq[0].amax : (cc == 1 ? a.fq[1].amax : (cc == 2 ? a.gs[0].amax : (cc == 3 ? a.gs[1].amax : a.gs[2].amax)));
        uint32_t m = 0u;
#pragma unroll
        for (int k = 0; k < kThreads / 64; ++k) {
            const uint32_t v = cc == 0 ? s_amax[0][k] : (cc == 1 ? s_amax[1][k] : (cc == 2 ? s_amax[2][k] : (cc == 3 ? s_amax[3][k] : s_amax[4][k])));
            m = m > v ? m : v;
        }
        if (slot && m != 0u) atomicMax(slot, m);
    }
    if (sums) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's atomics are performed before its ticket is drawn
        lds_barrier();
        if (tid < 3) {
            const bool want = tid == 0 ? a.gs[0].colsum != nullptr : (tid == 1 ? a.gs[1].colsum != nullptr : a.gs[2].colsum != nullptr);
            s_old[tid] = want ? __hip_atomic_fetch_add(a.ticket + tid * a.H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        }
        lds_barrier();
        if (t < 3 && cs && s_old[t] == (unsigned)a.B - 1u) {   // the batch's last arrival for this (tensor, head): read, re-zero, round
            if (cc == 0) __hip_atomic_store(a.ticket + t * a.H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long fx = __hip_atomic_exchange(a.acc + ((long)t * a.H + h) * 64 + cc, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cs[h * 64 + cc] = bf16_bits(fx_decode(fx, E, fxp));
        }
    }
    // ---- everything leaves (the gradient tiles are still in LDS: Gs, Vs, Ds are not touched by the column sums)
#pragma unroll
    for (int i = 0; i < kVecIters; ++i) {
        const int id = tid + i * kThreads;
        if (id < S * 8) {
            const int r = id >> 3, c = id & 7;
            const long off = (((long)b * S + r) * a.H + h) * kD + c * 8;
            *(uint4 *)(a.dq + off) = *(const uint4 *)(Gs + r * kRowD + c * 16);
            *(uint4 *)(a.dk + off) = *(const uint4 *)(Vs + r * kRowD + c * 16);
            *(uint4 *)(a.dv + off) = *(const uint4 *)(Ds + r * kRowD + c * 16);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (a.gs[t].out) *(uint4 *)(a.gs[t].out + off) = zq[i][t];
        }
    }
    stamp(10);
}

inline int launch_rc() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

template <typename ARGS, typename F_ROWS, typename F_SAT, typename F_INT>
int dispatch(const qt_format *fmt, const uint16_t *lut, F_ROWS rows, F_SAT sat, F_INT in) {
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut || !(fmt->p1 & 1)) return QT_ERR_BAD_DTYPE;
            rows();
            break;
        case QT_FMT_FP_SAT: sat(); break;
        case QT_FMT_INT: in(); break;
        default: return QT_ERR_BAD_DTYPE;
    }
    return launch_rc();
}

bool shape_ok(long batch, int heads, int positions, int head_dim) {
    return batch > 0 && heads > 0 && batch * heads <= 0x7FFFFFFF && head_dim == kD && positions >= 32 && positions <= kSMax && positions % 32 == 0;
}

}  // namespace

extern "C" {

int qt_attention_train_supported(long batch, int heads, int positions, int head_dim) { return shape_ok(batch, heads, positions, head_dim) ? 1 : 0; }

int qt_attention_train_bf16(const uint16_t *q_dev, const uint16_t *k_dev, const uint16_t *v_dev, long stride_b, long stride_s, long stride_h,
                            const uint16_t *mask_dev, long mask_sb, long mask_sh, long mask_sq, const qt_chain_stage *fqs, uint16_t *probs_dev,
                            uint16_t *out_dev, const uint8_t *drop_keep_dev, float drop_scale, long batch, int heads, int positions, int head_dim,
                            float scaling, const qt_format *fmt, const uint16_t *lut_dev, void *stream) {
    if (!q_dev || !k_dev || !v_dev || !fqs || !probs_dev || !out_dev || !fmt || !shape_ok(batch, heads, positions, head_dim)) return QT_ERR_BAD_ARG;
    for (int i = 0; i < 4; ++i)
        if (!fqs[i].out_dev) return QT_ERR_BAD_ARG;
    uintptr_t al = (uintptr_t)q_dev | (uintptr_t)k_dev | (uintptr_t)v_dev | (uintptr_t)mask_dev | (uintptr_t)probs_dev | (uintptr_t)out_dev;
    for (int i = 0; i < 5; ++i) al |= (uintptr_t)fqs[i].out_dev;
    if ((al & 15u) || ((stride_b | stride_s | stride_h) & 7) || (mask_dev && ((mask_sb | mask_sh | mask_sq) & 7))) return QT_ERR_UNALIGNED;
    AttnTrainFwdArgs a{};
    a.q = q_dev; a.k = k_dev; a.v = v_dev;
    a.qq = fqs[0].out_dev; a.kq = fqs[1].out_dev; a.vq = fqs[2].out_dev;
    a.sb = stride_b; a.ss = stride_s; a.sh = stride_h;
    a.mask = mask_dev; a.msb = mask_sb; a.msh = mask_sh; a.msq = mask_sq;
    a.probs = probs_dev; a.pq = fqs[3].out_dev;
    a.out = out_dev; a.oq = fqs[4].out_dev;
    a.H = heads; a.S = positions; a.scaling = scaling;
    if ((uintptr_t)drop_keep_dev & 7u) return QT_ERR_UNALIGNED;
    a.drop = drop_keep_dev; a.drop_scale = drop_scale;
    for (int i = 0; i < 5; ++i) a.fq[i] = FqDev{fqs[i].scale_f32_dev, fqs[i].amax_bits_dev};
#ifdef QT_TUNING_BUILD
    if (const char *e = getenv("QT_AT_STAMPS")) a.dbg = (unsigned long long *)strtoull(e, nullptr, 16);
#endif
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(batch * heads);
    return dispatch<AttnTrainFwdArgs>(
        fmt, lut_dev, [&] { attn_train_fwd_kernel<kFmtRows><<<grid, kThreads, 0, st>>>(a, *fmt, lut_dev); },
        [&] { attn_train_fwd_kernel<QT_FMT_FP_SAT><<<grid, kThreads, 0, st>>>(a, *fmt, lut_dev); },
        [&] { attn_train_fwd_kernel<QT_FMT_INT><<<grid, kThreads, 0, st>>>(a, *fmt, lut_dev); });
}

size_t qt_attention_train_backward_ws_bytes(int heads) {
    return heads > 0 ? (size_t)3 * heads * kD * sizeof(long long) + (size_t)3 * heads * sizeof(unsigned int) : 0;
}

int qt_attention_train_backward_bf16(const uint16_t *grad_out_dev, const uint16_t *qq_dev, const uint16_t *kq_dev, const uint16_t *vq_dev,
                                     long stride_b, long stride_s, long stride_h, const uint16_t *probs_dev, const uint16_t *pq_dev,
                                     const qt_chain_stage *fqs, uint16_t *grad_scores_dev, uint16_t *grad_q_dev, uint16_t *grad_k_dev,
                                     uint16_t *grad_v_dev, const qt_chain_stage *grad_fqs, uint16_t *const *colsum_out_devs, float colsum_max,
                                     void *ws_dev, size_t ws_bytes, const uint8_t *drop_keep_dev, float drop_scale, long batch, int heads,
                                     int positions, int head_dim, float scaling, const qt_format *fmt, const uint16_t *lut_dev, void *stream) {
    if (!grad_out_dev || !qq_dev || !kq_dev || !vq_dev || !probs_dev || !pq_dev || !fqs || !grad_q_dev || !grad_k_dev || !grad_v_dev || !fmt ||
        !shape_ok(batch, heads, positions, head_dim))
        return QT_ERR_BAD_ARG;
    const uintptr_t al = (uintptr_t)grad_out_dev | (uintptr_t)qq_dev | (uintptr_t)kq_dev | (uintptr_t)vq_dev | (uintptr_t)probs_dev |
                         (uintptr_t)pq_dev | (uintptr_t)grad_q_dev | (uintptr_t)grad_k_dev | (uintptr_t)grad_v_dev | (uintptr_t)grad_scores_dev |
                         (uintptr_t)fqs[0].out_dev | (uintptr_t)fqs[1].out_dev;
    if ((al & 15u) || ((stride_b | stride_s | stride_h) & 7)) return QT_ERR_UNALIGNED;
    AttnTrainBwdArgs a{};
    a.gy = grad_out_dev; a.qq = qq_dev; a.kq = kq_dev; a.vq = vq_dev;
    a.sb = stride_b; a.ss = stride_s; a.sh = stride_h;
    a.probs = probs_dev; a.pq = pq_dev;
    a.dq = grad_q_dev; a.dk = grad_k_dev; a.dv = grad_v_dev;
    a.g_out = fqs[0].out_dev; a.ds_out = grad_scores_dev; a.dsq_out = fqs[1].out_dev;
    a.H = heads; a.S = positions; a.scaling = scaling;
    for (int i = 0; i < 2; ++i) a.fq[i] = FqDev{fqs[i].scale_f32_dev, fqs[i].amax_bits_dev};
    a.B = (int)batch;
    if ((uintptr_t)drop_keep_dev & 7u) return QT_ERR_UNALIGNED;
    a.drop = drop_keep_dev; a.drop_scale = drop_scale;
    bool sums = false;
    for (int i = 0; i < 3; ++i) {
        if (!grad_fqs || !grad_fqs[i].out_dev) continue;
        if ((uintptr_t)grad_fqs[i].out_dev & 15u) return QT_ERR_UNALIGNED;
        a.gs[i] = GradStage{grad_fqs[i].scale_f32_dev, grad_fqs[i].amax_bits_dev, grad_fqs[i].out_dev, colsum_out_devs ? colsum_out_devs[i] : nullptr};
        sums = sums || a.gs[i].colsum;
    }
    if (sums) {
        if (!ws_dev || ws_bytes < qt_attention_train_backward_ws_bytes(heads) || !(colsum_max > 0.0f) || !(colsum_max < 3.0e38f)) return QT_ERR_BAD_ARG;
        if ((uintptr_t)ws_dev & 15u) return QT_ERR_UNALIGNED;
        a.colsum_max = colsum_max;
        a.acc = (long long *)ws_dev;
        a.ticket = (unsigned int *)((char *)ws_dev + (size_t)3 * heads * kD * sizeof(long long));
    }
#ifdef QT_TUNING_BUILD
    if (const char *e = getenv("QT_AT_STAMPS")) a.dbg = (unsigned long long *)strtoull(e, nullptr, 16);
#endif
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(batch * heads);
    return dispatch<AttnTrainBwdArgs>(
        fmt, lut_dev, [&] { attn_train_bwd_kernel<kFmtRows><<<grid, kThreads, 0, st>>>(a, *fmt, lut_dev); },
        [&] { attn_train_bwd_kernel<QT_FMT_FP_SAT><<<grid, kThreads, 0, st>>>(a, *fmt, lut_dev); },
        [&] { attn_train_bwd_kernel<QT_FMT_INT><<<grid, kThreads, 0, st>>>(a, *fmt, lut_dev); });
}

}  // extern "C"
