// qt_embedding_train.hip -- the weight gradient of nn.Embedding inside a TRAINING step (BASELINE configs[4]: RoBERTa's word / position /
// token-type tables, 2048 tokens), bit for bit torch's, without its serial walk.
//
// torch (embedding_dense_backward, the path for <= 3072 indices without scale_grad_by_freq: embedding_backward_feature_kernel) lets ONE
// workgroup per 64 features walk ALL indices in chunks of 16 rows, two barriers per chunk: 128 dependent steps for 2048 tokens --
// 134 us per table whatever its size, three tables per step.  What it computes per feature is
//     for each chunk c of 16 consecutive tokens, for each distinct index i in it (rows with the padding index excluded):
//         s = fp32 sum of that chunk's gradient rows with index i, in row order;   W[i] = bf16(float(W[i]) + float(bf16(s)))
// i.e. a fold, in chunk order, of per-chunk partial sums rounded to bf16.  Two launches reproduce exactly that:
//     embed_partials_kernel   one workgroup per chunk: the partial sum of every chunk leader (first row of its index in the chunk), bf16
//     embed_fold_kernel       one workgroup per token that is the FIRST occurrence of its index: the fold over that index's chunk
//                             leaders in ascending order, starting from zero
// grad_weight must arrive zero-filled (rows no token names stay zero), exactly as torch allocates it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_device.h"
#include "qt_chain.h"

namespace {

constexpr int kChunk = 16;             // rows per chunk: torch's BLOCKDIMY on ROCm
constexpr int kEmbBlock = 128;         // threads: one 16-byte vector (8 features) each
constexpr int kAhead = 16;             // partial rows a fold keeps in flight

__global__ __launch_bounds__(kEmbBlock) void embed_partials_kernel(const uint4 *__restrict__ grad, const long *__restrict__ ids, long n, int nvec,
                                                                   long pad, uint4 *__restrict__ part) {
    const int v = blockIdx.y * kEmbBlock + threadIdx.x;
    const long r0 = (long)blockIdx.x * kChunk;
    long id[kChunk];
    uint4 g[kChunk];
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
        id[r] = r0 + r < n ? ids[r0 + r] : pad;                // (rows past the end behave like padding rows: excluded)
        g[r] = (r0 + r < n && v < nvec && id[r] != pad) ? grad[(r0 + r) * nvec + v] : uint4{0u, 0u, 0u, 0u};
    }
    if (v >= nvec) return;
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
        if (id[r] == pad) continue;
        bool leader = true;
#pragma unroll
        for (int q = 0; q < r; ++q) leader = leader && (id[q] != id[r]);
        if (!leader) continue;                                  // (uniform over the workgroup: depends on the indices only)
        const uint32_t w[4] = {g[r].x, g[r].y, g[r].z, g[r].w};
        float s[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[2 * j] = bf_lo(w[j]); s[2 * j + 1] = bf_hi(w[j]); }
#pragma unroll
        for (int q = r + 1; q < kChunk; ++q) {
            if (id[q] != id[r]) continue;
            const uint32_t x[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[2 * j] += bf_lo(x[j]); s[2 * j + 1] += bf_hi(x[j]); }
        }
        part[(r0 + r) * nvec + v] = uint4{pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]), pack_bf16x2(s[4], s[5]), pack_bf16x2(s[6], s[7])};
    }
}

__global__ __launch_bounds__(kEmbBlock) void embed_fold_kernel(const uint4 *__restrict__ part, const long *__restrict__ ids, long n, int nvec, long pad,
                                                               uint4 *__restrict__ grad_weight) {
    __shared__ unsigned int s_bits[96];                         // occurrences of this token's index among rows i .. n - 1 (n <= 3072)
    __shared__ int s_ids[3072];                                 // the indices (table rows fit an int), staged once: every scan below reads LDS
    __shared__ int s_seen;
    const long i = blockIdx.x;
    const long idx = ids[i];
    if (idx == pad) return;
    if (threadIdx.x == 0) s_seen = 0;
    for (int w = threadIdx.x; w < 96; w += kEmbBlock) s_bits[w] = 0u;
#pragma unroll 8
    for (int j = threadIdx.x; j < (int)n; j += kEmbBlock) s_ids[j] = (int)ids[j];
    __syncthreads();
    bool seen = false;
    for (int j = threadIdx.x; j < (int)i; j += kEmbBlock) seen = seen || (s_ids[j] == (int)idx);
    if (seen) s_seen = 1;
    for (int j = (int)i + threadIdx.x; j < (int)n; j += kEmbBlock)
        if (s_ids[j] == (int)idx) atomicOr(&s_bits[j >> 5], 1u << (j & 31));      // (absolute row numbers: a chunk is an aligned half word)
    __syncthreads();
    if (s_seen) return;                                         // an earlier token owns this index
    // every chunk's leader (its first row with this index), in parallel: chunk c = bits 16 c .. 16 c + 15
    __shared__ int s_lead[3072 / kChunk];
    const int c_first = (int)(i / kChunk), c_end = (int)((n + kChunk - 1) / kChunk);
    for (int c = c_first + threadIdx.x; c < c_end; c += kEmbBlock) {
        const unsigned int hw = (s_bits[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu;
        s_lead[c] = hw ? c * kChunk + (__ffs(hw) - 1) : -1;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < nvec; v += kEmbBlock) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = 0.0f;
        for (int c0 = c_first; c0 < c_end; c0 += kAhead) {      // sixteen chunks' partial rows in flight, folded in chunk order
            uint4 p[kAhead];
            int lead[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                lead[u] = c0 + u < c_end ? s_lead[c0 + u] : -1;
                if (lead[u] >= 0) p[u] = part[(long)lead[u] * nvec + v];
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                if (lead[u] < 0) continue;
                const uint32_t pw[4] = {p[u].x, p[u].y, p[u].z, p[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t r = pack_bf16x2(w[2 * k] + bf_lo(pw[k]), w[2 * k + 1] + bf_hi(pw[k]));      // W += s, in bf16
                    w[2 * k] = bf_lo(r);
                    w[2 * k + 1] = bf_hi(r);
                }
            }
        }
        grad_weight[idx * nvec + v] = uint4{pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]), pack_bf16x2(w[4], w[5]), pack_bf16x2(w[6], w[7])};
    }
}

}  // namespace

extern "C" int qt_embedding_backward_bf16(const uint16_t *grad_dev, const long *ids_dev, long n, long cols, long padding_idx, long num_rows,
                                          uint16_t *partials_dev, uint16_t *grad_weight_dev, void *stream) {
    if (n == 0 || cols == 0) return QT_OK;
    if (!grad_dev || !ids_dev || !partials_dev || !grad_weight_dev || n < 0 || n > 3072 || cols < 8 || cols % 8 || num_rows < 1) return QT_ERR_BAD_ARG;
    if (((uintptr_t)grad_dev | (uintptr_t)partials_dev | (uintptr_t)grad_weight_dev) & 15u || ((uintptr_t)ids_dev & 7u)) return QT_ERR_UNALIGNED;
    const int nvec = (int)(cols / 8);
    hipStream_t st = (hipStream_t)stream;
    const dim3 g1((unsigned)((n + kChunk - 1) / kChunk), (unsigned)((nvec + kEmbBlock - 1) / kEmbBlock));
    embed_partials_kernel<<<g1, kEmbBlock, 0, st>>>((const uint4 *)grad_dev, ids_dev, n, nvec, padding_idx, (uint4 *)partials_dev);
    embed_fold_kernel<<<(unsigned)n, kEmbBlock, 0, st>>>((const uint4 *)partials_dev, ids_dev, n, nvec, padding_idx, (uint4 *)grad_weight_dev);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}
