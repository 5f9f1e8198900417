// qt_train_fanin.hip -- the gradients that meet at one tensor of a TRAINING step, added in one launch, each through the backward
// fake-quantizer that stands in front of it.
//
// A LayerNorm's output feeds several consumers (query / key / value and the residual add; the FFN's first dense layer and the residual
// add).  In the backward pass every consuming Linear's grad_input goes through the Linear's backward quantizer (`--quantize_backprop
// ...,residual`: quantize.py:147-148, 161-163 upstream: one observed E5M2 launch over [tokens, hidden]) and the autograd engine then
// adds the arrivals one by one (one launch each): 4 x (fake-quantizer + add) per layer.  This kernel evaluates
//     sum = (((first + y_0) + y_1) + ...),   y_i = fq_i(x_i)  or  x_i,
// every addition rounded to bf16 as torch's (fp32 add, one rounding) and taken in the order the engine takes them, every fq_i exactly
// qt_fake_quant_bf16 with its own scale and amax slot -- bit for bit the tensors the separate launches produce.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_device.h"
#include "qt_chain.h"

namespace {

constexpr int kFanBlock = 512, kFanMax = 4;

struct FaninArgs {
    const uint4 *first;
    const uint4 *x[kFanMax];
    uint4 *out[kFanMax];
    int fq[kFanMax];
    ChainStageDev st[kChainMax];       // (scale, amax) of item i; out / src unused
    uint4 *sum;
    size_t nvec;
};

__device__ __forceinline__ uint4 add_bf16x8(const uint4 &a, const uint4 &b) {
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    uint32_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = pack_bf16x2(bf_lo(aw[j]) + bf_lo(bw[j]), bf_hi(aw[j]) + bf_hi(bw[j]));
    return uint4{r[0], r[1], r[2], r[3]};
}

template <int KIND, int N>
__global__ __launch_bounds__(kFanBlock) void fanin_kernel(FaninArgs a, qt_format fmt, const uint16_t *__restrict__ lut) {
    __shared__ uint4 s_rows[KIND == kFmtRows ? 512 : 1];
    const Rounder<KIND> rnd = chain_rounder<KIND>(fmt, lut, s_rows, kFanBlock);
    float sc[N];
    uint32_t amax[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sc[i] = (a.fq[i] && a.st[i].scale) ? qt_bf2f(qt_f2bf(*a.st[i].scale)) : 1.0f;
        amax[i] = 0u;
    }
    for (size_t v = (size_t)blockIdx.x * kFanBlock + threadIdx.x; v < a.nvec; v += (size_t)gridDim.x * kFanBlock) {
        uint4 acc = a.first[v];
        uint4 x[N];
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = a.x[i][v];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint4 y = x[i];
            if (a.fq[i]) {
                const UniformDiv dv(sc[i]);
                y = chain_apply<KIND>(x[i], sc[i], dv, rnd, amax[i]);
                if (a.out[i]) a.out[i][v] = y;
            }
            acc = add_bf16x8(acc, y);
        }
        a.sum[v] = acc;
    }
    __shared__ uint32_t s_amax[N][kFanBlock / 64];
    chain_amax_commit<N, kFanBlock>(a.st, amax, s_amax);
}

template <int KIND>
void launch_n(const FaninArgs &a, int n, unsigned grid, const qt_format &fmt, const uint16_t *lut, hipStream_t st) {
    switch (n) {
        case 1: fanin_kernel<KIND, 1><<<grid, kFanBlock, 0, st>>>(a, fmt, lut); break;
        case 2: fanin_kernel<KIND, 2><<<grid, kFanBlock, 0, st>>>(a, fmt, lut); break;
        case 3: fanin_kernel<KIND, 3><<<grid, kFanBlock, 0, st>>>(a, fmt, lut); break;
        default: fanin_kernel<KIND, 4><<<grid, kFanBlock, 0, st>>>(a, fmt, lut); break;
    }
}

}  // namespace

extern "C" int qt_grad_fanin_bf16(const uint16_t *first_dev, const qt_fanin_item *items, int count, uint16_t *sum_dev, size_t n, const qt_format *fmt,
                                  const uint16_t *lut_dev, void *stream) {
    if (n == 0) return QT_OK;
    if (!first_dev || !items || !sum_dev || !fmt || count < 1 || count > kFanMax || n % 8) return QT_ERR_BAD_ARG;
    uintptr_t al = (uintptr_t)first_dev | (uintptr_t)sum_dev;
    FaninArgs a{};
    a.first = (const uint4 *)first_dev; a.sum = (uint4 *)sum_dev; a.nvec = n / 8;
    for (int i = 0; i < count; ++i) {
        if (!items[i].x_dev) return QT_ERR_BAD_ARG;
        al |= (uintptr_t)items[i].x_dev | (uintptr_t)items[i].out_dev;
        a.x[i] = (const uint4 *)items[i].x_dev;
        a.out[i] = (uint4 *)items[i].out_dev;
        a.fq[i] = items[i].fq ? 1 : 0;
        a.st[i] = ChainStageDev{items[i].fq ? items[i].scale_f32_dev : nullptr, items[i].fq ? items[i].amax_bits_dev : nullptr, nullptr, -1};
    }
    if (al & 15u) return QT_ERR_UNALIGNED;
    const size_t want = (a.nvec + kFanBlock - 1) / kFanBlock;
    const unsigned grid = (unsigned)(want < 256 ? want : 256);
    hipStream_t st = (hipStream_t)stream;
    switch (fmt->kind) {
        case QT_FMT_LUT:
            if (!lut_dev || !(fmt->p1 & 1)) return QT_ERR_BAD_DTYPE;
            launch_n<kFmtRows>(a, count, grid, *fmt, lut_dev, st);
            break;
        case QT_FMT_FP_SAT: launch_n<QT_FMT_FP_SAT>(a, count, grid, *fmt, lut_dev, st); break;
        case QT_FMT_INT: launch_n<QT_FMT_INT>(a, count, grid, *fmt, lut_dev, st); break;
        default: return QT_ERR_BAD_DTYPE;
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}
