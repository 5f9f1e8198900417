// qt_optimizer.hip -- the end of a fine-tuning step (H3: run_glue_no_trainer.py:655-668 upstream): clip_grad_norm_(max_norm) and the
// AdamW update of every parameter tensor, four launches for the whole model instead of torch's ~33 (8 + 8 + 8 multi-tensor launches,
// a norm clean-up and eight scalar kernels for the coefficient).
//
//   sumsq     one workgroup per 8192-element chunk of one gradient: the chunk's sum of squares (fp32), in a fixed order
//   norms     one wave per tensor: the chunks of a tensor added in a fixed order in fp64, the norm rounded to bf16 as
//             torch._foreach_norm returns it for bf16 gradients
//   finalize  one workgroup: total = bf16(sqrt(sum norm_t^2)); the coefficient exactly as
//             torch.nn.utils.clip_grad_norm_ forms it in bf16 -- t = bf16(total + 1e-6), r = bf16(1 / t), c = min(bf16(r * max_norm), 1);
//             every tensor's step count + 1 and its two bias corrections
//   apply     one workgroup per chunk: g' = bf16(g * c) (the in-place _foreach_mul_ of the clip; g itself is left alone -- the loop
//             drops the gradients right after the step), then torch's fused AdamW arithmetic (torch 2.10, ATen/native/cuda/
//             fused_adam_utils.cuh:27-98, ADAM_MODE::ADAMW): hyper-parameters in double, state in fp32, one rounding to bf16 per
//             stored value.  Reads param / grad / exp_avg / exp_avg_sq once, writes param / exp_avg / exp_avg_sq once: 14 B per element.
//
// The third-party arithmetic (torch's own optimizer) is restated in oracle/optimizer_oracle.py; the parity tests compare the three
// launches with that restatement and with torch.optim.AdamW(fused=True) behind torch's own clip_grad_norm_ on the same tensors.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "qt_device.h"

namespace {

constexpr int kOptBlock = 256, kOptVecs = 4, kOptChunk = kOptBlock * kOptVecs * 8;      // 8192 elements per workgroup

inline int opt_launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : (int)e;
}

__device__ __forceinline__ float bf_at(uint32_t w, int h) { return h ? qt_u2f(w & 0xFFFF0000u) : qt_u2f(w << 16); }
__device__ __forceinline__ float bf16_round(float x) { return qt_u2f(pack_bf16x2(x, 0.0f) << 16); }

// ---- sum of squares of one chunk -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kOptBlock) void adamw_sumsq_kernel(const qt_adamw_tensor *__restrict__ tensors, const int32_t *__restrict__ chunk_tensor,
                                                                float *__restrict__ partial) {
    const int t = chunk_tensor[blockIdx.x];
    const qt_adamw_tensor T = tensors[t];
    const long begin = ((long)blockIdx.x - T.first_chunk) * kOptChunk;
    const long n = min((long)kOptChunk, T.numel - begin);
    const uint16_t *g = (const uint16_t *)T.grad_dev + begin;
    float s = 0.0f;
    if (((uintptr_t)g & 15u) == 0) {
        uint4 v[kOptVecs];
#pragma unroll
        for (int u = 0; u < kOptVecs; ++u) {
            const long i = ((long)u * kOptBlock + threadIdx.x) * 8;
            v[u] = i + 8 <= n ? ((const uint4 *)g)[i / 8] : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < kOptVecs; ++u) {
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_at(w[j], 0), b = bf_at(w[j], 1);
                s += a * a;
                s += b * b;
            }
        }
        for (long i = (n / 8) * 8 + threadIdx.x; i < n; i += kOptBlock) {          // (the last chunk's ragged end)
            const float a = qt_u2f((uint32_t)g[i] << 16);
            s += a * a;
        }
    } else {
        for (long i = threadIdx.x; i < n; i += kOptBlock) {
            const float a = qt_u2f((uint32_t)g[i] << 16);
            s += a * a;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    __shared__ float s_w[kOptBlock / 64];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// ---- norms, coefficient, step counts ---------------------------------------------------------------------------------------------------
struct OptScalars {
    float coef;               // what every gradient is multiplied by (bf16 value); 1 when max_norm <= 0
    float total_norm;         // bf16 value
};

// one wave per tensor: the chunks of the tensor added in a fixed order (lanes stride over them, a butterfly in fp64), the norm rounded to
// bf16 as torch._foreach_norm returns it for bf16 gradients; its square goes to the finalize launch
__global__ __launch_bounds__(64) void adamw_tensor_norms_kernel(const qt_adamw_tensor *__restrict__ tensors, int ntensors, long nchunks,
                                                               const float *__restrict__ partial, double *__restrict__ norm_sq) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const long c0 = tensors[t].first_chunk, c1 = t + 1 < ntensors ? tensors[t + 1].first_chunk : nchunks;
    double s = 0.0;
    long c = c0 + lane;
    for (; c + 7 * 64 < c1; c += 8 * 64) {                            // eight loads in flight (the embedding matrix alone has ~4700 chunks)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[c + u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; c < c1; c += 64) s += (double)partial[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float norm = bf16_round(sqrtf((float)s));                   // fp32 sum, sqrt, result in the gradients' dtype
    if (lane == 0) norm_sq[t] = (double)norm * (double)norm;
}

__global__ __launch_bounds__(1024) void adamw_finalize_kernel(qt_adamw_tensor *__restrict__ tensors, int ntensors, const double *__restrict__ norm_sq,
                                                              float max_norm, float2 *__restrict__ bias_corr, OptScalars *__restrict__ scalars,
                                                              float *__restrict__ total_norm_out, int do_norm, int do_steps) {
    __shared__ double s_sq[1024];
    const bool clip = do_norm && max_norm > 0.0f;
    if (clip) {
        double acc = 0.0;                                            // thread i: tensors i, i + 1024, ... in that order
        for (int t = threadIdx.x; t < ntensors; t += 1024) acc += norm_sq[t];
        s_sq[threadIdx.x] = acc;
    }
    // step counts and bias corrections (fused_adam_utils.cuh:128-136: 1 - pow(beta, step) in double, sqrt of the second)
    for (int t = threadIdx.x; do_steps && t < ntensors; t += 1024) {
        double step = tensors[t].step;
        if (tensors[t].step_dev) {
            const float s1 = *tensors[t].step_dev + 1.0f;           // torch: _foreach_add_(state_steps, 1) in front of the kernel
            *tensors[t].step_dev = s1;
            step = (double)s1;
        }
        const double bc1 = 1.0 - pow(tensors[t].beta1, step), bc2 = 1.0 - pow(tensors[t].beta2, step);
        bias_corr[t] = float2{(float)bc1, (float)sqrt(bc2)};
    }
    __syncthreads();
    if (clip) {                                                      // a fixed-order tree over the 1024 slots
        for (int half = 512; half >= 1; half >>= 1) {
            if ((int)threadIdx.x < half) s_sq[threadIdx.x] += s_sq[threadIdx.x + half];
            __syncthreads();
        }
    }
    if (threadIdx.x == 0 && do_norm) {
        float coef = 1.0f, total = 0.0f;
        if (clip) {
            total = bf16_round(sqrtf((float)s_sq[0]));               // linalg.vector_norm over the stacked bf16 norms
            const float t1 = bf16_round(total + 1e-6f);              // clip_grad.py: max_norm / (total_norm + 1e-6), python scalar / tensor
            const float r = bf16_round(1.0f / t1);                   //   = tensor.reciprocal() * max_norm
            coef = bf16_round(r * max_norm);
            coef = coef > 1.0f ? 1.0f : coef;                        // torch.clamp(max=1.0): NaN stays NaN
        }
        scalars->coef = coef;
        scalars->total_norm = total;
        if (total_norm_out) *total_norm_out = total;
    }
}

// ---- the update ------------------------------------------------------------------------------------------------------------------------
struct AdamHyper {
    double lr, beta1, beta2, eps, wd;
    float bc1, bc2_sqrt, coef;
    bool clip;
};

// One element through torch's adam_math (ADAMW, no amsgrad, no maximize, no grad scaler); every mixed float / double step as written there.
__device__ __forceinline__ void adamw_one(float &param, float grad, float &exp_avg, float &exp_avg_sq, const AdamHyper &h) {
    if (h.clip) grad = bf16_round(grad * h.coef);                  // the clip's in-place multiplication, rounded to the gradient's dtype
    if (h.wd != 0.0) param = (float)((double)param - h.lr * h.wd * (double)param);
    exp_avg = (float)(h.beta1 * (double)exp_avg + (1.0 - h.beta1) * (double)grad);
    exp_avg_sq = (float)(h.beta2 * (double)exp_avg_sq + (1.0 - h.beta2) * (double)grad * (double)grad);
    const float step_size = (float)(h.lr / (double)h.bc1);
    const float denom = (float)((double)(sqrtf(exp_avg_sq) / h.bc2_sqrt) + h.eps);
    param -= step_size * exp_avg / denom;
}

__global__ __launch_bounds__(kOptBlock) void adamw_apply_kernel(const qt_adamw_tensor *__restrict__ tensors, const int32_t *__restrict__ chunk_tensor,
                                                                const float2 *__restrict__ bias_corr, const OptScalars *__restrict__ scalars, int clip) {
    const int t = chunk_tensor[blockIdx.x];
    const qt_adamw_tensor T = tensors[t];
    const long begin = ((long)blockIdx.x - T.first_chunk) * kOptChunk;
    const long n = min((long)kOptChunk, T.numel - begin);
    uint16_t *p = (uint16_t *)T.param_dev + begin, *m = (uint16_t *)T.exp_avg_dev + begin, *v = (uint16_t *)T.exp_avg_sq_dev + begin;
    const uint16_t *g = (const uint16_t *)T.grad_dev + begin;
    const float2 bc = bias_corr[t];
    AdamHyper h;
    h.lr = T.lr_dev ? (double)*T.lr_dev : T.lr;
    h.beta1 = T.beta1; h.beta2 = T.beta2; h.eps = T.eps; h.wd = T.weight_decay;
    h.bc1 = bc.x; h.bc2_sqrt = bc.y;
    h.clip = clip != 0;
    h.coef = clip ? scalars->coef : 1.0f;
    const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0;
    long done = 0;
    if (aligned) {
        uint4 pv[kOptVecs], gv[kOptVecs], mv[kOptVecs], vv[kOptVecs];
#pragma unroll
        for (int u = 0; u < kOptVecs; ++u) {
            const long i = ((long)u * kOptBlock + threadIdx.x) * 8;
            if (i + 8 <= n) {
                pv[u] = ((const uint4 *)p)[i / 8];
                gv[u] = ((const uint4 *)g)[i / 8];
                mv[u] = ((const uint4 *)m)[i / 8];
                vv[u] = ((const uint4 *)v)[i / 8];
            }
        }
#pragma unroll
        for (int u = 0; u < kOptVecs; ++u) {
            const long i = ((long)u * kOptBlock + threadIdx.x) * 8;
            if (i + 8 > n) continue;
            const uint32_t pw[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w}, gw[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
            const uint32_t mw[4] = {mv[u].x, mv[u].y, mv[u].z, mv[u].w}, vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
            uint32_t po[4], mo[4], vo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pp[2], mm[2], ss[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    pp[hh] = bf_at(pw[j], hh);
                    mm[hh] = bf_at(mw[j], hh);
                    ss[hh] = bf_at(vw[j], hh);
                    adamw_one(pp[hh], bf_at(gw[j], hh), mm[hh], ss[hh], h);
                }
                po[j] = pack_bf16x2(pp[0], pp[1]);
                mo[j] = pack_bf16x2(mm[0], mm[1]);
                vo[j] = pack_bf16x2(ss[0], ss[1]);
            }
            ((uint4 *)p)[i / 8] = uint4{po[0], po[1], po[2], po[3]};
            ((uint4 *)m)[i / 8] = uint4{mo[0], mo[1], mo[2], mo[3]};
            ((uint4 *)v)[i / 8] = uint4{vo[0], vo[1], vo[2], vo[3]};
        }
        done = (n / 8) * 8;
    }
    for (long i = done + threadIdx.x; i < n; i += kOptBlock) {       // ragged end, or a tensor whose storage is not 16-byte aligned
        float pp = qt_u2f((uint32_t)p[i] << 16), mm = qt_u2f((uint32_t)m[i] << 16), ss = qt_u2f((uint32_t)v[i] << 16);
        adamw_one(pp, qt_u2f((uint32_t)g[i] << 16), mm, ss, h);
        p[i] = (uint16_t)(pack_bf16x2(pp, 0.0f) & 0xFFFFu);
        m[i] = (uint16_t)(pack_bf16x2(mm, 0.0f) & 0xFFFFu);
        v[i] = (uint16_t)(pack_bf16x2(ss, 0.0f) & 0xFFFFu);
    }
}

}  // namespace

extern "C" long qt_clip_adamw_plan(qt_adamw_tensor *tensors_host, int ntensors, int32_t *chunk_tensor_out, long capacity) {
    if (!tensors_host || ntensors < 0) return QT_ERR_BAD_ARG;
    long chunks = 0;
    for (int t = 0; t < ntensors; ++t) {
        if (tensors_host[t].numel < 0) return QT_ERR_BAD_ARG;
        tensors_host[t].first_chunk = chunks;
        const long mine = (tensors_host[t].numel + kOptChunk - 1) / kOptChunk;
        if (chunk_tensor_out)
            for (long c = 0; c < mine && chunks + c < capacity; ++c) chunk_tensor_out[chunks + c] = (int32_t)t;
        chunks += mine;
    }
    return chunks;
}

extern "C" size_t qt_clip_adamw_ws_bytes(int ntensors, long nchunks) {
    if (ntensors < 0 || nchunks < 0) return 0;
    return (size_t)nchunks * sizeof(float) + (size_t)ntensors * (sizeof(float2) + sizeof(double)) + 64;
}

extern "C" int qt_clip_adamw_bf16(qt_adamw_tensor *tensors_dev, const int32_t *chunk_tensor_dev, int ntensors, long nchunks, float max_norm,
                                  float *total_norm_out_dev, void *ws_dev, size_t ws_bytes, int phases, void *stream) {
    if (ntensors == 0) return QT_OK;
    if (!tensors_dev || !chunk_tensor_dev || ntensors < 0 || nchunks < 0 || nchunks > 0x7FFFFFFFl || !ws_dev || !(phases & 3) || max_norm != max_norm)
        return QT_ERR_BAD_ARG;
    if (ws_bytes < qt_clip_adamw_ws_bytes(ntensors, nchunks)) return QT_ERR_BAD_ARG;
    if (((uintptr_t)ws_dev & 15u) || ((uintptr_t)tensors_dev & 7u)) return QT_ERR_UNALIGNED;
    OptScalars *scalars = (OptScalars *)ws_dev;
    float2 *bias_corr = (float2 *)((char *)ws_dev + 64);
    double *norm_sq = (double *)(bias_corr + ntensors);
    float *partial = (float *)(norm_sq + ntensors);
    hipStream_t st = (hipStream_t)stream;
    const bool clip = max_norm > 0.0f;
    // phases: 1 = the norm and the coefficient (left in the workspace; total_norm_out written), 2 = step counts + the update with the
    // coefficient the workspace holds.  3 = both, four launches.  A caller that must look at the norm first (error_if_nonfinite)
    // issues 1, reads total_norm_out, then 2.
    if ((phases & 1) && clip) {
        if (nchunks > 0) adamw_sumsq_kernel<<<(unsigned)nchunks, kOptBlock, 0, st>>>(tensors_dev, chunk_tensor_dev, partial);
        adamw_tensor_norms_kernel<<<(unsigned)ntensors, 64, 0, st>>>(tensors_dev, ntensors, nchunks, partial, norm_sq);
    }
    adamw_finalize_kernel<<<1, 1024, 0, st>>>(tensors_dev, ntensors, norm_sq, max_norm, bias_corr, scalars, total_norm_out_dev, phases & 1,
                                              (phases & 2) ? 1 : 0);
    if ((phases & 2) && nchunks > 0) adamw_apply_kernel<<<(unsigned)nchunks, kOptBlock, 0, st>>>(tensors_dev, chunk_tensor_dev, bias_corr, scalars, clip ? 1 : 0);
    return opt_launch_status();
}
