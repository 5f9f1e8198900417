// qt_value_codes.h -- fq_v(V) as FP8 codes, transposed to [d][key] with the keys of every 128-block permuted into the k-slot order of the
// attention kernel's P.V instruction (slot 16 g + 4 t + e <-> key 16 t + 4 g + e inside each half of 64; csrc/qt_attention_fp8.hip).
// One 256-thread workgroup per (batch * head, block of 128 keys); shared by the stand-alone pass (qt_value_codes_t) and the launch that
// also carries the rotary kernel (qt_rope_fq_value, csrc/qt_model_ops.hip).
#pragma once
#include "qt_device.h"

namespace {

template <bool E5M2, int D>
__device__ __forceinline__ void value_codes_block(uint8_t *tile /* D * 128 bytes of LDS */, const uint16_t *v, uint8_t *vt8, int kb, long bh, int H,
                                                  long Sk, long sb, long sh, long sk, const qt_format &fmt) {
    const int t = threadIdx.x;
    const long b = bh / H, h = bh % H;
#pragma unroll
    for (int it = 0; it < D / 16; ++it) {
        const int vi = it * 256 + t, key = vi / (D / 8), dv = vi % (D / 8);
        const uint4 in = *(const uint4 *)(v + b * sb + h * sh + ((long)kb * 128 + key) * sk + dv * 8);
        uint32_t o[4] = {in.x, in.y, in.z, in.w};
        const uint2 codes = fq8_hw_vec8<E5M2>(o, fmt);
        const int p = (key & 64) | (((key >> 2) & 3) << 4) | (((key >> 4) & 3) << 2) | (key & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[(dv * 8 + j) * 128 + p] = (uint8_t)((j < 4 ? codes.x >> (8 * j) : codes.y >> (8 * (j - 4))) & 0xFFu);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < D / 32; ++it) {
        const int ci = it * 256 + t, d = ci >> 3, ch = ci & 7;
        *(uint4 *)(vt8 + (bh * D + d) * Sk + (long)kb * 128 + ch * 16) = *(const uint4 *)(tile + d * 128 + ch * 16);
    }
}

}  // namespace
