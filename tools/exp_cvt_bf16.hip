// GPU experiment: does gfx950's packed bf16 -> FP8 conversion (v_cvt_scalef32_pk_fp8_bf16 / _bf8_bf16, scale 1.0)
// reproduce the reference's e4m3 / e5m2 value maps on all 65 536 bf16 inputs (as GEMM operand codes: the sign of a
// zero result does not matter for a product)?  Also prints how non-finite inputs and overflow come out.
//   hipcc --offload-arch=gfx950 -O3 -I quantized-training_amd/csrc tools/exp_cvt_bf16.hip -o /tmp/exp_cvt_bf16 && /tmp/exp_cvt_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "qt_formats.h"

typedef short v2s __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

__global__ void k(uint32_t *code_e4, uint32_t *code_e5) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536) return;
    const uint32_t w = (uint32_t)i | (0x3F80u << 16);         // element 0 = pattern i, element 1 = 1.0
    const v2bf src = __builtin_bit_cast(v2bf, w);
    v2s old = {0, 0};
    v2s r4 = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, src, 1.0f, false);
    v2s r5 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(old, src, 1.0f, false);
    code_e4[i] = __builtin_bit_cast(uint32_t, r4);
    code_e5[i] = __builtin_bit_cast(uint32_t, r5);
}

static float dec(uint32_t code, int mbits, int ebits, int bias, bool fn) {   // OCP FP8 byte -> float
    const int s = (code >> 7) & 1, e = (code >> mbits) & ((1 << ebits) - 1), m = code & ((1 << mbits) - 1);
    float v;
    if (fn) { if (e == 15 && m == 7) return NAN; }
    else if (e == 31) return m ? NAN : (s ? -INFINITY : INFINITY);
    if (e == 0) v = ldexpf((float)m, 1 - bias - mbits);
    else v = ldexpf((float)(m + (1 << mbits)), e - bias - mbits);
    return s ? -v : v;
}

int main() {
    uint32_t *d4, *d5;
    hipMalloc(&d4, 65536 * 4); hipMalloc(&d5, 65536 * 4);
    k<<<256, 256>>>(d4, d5);
    std::vector<uint32_t> h4(65536), h5(65536);
    hipMemcpy(h4.data(), d4, 65536 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h5.data(), d5, 65536 * 4, hipMemcpyDeviceToHost);
    int bad4 = 0, bad5 = 0, bad4_fin = 0, bad5_fin = 0, shown = 0, hi_bad = 0;
    for (int i = 0; i < 65536; ++i) {
        const uint32_t img = (uint32_t)i << 16;
        const uint32_t e4 = qt_fp_sat_u32(img, 3, -6, 448.0f), e5 = qt_fp_sat_u32(img, 2, -14, 57344.0f);
        const float g4 = dec(h4[i] & 0xFF, 3, 4, 7, true), g5 = dec(h5[i] & 0xFF, 2, 5, 15, false);
        const float x4 = qt_u2f(e4), x5 = qt_u2f(e5);
        const bool fin = (img & 0x7FFFFFFF) < 0x7F800000;
        const bool ok4 = (x4 != x4 && g4 != g4) || g4 == x4, ok5 = (x5 != x5 && g5 != g5) || g5 == x5;
        if (((h4[i] >> 8) & 0xFF) != 0x38 || ((h5[i] >> 8) & 0xFF) != 0x3C) hi_bad++;
        if (!ok4) { bad4++; bad4_fin += fin; if (shown < 16) { printf("e4m3 in %04x code %02x = %g, expected %g\n", i, h4[i] & 0xFF, g4, x4); shown++; } }
        if (!ok5) { bad5++; bad5_fin += fin; if (shown < 32) { printf("e5m2 in %04x code %02x = %g, expected %g\n", i, h5[i] & 0xFF, g5, x5); shown++; } }
    }
    printf("PKCVT e4m3 mismatches %d (finite inputs %d); e5m2 mismatches %d (finite inputs %d); second element wrong %d\n",
           bad4, bad4_fin, bad5, bad5_fin, hi_bad);
    printf("e4m3: +inf -> %02x, -inf -> %02x, nan -> %02x, 1e9 -> %02x, 449 -> %02x, 464(0x43e8) -> %02x, -tiny -> %02x\n",
           h4[0x7f80] & 0xFF, h4[0xff80] & 0xFF, h4[0x7fc0] & 0xFF, h4[0x4e6e] & 0xFF, h4[0x43e1] & 0xFF, h4[0x43e8] & 0xFF, h4[0x8001] & 0xFF);
    printf("e5m2: +inf -> %02x, -inf -> %02x, nan -> %02x, 1e9 -> %02x, 61440(0x4770) -> %02x, -tiny -> %02x\n",
           h5[0x7f80] & 0xFF, h5[0xff80] & 0xFF, h5[0x7fc0] & 0xFF, h5[0x4e6e] & 0xFF, h5[0x4770] & 0xFF, h5[0x8001] & 0xFF);
    return 0;
}
