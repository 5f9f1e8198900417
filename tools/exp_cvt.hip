// GPU experiment: does the gfx950 hardware fp8 conversion (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32, OCP)
// reproduce the reference's e4m3 / e5m2 value maps on all 65 536 bf16 inputs?
//   hipcc --offload-arch=gfx950 -O3 -I quantized-training_amd/csrc tools/exp_cvt.hip -o /tmp/exp_cvt && /tmp/exp_cvt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "qt_formats.h"

__global__ void k(uint32_t *out_e4, uint32_t *out_e5, uint32_t *raw_e4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536) return;
    float v = qt_u2f((uint32_t)i << 16);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
    float r = __builtin_amdgcn_cvt_f32_fp8(p, 0);
    out_e4[i] = qt_f2u(r);
    raw_e4[i] = (uint32_t)p & 0xFFu;
    int q = __builtin_amdgcn_cvt_pk_bf8_f32(v, v, 0, false);
    float r5 = __builtin_amdgcn_cvt_f32_bf8(q, 0);
    out_e5[i] = qt_f2u(r5);
}

int main() {
    uint32_t *d4, *d5, *dr;
    hipMalloc(&d4, 65536 * 4); hipMalloc(&d5, 65536 * 4); hipMalloc(&dr, 65536 * 4);
    k<<<256, 256>>>(d4, d5, dr);
    std::vector<uint32_t> h4(65536), h5(65536), hr(65536);
    hipMemcpy(h4.data(), d4, 65536 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h5.data(), d5, 65536 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), dr, 65536 * 4, hipMemcpyDeviceToHost);
    int bad4 = 0, bad5 = 0, bad4_fin = 0, bad5_fin = 0, shown = 0;
    for (int i = 0; i < 65536; ++i) {
        uint32_t img = (uint32_t)i << 16;
        uint32_t e4 = qt_fp_sat_u32(img, 3, -6, 448.0f), e5 = qt_fp_sat_u32(img, 2, -14, 57344.0f);
        bool nan_exp4 = (e4 & 0x7FFFFFFF) > 0x7F800000, nan_got4 = (h4[i] & 0x7FFFFFFF) > 0x7F800000;
        bool nan_exp5 = (e5 & 0x7FFFFFFF) > 0x7F800000, nan_got5 = (h5[i] & 0x7FFFFFFF) > 0x7F800000;
        bool finite_in = (img & 0x7FFFFFFF) < 0x7F800000;
        bool ok4 = (nan_exp4 && nan_got4) || (!nan_exp4 && !nan_got4 && (h4[i] == e4 || ((h4[i] | e4) << 1) == 0));
        bool ok5 = (nan_exp5 && nan_got5) || (!nan_exp5 && !nan_got5 && (h5[i] == e5 || ((h5[i] | e5) << 1) == 0));
        if (!ok4) { bad4++; if (finite_in) bad4_fin++; if (shown < 12) { printf("e4m3 in %04x got %08x (raw %02x) exp %08x\n", i, h4[i], hr[i], e4); shown++; } }
        if (!ok5) { bad5++; if (finite_in) bad5_fin++; if (shown < 24) { printf("e5m2 in %04x got %08x exp %08x\n", i, h5[i], e5); shown++; } }
    }
    printf("HWCVT e4m3 mismatches %d (finite inputs %d); e5m2 mismatches %d (finite inputs %d)\n", bad4, bad4_fin, bad5, bad5_fin);
    // sign of zero results
    printf("e4m3(-tiny 0x8001) -> %08x ; e4m3(-0) -> %08x ; e4m3(449.0 0x43e0+1) -> %08x ; e4m3(inf) -> %08x ; e4m3(1e9) -> %08x\n",
           h4[0x8001], h4[0x8000], h4[0x43e1], h4[0x7f80], h4[0x4e6e]);
    return 0;
}
