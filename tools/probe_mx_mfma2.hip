// Probe 2: for 8-bit operands of v_mfma_scale_f32_16x16x128_f8f6f4, which lane's scale governs byte i of lane group g?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
int main() {
    uint8_t ha[64][32], hb[64][32]; int hsa[64], hsb[64];
    v8i *da, *db; v4f* dc; int *dsa, *dsb;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    for (int which = 0; which < 2; ++which) {          // 0: A's scales vary, 1: B's scales vary
        printf("%s scale: governing lane group for (g, byte):\n", which ? "B" : "A");
        for (int g = 0; g < 4; ++g) {
            printf(" g=%d:", g);
            for (int i0 = 0; i0 < 32; i0 += 16) {
                memset(ha, 0x38, sizeof ha);             // 1.0 in e4m3
                memset(hb, 0, sizeof hb);
                for (int n = 0; n < 16; ++n) hb[n + 16 * g][i0 + n] = 0x38;      // column n: one-hot at (g, i0+n)
                for (int l = 0; l < 64; ++l) { hsa[l] = 127 + (which == 0 ? (l >> 4) : 0); hsb[l] = 127 + (which == 1 ? (l >> 4) : 0); }
                if (which == 1) {                        // swap roles: A one-hot rows, B ones
                    memset(hb, 0x38, sizeof hb); memset(ha, 0, sizeof ha);
                    for (int m = 0; m < 16; ++m) ha[m + 16 * g][i0 + m] = 0x38;
                }
                hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
                hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
                k<<<1, 64>>>(da, db, dc, dsa, dsb);
                float hc[64][4]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
                for (int j = 0; j < 16; ++j) {
                    // which == 0: C[0][n=j]; which == 1: C[m=j][0]
                    float v = which == 0 ? hc[j][0] : hc[(j >> 2) * 16 + 0][j & 3];
                    printf(" %d", (int)lround(log2f(v)));
                }
            }
            printf("\n");
        }
    }
    return 0;
}
