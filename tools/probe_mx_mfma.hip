// Probe: operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mx_mfma.hip -o gpurun_out/probe_mx && gpurun_out/probe_mx
// Layout confirmed by this probe (and probe_mx_mfma2.hip): lane l = (row r = l & 15, group g = l >> 4).
//   fp6 / fp4: the lane holds k = 32 g + [0, 32), element i in bits [6i, 6i+6) / [4i, 4i+4) of its registers;
//   fp8:       bytes 0-15 hold k = 16 g + [0, 16), bytes 16-31 hold k = 64 + 16 g + [0, 16);
//   scale:     byte 0 (op_sel 0) of lane (r, g)'s scale register is the E8M0 scale of row r, k in [32 g, 32 g + 32);
//   D:         col = l & 15, row = 4 * (l >> 4) + r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int FA, int FB>
__global__ void k(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, FA, FB, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}

static float decode(int fmt, unsigned code) {   // 0 e4m3, 1 e5m2, 2 e2m3, 3 e3m2, 4 e2m1
    int ebits, mbits, bias;
    switch (fmt) { case 0: ebits = 4; mbits = 3; bias = 7; break; case 1: ebits = 5; mbits = 2; bias = 15; break;
                   case 2: ebits = 2; mbits = 3; bias = 1; break; case 3: ebits = 3; mbits = 2; bias = 3; break;
                   default: ebits = 2; mbits = 1; bias = 1; }
    int nb = 1 + ebits + mbits;
    int s = (code >> (nb - 1)) & 1, e = (code >> mbits) & ((1 << ebits) - 1), m = code & ((1 << mbits) - 1);
    float v = e == 0 ? ldexpf((float)m, 1 - bias - mbits) : ldexpf((float)((1 << mbits) | m), e - bias - mbits);
    return s ? -v : v;
}
static int bitsof(int fmt) { return fmt < 2 ? 8 : (fmt < 4 ? 6 : 4); }

int run(int fa, int fb) {
    unsigned ca[16][128], cb[16][128]; int ea[16][4], eb[16][4];
    for (int r = 0; r < 16; ++r) for (int kk = 0; kk < 128; ++kk) {
        do { ca[r][kk] = rand() & ((1 << bitsof(fa)) - 1); } while (fa < 2 && std::isnan(decode(fa, ca[r][kk])) );
        do { cb[r][kk] = rand() & ((1 << bitsof(fb)) - 1); } while (fb < 2 && std::isnan(decode(fb, cb[r][kk])) );
        if (fa == 0 && (ca[r][kk] & 0x7F) == 0x7F) ca[r][kk] = 0x10;   // e4m3fn NaN
        if (fb == 0 && (cb[r][kk] & 0x7F) == 0x7F) cb[r][kk] = 0x10;
        if (fa == 1 && (ca[r][kk] & 0x7C) == 0x7C) ca[r][kk] = 0x11;   // e5m2 inf/nan
        if (fb == 1 && (cb[r][kk] & 0x7C) == 0x7C) cb[r][kk] = 0x11;
    }
    for (int r = 0; r < 16; ++r) for (int g = 0; g < 4; ++g) { ea[r][g] = 120 + rand() % 12; eb[r][g] = 122 + rand() % 8; }
    uint32_t ha[64][8], hb[64][8]; int hsa[64], hsb[64];
    memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
    for (int l = 0; l < 64; ++l) {
        int r = l & 15, g = l >> 4;
        for (int i = 0; i < 32; ++i) {
            int ba = bitsof(fa), bb = bitsof(fb);
            const int k8 = i < 16 ? 16 * g + i : 64 + 16 * g + (i - 16);
            uint64_t pos = (uint64_t)i * ba; unsigned v = ca[r][fa < 2 ? k8 : 32 * g + i];
            for (int bit = 0; bit < ba; ++bit) if (v >> bit & 1) ha[l][(pos + bit) / 32] |= 1u << ((pos + bit) % 32);
            pos = (uint64_t)i * bb; v = cb[r][fb < 2 ? k8 : 32 * g + i];
            for (int bit = 0; bit < bb; ++bit) if (v >> bit & 1) hb[l][(pos + bit) / 32] |= 1u << ((pos + bit) % 32);
        }
        hsa[l] = ea[r][g] | 0x55AA3300; hsb[l] = eb[r][g] | 0x11223300;   // junk in the upper bytes: only byte 0 may count
    }
    double ref[16][16];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double s = 0;
        for (int kk = 0; kk < 128; ++kk)
            s += (double)decode(fa, ca[m][kk]) * decode(fb, cb[n][kk]) * ldexp(1.0, ea[m][kk / 32] - 127) * ldexp(1.0, eb[n][kk / 32] - 127);
        ref[m][n] = s;
    }
    v8i *da, *db; v4f* dc; int *dsa, *dsb;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, 64 * 16); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
#define L(A, B) if (fa == A && fb == B) k<A, B><<<1, 64>>>(da, db, dc, dsa, dsb);
    L(0,0) L(1,1) L(0,1) L(1,0) L(2,2) L(3,3) L(4,4) L(0,4) L(4,0) L(2,4) L(3,0)
    float hc[64][4];
    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int col = l & 15, row = 4 * (l >> 4) + r;
        maxerr = fmax(maxerr, fabs(hc[l][r] - ref[row][col])); maxref = fmax(maxref, fabs(ref[row][col]));
    }
    printf("fmtA %d fmtB %d: max |err| %.3g (max |ref| %.3g) -> %s\n", fa, fb, maxerr, maxref, maxerr <= 1e-4 * maxref ? "LAYOUT OK" : "MISMATCH");
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dsa); hipFree(dsb);
    return maxerr <= 1e-4 * maxref ? 0 : 1;
}

int main() {
    srand(1);
    int bad = 0;
    int combos[][2] = {{0,0},{1,1},{0,1},{1,0},{2,2},{3,3},{4,4},{0,4},{4,0},{2,4},{3,0}};
    for (auto& c : combos) bad += run(c[0], c[1]);
    return bad;
}
