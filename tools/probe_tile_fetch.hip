// Probe: how fast can a CU pull operand tiles (rows of 128 bytes at a K-byte stride, as a GEMM k-loop does) -- LDS-DMA against
// register loads, by groups in flight and with / without a workgroup barrier per step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/probe_tile_fetch tools/probe_tile_fetch.hip && tools/build/probe_tile_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args { const uint8_t *src; uint32_t *sink; int K, nk, rows_total, wg_rows; };

// MODE 0: LDS-DMA into a ring of DEPTH + 1 slots; MODE 1: global_load_dwordx4 into registers (summed).  P pieces (8 rows x 128 B)
// per wave and step; DEPTH groups in flight behind the one waited for; BAR: workgroup barrier per step.
template <int MODE, int P, int DEPTH, bool BAR>
__global__ __launch_bounds__(512, 1) void fetch(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = (int)(((long)blockIdx.x * a.wg_rows) % a.rows_total);
    const uint8_t *g[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int row = (row0 + ((w * P + i) * 8 + (l >> 3)) % a.wg_rows) % a.rows_total;
        g[i] = a.src + (long)row * a.K + ((l & 7) << 4);
    }
    constexpr int kSlot = 8 * P * 1024;
    u32x4 sum = {0, 0, 0, 0};
    if constexpr (MODE == 0) {
        auto issue = [&](int kt, int slot) {
#pragma unroll
            for (int i = 0; i < P; ++i)
                __builtin_amdgcn_global_load_lds((glb_void *)(g[i] + (long)kt * 128), (lds_void *)(lds + slot * kSlot + (w * P + i) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d % a.nk, d);
        int slot = DEPTH;
        for (int kt = 0; kt < a.nk; ++kt) {
            issue((kt + DEPTH) % a.nk, slot);
            slot = slot == DEPTH ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * P) : "memory");
            if (BAR) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sum.x = ((volatile uint32_t *)lds)[t];
    } else {
        u32x4 buf[DEPTH + 1][P];
        auto issue = [&](int kt, int d) {
#pragma unroll
            for (int i = 0; i < P; ++i) buf[d][i] = __builtin_nontemporal_load((const u32x4 *)(g[i] + (long)kt * 128));
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d % a.nk, d);
        for (int kt0 = 0; kt0 < a.nk; kt0 += DEPTH + 1) {
#pragma unroll
            for (int d = 0; d <= DEPTH; ++d) {
                const int kt = kt0 + d;
                issue((kt + DEPTH) % a.nk, (d + DEPTH) % (DEPTH + 1));
#pragma unroll
                for (int i = 0; i < P; ++i) sum ^= buf[d][i];
                if (BAR) __builtin_amdgcn_s_barrier();
            }
        }
    }
    if (sum.x == 0x12345678u && sum.y == 0x9abcdef0u) a.sink[blockIdx.x * 512 + t] = sum.z + sum.w;
}

template <int MODE, int P, int DEPTH, bool BAR>
double run(const Args &a, int grid) {
    const int ldsb = MODE == 0 ? (DEPTH + 1) * 8 * P * 1024 : 0;
    hipFuncSetAttribute((const void *)fetch<MODE, P, DEPTH, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fetch<MODE, P, DEPTH, BAR><<<grid, 512, ldsb>>>(a);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) fetch<MODE, P, DEPTH, BAR><<<grid, 512, ldsb>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms * 1e3 / it;
}

int main() {
    const int K = 4096, rows_total = 16384;           // 64 MB of source: mostly cache-resident across launches (256 MB Infinity Cache)
    uint8_t *src; uint32_t *sink;
    hipMalloc(&src, (size_t)rows_total * K); hipMemset(src, 1, (size_t)rows_total * K);
    hipMalloc(&sink, 256 * 512 * 4 * 4);
    const int grid = 256;
    for (int shared = 0; shared < 2; ++shared) {
        // shared = 1: every 8 workgroups read the same rows (operand reuse through L2, as the row tiles of one B tile do)
        printf("%s\n", shared ? "-- rows shared by 8 workgroups" : "-- every workgroup its own rows");
#define ROW(MODE, P, DEPTH, BAR, name)                                                                                   \
    {                                                                                                                    \
        Args a{src, sink, K, K / 128, shared ? rows_total / 8 : rows_total, 64 * P};                                     \
        const double us = run<MODE, P, DEPTH, BAR>(a, grid);                                                             \
        const double kb = 8.0 * P * (K / 128);                                                                           \
        printf("%-44s %2d KB/step x %d steps  %7.1f us  %6.1f GB/s per CU\n", name, 8 * P, K / 128, us, kb * 1024 / us / 1e3); \
    }
        ROW(0, 4, 1, true, "LDS-DMA, 1 group ahead, barrier")
        ROW(0, 4, 2, true, "LDS-DMA, 2 groups ahead, barrier")
        ROW(0, 4, 3, true, "LDS-DMA, 3 groups ahead, barrier")
        ROW(0, 4, 3, false, "LDS-DMA, 3 groups ahead, no barrier")
        ROW(0, 6, 2, true, "LDS-DMA 48 KB, 2 groups ahead, barrier")
        ROW(0, 2, 4, true, "LDS-DMA 16 KB, 4 groups ahead, barrier")
        ROW(1, 4, 1, true, "register loads, 1 group ahead, barrier")
        ROW(1, 4, 2, true, "register loads, 2 groups ahead, barrier")
        ROW(1, 4, 3, true, "register loads, 3 groups ahead, barrier")
        ROW(1, 4, 3, false, "register loads, 3 groups ahead, no barrier")
        ROW(1, 6, 2, true, "register loads 48 KB, 2 groups ahead, barrier")
    }
    return 0;
}
