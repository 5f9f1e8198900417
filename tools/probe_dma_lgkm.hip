// GPU probe: does an in-flight global_load_lds (LDS-DMA) hold up `s_waitcnt lgkmcnt(N)`?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma_lgkm.hip -o tools/build/probe_dma_lgkm && tools/build/probe_dma_lgkm
// Each wave issues 8 LDS-DMA pieces from cold memory, then (a) waits lgkmcnt(0), (b) issues a ds_read and waits for it with
// lgkmcnt(0), (c) waits vmcnt(0); s_memtime stamps in between.  If the DMA counted on lgkmcnt, (a) would take as long as (c).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ __launch_bounds__(256) void k(const uint8_t *src, long long *out, size_t stride) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16 * 1024 * 4];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint8_t *p = src + ((size_t)blockIdx.x * 4 + w) * stride + l * 16;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((glb_void *)(p + i * 65536), (lds_void *)(lds + w * 16384 + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:60000\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(l * 4) : "memory");
    const long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = __builtin_amdgcn_s_memtime();
    if (l == 0) {
        long long *o = out + ((size_t)blockIdx.x * 4 + w) * 4;
        o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = v;
    }
}

int main() {
    const int blocks = 256;
    const size_t stride = 1 << 20;
    uint8_t *src; long long *out;
    hipMalloc(&src, (size_t)blocks * 4 * stride + (1 << 20)); hipMalloc(&out, blocks * 4 * 4 * sizeof(long long));
    hipMemset(src, 1, (size_t)blocks * 4 * stride);
    for (int rep = 0; rep < 2; ++rep) {
        k<<<blocks, 256>>>(src, out, stride);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(blocks * 4 * 4);
    hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < blocks * 4; ++i) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; }
    printf("DMALGKM cycles (mean over %d waves): issue 8 pieces + lgkmcnt(0): %.0f   ds_read + lgkmcnt(0): %.0f   then vmcnt(0): %.0f\n",
           blocks * 4, a / (blocks * 4), b / (blocks * 4), c / (blocks * 4));
    return 0;
}
