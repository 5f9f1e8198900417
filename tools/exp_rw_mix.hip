// Ceiling probe for the FP8-only weight pass: what does the memory system deliver for (a) read only, (b) write only,
// (c) 1:1 copy and (d) a 2:1 narrowing copy (two 16-B loads -> one 16-B store, no arithmetic to speak of) with the
// same launch geometry as fq8_kernel?  Rotating 8-tensor pool (> Infinity Cache), HIP events around 48 launches.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/exp_rw_mix tools/exp_rw_mix.hip && gpurun_out/exp_rw_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ x, uint32_t *out, size_t nvec) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const uint4 v = x[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ y, size_t nvec, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256)
        y[i] = uint4{seed, seed + 1, seed + 2, (uint32_t)i};
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ x, uint4 *__restrict__ y, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
// 2:1 -- keep the high byte of every 16-bit element
__global__ __launch_bounds__(256) void k_narrow(const uint4 *__restrict__ x, uint4 *__restrict__ y, size_t npair) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
        const uint4 a = x[2 * i], b = x[2 * i + 1];
        uint4 o;
        o.x = __builtin_amdgcn_perm(a.y, a.x, 0x07050301u);
        o.y = __builtin_amdgcn_perm(a.w, a.z, 0x07050301u);
        o.z = __builtin_amdgcn_perm(b.y, b.x, 0x07050301u);
        o.w = __builtin_amdgcn_perm(b.w, b.z, 0x07050301u);
        y[i] = o;
    }
}
// 2:1 with the loads of the NEXT iteration issued before this iteration's store (software pipelining)
__global__ __launch_bounds__(256) void k_narrow_pipe(const uint4 *__restrict__ x, uint4 *__restrict__ y, size_t npair) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    if (i >= npair) return;
    uint4 a = x[2 * i], b = x[2 * i + 1];
    while (true) {
        const size_t nx = i + step;
        uint4 na = a, nb = b;
        if (nx < npair) { na = x[2 * nx]; nb = x[2 * nx + 1]; }
        uint4 o;
        o.x = __builtin_amdgcn_perm(a.y, a.x, 0x07050301u);
        o.y = __builtin_amdgcn_perm(a.w, a.z, 0x07050301u);
        o.z = __builtin_amdgcn_perm(b.y, b.x, 0x07050301u);
        o.w = __builtin_amdgcn_perm(b.w, b.z, 0x07050301u);
        y[i] = o;
        if (nx >= npair) break;
        i = nx; a = na; b = nb;
    }
}

int main() {
    const size_t n = (size_t)4096 * 11008, pool = 8;          // bf16 elements per tensor
    const size_t in_bytes = n * 2, nvec = in_bytes / 16;
    uint8_t *x, *y;
    uint32_t *out;
    CK(hipMalloc(&x, in_bytes * pool));
    CK(hipMalloc(&y, in_bytes * pool));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0x3c, in_bytes * pool));
    CK(hipMemset(y, 0, in_bytes * pool));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int bpcs[] = {2, 4, 8, 16, 32, 64};
    for (int which = 0; which < 5; ++which) {
        for (int bpc : bpcs) {
            const unsigned blocks = 256u * bpc;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                for (int it = 0; it < 48; ++it) {
                    const uint4 *xi = (const uint4 *)(x + (it % pool) * in_bytes);
                    uint4 *yi = (uint4 *)(y + (it % pool) * in_bytes);
                    switch (which) {
                        case 0: k_read<<<blocks, 256>>>(xi, out, nvec); break;
                        case 1: k_write<<<blocks, 256>>>(yi, nvec, (uint32_t)it); break;
                        case 2: k_copy<<<blocks, 256>>>(xi, yi, nvec); break;
                        case 3: k_narrow<<<blocks, 256>>>(xi, yi, nvec / 2); break;
                        case 4: k_narrow_pipe<<<blocks, 256>>>(xi, yi, nvec / 2); break;
                    }
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double us = best / 48 * 1e3;
            const double bytes = which == 0 ? in_bytes : which == 1 ? in_bytes : which == 2 ? 2.0 * in_bytes : 1.5 * in_bytes;
            const char *names[] = {"read 16B", "write 16B", "copy 1:1", "narrow 2:1", "narrow 2:1 pipelined"};
            printf("%-22s blocks/CU %2d  %7.2f us  %7.1f GB/s\n", names[which], bpc, us, bytes / us / 1e3);
        }
    }
    return 0;
}
