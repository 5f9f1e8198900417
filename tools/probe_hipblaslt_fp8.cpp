// Probe: how good is hipBLASLt's default FP8 algorithm for the LLaMA shapes, against the best of its top heuristics?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_hipblaslt_fp8.cpp -lhipblaslt -o /tmp/probe_lt && /tmp/probe_lt
// D[M,N] (bf16, row-major) = X[M,K] (e4m3) . W[N,K]^T (e4m3), fp32 accumulation, no scale pointers.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e = (x); if (e != 0) { printf("error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)

int main() {
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    const int shapes[][3] = {{1024, 4096, 4096}, {1024, 11008, 4096}, {1024, 4096, 11008}, {1024, 32000, 4096}};
    size_t ws_size = 256u << 20;
    void *ws;
    CK(hipMalloc(&ws, ws_size));
    for (auto &s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        void *x, *w, *d;
        CK(hipMalloc(&x, (size_t)M * K)); CK(hipMalloc(&w, (size_t)N * K)); CK(hipMalloc(&d, (size_t)M * N * 2));
        CK(hipMemset(x, 0x38, (size_t)M * K)); CK(hipMemset(w, 0x30, (size_t)N * K));
        hipblasLtMatmulDesc_t desc;
        CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
        hipblasLtMatrixLayout_t la, lb, lc;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_8F_E4M3, K, N, K));      // W as column-major [K, N]
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_8F_E4M3, K, M, K));      // X as column-major [K, M]
        CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, N, M, N));         // D^T column-major [N, M]
        hipblasLtMatmulPreference_t pref;
        CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_size, sizeof ws_size));
        const int want = 64;
        std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
        int got = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, want, res.data(), &got));
        const float alpha = 1.f, beta = 0.f;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f, first = 0.f; int besti = -1;
        for (int i = 0; i < got; ++i) {
            if (res[i].workspaceSize > ws_size) continue;
            bool ok = true;
            for (int it = 0; it < 3 && ok; ++it)
                ok = hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, d, lc, d, lc, &res[i].algo, ws, ws_size, 0) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) continue;
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < 20; ++it)
                hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, d, lc, d, lc, &res[i].algo, ws, ws_size, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 20;
            if (i == 0) first = ms;
            if (ms < best) { best = ms; besti = i; }
        }
        printf("M%d N%d K%d: %d heuristics; default (first) %.1f us = %.0f TF; best #%d %.1f us = %.0f TF (%.1f %% faster)\n", M, N, K, got,
               first * 1e3, 2.0 * M * N * K / first / 1e9, besti, best * 1e3, 2.0 * M * N * K / best / 1e9, 100.0 * (first - best) / first);
        hipFree(x); hipFree(w); hipFree(d);
    }
    return 0;
}
