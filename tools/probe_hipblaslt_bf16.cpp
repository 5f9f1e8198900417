// Probe: hipBLASLt's first suggestion against the best of its own heuristic list for the bf16 GEMMs of the configs[4] training step
// (RoBERTa-base, 2048 tokens): forward y = x W^T + b, input gradient dx = g W, weight gradient dW = g^T x.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_hipblaslt_bf16.cpp -lhipblaslt -o tools/build/probe_lt_bf16 && tools/build/probe_lt_bf16
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e = (x); if (e != 0) { printf("error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)

struct Case { const char *name; int kind; int M, N, K; };      // kind 0 forward (+bias), 1 input gradient, 2 weight gradient

int main() {
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    const Case cases[] = {
        {"fwd q/k/v/o   ", 0, 2048, 768, 768},  {"fwd ffn up    ", 0, 2048, 3072, 768}, {"fwd ffn down  ", 0, 2048, 768, 3072},
        {"dgrad q/k/v/o ", 1, 2048, 768, 768},  {"dgrad ffn up  ", 1, 2048, 3072, 768}, {"dgrad ffn down", 1, 2048, 768, 3072},
        {"wgrad q/k/v/o ", 2, 2048, 768, 768},  {"wgrad ffn up  ", 2, 2048, 3072, 768}, {"wgrad ffn down", 2, 2048, 768, 3072},
    };
    size_t ws_size = 64u << 20;
    void *ws, *flush;
    CK(hipMalloc(&ws, ws_size));
    CK(hipMalloc(&flush, 256u << 20));
    double sum_first = 0, sum_best = 0;
    for (auto &c : cases) {
        const int M = c.M, N = c.N, K = c.K;
        void *x, *w, *g, *y, *dx, *dw, *bias;
        CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&g, (size_t)M * N * 2));
        CK(hipMalloc(&y, (size_t)M * N * 2)); CK(hipMalloc(&dx, (size_t)M * K * 2)); CK(hipMalloc(&dw, (size_t)N * K * 2)); CK(hipMalloc(&bias, (size_t)N * 2));
        CK(hipMemset(x, 0x3c, (size_t)M * K * 2)); CK(hipMemset(w, 0x3b, (size_t)N * K * 2)); CK(hipMemset(g, 0x3a, (size_t)M * N * 2)); CK(hipMemset(bias, 0, (size_t)N * 2));
        hipblasLtMatmulDesc_t desc;
        CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t ta, tb;
        hipblasLtMatrixLayout_t la, lb, lc;
        const void *A, *B; void *C;
        if (c.kind == 0) {            // y^T [N, M] = W (col-major [K, N], T) . x^T (col-major [K, M])
            ta = HIPBLAS_OP_T; tb = HIPBLAS_OP_N;
            CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K)); CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));
            CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, N, M, N));
            A = w; B = x; C = y;
            hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS; int32_t bt = HIP_R_16BF;
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof ep));
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt));
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias));
        } else if (c.kind == 1) {     // dx^T [K, M] = W^T (col-major [K, N], N) . g^T (col-major [N, M], N)
            ta = HIPBLAS_OP_N; tb = HIPBLAS_OP_N;
            CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K)); CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, N, M, N));
            CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, K, M, K));
            A = w; B = g; C = dx;
        } else {                      // dW^T [K, N] = x^T (col-major [K, M], N) . g (col-major [N, M], T)
            ta = HIPBLAS_OP_N; tb = HIPBLAS_OP_T;
            CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, M, K)); CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, N, M, N));
            CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, K, N, K));
            A = x; B = g; C = dw;
        }
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
        hipblasLtMatmulPreference_t pref;
        CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_size, sizeof ws_size));
        const int want = 48;
        std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
        int got = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, want, res.data(), &got));
        const float alpha = 1.f, beta = 0.f;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<std::pair<float, int>> times;
        float first_hot = 0.f, first_cold = 0.f;
        for (int i = 0; i < got; ++i) {
            if (res[i].workspaceSize > ws_size) continue;
            bool ok = true;
            for (int it = 0; it < 3 && ok; ++it)
                ok = hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_size, 0) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) continue;
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < 20; ++it) hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_size, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float hot = ms / 20 * 1e3f;
            // cold: another kernel's data through the caches first (a 256 MiB fill), one launch between events, best of 5
            float cold = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(flush, rep, 256u << 20, 0));
                CK(hipEventRecord(e0, 0));
                hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_size, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                cold = std::min(cold, ms * 1e3f);
            }
            if (i == 0) { first_hot = hot; first_cold = cold; }
            times.push_back({hot, i});
            if (i < 48) printf("    #%d hot %.1f cold %.1f us ws %zu\n", i, hot, cold, (size_t)res[i].workspaceSize);
        }
        std::sort(times.begin(), times.end());
        printf("%s M%d N%d K%d: %d suggestions; first %.1f us hot / %.1f cold; best hot #%d %.1f us (%.0f %% faster)\n", c.name, M, N, K, got, first_hot, first_cold,
               times[0].second, times[0].first, 100.0 * (first_hot - times[0].first) / first_hot);
        const int per_layer = (N == 768 && K == 768) ? 4 : 1;
        sum_first += first_hot * per_layer; sum_best += times[0].first * per_layer;
        hipFree(x); hipFree(w); hipFree(g); hipFree(y); hipFree(dx); hipFree(dw); hipFree(bias);
    }
    printf("per layer: first suggestions %.0f us, best suggestions %.0f us (x 12 layers: %.2f -> %.2f ms)\n", sum_first, sum_best, sum_first * 12e-3, sum_best * 12e-3);
    return 0;
}
