// FP8 GEMM on already fake-quantized operands through the vendor library (hipBLASLt), with the algorithm NAMED by the
// caller.  "hipBLASLt only for plain library GEMMs": this is one -- C[b][M][N] = A[b][M][K] . B^T (+ bias), OCP
// E4M3 / E5M2 bytes in, fp32 accumulation, bf16 out, no scaling.  The library's first heuristic is not always its
// fastest kernel for the LLaMA shapes (1024 x 11008 x 4096: 47.9 us first choice, 36.6 us best of its own top
// suggestions, tools/probe_hipblaslt_fp8.cpp), and torch._scaled_mm offers neither a choice nor batched operands, so
// the library is driven directly: the caller passes the index of the suggestion to run (0 = the library's first).
// Rounds 1-4 timed the suggestions inside the first call of every process; two ranks (or two boxes) could then settle on
// kernels with different fp32 summation orders.  Since round 5 the product never times anything: qt_fp8_gemm_tune is the
// tools' entry point (tools/tune_lt_algos.py), its result is committed as a table (fused._LT_ALGO_TABLE).
//
// The library is resolved at run time from the process (PyTorch has already loaded its libhipblaslt.so.1), so
// libqt_hip.so itself has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/qt_hip.h"

namespace {

struct Api {
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatrixLayoutSetAttribute) LayoutSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    decltype(&hipblasLtGetVersion) GetVersion = nullptr;      // optional
    bool ok = false;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
#define QT_SYM(field, name) a.field = (decltype(a.field))dlsym(h, name); if (!a.field) return;
        QT_SYM(Create, "hipblasLtCreate")
        QT_SYM(DescCreate, "hipblasLtMatmulDescCreate")
        QT_SYM(DescSet, "hipblasLtMatmulDescSetAttribute")
        QT_SYM(LayoutCreate, "hipblasLtMatrixLayoutCreate")
        QT_SYM(LayoutSet, "hipblasLtMatrixLayoutSetAttribute")
        QT_SYM(PrefCreate, "hipblasLtMatmulPreferenceCreate")
        QT_SYM(PrefSet, "hipblasLtMatmulPreferenceSetAttribute")
        QT_SYM(Heuristic, "hipblasLtMatmulAlgoGetHeuristic")
        QT_SYM(Matmul, "hipblasLtMatmul")
#undef QT_SYM
        a.GetVersion = (decltype(a.GetVersion))dlsym(h, "hipblasLtGetVersion");
        a.ok = true;
    });
    return a;
}

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    std::vector<hipblasLtMatmulHeuristicResult_t> cand;
};

using Key = std::tuple<int, int, int, long, int, int, int, int, int, long, long, long>;     // [8] = device ordinal
std::map<Key, Plan> g_plans;
std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;       // one library handle per device (created with that device current)

hipDataType fp8_type(int f) { return f == 1 ? HIP_R_8F_E5M2 : HIP_R_8F_E4M3; }

}  // namespace

// the plan of one problem (created on first use); QT_OK or an error code
static int plan_for(Api &L, hipblasLtHandle_t handle, const Key &key, int a_format, int b_format, int b_is_kn, const void *bias_bf16, long batch,
                    int M, int N, int K, long a_batch_stride, long b_batch_stride, long c_batch_stride, void *workspace, size_t workspace_bytes,
                    Plan *&out) {
    Plan &p = g_plans[key];
    out = &p;
    if (!p.desc) {
        // column-major view: C^T [N, M] = op(B-matrix) . A-matrix, with the A-matrix [K, M] (ld K)
        if (L.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return QT_ERR_BAD_ARG;
        const hipblasOperation_t ta = b_is_kn ? HIPBLAS_OP_N : HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta);
        L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb);
        if (bias_bf16) {
            const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
            const int32_t bt = HIP_R_16BF;
            L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof ep);
            L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt);
        }
        bool ok = true;
        if (b_is_kn) ok &= L.LayoutCreate(&p.la, fp8_type(b_format), N, K, N) == HIPBLAS_STATUS_SUCCESS;     // [K, N] row-major
        else ok &= L.LayoutCreate(&p.la, fp8_type(b_format), K, N, K) == HIPBLAS_STATUS_SUCCESS;             // [N, K] row-major
        ok &= L.LayoutCreate(&p.lb, fp8_type(a_format), K, M, K) == HIPBLAS_STATUS_SUCCESS;
        ok &= L.LayoutCreate(&p.lc, HIP_R_16BF, N, M, N) == HIPBLAS_STATUS_SUCCESS;
        if (!ok) return QT_ERR_BAD_ARG;
        if (batch > 1) {
            const int32_t bc = (int32_t)batch;
            const int64_t sa = b_batch_stride, sb = a_batch_stride, sc = c_batch_stride;
            L.LayoutSet(p.la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof bc);
            L.LayoutSet(p.lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof bc);
            L.LayoutSet(p.lc, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof bc);
            L.LayoutSet(p.la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sa, sizeof sa);
            L.LayoutSet(p.lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sb, sizeof sb);
            L.LayoutSet(p.lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sc, sizeof sc);
        }
        hipblasLtMatmulPreference_t pref;
        if (L.PrefCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return QT_ERR_BAD_ARG;
        uint64_t ws = workspace ? workspace_bytes : 0;
        L.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws);
        p.cand.resize(16);
        int got = 0;
        if (L.Heuristic(handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, (int)p.cand.size(), p.cand.data(), &got) != HIPBLAS_STATUS_SUCCESS || got < 1) {
            p.cand.clear();
            return QT_ERR_BAD_DTYPE;                          // no kernel for this problem: caller uses its other route
        }
        p.cand.resize(got);
    }
    if (p.cand.empty()) return QT_ERR_BAD_DTYPE;
    return QT_OK;
}

static int fp8_gemm(const uint8_t *a8, int a_format, const uint8_t *b8, int b_format, int b_is_kn, void *c_bf16, const void *bias_bf16, long batch,
                    int M, int N, int K, long a_batch_stride, long b_batch_stride, long c_batch_stride, void *workspace, size_t workspace_bytes,
                    int algo, int *best, float *us, int max_us, void *stream) {
    if ((long)batch * M * N == 0) return QT_OK;
    if (!a8 || !b8 || !c_bf16 || batch < 1 || M < 1 || N < 1 || K < 1 || a_format < 0 || a_format > 1 || b_format < 0 || b_format > 1)
        return QT_ERR_BAD_ARG;
    Api &L = api();
    if (!L.ok) return QT_ERR_NO_DEVICE;                      // library not available in this process
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return QT_ERR_NO_DEVICE;
    hipblasLtHandle_t &g_handle = g_handles[dev];
    if (!g_handle && L.Create(&g_handle) != HIPBLAS_STATUS_SUCCESS) return QT_ERR_NO_DEVICE;

    const Key key{a_format, b_format, b_is_kn, batch, M, N, K, bias_bf16 ? 1 : 0, dev, a_batch_stride, b_batch_stride, c_batch_stride};
    Plan *pp = nullptr;
    if (const int rc = plan_for(L, g_handle, key, a_format, b_format, b_is_kn, bias_bf16, batch, M, N, K, a_batch_stride, b_batch_stride,
                                c_batch_stride, workspace, workspace_bytes, pp))
        return rc;
    Plan &p = *pp;
    if (bias_bf16) L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias_bf16, sizeof bias_bf16);

    const float alpha = 1.0f, beta = 0.0f;
    const size_t have = workspace ? workspace_bytes : 0;
    auto run = [&](int i) {
        return L.Matmul(g_handle, p.desc, &alpha, b8, p.la, a8, p.lb, &beta, c_bf16, p.lc, c_bf16, p.lc, &p.cand[i].algo,
                        workspace, have, st);
    };
    if (best) {                                              // tools only: time every suggestion, report the fastest
        *best = 0;
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return QT_ERR_NO_DEVICE;
        float best_ms = 1e30f;
        for (int i = 0; i < (int)p.cand.size(); ++i) {
            if (us && i < max_us) us[i] = -1.0f;
            if (p.cand[i].workspaceSize > have) continue;
            bool ok = true;
            for (int it = 0; it < 3 && ok; ++it) ok = run(i) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) continue;
            (void)hipEventRecord(e0, st);
            for (int it = 0; it < 20; ++it) (void)run(i);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
            if (us && i < max_us) us[i] = ms * 1000.0f / 20.0f;
            if (ms < best_ms) { best_ms = ms; *best = i; }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return (int)p.cand.size();
    }
    // the named suggestion; one the library did not return, or one that wants more workspace than there is, falls back to its first
    int pick = (algo >= 0 && algo < (int)p.cand.size() && p.cand[algo].workspaceSize <= have) ? algo : 0;
    return run(pick) == HIPBLAS_STATUS_SUCCESS ? QT_OK : QT_ERR_BAD_ARG;
}

extern "C" int qt_fp8_gemm(const uint8_t *a8, int a_format, const uint8_t *b8, int b_format, int b_is_kn, void *c_bf16,
                           const void *bias_bf16, long batch, int M, int N, int K, long a_batch_stride, long b_batch_stride,
                           long c_batch_stride, void *workspace, size_t workspace_bytes, int algo, void *stream) {
    return fp8_gemm(a8, a_format, b8, b_format, b_is_kn, c_bf16, bias_bf16, batch, M, N, K, a_batch_stride, b_batch_stride, c_batch_stride,
                    workspace, workspace_bytes, algo, nullptr, nullptr, 0, stream);
}

extern "C" int qt_fp8_gemm_tune(const uint8_t *a8, int a_format, const uint8_t *b8, int b_format, int b_is_kn, void *c_bf16,
                                const void *bias_bf16, long batch, int M, int N, int K, long a_batch_stride, long b_batch_stride,
                                long c_batch_stride, void *workspace, size_t workspace_bytes, int *best, float *us, int max_us, void *stream) {
    if (!best) return QT_ERR_BAD_ARG;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing((hipStream_t)stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return QT_ERR_BAD_ARG;
    return fp8_gemm(a8, a_format, b8, b_format, b_is_kn, c_bf16, bias_bf16, batch, M, N, K, a_batch_stride, b_batch_stride, c_batch_stride,
                    workspace, workspace_bytes, 0, best, us, max_us, stream);
}

extern "C" int qt_fp8_gemm_library_version(void) {
    Api &L = api();
    if (!L.ok || !L.GetVersion) return 0;
    std::lock_guard<std::mutex> lock(g_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipblasLtHandle_t &h = g_handles[dev];
    if (!h && L.Create(&h) != HIPBLAS_STATUS_SUCCESS) return 0;
    int v = 0;
    return L.GetVersion(h, &v) == HIPBLAS_STATUS_SUCCESS ? v : 0;
}
