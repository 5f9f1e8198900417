/* CPU ORACLE (C part) -- TEST INFRASTRUCTURE ONLY; never linked into or called by the product.
 *
 * Plain-C restatement of the reference's per-tensor fake-quant pass, used (a) as the timed CPU
 * baseline of bench.py (`cpu_baseline`, kind "port") and (b) as a second checker for large
 * tensors where the numpy restatement (qt_oracle.py) would be slow.  It is itself checked against
 * qt_oracle.py (tests/test_oracle_golden.py::test_c_oracle_matches_numpy), which is pinned to the
 * reference's golden vectors.
 *
 * Follows  FusedAmaxObsFakeQuantFunction.forward   src/quantized_training/fake_quantize.py:217-248
 *          vmap                                    src/quantized_training/decomposed.py:146-163
 * The value map (65 536 uint16 bf16 patterns) is supplied by the caller (built by qt_oracle.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bf2f(uint16_t b) { return u2f((uint32_t)b << 16); }
static inline uint16_t f2bf(float f) {               /* torch float -> bfloat16: RNE, NaN stays NaN */
    uint32_t u = f2u(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

int qto_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* amax = torch.amax(torch.abs(input))   fake_quantize.py:223 ; returns the bf16 pattern of max|x|
 * (a NaN pattern if any input is NaN, like torch.amax). */
uint16_t qto_amax_bf16(const uint16_t *x, size_t n) {
    uint16_t m = 0;
#pragma omp parallel for reduction(max : m) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint16_t a = x[i] & 0x7FFF;
        if (a > m) m = a;
    }
    return m;
}

/* input = vmap(input / scale, qmap) * scale on a bf16 tensor   fake_quantize.py:245-246
 * (each op computed in fp32 and rounded to bf16, scale already cast to bf16). */
void qto_fake_quant_bf16(const uint16_t *x, uint16_t *y, size_t n, const uint16_t *qmap, uint16_t scale_bits) {
    const float s = bf2f(scale_bits);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint16_t q = f2bf(bf2f(x[i]) / s);            /* input / scale             */
        uint16_t r = qmap[q];                         /* decomposed.py:148-149,161 */
        y[i] = f2bf(bf2f(r) * s);                     /* ... * scale               */
    }
}

void qto_fake_quant_f32(const float *x, float *y, size_t n, const uint16_t *qmap, float s) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint32_t u = f2u(x[i] / s);
        uint16_t idx = (uint16_t)((u >> 16) | ((u & 0xFFFFu) != 0));   /* decomposed.py:151-153 */
        y[i] = bf2f(qmap[idx]) * s;
    }
}

float qto_amax_f32(const float *x, size_t n) {
    uint32_t m = 0;
#pragma omp parallel for reduction(max : m) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint32_t a = f2u(x[i]) & 0x7FFFFFFFu;
        if (a > m) m = a;
    }
    return u2f(m);
}
