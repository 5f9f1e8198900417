// qt_host.cpp -- host half of the C ABI: dtype-string parsing, the 65 536-entry value-map
// builder and the host evaluations of the rounding functions.
//
// Replaces get_quantization_map (src/quantized_training/fake_quantize.py:31-95 of the
// reference).  The reference builds each table by running torch ops over all 2^16 bf16
// patterns; here each family is a scalar function evaluated 65 536 times:
//   intN/uintN      qt_int_img           (fake_quantize.py:43-52)
//   e4m3/e5m2       qt_fp_sat_u32        (fp8.py:10-67)
//   fpN_eXmY        elemwise_core_bf16   (fp8.py:147-203 run in bf16 arithmetic, fake_quantize.py:63-80)
//   positN_ES       qt_posit_u32         (posit.py:6-67)
//   float16/...     IEEE casts           (fake_quantize.py:38-40)
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qt_formats.h"

namespace {

struct ParsedDtype {
    enum Family { IDENTITY, FLOAT16, INT, UINT, NV_FP8, MX_FLOAT, POSIT, BAD } family = BAD;
    int a = 0, b = 0, c = 0;   // INT/UINT: nbits; NV_FP8: a=mbits; MX_FLOAT: nbits,ebits,mbits; POSIT: nbits,es
    bool is_fp8_e4m3_name = false;
};

bool all_digits(const char *s, const char *e) {
    if (s == e) return false;
    for (; s < e; ++s)
        if (!isdigit((unsigned char)*s)) return false;
    return true;
}

// Parses exactly the dtype grammar of fake_quantize.py:43-86 (fullmatch semantics; intN/uintN and
// e4m3/e5m2 are case-insensitive there, fpN_eXmY and positN_ES are not).
ParsedDtype parse_dtype(const char *dtype) {
    ParsedDtype p;
    if (dtype == nullptr || dtype[0] == 0) {
        p.family = ParsedDtype::IDENTITY;
        return p;
    }
    size_t n = strlen(dtype);
    if (n > 63) return p;
    char low[64];
    for (size_t i = 0; i <= n; ++i) low[i] = (char)tolower((unsigned char)dtype[i]);
    if (!strcmp(dtype, "float32") || !strcmp(dtype, "bfloat16")) {
        p.family = ParsedDtype::IDENTITY;
        return p;
    }
    if (!strcmp(dtype, "float16")) {
        p.family = ParsedDtype::FLOAT16;
        return p;
    }
    if (!strncmp(low, "int", 3) && all_digits(low + 3, low + n)) {
        p.family = ParsedDtype::INT;
        p.a = atoi(low + 3);
        return p;
    }
    if (!strncmp(low, "uint", 4) && all_digits(low + 4, low + n)) {
        p.family = ParsedDtype::UINT;
        p.a = atoi(low + 4);
        return p;
    }
    const char *t = low;
    if (!strncmp(t, "fp8.", 4)) t += 4;
    if (!strcmp(t, "e4m3") || !strcmp(t, "e5m2")) {
        p.family = ParsedDtype::NV_FP8;
        p.a = t[1] == '4' ? 3 : 2;
        return p;
    }
    int nb = 0, eb = 0, mb = 0, consumed = 0;
    if (sscanf(dtype, "fp%d_e%dm%d%n", &nb, &eb, &mb, &consumed) == 3 && (size_t)consumed == n &&
        isdigit((unsigned char)dtype[2])) {
        if (!(nb == eb + mb + 1 || nb == eb + mb)) return p;     // assert at fake_quantize.py:65
        p.family = ParsedDtype::MX_FLOAT;
        p.a = nb; p.b = eb; p.c = mb;
        p.is_fp8_e4m3_name = !strcmp(dtype, "fp8_e4m3");
        return p;
    }
    int es = 0;
    consumed = 0;
    if (sscanf(dtype, "posit%d_%d%n", &nb, &es, &consumed) == 2 && (size_t)consumed == n &&
        isdigit((unsigned char)dtype[5])) {
        p.family = ParsedDtype::POSIT;
        p.a = nb; p.b = es;
        return p;
    }
    return p;
}

inline float rbf(float f) { return qt_bf2f(qt_f2bf(f)); }

// torch.clamp(x, lo, hi) on a bf16 tensor: bounds are cast to the tensor dtype first.
inline float clamp_t(float v, float lo, float hi) {
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

// _quantize_elemwise_core(A, bits, exp_bits, max_norm, round="even", saturate_normals=True)
// with A a bf16 tensor: every torch op result is rounded to bf16 (fp8.py:147-203).
uint16_t elemwise_core_bf16(uint16_t in, int bits, int ebits, float max_norm) {
    const float A = qt_bf2f(in);
    const float t = rbf(fabsf(A) + (A == 0.0f ? 1.0f : 0.0f));                 // :174-175
    float pe = floorf(rbf(log2f(t)));                                          // :174
    const float min_exp = (float)(-(1 << (ebits - 1)) + 2);                    // :178
    if (pe < min_exp) pe = min_exp;                                            // :179
    const float p2 = rbf((float)exp2((double)pe));                             // 2 ** private_exp
    const float up = (float)(1 << (bits - 2));
    float out = rbf(rbf(A / p2) * up);                                         // _safe_lshift :90-94
    const float a = fabsf(out);                                                // _round_mantissa "even" :123-127
    const float am = rbf(a - 0.5f);
    const float rem = am - 2.0f * floorf(am / 2.0f);
    const float mask = (rem == 0.0f) ? 1.0f : 0.0f;
    const float fl = rbf(floorf(rbf(a + 0.5f)) - mask);
    const float sgn = (float)(out > 0.0f) - (float)(out < 0.0f);
    out = rbf(sgn * fl);
    out = rbf(rbf(out / up) * p2);                                             // _safe_rshift :97-101
    const float mx = rbf(max_norm);
    out = clamp_t(out, -mx, mx);                                               // :193
    if (A == INFINITY) out = INFINITY;                                         // :199-200
    if (A == -INFINITY) out = -INFINITY;
    return qt_f2bf(out);
}

float posit_threshold(int nbits, int es) {
    // math.pow(2, math.floor(-(nbits - 1) * (1 << es) + 2 ** (es - 1)))   posit.py:52
    double e = floor(-(double)(nbits - 1) * (double)(1 << es) + pow(2.0, (double)(es - 1)));
    return (float)pow(2.0, e);
}

int fp8_emin(float fp8_min) { return (int)floor(log2((double)fp8_min)); }     // fp8.py:20

}  // namespace

extern "C" {

int qt_abi_version(void) { return QT_ABI_VERSION; }

const char *qt_status_string(int code) {
    switch (code) {
        case QT_OK: return "ok";
        case QT_ERR_BAD_DTYPE: return "unsupported dtype";
        case QT_ERR_BAD_ARG: return "bad argument";
        case QT_ERR_UNALIGNED: return "pointer not aligned to the element size";
        case QT_ERR_NO_DEVICE: return "no HIP device";
        default: return code > 0 ? "HIP runtime error (hipError_t)" : "unknown error";
    }
}

static int format_basic(const char *dtype, qt_format *out) {
    if (!out) return QT_ERR_BAD_ARG;
    qt_format f = {QT_FMT_LUT, 0, 0, 0.0f, 0.0f};
    ParsedDtype p = parse_dtype(dtype);
    switch (p.family) {
        case ParsedDtype::BAD: return QT_ERR_BAD_DTYPE;
        case ParsedDtype::IDENTITY: f.kind = QT_FMT_IDENTITY; break;
        case ParsedDtype::NV_FP8:
            f.kind = QT_FMT_FP_SAT;
            f.p0 = p.a;
            f.p1 = p.a == 3 ? -6 : -14;
            f.fhi = p.a == 3 ? 448.0f : 57344.0f;
            f.flo = -f.fhi;
            break;
        case ParsedDtype::INT:
        case ParsedDtype::UINT:
            if (p.a >= 1 && p.a <= 24) {
                f.kind = QT_FMT_INT;
                double lo = p.family == ParsedDtype::INT ? -ldexp(1.0, p.a - 1) : 0.0;
                double hi = p.family == ParsedDtype::INT ? ldexp(1.0, p.a - 1) - 1.0 : ldexp(1.0, p.a) - 1.0;
                f.flo = rbf((float)lo);
                f.fhi = rbf((float)hi);
                f.p0 = p.a;
            }
            break;
        default: break;   // table only
    }
    *out = f;
    return QT_OK;
}

// Table formats whose map is odd-symmetric -- map[-v] = -map[v] for every finite non-zero result, zero results +0 and NaN
// results canonical on both sides (posit, fpN_eXmY, NormalFloat ...) -- are marked p0 = 1: the device kernels then stage
// only the non-negative half of the table (64 KiB of LDS instead of 128) and put the sign back on the looked-up value.
int qt_format_for(const char *dtype, qt_format *out) {
    const int rc = format_basic(dtype, out);
    if (rc != QT_OK || out->kind != QT_FMT_LUT) return rc;
    static thread_local uint16_t map[QT_MAP_ENTRIES];
    if (qt_build_map(dtype, map) != QT_OK) return QT_OK;
    bool odd = true;
    for (uint32_t i = 0; i < 0x8000u && odd; ++i) {
        const uint16_t t = map[i], n = map[i | 0x8000u];
        const uint16_t want = (uint16_t)(t - 1u) < 0x7F80u ? (uint16_t)(t | 0x8000u) : t;      // signed unless zero / NaN
        odd = n == want && (t & 0x8000u) == 0;
    }
    out->p0 = odd ? 1 : 0;
    return QT_OK;
}

uint16_t qt_format_apply_host(const qt_format *fmt, uint16_t b) {
    return (uint16_t)(qt_apply_format_img(*fmt, (uint32_t)b << 16) >> 16);
}

int qt_build_map(const char *dtype, uint16_t *out) {
    if (!out) return QT_ERR_BAD_ARG;
    ParsedDtype p = parse_dtype(dtype);
    if (p.family == ParsedDtype::BAD) return QT_ERR_BAD_DTYPE;
    qt_format f;
    format_basic(dtype, &f);
    float max_norm = 0.0f, thr = 0.0f;
    int mb = 0;
    if (p.family == ParsedDtype::INT || p.family == ParsedDtype::UINT) {
        if (p.a < 1 || p.a > 24) return QT_ERR_BAD_DTYPE;
    } else if (p.family == ParsedDtype::MX_FLOAT) {
        if (p.b < 1 || p.b > 8 || p.c < 0 || p.c > 20) return QT_ERR_BAD_DTYPE;
        mb = p.c + 2;                                                          // fake_quantize.py:71
        int emax = p.b > 4 ? (1 << (p.b - 1)) - 1 : (1 << (p.b - 1));          // :72
        if (!p.is_fp8_e4m3_name)
            max_norm = (float)(ldexp(1.0, emax) * (double)((1 << (mb - 1)) - 1) / (double)(1 << (mb - 2)));   // :74
        else
            max_norm = (float)(ldexp(1.0, emax) * 1.75);                       // :76
    } else if (p.family == ParsedDtype::POSIT) {
        if (p.a < 3 || p.a > 24 || p.b < 0 || p.b > 4 || ((p.a - 2) << p.b) > 126) return QT_ERR_BAD_DTYPE;
        thr = posit_threshold(p.a, p.b);
    }
    for (uint32_t i = 0; i < QT_MAP_ENTRIES; ++i) {
        const uint16_t b = (uint16_t)i;
        uint16_t r;
        switch (p.family) {
            case ParsedDtype::FLOAT16: {
                float v = qt_bf2f(b);
                r = qt_f2bf((float)(_Float16)v);
                break;
            }
            case ParsedDtype::MX_FLOAT:
                r = elemwise_core_bf16(p.a == p.b + p.c ? (uint16_t)(b & 0x7FFF) : b, mb, p.b, max_norm);   // :68-69
                break;
            case ParsedDtype::POSIT:
                r = qt_f2bf(qt_u2f(qt_posit_u32((uint32_t)b << 16, p.a, p.b, thr)));
                break;
            default:
                r = (uint16_t)(qt_apply_format_img(f, (uint32_t)b << 16) >> 16);
                break;
        }
        if ((r & 0x7FFF) > 0x7F80) r = QT_NAN16;
        out[i] = r;
    }
    return QT_OK;
}

int qt_round_fp8_host(const float *x, float *y, size_t n, int mbits, float fp8_max, float fp8_min) {
    if ((!x || !y) && n) return QT_ERR_BAD_ARG;
    if (mbits < 1 || mbits > 22 || !(fp8_min > 0.0f)) return QT_ERR_BAD_ARG;
    const int emin = fp8_emin(fp8_min);
    for (size_t i = 0; i < n; ++i) y[i] = qt_u2f(qt_fp_sat_u32(qt_f2u(x[i]), mbits, emin, fp8_max));
    return QT_OK;
}

int qt_round_posit_host(const float *x, float *y, size_t n, int nbits, int es) {
    if ((!x || !y) && n) return QT_ERR_BAD_ARG;
    if (nbits < 3 || nbits > 24 || es < 0 || es > 4 || ((nbits - 2) << es) > 126) return QT_ERR_BAD_ARG;
    const float thr = posit_threshold(nbits, es);
    for (size_t i = 0; i < n; ++i) y[i] = qt_u2f(qt_posit_u32(qt_f2u(x[i]), nbits, es, thr));
    return QT_OK;
}

int qt_posit_quantize_host(const float *x, float *y, int32_t *pbits, size_t n, int nbits, int es, int round_to_even) {
    if ((!x || !y) && n) return QT_ERR_BAD_ARG;
    if (nbits < 3 || nbits > 24 || es < 0 || es > 4 || ((nbits - 2) << es) > 126) return QT_ERR_BAD_ARG;
    const float thr = round_to_even ? posit_threshold(nbits, es) : 0.0f;
    for (size_t i = 0; i < n; ++i) {
        int32_t pb;
        y[i] = qt_u2f(qt_posit_bits_u32(qt_f2u(x[i]), nbits, es, thr, &pb));
        if (pbits) pbits[i] = pb;
    }
    return QT_OK;
}

}  // extern "C"

// ---- row parameters of a value map (qt_linear_fqt_bf16: the weight fake-quantizer inside the bf16 GEMM) -----------------
// A "row" is the 128 bf16 patterns that share sign and exponent (index = bits >> 7).  Within a row almost every value map of
// the reference is a round-to-nearest-even onto a power-of-two grid followed by a clamp -- which the fp32 adder evaluates as
//     t = float_from_bits(|x| bits + D);   z = (t + C) - C;   y = min(max(z, lo), hi)            (C = 1.5 * 2^T, or 0)
// with a tiny integer nudge D of the pattern for the rows whose tie goes the other way (a posit's last bit may be an exponent
// bit: posit.py:45-60) or whose threshold is an artefact of bf16 arithmetic (fp8.py:147-203).  The fit is found by search and
// accepted only if it reproduces the map on all 128 patterns of the row, so the table is exact by construction; rows that do
// not fit (the non-finite inputs, and a few rows at the far ends of some formats) are flagged -- C's lowest bit -- and a
// kernel that meets one redoes its tile with the map itself.
namespace {

inline float rp_eval(uint32_t mag_img, int32_t d, float c, float lo, float hi) {
    volatile float t = qt_u2f(mag_img + (uint32_t)d);      // volatile: the two roundings must stay two roundings
    volatile float s = t + c;
    float z = s - c;
    z = z < lo ? lo : z;
    z = z > hi ? hi : z;
    return z;
}

// fits one row: `expect[m]` = |map value| as fp32 for mantissa m (any NaN: no fit)
bool rp_fit_row(uint32_t exp8, const float *expect, uint32_t out[4]) {
    float lo = expect[0], hi = expect[0];
    for (int m = 0; m < 128; ++m) {
        if (expect[m] != expect[m]) return false;
        lo = expect[m] < lo ? expect[m] : lo;
        hi = expect[m] > hi ? expect[m] : hi;
    }
    if (hi > 3.0e38f) return false;
    static const int kD[17] = {0, 0x4000, -0x4000, 0x8000, -0x8000, 0xC000, -0xC000, 0x10000, -0x10000, 0x14000, -0x14000,
                               0x18000, -0x18000, 0x1C000, -0x1C000, 0x20000, -0x20000};
    for (int di = 0; di < 17; ++di) {
        for (int T = -62; T <= 70; ++T) {                  // T == -62: no rounding step (C = 0)
            const float c = T == -62 ? 0.0f : (float)ldexp(1.5, T);
            bool ok = true;
            for (int m = 0; m < 128 && ok; ++m) {
                const uint32_t img = ((exp8 << 7) | (uint32_t)m) << 16;
                ok = rp_eval(img, kD[di], c, lo, hi) == expect[m];
            }
            if (ok) {
                out[0] = (uint32_t)kD[di]; out[1] = qt_f2u(c); out[2] = qt_f2u(lo); out[3] = qt_f2u(hi);
                return (out[1] & 1u) == 0u;
            }
        }
    }
    return false;
}

}  // namespace

extern "C" {

int qt_build_rowparams(const uint16_t *map, qt_rowparams *out) {
    if (!map || !out) return QT_ERR_BAD_ARG;
    memset(out, 0, sizeof(*out));
    // sign rule for negative inputs: every non-zero finite result negative (the input's sign is copied) or every one positive
    int neg = 0, pos = 0;
    for (uint32_t i = 0x8000; i < 0x10000; ++i) {
        const uint16_t r = map[i];
        if ((r & 0x7FFF) == 0 || (r & 0x7FFF) > 0x7F80) continue;
        (r & 0x8000) ? ++neg : ++pos;
    }
    for (uint32_t i = 0; i < 0x8000; ++i) {                // a positive input must not produce a negative value
        const uint16_t r = map[i];
        if ((r & 0x7FFF) != 0 && (r & 0x7FFF) <= 0x7F80 && (r & 0x8000)) { neg = pos = 1; break; }
    }
    const bool sign_ok = !(neg && pos);
    out->sign_mask = neg ? 0x80008000u : 0u;
    bool same = true;                                       // |map(-x)| == |map(x)| everywhere: index rows by exponent only
    for (uint32_t i = 0; i < 0x8000 && same; ++i) {
        const uint16_t a = map[i] & 0x7FFF, b = map[i | 0x8000] & 0x7FFF;
        same = a == b || (a > 0x7F80 && b > 0x7F80);
    }
    out->signed_rows = same ? 0 : 1;
    {   // zero results of negative (non-zero) inputs: +0 (e4m3, posit: fp8.py:33-35), -0 (intN: torch.round keeps the sign), or both
        int pz = 0, nz = 0;
        for (uint32_t i = 0x8001; i < 0x10000; ++i) {
            if (map[i] == 0x0000) ++pz;
            if (map[i] == 0x8000) ++nz;
        }
        out->zero_sign = (pz && nz) ? 2 : (nz ? 1 : 0);
    }
    for (uint32_t row = 0; row < 512; ++row) {
        float expect[128];
        for (int m = 0; m < 128; ++m) {
            const uint16_t r = map[(row << 7) | (uint32_t)m];
            expect[m] = qt_bf2f((uint16_t)(r & 0x7FFF));
            if ((r & 0x7FFF) > 0x7F80) expect[m] = qt_u2f(QT_NAN32);
        }
        uint32_t *p = out->row[row];
        const bool fit = sign_ok && (row & 0xFF) != 0xFF && rp_fit_row(row & 0xFF, expect, p);
        if (!fit) {
            // D = 0 and lo = hi = the result of the row's first input (mantissa 0; row 0: of zero itself): callers that branch to the
            // map for a flagged row may skip the branch for that one input -- exact zeros are the common case of a flagged row 0.
            // D = 1: the first result is a NaN or carries a sign, which a clamp cannot produce; every input of the row takes the branch.
            const bool usable = expect[0] == expect[0] && !(map[row << 7] & 0x8000);   // (the clamp carries no sign either)
            const uint32_t first = usable ? qt_f2u(expect[0]) : 0u;
            p[0] = usable ? 0u : 1u; p[1] = 1u; p[2] = first; p[3] = first;
            out->flagged[row] = 1;
            out->n_flagged += 1;
        }
    }
    return QT_OK;
}

uint16_t qt_rowparams_apply_host(const qt_rowparams *rp, uint16_t b, int *flagged) {
    const uint32_t row = rp->signed_rows ? (uint32_t)(b >> 7) : (uint32_t)((b >> 7) & 0xFF);
    const uint32_t *p = rp->row[row];
    if (flagged) *flagged = (int)(p[1] & 1u);
    const float z = rp_eval((uint32_t)(b & 0x7FFF) << 16, (int32_t)p[0], qt_u2f(p[1]), qt_u2f(p[2]), qt_u2f(p[3]));
    return (uint16_t)((qt_f2u(z) >> 16) | (b & (uint16_t)(rp->sign_mask & 0x8000u)));
}

}  // extern "C"

extern "C" {

// helpers shared with the device half (qt_elementwise.hip)
float qt_internal_posit_threshold(int nbits, int es) { return posit_threshold(nbits, es); }
int qt_internal_fp8_emin(float fp8_min) { return fp8_emin(fp8_min); }

}  // extern "C"
