// Element codes of the block-scaled formats the matrix instruction takes (OCP MX v1.0 element formats).
#pragma once
#include <stdint.h>

#include "qt_formats.h"

namespace qt_mx {

__host__ __device__ constexpr int elem_bits(int f) { return f < 2 ? 8 : (f < 4 ? 6 : 4); }

// value -> code of format f (QT_MX_*).  A value the format holds exactly converts exactly; anything else sets `bad`.
__device__ __forceinline__ uint32_t encode_elem(int f, float v, bool &bad) {
    const uint32_t u = qt_f2u(v), s = u >> 31, au = u & 0x7FFFFFFFu;
    const float m = qt_u2f(au);
    const int E = (int)(au >> 23) - 127;
    int ebits, mbits, bias;
    float maxv;
    switch (f) {
        case 0: ebits = 4; mbits = 3; bias = 7; maxv = 448.f; break;
        case 1: ebits = 5; mbits = 2; bias = 15; maxv = 57344.f; break;
        case 2: ebits = 2; mbits = 3; bias = 1; maxv = 7.5f; break;
        case 3: ebits = 3; mbits = 2; bias = 3; maxv = 28.f; break;
        default: ebits = 2; mbits = 1; bias = 1; maxv = 6.f; break;
    }
    if (!(m <= maxv)) { bad = true; return 0; }          // also catches NaN
    uint32_t code;
    float back;
    if (E < 1 - bias) {                                  // subnormal of the target: multiples of 2^(1 - bias - mbits)
        const float q = m * qt_u2f((uint32_t)(127 - (1 - bias - mbits)) << 23);
        code = (uint32_t)q;
        back = (float)code * qt_u2f((uint32_t)(127 + (1 - bias - mbits)) << 23);
    } else {
        const uint32_t mant = (au >> (23 - mbits)) & ((1u << mbits) - 1u);
        code = ((uint32_t)(E + bias) << mbits) | mant;
        back = qt_u2f(au & ~((1u << (23 - mbits)) - 1u));
    }
    if (back != m) bad = true;
    (void)ebits;
    return code | (s << (ebits + mbits));
}


}  // namespace qt_mx
