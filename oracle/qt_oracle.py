"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference's fake-quantization hot path.  It exists so
that the HIP kernels can be checked against an independent statement of the
algorithm on hosts where the reference itself is absent (the GPU box).

  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import this package.  The product (quantized-training_amd/) never does; it
    raises if its HIP library is missing.
  * Parity is PINNED: every function below is checked against golden vectors
    produced by running the reference in the build container
    (tests/golden/gen_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).

All file:line citations are relative to the upstream checkout
(src/quantized_training/...).  Values travel as bit patterns: bf16 tensors are
uint16 arrays, fp32 tensors are float32 arrays (or their uint32 view).
"""
import math
import re

import numpy as np

U16 = np.uint16
U32 = np.uint32
F32 = np.float32


# --------------------------------------------------------------------------
# bf16 helpers (value <-> bit pattern); torch semantics: RNE, NaN stays NaN
# --------------------------------------------------------------------------
def bf16_to_f32(bits):
    return (np.asarray(bits, dtype=U16).astype(U32) << U32(16)).view(F32)


def f32_to_bf16(x):
    """float32 -> bf16 bits, round-to-nearest-even, NaN -> quiet NaN (0x7FC0)."""
    x = np.ascontiguousarray(x, dtype=F32)
    b = x.view(U32)
    r = ((b + U32(0x7FFF) + ((b >> U32(16)) & U32(1))) >> U32(16)).astype(U16)
    return np.where(np.isnan(x), U16(0x7FC0), r).astype(U16)


def rbf(x):
    """Round a float32 array to the nearest bf16-representable float32."""
    return bf16_to_f32(f32_to_bf16(x))


def all_bf16_patterns():
    return np.arange(65536, dtype=U32).astype(U16)


def canon_nan16(bits):
    bits = np.array(bits, dtype=U16, copy=True)
    nan = ((bits & 0x7F80) == 0x7F80) & ((bits & 0x007F) != 0)
    bits[nan] = 0x7FC0
    return bits


def canon_nan32(bits):
    bits = np.array(bits, dtype=U32, copy=True)
    nan = ((bits & 0x7F800000) == 0x7F800000) & ((bits & 0x007FFFFF) != 0)
    bits[nan] = 0x7FC00000
    return bits


# --------------------------------------------------------------------------
# A2: NVIDIA-style FP8 rounding on the fp32 image (fp8.py:10-67)
# --------------------------------------------------------------------------
def quantize_to_fp8(x, mbits, fp8_max, fp8_min):
    """fp8.py:10-37 (e4m3: mbits=3, max 448, min 2^-6) / :40-67 (e5m2: 2, 57344, 2^-14).
    x: float32 array; returns float32 array."""
    x = np.ascontiguousarray(x, dtype=F32)
    raw = x.view(U32).astype(np.int64)
    sign = raw & 0x80000000
    exp = ((raw & 0x7F800000) >> 23) - 127                      # :17
    frac = (raw & 0x7FFFFF) | 0x800000                           # :18
    min_exp = int(math.floor(math.log2(fp8_min)))                # :20
    nf = 23 - mbits + np.clip(min_exp - exp, 0, None)            # :21
    nfs = np.minimum(nf, 40)                                     # int64 shifts stay defined; tiny inputs are flushed below
    lb = (frac & (1 << nfs)) != 0                                # :22
    gb = (frac & (1 << (nfs - 1))) != 0                          # :23
    sb = (frac & ((1 << (nfs - 1)) - 1)) != 0                    # :24
    rb = (lb & gb) | (gb & sb)                                   # :25
    nfc = np.minimum(nf, 23)                                     # :27
    mag = (raw & 0x7FFFFFFF) & ~((1 << nfc) - 1)                 # :28
    mag = np.where(rb, mag + (1 << nfc), mag)                    # :29
    out = (mag | sign).astype(U32).view(F32)
    with np.errstate(invalid="ignore"):
        out = np.clip(out, -F32(fp8_max), F32(fp8_max))          # :32
        out = np.where(np.abs(x) <= F32(fp8_min * 2.0 ** -(mbits + 1)), F32(0), out)   # :33
        out = np.where(x == 0, F32(0), out)                      # :35
    out = np.where(np.isfinite(x), out, F32(np.nan))             # :36
    return out.astype(F32)


def quantize_to_fp8_e4m3(x):
    return quantize_to_fp8(x, 3, 448.0, 2.0 ** -6)


def quantize_to_fp8_e5m2(x):
    return quantize_to_fp8(x, 2, 57344.0, 2.0 ** -14)


# --------------------------------------------------------------------------
# A4: posit<nbits,es> rounding (posit.py:6-67)
# --------------------------------------------------------------------------
def quantize_to_posit(x, nbits, es):
    """posit.py:6-67 with round_to_even=True.  x: float32 array -> float32 array."""
    x = np.ascontiguousarray(x, dtype=F32)
    raw = x.view(U32).astype(np.int64)
    scale = ((raw & 0x7F800000) >> 23) - 127                     # :14
    frac = raw & 0x7FFFFF                                        # :15
    r = scale >= 0                                               # :16
    max_scale = (nbits - 2) * (1 << es)                          # :18
    dominated = np.where(r, scale > max_scale, scale < -max_scale)   # :19
    run = np.where(r, 1 + (scale >> es), -(scale >> es))         # :21 (arithmetic shift == floor)
    run_s = np.minimum(run, 30)                                  # keep shifts defined; dominated lanes ignore rb
    regime = np.where(r, (1 << (run_s + 1)) - 1, 0) ^ 1          # :22
    exponent = scale % (1 << es)                                 # :23 (floor-mod)
    pt = (regime << (23 + es)) | (exponent << 23) | frac         # :24
    ln = 2 + run_s + es + 23                                     # :27
    sh = np.clip(ln - nbits, 1, 62)
    lbm = 1 << sh                                                # :28
    gbm = lbm >> 1                                               # :29
    sbm = gbm - 1                                                # :30
    lb = (pt & lbm) != 0
    gb = (pt & gbm) != 0
    sb = (pt & sbm) != 0
    rb = ((lb & gb) | (gb & sb)) & ~dominated                    # :35
    ne = np.clip(2 + run + es - nbits, 0, es)                    # :38
    sc = scale & ~((1 << ne) - 1)                                # :39 (two's complement and)
    sc = np.clip(sc, -max_scale, max_scale)                      # :40
    nf = np.clip(2 + run + es + 23 - nbits, 0, 23)               # :43
    fr = frac & ~((1 << nf) - 1)                                 # :44
    out = ((sc + 127) << 23) | fr                                # :46
    out = np.where(rb, out + (1 << (nf + ne)), out)              # :47
    with np.errstate(invalid="ignore", over="ignore"):
        mag = (out & 0xFFFFFFFF).astype(U32).view(F32)
        out = mag * np.sign(x)                                   # :48
        thr = math.pow(2, math.floor(-(nbits - 1) * (1 << es) + 2 ** (es - 1)))   # :52
        out = np.where(np.abs(x) < F32(thr), F32(0), out)        # :53
        out = np.where(x == 0, F32(0), out)                      # :56
    out = np.where(np.isfinite(x), out, F32(np.nan))             # :57
    return out.astype(F32)


# --------------------------------------------------------------------------
# A3: MX-library float rounding evaluated in bf16 arithmetic (fp8.py:147-203
#     called on a bf16 tensor from fake_quantize.py:63-80)
# --------------------------------------------------------------------------
def _clamp(a, lo, hi):
    """torch.clamp: min(max(a, lo), hi) with std::max/min operand order (keeps -0.0, NaN)."""
    with np.errstate(invalid="ignore"):
        a = np.where(a < lo, lo, a)
        a = np.where(a > hi, hi, a)
    return a


def _sign(a):
    with np.errstate(invalid="ignore"):
        s = (a > 0).astype(F32) - (a < 0).astype(F32)
    return np.where(np.isnan(a), F32(np.nan), s).astype(F32)


def quantize_elemwise_core_bf16(bits_in, bits, exp_bits, max_norm):
    """_quantize_elemwise_core(A:bf16, bits, exp_bits, max_norm, round='even',
    saturate_normals=True) -- fp8.py:147-203; every torch op rounds to bf16."""
    A = bf16_to_f32(bits_in)
    with np.errstate(all="ignore"):
        absA = np.abs(A)
        t = rbf(absA + (A == 0).astype(F32))                       # :174-175
        pe = np.floor(rbf(np.log2(t).astype(F32)))                 # :174
        min_exp = -(2 ** (exp_bits - 1)) + 2                       # :178
        pe = np.where(pe < min_exp, F32(min_exp), pe).astype(F32)  # :179 (NaN stays NaN)
        p2 = rbf(np.exp2(pe.astype(np.float64)).astype(F32))       # 2 ** private_exp  (:94)
        out = rbf(rbf(A / p2) * F32(2 ** (bits - 2)))              # _safe_lshift :90-94
        # _round_mantissa(..., 'even')  :123-127
        a = np.abs(out)
        am = rbf(a - F32(0.5))
        rem = am - F32(2) * np.floor(am / F32(2))                  # python-style remainder, exact
        mask = (rem == 0).astype(F32)
        mask = np.where(np.isnan(am), F32(0), mask)
        fl = rbf(np.floor(rbf(a + F32(0.5))) - mask)
        out = rbf(_sign(out) * fl)
        out = rbf(rbf(out / F32(2 ** (bits - 2))) * p2)            # _safe_rshift :97-101
        out = _clamp(out, -rbf(np.array(max_norm, F32)), rbf(np.array(max_norm, F32)))   # :193
        out = np.where(A == np.inf, F32(np.inf), out)              # :199
        out = np.where(A == -np.inf, F32(-np.inf), out)            # :200
    return f32_to_bf16(out.astype(F32))


# --------------------------------------------------------------------------
# A1: the 65 536-entry value map (fake_quantize.py:31-95)
# --------------------------------------------------------------------------
def get_quantization_map(dtype):
    """Returns uint16[65536]: bf16 bits of Q_dtype(bf16_from_bits(i)).  NaNs canonical (0x7FC0)."""
    idx = all_bf16_patterns()
    vals = bf16_to_f32(idx)
    if dtype is None:                                              # :34-35
        return canon_nan16(idx)
    if dtype in ("float32", "bfloat16"):                           # :38-40
        return canon_nan16(idx)
    if dtype == "float16":
        with np.errstate(over="ignore", invalid="ignore"):
            return canon_nan16(f32_to_bf16(vals.astype(np.float16).astype(F32)))
    m = re.fullmatch(r"int(\d+)", dtype, re.IGNORECASE)            # :43-46
    u = re.fullmatch(r"uint(\d+)", dtype, re.IGNORECASE)           # :49-52
    if m or u:
        n = int((m or u).group(1))
        lo, hi = (-(2 ** (n - 1)), 2 ** (n - 1) - 1) if m else (0, 2 ** n - 1)
        lo = rbf(np.array(lo, F32))                                # clamp bounds are cast to the tensor dtype
        hi = rbf(np.array(hi, F32))
        with np.errstate(invalid="ignore"):
            r = np.rint(vals)                                      # round-half-even, exact in bf16
        return canon_nan16(f32_to_bf16(_clamp(r, lo, hi)))
    m = re.fullmatch(r"(?:fp8\.)?(e4m3|e5m2)", dtype, re.IGNORECASE)   # :55-60
    if m:
        f = quantize_to_fp8_e4m3 if m.group(1).lower() == "e4m3" else quantize_to_fp8_e5m2
        return canon_nan16(f32_to_bf16(f(vals)))
    m = re.fullmatch(r"fp(\d+)_e(\d+)m(\d+)", dtype)               # :63-80
    if m:
        nbits, ebits, mbits = map(int, m.groups())
        assert nbits == ebits + mbits + 1 or nbits == ebits + mbits
        src = idx
        if nbits == ebits + mbits:                                 # unsigned: abs first (:68-69)
            src = idx & U16(0x7FFF)
        mb = mbits + 2
        emax = 2 ** (ebits - 1) - 1 if ebits > 4 else 2 ** (ebits - 1)
        if dtype != "fp8_e4m3":
            max_norm = 2 ** emax * float(2 ** (mb - 1) - 1) / 2 ** (mb - 2)
        else:
            max_norm = 2 ** emax * 1.75
        return canon_nan16(quantize_elemwise_core_bf16(src, mb, ebits, max_norm))
    m = re.fullmatch(r"posit(\d+)_(\d+)", dtype)                   # :84-86
    if m:
        return canon_nan16(f32_to_bf16(quantize_to_posit(vals, int(m.group(1)), int(m.group(2)))))
    m = re.fullmatch(r"nf(\d+)(?:_(\d+))?", dtype)                # :90-93 (flattened: values[indices])
    if m:
        return canon_nan16(nf_value_map(int(m.group(1)), int(m.group(2)) if m.group(2) else None))
    raise ValueError(f"Unsupported dtype: {dtype}")                # :95


# --------------------------------------------------------------------------
# NormalFloat code books (normal_float.py:4-62): 2^k levels at evenly spaced N(0,1) quantiles
# --------------------------------------------------------------------------
def nf_levels(k=4, int_bits=None, offset=0.9677083):
    from scipy.stats import norm
    half = 2 ** (k - 1)

    def lin(steps):          # torch.linspace(offset, 0.5, steps) in float32 (two-sided evaluation)
        step = (F32(0.5) - F32(offset)) / F32(steps - 1)
        i = np.arange(steps, dtype=F32)
        return np.where(np.arange(steps) < steps // 2, F32(offset) + step * i,
                        F32(0.5) - step * (F32(steps - 1) - i)).astype(F32)

    v = np.concatenate([norm.ppf(lin(half + 1)[:-1]), [0.0], -norm.ppf(lin(half)[:-1])]).astype(F32)   # :13-16
    v = np.sort(v)
    v = (v / v.max()).astype(F32)                                                                        # :26-27
    if int_bits is not None:
        v = np.rint(v * F32(2 ** (int_bits - 1) - 1)).astype(F32)                                        # :52-54
    return v


def nf_value_map(k=4, int_bits=None):
    """bf16 bits of values[argmin |values - clamp(x)|] for every bf16 pattern (normal_float.py:56-60)."""
    levels = rbf(nf_levels(k, int_bits))                                                                 # values.to(bf16)
    x = bf16_to_f32(all_bf16_patterns())
    with np.errstate(invalid="ignore"):
        xc = _clamp(x, levels.min(), levels.max())
        dist = np.abs(rbf(levels[None, :] - xc[:, None]))                                                # bf16 subtraction
        dist = np.where(np.isnan(dist), F32(np.inf), dist)
        nanrow = np.isnan(xc)
        idx = np.argmin(dist, axis=1)
    out = f32_to_bf16(levels[idx])
    # torch.argmin on an all-NaN row returns the first NaN position (index 0)
    out[nanrow] = f32_to_bf16(levels[:1])[0]
    return out


# --------------------------------------------------------------------------
# A5: vmap (decomposed.py:146-163)
# --------------------------------------------------------------------------
def vmap_index_f32(x):
    """decomposed.py:151-153: hi16(bits) | (lo16 != 0)"""
    b = np.ascontiguousarray(x, dtype=F32).view(U32)
    return ((b >> U32(16)) | ((b & U32(0xFFFF)) != 0).astype(U32)).astype(U16)


def vmap_bf16(xbits, qmap):
    """bf16 tensor (uint16 bits) -> bf16 bits (decomposed.py:148-149,159-161)."""
    return np.asarray(qmap, dtype=U16)[np.asarray(xbits, dtype=U16)]


def vmap_f32(x, qmap):
    """float32 tensor -> float32 tensor (qmap value cast to the output dtype)."""
    return bf16_to_f32(np.asarray(qmap, dtype=U16)[vmap_index_f32(x)])


def vmap_f16(xbits_f16, qmap):
    """float16 bits -> float16 bits: index through the fp32 image, value cast bf16->fp16."""
    x = np.asarray(xbits_f16, dtype=U16).view(np.float16).astype(F32)
    with np.errstate(over="ignore", invalid="ignore"):
        return vmap_f32(x, qmap).astype(np.float16).view(U16)


# --------------------------------------------------------------------------
# A6/A7: quantize / dequantize / fake-quant with a scale, in the input dtype
# --------------------------------------------------------------------------
def fq_bf16(xbits, qmap, scale_bits):
    """y = vmap(x / s, qmap) * s on bf16 tensors; s is bf16 (bits), broadcastable.
    fake_quantize.py:245-246: each op computed in fp32 and rounded to bf16."""
    x = bf16_to_f32(xbits)
    s = bf16_to_f32(scale_bits)
    with np.errstate(all="ignore"):
        q = vmap_bf16(f32_to_bf16((x / s).astype(F32)), qmap)
        return f32_to_bf16((bf16_to_f32(q) * s).astype(F32))


def fq_f32(x, qmap, scale):
    x = np.ascontiguousarray(x, dtype=F32)
    s = np.asarray(scale, dtype=F32)
    with np.errstate(all="ignore"):
        return (vmap_f32((x / s).astype(F32), qmap) * s).astype(F32)


def quantize_f32(x, qmap, scale, zero_point=None):
    """decomposed.py:205-210"""
    with np.errstate(all="ignore"):
        t = (np.asarray(x, F32) / np.asarray(scale, F32)).astype(F32)
        if zero_point is not None:
            t = (t + np.asarray(zero_point, F32)).astype(F32)
    return vmap_f32(t, qmap)


def quantize_bf16(xbits, qmap, scale_bits):
    with np.errstate(all="ignore"):
        t = f32_to_bf16((bf16_to_f32(xbits) / bf16_to_f32(scale_bits)).astype(F32))
    return vmap_bf16(t, qmap)


def dequantize_f32(x, scale, zero_point=None, input_qmap=None, output_qmap=None):
    """decomposed.py:246-262"""
    x = np.asarray(x, F32)
    if input_qmap is not None:
        x = vmap_f32(x, input_qmap)
    with np.errstate(all="ignore"):
        if zero_point is not None:
            x = (x - np.asarray(zero_point, F32)).astype(F32)
        y = (x * np.asarray(scale, F32)).astype(F32)
    if output_qmap is not None:
        y = vmap_f32(y, output_qmap)
    return y


def dequantize_bf16(xbits, scale_bits, input_qmap=None, output_qmap=None):
    if input_qmap is not None:
        xbits = vmap_bf16(xbits, input_qmap)
    with np.errstate(all="ignore"):
        y = f32_to_bf16((bf16_to_f32(xbits) * bf16_to_f32(scale_bits)).astype(F32))
    if output_qmap is not None:
        y = vmap_bf16(y, output_qmap)
    return y


class FakeQuantState:
    """The buffers FusedAmaxObsFakeQuantFunction mutates (fake_quantize.py:304-312)."""

    def __init__(self, amax_history_len, quant_max, observer=True, ch_axis=None, per_channel=False, pow2=False):
        self.amax_history = np.zeros((0,), F32)
        self.scale = np.ones((1,), F32)
        self.L = amax_history_len
        self.quant_max = quant_max
        self.observer_enabled = observer
        self.fake_quant_enabled = True
        self.ch_axis = ch_axis
        self.per_channel = per_channel
        self.pow2 = pow2


def _amax_f32(x, st):
    """fake_quantize.py:218-223; NaN propagates like torch.amax."""
    a = np.abs(x)
    if st.per_channel:
        ax = st.ch_axis + x.ndim if st.ch_axis < 0 else st.ch_axis
        dims = tuple(i for i in range(x.ndim) if i != ax)
        return np.max(a, axis=dims, keepdims=True).astype(F32)       # np.max propagates NaN
    return np.max(a).astype(F32).reshape(())


def fake_quant_forward(x, is_bf16, qmap, st):
    """fake_quantize.py:217-248.  x: float32 array holding the tensor VALUES (for bf16 tensors the
    values are bf16-representable).  Returns values as float32 (bf16-representable when is_bf16)."""
    x = np.ascontiguousarray(x, dtype=F32)
    if st.observer_enabled:
        amax_cur = _amax_f32(x, st)                                    # :218-223 (exact in either dtype)
        if st.amax_history.size == 0:                                  # :225-228
            st.amax_history = np.zeros((st.L,) + amax_cur.shape, F32)
            st.scale = np.ones(amax_cur.shape, F32)
        amax = np.max(st.amax_history, axis=0)                         # :230
        if st.amax_history.shape[0] > 1:                               # :232-234
            st.amax_history = np.roll(st.amax_history, -1, axis=0)
        st.amax_history[0] = amax_cur                                  # :235
        with np.errstate(all="ignore"):
            sf = (amax / F32(st.quant_max)).astype(F32)                # :237
            sf = np.where(amax > 0.0, sf, st.scale)                    # :238
            sf = np.where(np.isfinite(amax), sf, st.scale)             # :239
            if st.pow2:                                                # :240-241
                sf = np.exp2(np.ceil(np.log2(sf.astype(F32)))).astype(F32)
        st.scale = sf.astype(F32)                                      # :242
    if not st.fake_quant_enabled:
        return x
    if is_bf16:                                                        # :245-246
        sb = f32_to_bf16(st.scale)
        return bf16_to_f32(fq_bf16(f32_to_bf16(x), qmap, np.broadcast_to(sb, x.shape) if sb.ndim == x.ndim else sb))
    return fq_f32(x, qmap, st.scale)


# --------------------------------------------------------------------------
# H1: WikiText sliding-window schedule (examples/language_modeling/wikitext.py:143-165)
# --------------------------------------------------------------------------
def wikitext_windows(seq_len, max_length, stride):
    rows, prev_end = [], 0
    for begin in range(0, seq_len - max_length, stride):
        end = min(begin + max_length, seq_len)
        rows.append((begin, end, end - prev_end))
        prev_end = end
        if end == seq_len:
            break
    return rows


# --------------------------------------------------------------------------
# Block-scaled formats: microscaling and group-wise affine fake-quant
#   MXFakeQuantFunction            fake_quantize.py:98-133
#   calculate_mx_qparam/quantize_mx decomposed.py:365-448,  _reshape_to_blocks mx_utils.py:58-118
#   GroupWiseAffineFakeQuantFunction fake_quantize.py:136-194
# Every torch op on a bf16 tensor rounds its result to bf16; `rd` applies that rounding.
# --------------------------------------------------------------------------
def _to_blocks(x, axis, bs):
    """[..., n, ...] -> [..., nblk, bs, ...] zero-padded along `axis` (mx_utils.py:58-118)."""
    ax = axis % x.ndim
    n = x.shape[ax]
    pad = (-n) % bs
    if pad:
        w = [(0, 0)] * x.ndim
        w[ax] = (0, pad)
        x = np.pad(x, w)
    shp = list(x.shape)
    shp[ax:ax + 1] = [x.shape[ax] // bs, bs]
    return x.reshape(shp), ax


def _expand_blocks(s, ax, bs, n):
    return np.repeat(s, bs, axis=ax).take(np.arange(n), axis=ax)


def mx_fake_quant(x, is_bf16, qmap, axis, block_size, quant_max, pow2=False, scale_qmap=None):
    """Returns (y, scale) like MXFakeQuantFunction.forward + the `scale` buffer it fills."""
    rd = rbf if is_bf16 else (lambda a: np.asarray(a, F32))
    x = np.ascontiguousarray(x, dtype=F32)
    blocks, ax = _to_blocks(x, axis, block_size)
    with np.errstate(all="ignore"):
        amax = np.max(np.abs(blocks), axis=ax + 1)
        if not pow2:
            scale = rd((amax / F32(quant_max)).astype(F32))                                # decomposed.py:414-415
            if scale_qmap is not None:                                                     # :418-419
                scale = bf16_to_f32(vmap_bf16(f32_to_bf16(scale), scale_qmap)) if is_bf16 else vmap_f32(scale, scale_qmap)
        else:
            t = rd(amax + F32(2.0 ** -126) * (amax == 0))                                  # mx_utils.py:43-47
            e = np.floor(rd(np.log2(t).astype(F32)))
            e = rd(e - F32(math.floor(math.log2(quant_max))))                              # decomposed.py:405
            scale = rd(np.exp2(e.astype(np.float64)).astype(F32))                          # :411
        scale = np.where(scale > 0.0, scale, F32(1.0)).astype(F32)                         # :421
        se = _expand_blocks(scale, ax, block_size, x.shape[ax])
        if is_bf16:
            q = bf16_to_f32(vmap_bf16(f32_to_bf16((x / se).astype(F32)), qmap))            # quantize, decomposed.py:205-210
        else:
            q = vmap_f32((x / se).astype(F32), qmap)
        y = rd((q * se).astype(F32))                                                       # fake_quantize.py:128
    return y, scale


def group_wise_affine_fake_quant(x, is_bf16, axis, block_size, quant_min, quant_max):
    """Returns (y, scale, zero_point) like GroupWiseAffineFakeQuantFunction.forward (no scale_qmap)."""
    rd = rbf if is_bf16 else (lambda a: np.asarray(a, F32))
    x = np.ascontiguousarray(x, dtype=F32)
    blocks, ax = _to_blocks(x, axis, block_size)
    with np.errstate(all="ignore"):
        mn = np.min(blocks, axis=ax + 1)                                                   # fake_quantize.py:166-167
        mx = np.max(blocks, axis=ax + 1)
        sf = rd(rd(mx - mn) / F32(quant_max - quant_min))                                  # :169
        sf = np.where(sf > 0.0, sf, F32(1.0)).astype(F32)                                  # :170
        zp = rd(rd(rd(-mn) / sf) + F32(quant_min))                                         # :171
        se = _expand_blocks(sf, ax, block_size, x.shape[ax])
        ze = _expand_blocks(zp, ax, block_size, x.shape[ax])
        q = _clamp(np.rint(rd(rd(x / se) + ze)), F32(quant_min), F32(quant_max))           # :185
        y = rd(rd(q - ze) * se)                                                            # :188
    return y.astype(F32), sf, zp


# ---- packed block-scaled operands (OCP MX element codes + E8M0 scales) --------------------------------------
# What linear_mx / matmul_mx consume is (values, block scales) (decomposed.py:304-363).  The matrix instruction
# wants the same numbers as element codes and one exponent byte per 32 elements; this is that re-encoding,
# restated on the host so the device packer can be checked bit for bit.
MX_ELEM = {"fp8_e4m3": (0, 4, 3, 7), "fp8_e5m2": (1, 5, 2, 15), "fp6_e2m3": (2, 2, 3, 1), "fp6_e3m2": (3, 3, 2, 3),
           "fp4_e2m1": (4, 2, 1, 1)}      # name -> (hardware format id, exponent bits, mantissa bits, bias)


def mx_decode(code, fmt):
    """Element code -> float64 value (OCP MX v1.0 element formats; no Inf/NaN codes are produced by mx_encode)."""
    _, eb, mb, bias = MX_ELEM[fmt]
    code = np.asarray(code, dtype=np.int64)
    s = (code >> (eb + mb)) & 1
    e = (code >> mb) & ((1 << eb) - 1)
    m = code & ((1 << mb) - 1)
    v = np.where(e == 0, np.ldexp(m.astype(np.float64), 1 - bias - mb), np.ldexp(((1 << mb) | m).astype(np.float64), e - bias - mb))
    return np.where(s == 1, -v, v)


def mx_encode(values, fmt):
    """float values that are exactly representable in `fmt` -> element codes; raises if one is not."""
    _, eb, mb, bias = MX_ELEM[fmt]
    v = np.asarray(values, dtype=np.float64)
    table = mx_decode(np.arange(1 << (eb + mb)), fmt)          # non-negative half, increasing
    finite = table[: {"fp8_e4m3": 0x7F, "fp8_e5m2": 0x7C}.get(fmt, len(table))]     # drop the NaN / Inf codes of the 8-bit formats
    idx = np.searchsorted(finite, np.abs(v))
    idx = np.clip(idx, 0, len(finite) - 1)
    if not np.array_equal(finite[idx], np.abs(v)):
        raise ValueError(f"value not representable in {fmt}")
    return (idx | (np.signbit(v).astype(np.int64) << (eb + mb))).astype(np.uint32)


def mx_pack(values, scales, fmt, block_size=32):
    """values [rows, K], scales [rows, K / block_size] (powers of two) -> (codes uint8 [rows, K*bits/8], e8m0 uint8 [rows, K/32]).
    Element i of a row occupies bits [i*bits, (i+1)*bits) of the row, little-endian."""
    _, eb, mb, _ = MX_ELEM[fmt]
    bits = 1 + eb + mb
    values = np.asarray(values, dtype=np.float64)
    rows, K = values.shape
    codes = mx_encode(values, fmt).astype(np.uint64)
    bitpos = np.arange(K, dtype=np.uint64) * np.uint64(bits)
    out = np.zeros((rows, K * bits // 8), dtype=np.uint8)
    for b in range(bits):
        pos = bitpos + np.uint64(b)
        byte, off = (pos >> np.uint64(3)).astype(np.int64), (pos & np.uint64(7)).astype(np.uint8)
        bitv = ((codes >> np.uint64(b)) & np.uint64(1)).astype(np.uint8)
        np.bitwise_or.at(out, (np.arange(rows)[:, None], byte[None, :]), bitv << off[None, :])
    s = np.asarray(scales, dtype=np.float64)
    m, e = np.frexp(s)                                            # s = m * 2^e, m in [0.5, 1)
    if not np.all(m == 0.5):
        raise ValueError("scale is not a power of two")
    e8 = (e - 1 + 127)
    if np.any(e8 < 0) or np.any(e8 > 254):
        raise ValueError("scale outside E8M0")
    e8 = np.repeat(e8.astype(np.uint8), block_size // 32, axis=1)
    return out, e8


# ---- attention-score path with every bf16 rounding point explicit ---------------------------------------------------------
# modules/quantizable/modeling_bert.py:118, 142-158 (and the HF LLaMA eager path the LLaMA twin follows):
#     scores = qk_matmul(q, k^T)                     bf16 tensor (fp32-accumulated GEMM, ONE rounding)
#     t      = attn_scaling(scores, scaling)         bf16 x python float -> bf16 (ONE rounding of the fp32 product)
#     u      = t + mask                              bf16 + bf16 -> bf16 (ONE rounding)
#     p      = softmax(u) in fp32, then bf16         (ONE rounding; nn.Softmax on bf16 computes in fp32 as well)
#     pq     = fake-quant(p)                         value map on the bf16 pattern (av_matmul's input hook, quantize.py:128-140)
#     out    = av_matmul(pq, v)                      bf16 (fp32-accumulated GEMM, ONE rounding)
# exp and every sum are evaluated in float64 here, so this is the chain's value with exact transcendental / reduction
# arithmetic; an implementation in fp32 differs from it only where its ~1e-7 relative error straddles a bf16 rounding
# boundary -- a few elements in 10^4, each by one bf16 ULP -- whereas a missing or extra rounding point moves a large
# share of the elements.
def _rbf64(x):
    """float64 -> nearest bf16 value (as float64), through fp32 like the hardware path: the fp32 rounding is exact for
    products / sums of two bf16 values, so double rounding cannot occur for the single-operation steps below."""
    return bf16_to_f32(f32_to_bf16(np.asarray(x, dtype=np.float64).astype(np.float32))).astype(np.float64)


def scale_and_mask(scores_bits, mask_bits, scaling):
    """The softmax's input in bf16: bf16(bf16(score x scaling) + mask), each step ONE rounding of an exactly representable fp32
    result (modeling_bert.py:142-146: MulFunctional with a python float, then `+ attention_mask`).  Returns float64 values."""
    s = bf16_to_f32(scores_bits).astype(np.float64)
    t = _rbf64(s.astype(np.float32) * np.float32(scaling))
    if mask_bits is not None:
        m = bf16_to_f32(mask_bits).astype(np.float64)
        with np.errstate(over="ignore"):
            t = _rbf64(t + m)
    return t


def softmax_fq(scores_bits, mask_bits, scaling, qmap):
    """scores_bits: uint16 [..., rows, cols] bf16 patterns; mask_bits: broadcastable uint16 array or None; scaling: python
    float (multiplied as fp32, like torch's bf16-tensor x scalar); qmap: uint16[65536] or None.  Returns (p_bits, pq_bits).
    Pinned to upstream's BertSelfAttention twin by tests/test_oracle_golden.py::test_attention_chain (tests/golden/attn_chain.*)."""
    t = scale_and_mask(scores_bits, mask_bits, scaling)
    mx = t.max(axis=-1, keepdims=True)
    e = np.exp(t - mx)
    p = e / e.sum(axis=-1, keepdims=True)
    p_bits = f32_to_bf16(p.astype(np.float32))
    return p_bits, (vmap_bf16(p_bits, qmap) if qmap is not None else p_bits)


def softmax_fq_f32(scores, mask, scaling, qmap):
    """The same chain on an fp32 model (no bf16 rounding points; the probabilities' fake-quantizer indexes the map with
    hi16 | sticky, decomposed.py:151-153): scores / mask float32 arrays.  exp and the sums in float64, ONE rounding to fp32.
    Returns (p float32, pq float32)."""
    t = (scores.astype(np.float32) * np.float32(scaling)).astype(np.float32)
    if mask is not None:
        with np.errstate(over="ignore"):
            t = (t + mask.astype(np.float32)).astype(np.float32)
    t = t.astype(np.float64)
    mx = t.max(axis=-1, keepdims=True)
    e = np.exp(t - mx)
    p = (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)
    return p, (vmap_f32(p, qmap) if qmap is not None else p)


def attention_fq(q_bits, k_bits, v_bits, mask_bits, scaling, qmap):
    """q [B,H,Sq,D], k / v [B,H,Sk,D] as uint16 bf16 patterns (already fake-quantized); returns the attention output
    [B,H,Sq,D] as bf16 patterns plus the quantized probabilities' patterns."""
    q = bf16_to_f32(q_bits).astype(np.float64)
    k = bf16_to_f32(k_bits).astype(np.float64)
    v = bf16_to_f32(v_bits).astype(np.float64)
    scores = f32_to_bf16(np.einsum("bhqd,bhkd->bhqk", q, k).astype(np.float32))
    _, pq_bits = softmax_fq(scores, mask_bits, scaling, qmap)
    pq = bf16_to_f32(pq_bits).astype(np.float64)
    out = np.einsum("bhqk,bhkd->bhqd", pq, v)
    return f32_to_bf16(out.astype(np.float32)), pq_bits
