"""CPU-side checks of the C-ABI library (no kernel launches): it loads without a GPU, exports every
symbol include/qt_hip.h declares, and its host functions (value-map builder, closed-form
descriptors, fp8 / posit rounding) are bit-exact against the golden vectors."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from quantized_training import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
MAPS = np.load(os.path.join(G, "maps.npz"))


def test_library_loads_and_exports_every_declared_symbol():
    L = _native.lib()
    assert L.qt_abi_version() == 3
    header = open(os.path.join(ROOT, "include", "qt_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(qt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in qt_hip.h but not exported"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert L.qt_status_string(0) == b"ok"


_C_KEYS = sorted(k for k in MAPS.files if "__" not in k and not k.startswith("nf"))


@pytest.mark.parametrize("key", _C_KEYS)
def test_build_map_bit_exact(key):
    m = _native.build_map_u16(None if key == "none" else key)
    assert np.array_equal(m, MAPS[key])
    sha = json.load(open(os.path.join(G, "maps_sha256.json")))[key]
    assert hashlib.sha256(m.astype("<u2").tobytes()).hexdigest() == sha


@pytest.mark.parametrize("key", _C_KEYS)
def test_closed_form_descriptor_equals_table(key):
    """Every closed-form descriptor the kernels may use instead of the table reproduces the table on
    all 65 536 inputs."""
    dt = None if key == "none" else key
    f = _native.format_for(dt)
    if f.kind == _native.QT_FMT_LUT:
        pytest.skip("table-only dtype")
    L = _native.lib()
    got = np.fromiter((L.qt_format_apply_host(ctypes.byref(f), i) for i in range(65536)), dtype=np.uint16, count=65536)
    assert np.array_equal(got, MAPS[key])


def test_expected_closed_forms():
    assert _native.format_for("e4m3").kind == _native.QT_FMT_FP_SAT
    assert _native.format_for("fp8.E5M2").kind == _native.QT_FMT_FP_SAT
    assert _native.format_for("INT8").kind == _native.QT_FMT_INT
    assert _native.format_for("posit8_1").kind == _native.QT_FMT_LUT
    assert _native.format_for("fp8_e4m3").kind == _native.QT_FMT_LUT
    assert _native.format_for(None).kind == _native.QT_FMT_IDENTITY


@pytest.mark.parametrize("bad", ["bogus", "int", "fp9_e4m3", "posit8", "e4m3x", "fp_e4m3", "nf4", "INT8 "])
def test_unknown_dtype_is_value_error(bad):
    with pytest.raises(ValueError):
        _native.build_map_u16(bad)
    with pytest.raises(ValueError):
        _native.format_for(bad)


def _canon32(b):
    b = b.copy()
    nan = ((b & 0x7F800000) == 0x7F800000) & ((b & 0x7FFFFF) != 0)
    b[nan] = 0x7FC00000
    return b


def test_host_rounding_functions():
    d = np.load(os.path.join(G, "direct_fns.npz"))
    L = _native.lib()
    x = d["x"].view(np.float32).copy()
    y = np.empty_like(x)
    assert L.qt_round_fp8_host(x.ctypes.data, y.ctypes.data, x.size, 3, 448.0, 2.0 ** -6) == 0
    assert np.array_equal(_canon32(y.view(np.uint32)), d["e4m3"])
    assert L.qt_round_fp8_host(x.ctypes.data, y.ctypes.data, x.size, 2, 57344.0, 2.0 ** -14) == 0
    assert np.array_equal(_canon32(y.view(np.uint32)), d["e5m2"])
    for nb, es in [(8, 0), (8, 1), (8, 2), (16, 1), (16, 2), (6, 1)]:
        assert L.qt_round_posit_host(x.ctypes.data, y.ctypes.data, x.size, nb, es) == 0
        assert np.array_equal(_canon32(y.view(np.uint32)), d[f"posit{nb}_{es}"]), (nb, es)
    assert L.qt_round_posit_host(x.ctypes.data, y.ctypes.data, x.size, 2, 0) < 0
    assert L.qt_round_fp8_host(None, None, 4, 3, 448.0, 2.0 ** -6) < 0


def test_bad_arguments_are_reported_not_thrown():
    L = _native.lib()
    f = _native.format_for("e4m3")
    assert L.qt_fake_quant_bf16(None, None, 16, ctypes.byref(f), None, None, None, None) == -2
    assert L.qt_fake_quant_bf16(None, None, 0, ctypes.byref(f), None, None, None, None) == 0
    assert L.qt_scale_update(None, 1, 1, None, 1.0, 0, None) == -2
    assert L.qt_linear_fq_bf16(None, None, None, None, 0, 0, 8, None, None, None) == 0
    assert L.qt_linear_fq_bf16(None, None, None, None, 4, 4, 8, None, None, None) == -2


# ---- row form of a value map (qt_build_rowparams): what qt_linear_fqt_bf16 applies to its weight operand ------------------
_ROW_KEYS = [k for k in _C_KEYS if k not in ("none", "float32")]


@pytest.mark.parametrize("key", _ROW_KEYS)
def test_row_form_reproduces_the_value_map(key):
    """Every row the builder does not flag reproduces the oracle-pinned map (tests/golden/maps.npz = upstream's
    get_quantization_map, fake_quantize.py:31-95) on all of its 128 inputs, bit for bit up to the sign of a zero (which no product
    sees); the flagged rows are exactly the ones the kernel redoes with the map itself, and there are few of them."""
    m = MAPS[key].astype(np.uint16)
    rp = _native.build_rowparams(m)
    L = _native.lib()
    f = ctypes.c_int(0)
    flagged = np.ctypeslib.as_array(rp.flagged).astype(bool)
    assert flagged.sum() == rp.n_flagged
    assert flagged[255] and flagged[511]                      # non-finite inputs never take the row form
    got = np.empty(65536, dtype=np.uint16)
    fl = np.empty(65536, dtype=bool)
    for b in range(65536):
        got[b] = L.qt_rowparams_apply_host(ctypes.byref(rp), b, ctypes.byref(f))
        fl[b] = bool(f.value)
    row = np.arange(65536) >> 7
    expect_flag = flagged[row] if rp.signed_rows else flagged[row & 0xFF]
    assert np.array_equal(fl, expect_flag)
    same = (got == m) | (((got | m) & 0x7FFF) == 0)
    assert same[~fl].all(), np.flatnonzero(~same & ~fl)[:8]
    # a flagged row still answers for its FIRST input (mantissa 0; row 0: exact zero, the commonest input of all) -- {0, 1, y0, y0} with
    # y0 the map's value there -- unless that value is a NaN or carries a sign, which a clamp cannot produce: {1, 1, 0, 0}
    rows = np.ctypeslib.as_array(rp.row).reshape(512, 4).astype(np.uint32)
    for r in np.flatnonzero(flagged):
        first = int(m[r << 7])
        usable = (first & 0x7FFF) <= 0x7F80 and not (first & 0x8000)
        assert rows[r, 1] == 1
        if usable:
            assert rows[r, 0] == 0 and rows[r, 2] == rows[r, 3] == (first << 16), (key, r)
        else:
            assert rows[r, 0] == 1, (key, r)
    # the rows weights live in (2^-40 .. 2^15) are all covered for the formats of BASELINE.json and the README tables
    if key in ("posit8_0", "posit8_1", "posit8_2", "int8", "int4", "e4m3", "e5m2", "fp8_e4m3", "fp6_e3m2", "fp6_e2m3", "fp4_e2m1"):
        assert not flagged[127 - 40:127 + 15].any()
        assert rp.n_flagged <= 8


def test_row_form_sign_rules():
    """Odd maps copy the input's sign, unsigned formats (fp8_e5m3, uintN) keep results positive, intN needs rows by sign."""
    rp = _native.build_rowparams(MAPS["posit8_1"].astype(np.uint16))
    assert rp.sign_mask == 0x80008000 and rp.signed_rows == 0
    rp = _native.build_rowparams(MAPS["int8"].astype(np.uint16))
    assert rp.sign_mask == 0x80008000 and rp.signed_rows == 1     # -128 has no positive twin
    rp = _native.build_rowparams(MAPS["uint8"].astype(np.uint16))
    assert rp.sign_mask == 0 and rp.signed_rows == 1
    rp = _native.build_rowparams(MAPS["fp8_e5m3"].astype(np.uint16))
    assert rp.sign_mask == 0 and rp.signed_rows == 0
