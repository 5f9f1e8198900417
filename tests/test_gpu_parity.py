"""GPU parity: the HIP kernels (called through the C ABI of include/qt_hip.h) against the CPU oracle
and the committed golden vectors.  Bit-exact for every elementwise path (NaNs compared as NaN);
the GEMMs are compared within an fp32-accumulation tolerance stated in the test.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import ctypes
import math
import json
import os

import numpy as np
import pytest
import torch

from oracle import qt_oracle as o

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nv():
    from quantized_training import _native
    assert torch.cuda.is_available(), "these tests need the GPU"
    _native.lib()
    return _native


def dev_u16(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda()


def dev_f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host_u16(t):
    return t.cpu().numpy().view(np.uint16)


def host_u32(t):
    return t.cpu().numpy().view(np.uint32)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


LUT_FMT = (0, 0, 0, 0.0, 0.0)

DTYPES = ["int8", "int4", "uint8", "e4m3", "e5m2", "fp8_e4m3", "fp8_e5m2", "fp6_e3m2", "fp4_e2m1",
          "posit8_0", "posit8_1", "posit8_2", "posit16_1"]


def run_fq_bf16(nv, xbits, dtype, scale, force_lut=False, observe=False):
    L = nv.lib()
    x = dev_u16(xbits)
    y = torch.empty_like(x)
    lut = dev_u16(nv.build_map_u16(dtype))
    fmt = nv.QtFormat(*LUT_FMT) if force_lut else nv.format_for(dtype)
    s = dev_f32(np.array([scale], np.float32))
    amax = torch.zeros(1, dtype=torch.int32, device="cuda") if observe else None
    nv.check(L.qt_fake_quant_bf16(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), lut.data_ptr(),
                                  s.data_ptr(), amax.data_ptr() if observe else None, stream()), "fq")
    torch.cuda.synchronize()
    return host_u16(y), (host_u32(amax)[0] if observe else None)


def run_fq_f32(nv, x, dtype, scale, force_lut=False, observe=False):
    L = nv.lib()
    xd = dev_f32(x)
    y = torch.empty_like(xd)
    lut = dev_u16(nv.build_map_u16(dtype))
    fmt = nv.QtFormat(*LUT_FMT) if force_lut else nv.format_for(dtype)
    s = dev_f32(np.array([scale], np.float32))
    amax = torch.zeros(1, dtype=torch.int32, device="cuda") if observe else None
    nv.check(L.qt_fake_quant_f32(xd.data_ptr(), y.data_ptr(), xd.numel(), ctypes.byref(fmt), lut.data_ptr(),
                                 s.data_ptr(), amax.data_ptr() if observe else None, stream()), "fq")
    torch.cuda.synchronize()
    return host_u32(y), (host_u32(amax)[0] if observe else None)


def expect_bf16(xbits, dtype, scale):
    qmap = o.get_quantization_map(dtype)
    sb = o.f32_to_bf16(np.array([scale], np.float32))
    return o.canon_nan16(o.fq_bf16(xbits, qmap, sb))


def expect_f32(x, dtype, scale):
    qmap = o.get_quantization_map(dtype)
    return o.canon_nan32(o.fq_f32(x, qmap, np.float32(scale)).view(np.uint32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scale", [1.0, 0.037, 0.125, 3.5, 1e-3])
def test_exhaustive_bf16_all_patterns(nv, dtype, scale):
    """All 65 536 bf16 inputs, closed form (where there is one) and table kernels."""
    x = o.all_bf16_patterns()
    exp = expect_bf16(x, dtype, scale)
    for force_lut in (False, True):
        got, _ = run_fq_bf16(nv, x, dtype, scale, force_lut)
        assert np.array_equal(o.canon_nan16(got), exp), (dtype, scale, force_lut)


@pytest.mark.parametrize("dtype", ["posit8_1", "posit8_2", "posit8_0", "fp8_e4m3", "fp6_e3m2", "fp6_e2m3", "fp4_e2m1", "posit16_1", "fp8_e5m3"])
@pytest.mark.parametrize("scale", [1.0, 0.037, 3.5])
def test_row_form_pass_equals_the_value_map(nv, dtype, scale):
    """Table formats take the ROW FORM of their map in the elementwise pass (qt_format.p1 bit 0: the row words of qt_build_rowparams
    sit behind the 65 536 map entries; csrc/qt_device.h Rounder<kFmtRows>): all 65 536 bf16 inputs, tiled to a tensor large enough for
    the streaming kernel, bit for bit the oracle's fake-quant -- flagged rows (through the map in global memory), the sign of zero
    results and NaN included; and the same through the module, which is how the product reaches it."""
    import quantized_training as qt
    from quantized_training.fake_quantize import _launch_format
    x = np.tile(o.all_bf16_patterns(), 40)                        # 2.6 M elements
    exp = np.tile(expect_bf16(o.all_bf16_patterns(), dtype, scale), 40)
    lut = qt.get_quantization_map(dtype, torch.device("cuda"))
    fmt = _launch_format(nv.format_for(dtype), lut)
    assert fmt.kind == nv.QT_FMT_LUT and (fmt.p1 & 1), "the row form covers this map"
    xd = torch.from_numpy(x.view(np.int16)).cuda()
    yd = torch.empty_like(xd)
    sc = torch.tensor([scale], dtype=torch.float32, device="cuda")
    for n in (x.size, 65536, 4096 + 8):
        nv.check(nv.lib().qt_fake_quant_bf16(xd.data_ptr(), yd.data_ptr(), n, ctypes.byref(fmt), lut.data_ptr(), sc.data_ptr(), None, stream()), "fq")
        got = host_u16(yd[:n])
        assert np.array_equal(o.canon_nan16(got), exp[:n]), (dtype, scale, n)
    if scale == 1.0:
        fq = qt.FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
        y = fq(xd.view(torch.bfloat16))
        assert np.array_equal(o.canon_nan16(host_u16(y.view(torch.int16))), exp)
        # a module built on the host and moved (`model.to("cuda")` copies the plain map into the buffer) takes the cached device
        # map -- and with it the row form -- on its first call
        moved = qt.FusedAmaxObsFakeQuantize(dtype=dtype).to("cuda")
        assert getattr(moved.qmap, "_qt_rows", 0) == 0
        y = moved(xd.view(torch.bfloat16))
        assert moved.qmap is qt.get_quantization_map(dtype, xd.device) and (_launch_format(moved._qt_format, moved.qmap).p1 & 1)
        assert np.array_equal(o.canon_nan16(host_u16(y.view(torch.int16))), exp)
        # strided rows (qt_fake_quant_rows_bf16: a [B, S, H, D] projection seen as [B, H, S, D], and K^T)
        t = xd.view(torch.bfloat16)[:4 * 64 * 8 * 128].view(4, 64, 8, 128).transpose(1, 2)
        e = torch.from_numpy(exp[:t.numel()].view(np.int16)).view(4, 64, 8, 128).transpose(1, 2)
        for view, want in ((t, e), (t.transpose(-1, -2), e.transpose(-1, -2))):
            assert not view.is_contiguous()
            got = fq(view)
            assert np.array_equal(o.canon_nan16(host_u16(got.contiguous().view(torch.int16))).reshape(-1),
                                  want.contiguous().numpy().view(np.uint16).reshape(-1)), dtype


@pytest.mark.parametrize("dtype", ["e4m3", "int8"])
def test_division_by_many_scales(nv, dtype):
    """x / s must be torch's correctly rounded fp32 division for every scale: sweep scales across the
    fast (refined reciprocal) and the guarded (full division) ranges, bf16 (all patterns) and fp32."""
    rng = np.random.default_rng(42)
    scales = list(2.0 ** rng.uniform(-24, 19, 24)) + [2.0 ** -25, 2.0 ** 20, 2.0 ** -30, 2.0 ** 40, 1e-38, 3e38,
                                                      0.1, 1.0 / 3.0, 448.0, 1.0 / 448.0, 127.0, 5e-7]
    xb = o.all_bf16_patterns()
    x32 = _sample_f32(1 << 15, 9)
    for s in scales:
        s = float(np.float32(s))
        got, _ = run_fq_bf16(nv, xb, dtype, s)
        assert np.array_equal(o.canon_nan16(got), expect_bf16(xb, dtype, s)), ("bf16", s)
        got, _ = run_fq_f32(nv, x32, dtype, s)
        assert np.array_equal(o.canon_nan32(got), expect_f32(x32, dtype, s)), ("f32", s)


@pytest.mark.parametrize("dtype", ["int8", "e4m3", "e5m2", "posit8_1", "fp8_e4m3", "posit16_1", "uint8"])
def test_closed_form_equals_table_on_device(nv, dtype):
    x = o.all_bf16_patterns()
    a, _ = run_fq_bf16(nv, x, dtype, 1.0, False)
    b, _ = run_fq_bf16(nv, x, dtype, 1.0, True)
    assert np.array_equal(o.canon_nan16(a), o.canon_nan16(b))
    assert np.array_equal(o.canon_nan16(a), o.get_quantization_map(dtype))


def _sample_f32(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(n // 2).astype(np.float32) * np.float32(10.0) ** rng.uniform(-3, 3, n // 2).astype(np.float32)
    b = (np.sign(rng.standard_normal(n - n // 2)) * 2.0 ** rng.uniform(-30, 30, n - n // 2)).astype(np.float32)
    x = np.concatenate([a, b]).astype(np.float32)
    u = x.view(np.uint32).copy()
    u[::7] &= 0xFFFF0000                       # exactly bf16-representable
    u[1::7] = (u[1::7] & 0xFFFF0000) | 1       # sticky fold
    x = u.view(np.float32)
    x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 100.7, 127.5, -128.5]
    return x


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scale", [1.0, 0.0123, 2.0])
def test_f32_inputs(nv, dtype, scale):
    x = _sample_f32(1 << 16, 3)
    exp = expect_f32(x, dtype, scale)
    for force_lut in (False, True):
        got, _ = run_fq_f32(nv, x, dtype, scale, force_lut)
        assert np.array_equal(o.canon_nan32(got), exp), (dtype, scale, force_lut)


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 4095, 4096, 4097, (1 << 21) + 13, (1 << 22) + 8 * 1024 * 5 + 3])
@pytest.mark.parametrize("dtype", ["e4m3", "posit8_1"])
def test_sizes_and_tails_bf16(nv, n, dtype):
    """Empty, ragged and multi-tile sizes: vector body, ragged last tile, scalar tail, table-in-LDS kernel."""
    rng = np.random.default_rng(n + 1)
    x = rng.integers(0, 65536, n, dtype=np.uint32).astype(np.uint16)
    if n == 0:
        L = nv.lib()
        fmt = nv.format_for(dtype)
        assert L.qt_fake_quant_bf16(None, None, 0, ctypes.byref(fmt), None, None, None, stream()) == 0
        return
    for scale in (1.0, 0.3):
        got, amax = run_fq_bf16(nv, x, dtype, scale, observe=True)
        assert np.array_equal(o.canon_nan16(got), expect_bf16(x, dtype, scale)), (n, scale)
        finite = x[(x & 0x7FFF) <= 0x7F80]
        nan_in = np.any((x & 0x7FFF) > 0x7F80)
        if nan_in:
            assert (amax & 0x7FFFFFFF) > 0x7F800000
        else:
            assert amax == (int((finite & 0x7FFF).max()) << 16)


@pytest.mark.parametrize("n", [5, 4099, (1 << 21) + 5])
def test_sizes_and_tails_f32(nv, n):
    x = _sample_f32(max(n, 16), n)[:n]
    x = np.where(np.isnan(x), np.float32(1.0), x).astype(np.float32)
    for dtype in ("e4m3", "posit8_1", "int8"):
        got, amax = run_fq_f32(nv, x, dtype, 0.25, observe=True)
        assert np.array_equal(o.canon_nan32(got), expect_f32(x, dtype, 0.25))
        assert amax == int(np.abs(x).max().view(np.uint32))


def test_unaligned_pointers(nv):
    """Views at odd element offsets take the element-granular kernel."""
    L = nv.lib()
    xb = np.random.default_rng(0).integers(0, 65536, 10001, dtype=np.uint32).astype(np.uint16)
    x = dev_u16(xb)
    y = torch.zeros_like(x)
    fmt = nv.format_for("e4m3")
    nv.check(L.qt_fake_quant_bf16(x.data_ptr() + 2, y.data_ptr() + 6, 9000, ctypes.byref(fmt), None, None, None, stream()), "fq")
    torch.cuda.synchronize()
    assert np.array_equal(o.canon_nan16(host_u16(y)[3:9003]), expect_bf16(xb[1:9001], "e4m3", 1.0))


def test_vmap_golden(nv):
    d = np.load(os.path.join(G, "vmap.npz"))
    L = nv.lib()
    for dt in ["int8", "int4", "e4m3", "e5m2", "fp8_e4m3", "fp4_e2m1", "posit8_1", "posit16_1"]:
        lut = dev_u16(nv.build_map_u16(dt))
        fmt = nv.format_for(dt)
        x = dev_u16(d["xb"]); y = torch.empty_like(x)
        nv.check(L.qt_vmap_bf16(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), lut.data_ptr(), stream()), "vmap")
        assert np.array_equal(o.canon_nan16(host_u16(y)), d[f"yb_{dt}"]), dt
        x = torch.from_numpy(d["x32"].view(np.float32)).cuda(); y = torch.empty_like(x)
        nv.check(L.qt_vmap_f32(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), lut.data_ptr(), stream()), "vmap")
        assert np.array_equal(o.canon_nan32(host_u32(y)), d[f"y32_{dt}"]), dt
        x = dev_u16(d["xh"]); y = torch.empty_like(x)
        nv.check(L.qt_vmap_f16(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), lut.data_ptr(), stream()), "vmap")
        got, exp = host_u16(y), d[f"yh_{dt}"]
        nan = ((exp & 0x7C00) == 0x7C00) & ((exp & 0x3FF) != 0)
        assert np.array_equal(got[~nan], exp[~nan]), dt
        assert np.all(((got[nan] & 0x7C00) == 0x7C00) & ((got[nan] & 0x3FF) != 0))


def test_direct_rounding_golden(nv):
    d = np.load(os.path.join(G, "direct_fns.npz"))
    L = nv.lib()
    x = torch.from_numpy(d["x"].view(np.float32)).cuda()
    y = torch.empty_like(x)
    nv.check(L.qt_round_fp8_f32(x.data_ptr(), y.data_ptr(), x.numel(), 3, 448.0, 2.0 ** -6, stream()), "fp8")
    assert np.array_equal(o.canon_nan32(host_u32(y)), d["e4m3"])
    nv.check(L.qt_round_fp8_f32(x.data_ptr(), y.data_ptr(), x.numel(), 2, 57344.0, 2.0 ** -14, stream()), "fp8")
    assert np.array_equal(o.canon_nan32(host_u32(y)), d["e5m2"])
    for nb, es in [(8, 0), (8, 1), (8, 2), (16, 1), (16, 2), (6, 1)]:
        nv.check(L.qt_round_posit_f32(x.data_ptr(), y.data_ptr(), x.numel(), nb, es, stream()), "posit")
        assert np.array_equal(o.canon_nan32(host_u32(y)), d[f"posit{nb}_{es}"]), (nb, es)


def test_ops_quantize_dequantize_golden(nv):
    """torch.ops.quantized_ops.quantize / dequantize on device tensors against the reference's outputs."""
    import quantized_training as qt
    d = np.load(os.path.join(G, "quant_dequant.npz"))
    cases = json.load(open(os.path.join(G, "quant_dequant.json")))
    q = torch.ops.quantized_ops
    for c in cases:
        n = c["name"]
        qmap = qt.get_quantization_map(c["dtype"], "cuda")
        if c["in"] == "f32":
            x = torch.from_numpy(d["x32"].view(np.float32)).cuda()
            s = torch.from_numpy(d[n + "_scale"].view(np.float32)).cuda()
            can = lambda t: o.canon_nan32(host_u32(t))  # noqa: E731
        else:
            x = dev_u16(d["xb"]).view(torch.bfloat16)
            s = dev_u16(d[n + "_scale"]).view(torch.bfloat16)
            can = lambda t: o.canon_nan16(host_u16(t.view(torch.int16)))  # noqa: E731
        yq = q.quantize(x, s, None, None, None, qmap)
        assert np.array_equal(can(yq), d[n + "_q"]), n
        assert np.array_equal(can(q.dequantize(yq, s, None, None, None, None, None)), d[n + "_dq"]), n
        assert np.array_equal(can(q.dequantize(x, s, None, None, None, qmap, qmap)), d[n + "_dq2"]), n
    qmap = qt.get_quantization_map("uint8", "cuda")
    x = torch.from_numpy(d["x32"].view(np.float32)).cuda()
    s, z = torch.tensor([0.05], device="cuda"), torch.tensor([128.0], device="cuda")
    yq = q.quantize(x, s, z, None, None, qmap)
    assert np.array_equal(o.canon_nan32(host_u32(yq)), d["zp_q"])
    assert np.array_equal(o.canon_nan32(host_u32(q.dequantize(yq, s, z, None, None, None, None))), d["zp_dq"])


FQ_META = json.load(open(os.path.join(G, "fake_quant.json")))


@pytest.mark.parametrize("case", FQ_META, ids=[c["name"] for c in FQ_META])
def test_module_traces_golden(nv, case):
    """FusedAmaxObsFakeQuantize on device tensors reproduces the reference's multi-call traces:
    outputs, scale and amax_history after every call (delayed scaling, per-channel, pow2, NaN / zero guards)."""
    from dataclasses import asdict
    import quantized_training as qt
    d = np.load(os.path.join(G, "fake_quant.npz"))
    kw = asdict(qt.QuantizationSpec.from_str(case["spec"]))
    m = qt.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=case["pow2"], device="cuda")
    bf16 = case["in"] == "bf16"
    for ci in range(case["n_calls"]):
        k = f"{case['name']}__{ci}__"
        if bf16:
            x = dev_u16(d[k + "x"]).view(torch.bfloat16).reshape(case["shape"])
        else:
            x = torch.from_numpy(d[k + "x"].view(np.float32)).cuda().reshape(case["shape"])
        with torch.no_grad():
            y = m(x)
        assert y.dtype == x.dtype and y.shape == x.shape
        if bf16:
            assert np.array_equal(o.canon_nan16(host_u16(y.contiguous().view(torch.int16))).reshape(-1), d[k + "y"].reshape(-1)), ci
        else:
            assert np.array_equal(o.canon_nan32(host_u32(y.contiguous())).reshape(-1), d[k + "y"].reshape(-1)), ci
        assert np.array_equal(host_u32(m.scale.reshape(-1)), d[k + "scale"]), (ci, m.scale)
        if m._observe:
            assert np.array_equal(o.canon_nan32(host_u32(m.amax_history.reshape(-1))), d[k + "hist"]), ci
            assert list(m.amax_history.shape) == case["calls"][ci]["hist_shape"]
            assert list(m.scale.shape) == case["calls"][ci]["scale_shape"]
    assert sorted(m.state_dict().keys()) == sorted(case["state_dict"].keys())


def _oracle_fq_matrix(xbits, dtype, scale):
    return o.bf16_to_f32(o.fq_bf16(xbits, o.get_quantization_map(dtype), o.f32_to_bf16(np.array([scale], np.float32))))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (100, 72, 136), (1, 16, 8), (333, 130, 1000)])
@pytest.mark.parametrize("wdtype,wscale", [("e4m3", 1.0), ("int8", 0.02), ("posit8_1", 1.0), (None, 1.0)])
def test_linear_fq_gemm(nv, M, N, K, wdtype, wscale):
    """y = x @ fq(W)^T + b: operands quantized exactly like the oracle, fp32 accumulation.
    Tolerance: |err| <= 2^-7 * |y| + 1e-2 * sqrt(K) * 2^-8 (one bf16 output rounding + accumulation order)."""
    L = nv.lib()
    rng = np.random.default_rng(M * 131 + N * 7 + K)
    xb = o.f32_to_bf16(rng.standard_normal((M, K)).astype(np.float32))
    wb = o.f32_to_bf16((rng.standard_normal((N, K)) * 0.05).astype(np.float32))
    bb = o.f32_to_bf16(rng.standard_normal(N).astype(np.float32))
    x, w, b = dev_u16(xb), dev_u16(wb), dev_u16(bb)
    y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    qx = nv.QtOperandQ(); qx.fmt = nv.QtFormat(nv.QT_FMT_IDENTITY, 0, 0, 0.0, 0.0)
    qw = nv.QtOperandQ()
    amax = torch.zeros(1, dtype=torch.int32, device="cuda")
    keep = []
    if wdtype is None:
        qw.fmt = nv.QtFormat(nv.QT_FMT_IDENTITY, 0, 0, 0.0, 0.0)
        wq = o.bf16_to_f32(wb)
    else:
        qw.fmt = nv.format_for(wdtype)
        lut = dev_u16(nv.build_map_u16(wdtype)); keep.append(lut)
        qw.lut_dev = lut.data_ptr()
        s = dev_f32(np.array([wscale], np.float32)); keep.append(s)
        qw.scale_f32_dev = s.data_ptr()
        qw.amax_bits_dev = amax.data_ptr()
        wq = _oracle_fq_matrix(wb, wdtype, wscale)
    nv.check(L.qt_linear_fq_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K,
                                 ctypes.byref(qx), ctypes.byref(qw), stream()), "linear")
    torch.cuda.synchronize()
    got = o.bf16_to_f32(host_u16(y))
    ref = o.bf16_to_f32(xb).astype(np.float64) @ wq.astype(np.float64).T + o.bf16_to_f32(bb).astype(np.float64)
    tol = 2.0 ** -7 * np.abs(ref) + 1e-2 * np.sqrt(K) * 2.0 ** -8
    assert np.all(np.abs(got - ref) <= tol), float(np.max(np.abs(got - ref) - tol))
    if wdtype is not None:
        assert host_u32(amax)[0] == (int((wb & 0x7FFF).max()) << 16)


@pytest.mark.parametrize("nn_layout", [False, True])
def test_bmm_fq_gemm(nv, nn_layout):
    """Batched QK^T-style (k-contiguous B) and AV-style (n-contiguous B) products with both operands quantized."""
    L = nv.lib()
    B, M, N, K = 3, 200, 136, 72
    rng = np.random.default_rng(5)
    ab = o.f32_to_bf16(rng.standard_normal((B, M, K)).astype(np.float32))
    if nn_layout:
        bb = o.f32_to_bf16(rng.standard_normal((B, K, N)).astype(np.float32))
        ldb_k, ldb_n = N, 1
    else:
        bb = o.f32_to_bf16(rng.standard_normal((B, N, K)).astype(np.float32))
        ldb_k, ldb_n = 1, K
    a, b = dev_u16(ab), dev_u16(bb)
    y = torch.empty((B, M, N), dtype=torch.int16, device="cuda")
    lut = dev_u16(nv.build_map_u16("e4m3"))
    amax_a = torch.zeros(1, dtype=torch.int32, device="cuda")
    amax_b = torch.zeros(1, dtype=torch.int32, device="cuda")
    qa = nv.QtOperandQ(); qa.fmt = nv.format_for("e4m3"); qa.amax_bits_dev = amax_a.data_ptr()
    qb = nv.QtOperandQ(); qb.fmt = nv.QtFormat(*LUT_FMT); qb.lut_dev = lut.data_ptr(); qb.amax_bits_dev = amax_b.data_ptr()
    nv.check(L.qt_bmm_fq_bf16(a.data_ptr(), b.data_ptr(), y.data_ptr(), B, M, N, K, K, M * K, ldb_k, ldb_n, N * K,
                              ctypes.byref(qa), ctypes.byref(qb), stream()), "bmm")
    torch.cuda.synchronize()
    aq = _oracle_fq_matrix(ab, "e4m3", 1.0).astype(np.float64)
    bq = _oracle_fq_matrix(bb, "e4m3", 1.0).astype(np.float64)
    ref = aq @ (bq if nn_layout else bq.transpose(0, 2, 1))
    got = o.bf16_to_f32(host_u16(y))
    tol = 2.0 ** -7 * np.abs(ref) + 1e-2 * np.sqrt(K) * 2.0 ** -8
    assert np.all(np.abs(got - ref) <= tol)
    assert host_u32(amax_a)[0] == (int((ab & 0x7FFF).max()) << 16)
    assert host_u32(amax_b)[0] == (int((bb & 0x7FFF).max()) << 16)


def test_full_size_properties(nv):
    """LLaMA-2-7B sized tensor (4096 x 11008 bf16): size-independent properties --
    idempotence fq(fq(x)) == fq(x), the output only takes map values, and amax == max|x|."""
    L = nv.lib()
    torch.manual_seed(0)
    x = (torch.randn(4096, 11008, device="cuda") * 0.02).bfloat16()
    for dtype in ("e4m3", "posit8_1"):
        fmt = nv.format_for(dtype)
        lut = dev_u16(nv.build_map_u16(dtype))
        y = torch.empty_like(x); z = torch.empty_like(x)
        amax = torch.zeros(1, dtype=torch.int32, device="cuda")
        nv.check(L.qt_fake_quant_bf16(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), lut.data_ptr(), None,
                                      amax.data_ptr(), stream()), "fq")
        nv.check(L.qt_fake_quant_bf16(y.data_ptr(), z.data_ptr(), x.numel(), ctypes.byref(fmt), lut.data_ptr(), None,
                                      None, stream()), "fq")
        assert torch.equal(y.view(torch.int16), z.view(torch.int16))
        assert amax.view(torch.float32).item() == x.abs().max().float().item()
        # spot-check 1M elements against the oracle
        idx = torch.randint(0, x.numel(), (1 << 20,), device="cuda")
        xs = host_u16(x.view(-1)[idx].view(torch.int16))
        ys = host_u16(y.view(-1)[idx].view(torch.int16))
        assert np.array_equal(ys, expect_bf16(xs, dtype, 1.0))
        vals = torch.unique(y.view(torch.int16)).cpu().numpy().view(np.uint16)
        assert np.all(np.isin(vals, o.get_quantization_map(dtype)))


@pytest.mark.parametrize("dtype,tdtype", [("e4m3", torch.float8_e4m3fn), ("e5m2", torch.float8_e5m2)])
@pytest.mark.parametrize("scale", [1.0, 0.25, 0.037])
def test_fp8_side_output(nv, dtype, tdtype, scale):
    """qt_fake_quant_bf16_fp8: the FP8 bytes decode (OCP, torch's float8 types) to exactly the oracle's
    quantized code q = map[x / s]; the optional bf16 output is the usual fake-quantized tensor."""
    L = nv.lib()
    xb = o.all_bf16_patterns()
    x = dev_u16(xb)
    fmt = nv.format_for(dtype)
    s = dev_f32(np.array([scale], np.float32))
    qmap = o.get_quantization_map(dtype)
    sb = o.f32_to_bf16(np.array([scale], np.float32))
    q_exp = o.canon_nan16(o.quantize_bf16(xb, qmap, sb))
    for both, obs in ((True, True), (False, True), (True, False), (False, False)):
        # without the observer and at scale 1 both variants take the hardware conversion (finite vectors) or the closed
        # form (vectors holding a NaN / Inf pattern); the pattern sweep puts both kinds next to each other
        y = torch.zeros_like(x)
        y8 = torch.zeros(x.numel(), dtype=torch.uint8, device="cuda")
        amax = torch.zeros(1, dtype=torch.int32, device="cuda")
        nv.check(L.qt_fake_quant_bf16_fp8(x.data_ptr(), y.data_ptr() if both else None, y8.data_ptr(), x.numel(),
                                          ctypes.byref(fmt), s.data_ptr() if (obs or scale != 1.0) else None,
                                          amax.data_ptr() if obs else None, stream()), "fq8")
        torch.cuda.synchronize()
        dec = y8.view(tdtype).float().bfloat16().view(torch.int16)
        assert np.array_equal(o.canon_nan16(host_u16(dec)), q_exp), (dtype, scale, both, obs)
        if both:
            assert np.array_equal(o.canon_nan16(host_u16(y)), expect_bf16(xb, dtype, scale)), (dtype, scale, obs)
        if obs:
            assert (host_u32(amax)[0] & 0x7FFFFFFF) > 0x7F800000      # the pattern sweep contains NaNs


def test_fp8_linear_path(nv):
    """quantize(model) with e4m3 act+weight (no qs): the QAT Linear runs activation pass (bf16 + FP8),
    FP8-only weight pass and an FP8 GEMM; result equals the oracle-quantized operands' product within
    fp32-accumulation + one bf16 output rounding."""
    import quantized_training as qt
    from quantized_training import fused
    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 384).cuda()
    model = torch.nn.Sequential(lin)
    args = qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16"])
    qt.quantize(model, args)
    x = (torch.randn(4, 96, 512, device="cuda") * 2).bfloat16()
    with torch.no_grad():
        y = model(x)
        assert model[0].activation_pre_process["0"]._emit_fp8 == "both"
        qmap = o.get_quantization_map("e4m3")
        fq = lambda t: torch.from_numpy(o.bf16_to_f32(o.vmap_bf16(host_u16(t.contiguous().view(torch.int16)), qmap))).cuda()  # noqa: E731
        ref = torch.nn.functional.linear(fq(x).double(), fq(model[0].weight).double(), model[0].bias.double())
    err = (y.double() - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 1e-2 * (512 ** 0.5) * 2.0 ** -8
    assert bool((err <= tol).all()), float((err - tol).max())
    os.environ["QT_FP8_GEMM"] = "0"
    try:
        with torch.no_grad():
            y2 = model(x)
    finally:
        del os.environ["QT_FP8_GEMM"]
    assert bool(((y2.double() - ref).abs() <= tol).all())


def _bf16_ulp_diff(a, b):
    """|a - b| in units of bf16 ULPs of the larger magnitude (a, b bf16 tensors, finite)."""
    ai = a.view(torch.int16).to(torch.int32) & 0xFFFF
    bi = b.view(torch.int16).to(torch.int32) & 0xFFFF
    key = lambda t: torch.where(t >= 0x8000, 0x8000 - t, t)  # noqa: E731  monotone integer key
    return (key(ai) - key(bi)).abs()


@pytest.mark.parametrize("shape,mask_kind", [((2, 4, 128, 128), "causal"), ((1, 32, 1024, 1024), "causal"),
                                              ((16, 12, 384, 384), "padding"), ((2, 3, 40, 72), None)])
def test_softmax_fq(nv, shape, mask_kind):
    """qt_softmax_fq_bf16 against the oracle's restatement of the chain with every bf16 rounding point explicit and exp / row
    sum in float64 (oracle.softmax_fq; modules/quantizable/modeling_bert.py:142-158).  The kernel's fp32 exp and sum can
    only differ where their ~1e-7 error straddles a bf16 rounding boundary: measured <= 7e-6 of the elements, one ULP
    (tools/exp_oracle_attention.py); a missing or extra rounding point would move percents of them.  The fake-quantized
    output is then EXACTLY the value map applied to the kernel's own probabilities, for every format, and the FP8-code
    variant holds exactly the codes of those values."""
    L = nv.lib()
    B, H, Q, C = shape
    torch.manual_seed(1)
    scores = (torch.randn(shape, device="cuda") * 3).bfloat16()
    scaling = 0.08838834764831845
    mask = None
    minv = torch.finfo(torch.bfloat16).min
    if mask_kind == "causal":
        mask = torch.full((Q, C), minv, device="cuda").triu(1).bfloat16()[None, None]
    elif mask_kind == "padding":
        mask = torch.zeros(B, 1, 1, C, device="cuda", dtype=torch.bfloat16)
        mask[:, :, :, C - 37:] = minv
    msb = msh = msq = 0
    if mask is not None:
        msb = mask.stride(0) if mask.shape[0] > 1 else 0
        msq = mask.stride(2) if mask.shape[2] > 1 else 0
    exp_p, _ = o.softmax_fq(host_u16(scores.view(torch.int16)), host_u16(mask.view(torch.int16)) if mask is not None else None, scaling, None)
    plain = None
    for dtype in (None, "e4m3", "e5m2", "posit8_1", "int8"):
        fmt = nv.format_for(dtype)
        lut = dev_u16(nv.build_map_u16(dtype))
        out = torch.empty_like(scores)
        amax = torch.zeros(1, dtype=torch.int32, device="cuda")
        nv.check(L.qt_softmax_fq_bf16(scores.data_ptr(), mask.data_ptr() if mask is not None else None, out.data_ptr(),
                                      B, H, Q, C, msb, msh, msq, scaling, ctypes.byref(fmt), lut.data_ptr(), None,
                                      amax.data_ptr(), stream()), "softmax")
        torch.cuda.synchronize()
        got = host_u16(out.view(torch.int16))
        if dtype is None:
            plain = got
            ulp = np.abs(got.astype(np.int32) - exp_p.astype(np.int32))            # probabilities: same sign, integer order
            assert int(ulp.max()) <= 1 and float((ulp > 0).mean()) <= 1e-4, (int(ulp.max()), float((ulp > 0).mean()))
            pmax = float(o.bf16_to_f32(got).max())
            assert amax.view(torch.float32).item() == pmax                          # the observer sees the kernel's own maximum
        else:
            assert np.array_equal(got, o.vmap_bf16(plain, o.get_quantization_map(dtype))), dtype
        if dtype in ("e4m3", "e5m2"):
            out8 = torch.empty(shape, dtype=torch.uint8, device="cuda")
            nv.check(L.qt_softmax_fq_bf16_fp8(scores.data_ptr(), mask.data_ptr() if mask is not None else None, None, out8.data_ptr(),
                                              B, H, Q, C, msb, msh, msq, scaling, ctypes.byref(fmt), stream()), "softmax fp8")
            dec = out8.view(torch.float8_e4m3fn if dtype == "e4m3" else torch.float8_e5m2).float().bfloat16()
            assert np.array_equal(host_u16(dec.view(torch.int16)), got), dtype


def test_llama_attention_fused_vs_unfused(nv):
    """Tiny LLaMA through quantize(): the fused score path (one HIP pass) and the unfused module chain
    give the same logits up to the 1-ULP softmax caveat; hooks, names and element counts are unchanged."""
    import quantized_training as qt
    from quantized_training import harness
    from quantized_training.fake_quantize import STATS
    model = harness.build_causal_lm("llama-tiny", device="cuda", seed=0)
    args = qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16", "--quantize_forward", "gemm"])
    qt.quantize(model, args)
    ids = torch.randint(0, 512, (1, 256), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    with torch.no_grad():
        model(ids)                                     # creates the per-tensor fake-quantizers
        STATS.reset(); a = model(ids).logits.float(); n_fused = STATS.elements
        os.environ["QT_FUSED_SOFTMAX"] = "0"
        try:
            STATS.reset(); b = model(ids).logits.float(); n_unfused = STATS.elements
        finally:
            del os.environ["QT_FUSED_SOFTMAX"]
    assert n_fused == n_unfused
    assert float((a - b).abs().max()) <= 0.05 * float(b.abs().max())
    names = [n for n, _ in model.named_modules() if n.endswith("av_matmul.activation_pre_process.0")]
    assert len(names) == 2


@pytest.mark.parametrize("spec", ["e4m3", "posit8_1", "int8,qs=per_tensor_symmetric"])
def test_permuted_views_through_module(nv, spec):
    """Attention-style permuted views ([B,S,H,D] storage seen as [B,H,S,D], and its K^T transpose): the
    module returns the same VALUES as on the contiguous copy, bit-exactly, observer state included."""
    from dataclasses import asdict
    import quantized_training as qt
    torch.manual_seed(3)
    base = (torch.randn(2, 48, 4, 64, device="cuda") * 2).bfloat16()        # [B, S, H, D]
    q_view = base.transpose(1, 2)                                            # [B, H, S, D], rows contiguous
    kt_view = q_view.transpose(2, 3)                                         # [B, H, D, S], transposed pair
    for view in (q_view, kt_view, base[:, :, 1:3, :].transpose(1, 2)):
        kw = asdict(qt.QuantizationSpec.from_str(spec))
        m1 = qt.FusedAmaxObsFakeQuantize(**kw, device="cuda")
        m2 = qt.FusedAmaxObsFakeQuantize(**kw, device="cuda")
        for rep in range(2):
            y1 = m1(view * (1 + rep))
            y2 = m2((view * (1 + rep)).contiguous())
            assert y1.shape == y2.shape
            assert torch.equal(y1.contiguous().view(torch.int16), y2.view(torch.int16))
            assert torch.equal(m1.scale, m2.scale) and torch.equal(m1.amax_history, m2.amax_history)
        xb = host_u16((view * 2).contiguous().view(torch.int16))
        st = o.FakeQuantState(kw["amax_history_len"], kw["quant_max"], observer=kw["qscheme"] is not None)
        qmap = o.get_quantization_map(spec.split(",")[0])
        o.fake_quant_forward(o.bf16_to_f32(host_u16(view.contiguous().view(torch.int16))).reshape(view.shape), True, qmap, st)
        exp = o.f32_to_bf16(o.fake_quant_forward(o.bf16_to_f32(xb).reshape(view.shape), True, qmap, st))
        assert np.array_equal(o.canon_nan16(host_u16(y1.contiguous().view(torch.int16))).reshape(-1), o.canon_nan16(exp).reshape(-1))


MX_META = json.load(open(os.path.join(G, "mx.json")))


@pytest.mark.parametrize("case", MX_META, ids=[c["name"] for c in MX_META])
def test_block_scaled_module_golden(nv, case):
    """Microscaling (fused HIP pass for last-axis blocks, composite otherwise) and group-wise affine on device
    tensors reproduce the reference's outputs and block scales bit-exactly."""
    from dataclasses import asdict
    import quantized_training as qt
    d = np.load(os.path.join(G, "mx.npz"))
    n = case["name"]
    kw = asdict(qt.QuantizationSpec.from_str(case["spec"]))
    m = qt.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=case["pow2"], device="cuda")
    bf16 = case["in"] == "bf16"
    if bf16:
        x = dev_u16(d[n + "__x"]).view(torch.bfloat16).reshape(case["shape"])
    else:
        x = torch.from_numpy(d[n + "__x"].view(np.float32)).cuda().reshape(case["shape"])
    y = m(x).contiguous()
    got = o.canon_nan16(host_u16(y.view(torch.int16))) if bf16 else o.canon_nan32(host_u32(y))
    assert np.array_equal(got.reshape(-1), d[n + "__y"].reshape(-1))
    assert list(m.scale.shape) == case["scale_shape"]
    assert np.array_equal(o.canon_nan32(host_u32(m.scale.float().reshape(-1))), d[n + "__scale"])


@pytest.mark.parametrize("dtype,bs,io", [("int8", 32, "bf16"), ("int6", 64, "bf16"), ("fp4_e2m1", 16, "bf16"),
                                          ("int8", 32, "f32"), ("fp8_e4m3", 128, "bf16"), ("int4", 8, "f32")])
def test_mx_fused_kernel_vs_oracle(nv, dtype, bs, io):
    """qt_fake_quant_mx_* on a 1 M-element tensor with per-row magnitudes spanning 1e-3..1e2, zero blocks and
    an Inf: outputs and block scales equal the oracle's."""
    L = nv.lib()
    rng = np.random.default_rng(bs)
    rows, cols = 2048, 512
    x = (rng.standard_normal((rows, cols)) * 10.0 ** rng.uniform(-3, 2, (rows, 1))).astype(np.float32)
    x[5, :bs] = 0.0
    x[7, 3] = np.inf
    qmax = {"int8": 127.0, "int6": 31.0, "int4": 7.0, "fp4_e2m1": 6.0, "fp8_e4m3": 448.0}[dtype]
    qmap = o.get_quantization_map(dtype)
    fmt = nv.format_for(dtype)
    lut = dev_u16(nv.build_map_u16(dtype))
    if io == "bf16":
        xb = o.f32_to_bf16(x)
        xd = dev_u16(xb.reshape(-1)); yd = torch.empty_like(xd)
        sf = torch.empty(rows * cols // bs, dtype=torch.int16, device="cuda")
        nv.check(L.qt_fake_quant_mx_bf16(xd.data_ptr(), yd.data_ptr(), sf.data_ptr(), rows, cols, bs, ctypes.byref(fmt),
                                         lut.data_ptr(), qmax, None, stream()), "mx")
        ey, es = o.mx_fake_quant(o.bf16_to_f32(xb), True, qmap, -1, bs, qmax)
        assert np.array_equal(o.canon_nan16(host_u16(yd)), o.canon_nan16(o.f32_to_bf16(ey)).reshape(-1))
        assert np.array_equal(o.canon_nan16(host_u16(sf)), o.canon_nan16(o.f32_to_bf16(es)).reshape(-1))
    else:
        xd = dev_f32(x.reshape(-1)); yd = torch.empty_like(xd)
        sf = torch.empty(rows * cols // bs, dtype=torch.float32, device="cuda")
        nv.check(L.qt_fake_quant_mx_f32(xd.data_ptr(), yd.data_ptr(), sf.data_ptr(), rows, cols, bs, ctypes.byref(fmt),
                                        lut.data_ptr(), qmax, None, stream()), "mx")
        ey, es = o.mx_fake_quant(x, False, qmap, -1, bs, qmax)
        assert np.array_equal(o.canon_nan32(host_u32(yd)), o.canon_nan32(ey.astype(np.float32).view(np.uint32)).reshape(-1))
        assert np.array_equal(o.canon_nan32(host_u32(sf)), o.canon_nan32(es.astype(np.float32).view(np.uint32)).reshape(-1))


@pytest.mark.parametrize("B,H,Sq,Sk,D,mask_kind", [(1, 4, 128, 128, 128, "causal"), (2, 3, 200, 200, 64, "padding"),
                                                    (1, 2, 64, 320, 128, None), (1, 32, 1024, 1024, 128, "causal")])
@pytest.mark.parametrize("pdtype", [None, "e4m3", "posit8_1"])
def test_fused_attention_kernel(nv, B, H, Sq, Sk, D, mask_kind, pdtype):
    """qt_attention_fq_bf16 against oracle.attention_fq: the module chain (bf16 QK^T -> x scaling -> + mask -> fp32 softmax ->
    bf16 -> fake-quant -> bf16 P.V; modeling_bert.py:118-158) with every rounding point explicit and all sums / exp in
    float64.  The kernel's fp32 accumulations differ only where they straddle a rounding boundary, so almost every output
    element is IDENTICAL: measured <= 8e-4 of the elements differ at all (tools/exp_oracle_attention.py); where a probability
    lands on the other side of a boundary of its 8-bit format one output row moves by at most that probability's step."""
    L = nv.lib()
    torch.manual_seed(B * 7 + H)
    qmap_in = torch.from_numpy(o.get_quantization_map("e4m3").view(np.int16)).cuda().view(torch.bfloat16)
    fqin = lambda t: qmap_in[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q = fqin((torch.randn(B, H, Sq, D, device="cuda")).bfloat16())
    k = fqin((torch.randn(B, H, Sk, D, device="cuda")).bfloat16())
    v = fqin((torch.randn(B, H, Sk, D, device="cuda")).bfloat16())
    scaling = D ** -0.5
    mask = None
    minv = torch.finfo(torch.bfloat16).min
    msb = msq = 0
    if mask_kind == "causal":
        mask = torch.full((Sq, Sk), minv, device="cuda").triu(1).bfloat16()[None, None]
        msq = mask.stride(2)
    elif mask_kind == "padding":
        mask = torch.zeros(B, 1, 1, Sk, device="cuda", dtype=torch.bfloat16)
        mask[:, :, :, Sk - 29:] = minv
        msb = mask.stride(0)
    fmt = nv.format_for(pdtype)
    lut = dev_u16(nv.build_map_u16(pdtype))
    out = torch.empty(B, Sq, H, D, dtype=torch.bfloat16, device="cuda")
    amax = torch.zeros(1, dtype=torch.int32, device="cuda")
    nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr() if mask is not None else None,
                                    out.data_ptr(), B, H, Sq, Sk, D, msb, 0, msq, scaling, ctypes.byref(fmt), lut.data_ptr(),
                                    None, amax.data_ptr(), stream()), "attention")
    torch.cuda.synchronize()
    u16 = lambda t: host_u16(t.contiguous().view(torch.int16))  # noqa: E731
    exp, pq = o.attention_fq(u16(q), u16(k), u16(v), u16(mask) if mask is not None else None, scaling,
                             o.get_quantization_map(pdtype) if pdtype else None)
    got = u16(out.permute(0, 2, 1, 3))
    assert float((got != exp).mean()) <= 2e-3, float((got != exp).mean())
    ev = o.bf16_to_f32(exp)
    err = np.abs(o.bf16_to_f32(got) - ev) / (np.abs(ev).max(axis=-1, keepdims=True) + 1e-30)
    assert float(err.max()) <= 0.08, float(err.max())
    # the observer saw the largest probability BEFORE fake-quantization; the oracle's quantized maximum brackets it
    pmax = float(o.bf16_to_f32(pq).max())
    assert abs(amax.view(torch.float32).item() - pmax) <= 0.07 * pmax
    if mask is not None:
        # the same launch with the mask's row extents (qt_mask_row_live_checked): a causal / right-padding mask is then not read at all --
        # values and dead key tiles come from the extents -- and the result is the same bit for bit; so is it for an irregular mask
        # (the device flag keeps the kernel on the mask)
        for irregular in (False, True):
            mk = mask.clone()
            if irregular:
                mk[..., 0, 3] = -1.0
            mrows = mk.shape[0] * mk.shape[1] * mk.shape[2]
            rl = torch.empty(mrows + 1, dtype=torch.int32, device="cuda")
            nv.check(L.qt_mask_row_live_checked(mk.data_ptr(), mrows, Sk, Sk, rl.data_ptr(), rl.data_ptr() + 4 * mrows, stream()), "row_live")
            assert int(rl[-1]) == int(irregular)
            lsb, lsq = (mk.shape[2] if mk.shape[0] > 1 else 0), (1 if mk.shape[2] > 1 else 0)
            ref_o, live_o = torch.empty_like(out), torch.empty_like(out)
            nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mk.data_ptr(), ref_o.data_ptr(), B, H, Sq, Sk, D, msb, 0, msq,
                                            scaling, ctypes.byref(fmt), lut.data_ptr(), None, None, stream()), "attention")
            nv.check(L.qt_attention_fq_live_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mk.data_ptr(), live_o.data_ptr(), B, H, Sq, Sk, D, msb, 0,
                                                 msq, scaling, ctypes.byref(fmt), lut.data_ptr(), None, None, 0, rl.data_ptr(), lsb, 0, lsq,
                                                 rl.data_ptr() + 4 * mrows, stream()), "attention live")
            assert torch.equal(ref_o.view(torch.int16), live_o.view(torch.int16)), (mask_kind, irregular)
    if pdtype == "posit8_1":
        # table formats: with the row form behind the map (qt_format.p1 bit 0) the kernel evaluates the probabilities' fake-quantizer
        # from a row table in LDS instead of gathering from the map -- the same function, so the same output bit for bit
        import quantized_training as qt
        from quantized_training.fake_quantize import _launch_format
        for dt in ("posit8_1", "posit8_2", "fp4_e2m1"):
            m = qt.get_quantization_map(dt, torch.device("cuda"))
            f_rows = _launch_format(nv.format_for(dt), m)
            assert f_rows.p1 & 1
            outs = []
            for f in (nv.format_for(dt), f_rows):
                o2 = torch.empty_like(out)
                nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr() if mask is not None else None,
                                                o2.data_ptr(), B, H, Sq, Sk, D, msb, 0, msq, scaling, ctypes.byref(f), m.data_ptr(),
                                                None, None, stream()), "attention")
                outs.append(o2)
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), dt


@pytest.mark.parametrize("B,H,S,D", [(1, 40, 1024, 128), (2, 5, 256, 128)])
def test_fused_attention_kernel_on_posit8_2_inputs_in_row_form(nv, B, H, S, D):
    """BASELINE configs[3]'s own attention core (parity gap 1): posit(8,2)-VALUED q / k / v (the rotary producer's
    output format there), the probabilities' fake-quantizer posit(8,2) in its ROW FORM (what fake_quantize._launch_format hands the
    kernel in the window), 40 heads x 1024 x 1024, causal -- against oracle.attention_fq, not against another path of the same
    kernel.  Both launch forms: with the mask read, and with the mask's row extents (qt_attention_fq_live_bf16, the window's)."""
    import quantized_training as qt
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    torch.manual_seed(3 + H)
    qmap = o.get_quantization_map("posit8_2")
    qmap_dev = torch.from_numpy(qmap.view(np.int16)).cuda().view(torch.bfloat16)
    fqin = lambda t: qmap_dev[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q = fqin((torch.randn(B, H, S, D, device="cuda") * 1.5).bfloat16())
    k = fqin((torch.randn(B, H, S, D, device="cuda") * 1.5).bfloat16())
    v = fqin((torch.randn(B, H, S, D, device="cuda")).bfloat16())
    scaling = D ** -0.5
    minv = torch.finfo(torch.bfloat16).min
    mask = torch.full((S, S), minv, device="cuda").triu(1).bfloat16()[None, None]
    m = qt.get_quantization_map("posit8_2", torch.device("cuda"))
    f_rows = _launch_format(nv.format_for("posit8_2"), m)
    assert f_rows.p1 & 1                                                        # the row form is what runs
    out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr(), out.data_ptr(), B, H, S, S, D, 0, 0, mask.stride(2),
                                    scaling, ctypes.byref(f_rows), m.data_ptr(), None, None, stream()), "attention")
    rl = torch.empty(S + 1, dtype=torch.int32, device="cuda")
    nv.check(L.qt_mask_row_live_checked(mask.data_ptr(), S, S, S, rl.data_ptr(), rl.data_ptr() + 4 * S, stream()), "row_live")
    live = torch.empty_like(out)
    nv.check(L.qt_attention_fq_live_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr(), live.data_ptr(), B, H, S, S, D, 0, 0, mask.stride(2),
                                         scaling, ctypes.byref(f_rows), m.data_ptr(), None, None, 0, rl.data_ptr(), 0, 0, 1,
                                         rl.data_ptr() + 4 * S, stream()), "attention live")
    torch.cuda.synchronize()
    assert int(rl[-1]) == 0 and torch.equal(out.view(torch.int16), live.view(torch.int16))
    u16 = lambda t: host_u16(t.contiguous().view(torch.int16))  # noqa: E731
    exp, pq = o.attention_fq(u16(q), u16(k), u16(v), u16(mask), scaling, qmap)
    got = u16(out.permute(0, 2, 1, 3))
    differ = float((got != exp).mean())
    assert differ <= 2e-3, differ
    ev = o.bf16_to_f32(exp)
    err = np.abs(o.bf16_to_f32(got) - ev) / (np.abs(ev).max(axis=-1, keepdims=True) + 1e-30)
    assert float(err.max()) <= 0.08, float(err.max())
    # every quantized probability the oracle produced is a posit(8,2) value (the map is idempotent on its image)
    assert np.array_equal(o.vmap_bf16(pq.reshape(-1)[:: 97], qmap), pq.reshape(-1)[:: 97])


@pytest.mark.parametrize("B,S,H", [(1, 1024, 40), (2, 128, 3), (3, 256, 5)])
@pytest.mark.parametrize("dtype", ["posit8_2", "fp6_e3m2"])
def test_rope_map_value_is_the_two_launches(nv, B, S, H, dtype):
    """qt_rope_map_value (the table-format rotary kernel and the attention core's value pass in ONE launch, 64-key value blocks) writes
    exactly what qt_rope_map_bf16 and qt_value_t_rows write as two launches, with q / k / v the column slices of one [B * S, 3 H D]
    projection product (LLaMA-2-13B's shape first); and refuses what the value pass does not take."""
    import quantized_training as qt
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    D = 128
    g = torch.Generator(device="cuda").manual_seed(3 + S)
    m = qt.get_quantization_map(dtype, torch.device("cuda"))
    f = _launch_format(nv.format_for(dtype), m)
    qkv = (torch.randn(B * S, 3 * H * D, device="cuda", generator=g) * 2).bfloat16()
    qkv.view(-1)[::1013] = 0.0                                                        # exact zeros: the flagged row 0 of posit maps
    q = qkv[:, :H * D].view(B, S, H, D).transpose(1, 2)
    k = qkv[:, H * D:2 * H * D].view(B, S, H, D).transpose(1, 2)
    v = qkv[:, 2 * H * D:].view(B, S, H, D).transpose(1, 2)
    ang = torch.rand(B, S, D, device="cuda", generator=g) * 6.28
    cos, sin = ang.cos().bfloat16(), ang.sin().bfloat16()
    rs = 3 * H * D
    outs = []
    for merged in (False, True):
        qo, ko = torch.empty(B, H, S, D, dtype=torch.bfloat16, device="cuda"), torch.empty(B, H, S, D, dtype=torch.bfloat16, device="cuda")
        vt = torch.empty(B, H, D, S, dtype=torch.bfloat16, device="cuda")
        if merged:
            nv.check(L.qt_rope_map_value(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), qo.data_ptr(), ko.data_ptr(), B, S, H, H, D, rs, rs,
                                         ctypes.byref(f), m.data_ptr(), 0, 0, v.data_ptr(), vt.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
                                         stream()), "qt_rope_map_value")
        else:
            nv.check(L.qt_rope_map_bf16(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), qo.data_ptr(), ko.data_ptr(), B, S, H, H, D, rs, rs,
                                        ctypes.byref(f), m.data_ptr(), 0, 0, stream()), "qt_rope_map_bf16")
            nv.check(L.qt_value_t_rows(v.data_ptr(), vt.data_ptr(), B, H, S, D, v.stride(0), v.stride(1), v.stride(2), ctypes.byref(f), m.data_ptr(),
                                       stream()), "qt_value_t_rows")
        outs.append((qo, ko, vt))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    # ... and with a weight's fake-quant pass as the launch's third job (qt_rope_map_value_weight): the same three tensors, and fq(W) bit for
    # bit the oracle's value map (all 65 536 patterns in front, a ragged count of vectors behind them)
    W = (torch.randn(H * D + 3, 5 * 8 + H * D, device="cuda", generator=g) * 0.1).bfloat16().contiguous()
    W.view(torch.int16).view(-1)[:65536] = torch.arange(65536, device="cuda", dtype=torch.int32).to(torch.int16)
    wq = torch.empty_like(W)
    qo3, ko3 = torch.empty_like(outs[0][0]), torch.empty_like(outs[0][1])
    vt3 = torch.empty_like(outs[0][2])
    nv.check(L.qt_rope_map_value_weight(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), qo3.data_ptr(), ko3.data_ptr(), B, S, H, H, D, rs,
                                        rs, ctypes.byref(f), m.data_ptr(), 0, 0, v.data_ptr(), vt3.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
                                        W.data_ptr(), wq.data_ptr(), W.numel(), stream()), "qt_rope_map_value_weight")
    for a, b in zip((qo3, ko3, vt3), outs[0]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    exp_w = o.canon_nan16(o.vmap_bf16(host_u16(W.view(torch.int16)), o.get_quantization_map(dtype)))
    assert np.array_equal(o.canon_nan16(host_u16(wq.view(torch.int16))), exp_w)
    qo, ko, vt = outs[0]
    args = (q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), qo.data_ptr(), ko.data_ptr())
    tail = (ctypes.byref(f), m.data_ptr(), 0, 0, v.data_ptr(), vt.data_ptr(), v.stride(0), v.stride(1), v.stride(2), stream())
    assert L.qt_rope_map_value(*args, B, S, H, H, 64, rs, rs, *tail) == nv.QT_ERR_BAD_ARG                 # head_dim 64: no value pass for it
    plain = nv.format_for(dtype)
    assert L.qt_rope_map_value(*args, B, S, H, H, D, rs, rs, ctypes.byref(plain), *tail[1:]) == nv.QT_ERR_BAD_ARG
    assert L.qt_rope_map_value(*args, B, S, H, H, D, rs, rs, *tail[:4], v.data_ptr() + 2, *tail[5:]) == nv.QT_ERR_UNALIGNED


@pytest.mark.parametrize("B,H,S,D,mask_kind", [(1, 40, 1024, 128, "causal"), (2, 5, 256, 128, "causal"), (2, 3, 384, 128, "padding"), (1, 4, 128, 128, None),
                                               (1, 2, 640, 128, "full")])
@pytest.mark.parametrize("pdtype", ["posit8_2", "posit8_1", "fp4_e2m1"])
def test_attention_rows_split_kernel(nv, B, H, S, D, mask_kind, pdtype):
    """qt_attention_rows_bf16 (+ qt_value_t_rows): the one-launch attention core for stateless TABLE formats with the score strip in
    registers (round 4) against oracle.attention_fq -- table-format-valued q / k / v, the probabilities' fake-quantizer in its row form,
    causal / right-padding masks through their row extents (never read), an irregular mask read in full, no mask; same bounds as the
    other attention cores (<= 2e-3 of the outputs differ, <= 0.08 of the row maximum).  The value pass is checked on its own: fq(v)
    transposed with the keys of every 32-chunk in the k-slot order.  With out_fq the epilogue applies the same value map."""
    import quantized_training as qt
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    torch.manual_seed(11 + H + S)
    qmap = o.get_quantization_map(pdtype)
    qmap_dev = torch.from_numpy(qmap.view(np.int16)).cuda().view(torch.bfloat16)
    fqin = lambda t: qmap_dev[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q = fqin((torch.randn(B, H, S, D, device="cuda") * 1.5).bfloat16())
    k = fqin((torch.randn(B, H, S, D, device="cuda") * 1.5).bfloat16())
    v_raw = torch.randn(B, S, H, D, device="cuda").bfloat16().transpose(1, 2)         # [B, H, S, D] view of a [B, S, H, D] buffer, unquantized
    v = fqin(v_raw.contiguous())
    scaling = D ** -0.5
    minv = torch.finfo(torch.bfloat16).min
    mask, msb, msq = None, 0, 0
    if mask_kind == "causal":
        mask = torch.full((S, S), minv, device="cuda").triu(1).bfloat16()[None, None]
        msq = mask.stride(2)
    elif mask_kind == "padding":
        mask = torch.zeros(B, 1, 1, S, device="cuda", dtype=torch.bfloat16)
        mask[:, :, :, S - 29:] = minv
        msb = mask.stride(0)
    elif mask_kind == "full":
        mask = (torch.randn(1, 1, S, S, device="cuda") * 2).bfloat16()
        mask[..., S // 2:] = minv
        mask[..., 5, 3] = -1.0
        msq = mask.stride(2)
    m = qt.get_quantization_map(pdtype, torch.device("cuda"))
    f_rows = _launch_format(nv.format_for(pdtype), m)
    assert f_rows.p1 & 1
    # the value pass: fq_v + transpose + slot permutation
    vt = torch.empty(B, H, D, S, dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_value_t_rows(v_raw.data_ptr(), vt.data_ptr(), B, H, S, D, v_raw.stride(0), v_raw.stride(1), v_raw.stride(2), ctypes.byref(f_rows),
                               m.data_ptr(), stream()), "qt_value_t_rows")
    key = torch.arange(S, device="cuda")
    slot = (key & ~31) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3)
    want_vt = torch.empty_like(vt)
    want_vt[..., slot] = v.transpose(2, 3)
    assert torch.equal(vt.view(torch.int16), want_vt.view(torch.int16))
    rl = irr = None
    lsb = lsq = 0
    if mask is not None:
        mrows = mask.shape[0] * mask.shape[1] * mask.shape[2]
        rlbuf = torch.empty(mrows + 1, dtype=torch.int32, device="cuda")
        nv.check(L.qt_mask_row_live_checked(mask.data_ptr(), mrows, S, S, rlbuf.data_ptr(), rlbuf.data_ptr() + 4 * mrows, stream()), "row_live")
        assert int(rlbuf[-1]) == int(mask_kind == "full")
        rl, irr = rlbuf.data_ptr(), rlbuf.data_ptr() + 4 * mrows
        lsb, lsq = (mask.shape[2] if mask.shape[0] > 1 else 0), (1 if mask.shape[2] > 1 else 0)
    outs = []
    for out_fq in (0, 1):
        out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device="cuda")
        nv.check(L.qt_attention_rows_bf16(q.data_ptr(), k.data_ptr(), vt.data_ptr(), mask.data_ptr() if mask is not None else None, msb, 0, msq,
                                          rl, lsb, 0, lsq, irr, out.data_ptr(), out_fq, ctypes.byref(f_rows), m.data_ptr(), B, H, S, S, D, scaling,
                                          stream()), "qt_attention_rows_bf16")
        outs.append(out)
    torch.cuda.synchronize()
    u16 = lambda t: host_u16(t.contiguous().view(torch.int16))  # noqa: E731
    exp, pq = o.attention_fq(u16(q), u16(k), u16(v), u16(mask) if mask is not None else None, scaling, qmap)
    got = u16(outs[0].permute(0, 2, 1, 3))
    differ = float((got != exp).mean())
    assert differ <= 2e-3, differ
    ev = o.bf16_to_f32(exp)
    err = np.abs(o.bf16_to_f32(got) - ev) / (np.abs(ev).max(axis=-1, keepdims=True) + 1e-30)
    assert float(err.max()) <= 0.08, float(err.max())
    # out_fq: the value map applied to the unquantized result, bit for bit
    assert np.array_equal(o.canon_nan16(u16(outs[1])), o.canon_nan16(o.vmap_bf16(u16(outs[0]), qmap)))
    # and against round 3's two-pass kernel on the same inputs: the same rounding points, fp32 sums in another order
    if mask_kind != "full":
        old = torch.empty_like(outs[0])
        nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr() if mask is not None else None, old.data_ptr(), B, H, S, S, D,
                                        msb, 0, msq, scaling, ctypes.byref(f_rows), m.data_ptr(), None, None, stream()), "attention")
        assert float((old.view(torch.int16) != outs[0].view(torch.int16)).float().mean()) <= 4e-3


def test_llama_fused_attention_vs_module_chain(nv):
    """Tiny LLaMA (head_dim 64) through quantize(): fused attention core vs the unfused module chain."""
    import quantized_training as qt
    from quantized_training import harness
    from quantized_training.fake_quantize import STATS
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      intermediate_size=512, vocab_size=512, attn_implementation="eager")
    model = LlamaForCausalLM(cfg).cuda().bfloat16().eval()
    qt.quantize(model, qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16"]))
    ids = torch.randint(0, 512, (2, 192), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    with torch.no_grad():
        model(ids)
        STATS.reset(); a = model(ids).logits.float(); n_fused = STATS.elements
        os.environ["QT_FUSED_ATTENTION"] = "0"
        try:
            STATS.reset(); b = model(ids).logits.float(); n_unfused = STATS.elements
        finally:
            del os.environ["QT_FUSED_ATTENTION"]
    assert n_fused == n_unfused
    assert float((a - b).abs().max()) <= 0.05 * float(b.abs().max())


# ---- block-scaled GEMMs on the scaled matrix instruction (qt_mx_pack / qt_mx_gemm) ----------------------------
MX_FMT_ID = {"fp8_e4m3": 0, "fp8_e5m2": 1, "fp6_e2m3": 2, "fp6_e3m2": 3, "fp4_e2m1": 4}


def _mx_operand(rows, K, fmt, block, dtype, seed, sigma=1.0):
    """Random operand through the engine's own quantize_mx: (block scales, element values) as the converted graphs hold them."""
    import quantized_training as qt
    from quantized_training.fake_quantize import get_quantization_map
    from quantized_training.quantizer.quantizer import get_quant_min_max
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(rows, K, device="cuda", generator=g) * sigma).to(dtype)
    qmap = get_quantization_map(fmt, "cuda")
    qmax = get_quant_min_max(fmt)[1]
    scale, q = torch.ops.quantized_ops.quantize_mx(x, qmap, [-1], block, float(qmax), True, None, None)
    return scale, q


def _pack(nv, q, scale, fmt, block, transpose=False):
    L = nv.lib()
    if transpose:                                     # q is [K, N]: pack as [N, K]
        K, rows = q.shape
        xs = (0, 1, q.stride(0))
        ss = (0, 1, scale.stride(0))
    else:
        rows, K = q.shape
        xs = (0, q.stride(0), 1)
        ss = (0, scale.stride(0), 1)
    bits = {0: 8, 1: 8, 2: 6, 3: 6, 4: 4}[MX_FMT_ID[fmt]]
    codes = torch.empty(rows, K * bits // 8, dtype=torch.uint8, device="cuda")
    e8 = torch.empty(rows, K // 32, dtype=torch.uint8, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    nv.check(L.qt_mx_pack(q.data_ptr(), scale.data_ptr(), int(q.dtype == torch.float32), codes.data_ptr(), e8.data_ptr(),
                          1, rows, K, *xs, *ss, block, MX_FMT_ID[fmt], bad.data_ptr(), stream()), "qt_mx_pack")
    return codes, e8, int(bad.item())


@pytest.mark.parametrize("fmt", sorted(MX_FMT_ID))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_mx_pack_matches_host_restatement(nv, fmt, dtype):
    for block in (32, 64):
        scale, q = _mx_operand(24, 256, fmt, block, dtype, seed=block, sigma=3.0)
        codes, e8, bad = _pack(nv, q, scale, fmt, block)
        assert bad == 0
        want_codes, want_e8 = o.mx_pack(q.float().cpu().numpy(), scale.float().cpu().numpy(), fmt, block)
        assert np.array_equal(codes.cpu().numpy(), want_codes)
        assert np.array_equal(e8.cpu().numpy(), want_e8)
    # a value outside the format, or a scale that is not a power of two, raises the flag instead of packing garbage
    q2 = q.clone(); q2[3, 5] = 0.3
    assert _pack(nv, q2, scale, fmt, 64)[2] == 1
    s2 = scale.clone(); s2[1, 1] = 3.0
    assert _pack(nv, q, s2, fmt, 64)[2] == 1


@pytest.mark.parametrize("fa,fb", [("fp8_e4m3", "fp8_e4m3"), ("fp8_e4m3", "fp8_e5m2"), ("fp8_e5m2", "fp8_e4m3"),
                                   ("fp8_e5m2", "fp8_e5m2"), ("fp6_e2m3", "fp6_e2m3"), ("fp6_e3m2", "fp6_e3m2"),
                                   ("fp4_e2m1", "fp4_e2m1"), ("fp8_e4m3", "fp4_e2m1"), ("fp6_e2m3", "fp4_e2m1"),
                                   ("fp6_e3m2", "fp4_e2m1")])
@pytest.mark.parametrize("shape", [(128, 128, 128), (200, 328, 448), (1, 64, 64), (1024, 512, 1024), (3600, 3400, 256),
                                   (1000, 2768, 384), (520, 11008, 128)])
def test_mx_gemm_vs_dequantized_reference(nv, fa, fb, shape):
    """C = (a * expand(sa)) @ (b * expand(sb))^T exactly as linear_mx states it.  Products of MX elements and
    power-of-two scales are exact; the instruction adds the 128 products of a step in an aligned fixed-point tree that
    keeps fewer bits than an fp32 chain (measured: up to 2^-16 of sum|a||b|), so the bound is |err| <= 2^-14 * sum|a||b|
    for the fp32 result; the bf16 result adds one rounding (2^-9 relative, 2^-8 allowed).  The shapes reach every
    kernel variant: register-staged (ragged K), one-stage and two-stage LDS-DMA at 128 x 128, the 256 x 256 ring
    kernel with ragged M and N (3600 x 3400: 15 x 14 tiles), and for the 8-bit pairs the 256 x (16 nt) kernel with one
    column group per tile (1024 x 512), ragged M with tiles of 2 / 3 groups (1000 x 2768) and of 11 / 10 groups in a single
    k step (520 x 11008 x 128)."""
    L = nv.lib()
    M, N, K = shape
    sa, qa = _mx_operand(M, K, fa, 32, torch.float32, seed=1, sigma=2.0)
    sb, qb = _mx_operand(N, K, fb, 32, torch.float32, seed=2, sigma=0.5)
    ca, ea, bad_a = _pack(nv, qa, sa, fa, 32)
    cb, eb, bad_b = _pack(nv, qb, sb, fb, 32)
    assert bad_a == 0 and bad_b == 0
    da = (qa * sa.repeat_interleave(32, 1)).double()
    db = (qb * sb.repeat_interleave(32, 1)).double()
    ref = da @ db.t()
    mag = da.abs() @ db.abs().t()
    bias = torch.randn(N, device="cuda")
    for out_dtype in (torch.float32, torch.bfloat16):
        for use_bias in (False, True):
            c = torch.full((M, N), float("nan"), dtype=out_dtype, device="cuda")
            b = bias.to(out_dtype) if use_bias else None
            nv.check(L.qt_mx_gemm(ca.data_ptr(), ea.data_ptr(), MX_FMT_ID[fa], cb.data_ptr(), eb.data_ptr(), MX_FMT_ID[fb],
                                  c.data_ptr(), int(out_dtype == torch.float32), b.data_ptr() if use_bias else None,
                                  1, M, N, K, 0, 0, stream()), "qt_mx_gemm")
            want = ref + (b.double() if use_bias else 0.0)
            err = (c.double() - want).abs()
            tol = 2.0 ** -14 * mag + (2.0 ** -8 * want.abs() if out_dtype == torch.bfloat16 else 0.0) + 1e-30
            assert bool((err <= tol).all()), float((err / tol).max())


@pytest.mark.parametrize("tm", [256, 128])
@pytest.mark.parametrize("fa,fb", [("fp8_e4m3", "fp8_e4m3"), ("fp8_e5m2", "fp8_e4m3")])
@pytest.mark.parametrize("shape", [(1024, 512, 1024), (1000, 2768, 384), (520, 11008, 128), (300, 4096, 256), (256, 16, 128)])
def test_mx_gemm_wide_kernel_both_heights(nv, fa, fb, shape, tm, monkeypatch):
    """The TM x (16 nt) kernel forced for both row-tile heights (the host otherwise picks one per shape): ragged M, tiles
    of 1 ... 12 / 16 column groups, a single k step, batch of 2 with operand batch strides, bias, both output dtypes."""
    monkeypatch.setenv("QT_MX_WIDE", "1")
    monkeypatch.setenv("QT_MX_WIDE_TM", str(tm))
    L = nv.lib()
    M, N, K = shape
    B = 2
    sa, qa = _mx_operand(B * M, K, fa, 32, torch.float32, seed=3, sigma=2.0)
    sb, qb = _mx_operand(B * N, K, fb, 32, torch.float32, seed=4, sigma=0.5)
    ca, ea, bad_a = _pack(nv, qa, sa, fa, 32)
    cb, eb, bad_b = _pack(nv, qb, sb, fb, 32)
    assert bad_a == 0 and bad_b == 0
    da = (qa * sa.repeat_interleave(32, 1)).double().view(B, M, K)
    db = (qb * sb.repeat_interleave(32, 1)).double().view(B, N, K)
    ref = torch.matmul(da, db.transpose(1, 2))
    mag = torch.matmul(da.abs(), db.abs().transpose(1, 2))
    bias = torch.randn(N, device="cuda")
    for out_dtype in (torch.float32, torch.bfloat16):
        c = torch.full((B, M, N), float("nan"), dtype=out_dtype, device="cuda")
        b = bias.to(out_dtype)
        nv.check(L.qt_mx_gemm(ca.data_ptr(), ea.data_ptr(), MX_FMT_ID[fa], cb.data_ptr(), eb.data_ptr(), MX_FMT_ID[fb],
                              c.data_ptr(), int(out_dtype == torch.float32), b.data_ptr(), B, M, N, K, M, N, stream()), "qt_mx_gemm")
        want = ref + b.double()
        err = (c.double() - want).abs()
        tol = 2.0 ** -14 * mag + (2.0 ** -8 * want.abs() if out_dtype == torch.bfloat16 else 0.0) + 1e-30
        assert bool((err <= tol).all()), float((err / tol).max())


def test_mx_gemm_batched_and_transposed_operand(nv):
    """matmul_mx's second operand arrives as [K, N] with blocks along K (axis -2): packed through strides as [N, K]."""
    L = nv.lib()
    B, M, N, K = 3, 96, 80, 64
    import quantized_training as qt
    from quantized_training.fake_quantize import get_quantization_map
    qmap = get_quantization_map("fp8_e4m3", "cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(B, M, K, device="cuda", generator=g).bfloat16()
    b = torch.randn(B, K, N, device="cuda", generator=g).bfloat16()
    sa, qa = torch.ops.quantized_ops.quantize_mx(a, qmap, [-1], 32, 448.0, True, None, None)
    sb = torch.ops.quantized_ops.calculate_mx_qparam(b, [-2], 32, 448.0, True)
    qb = torch.ops.quantized_ops.quantize(b, sb, None, [-2], 32, qmap)
    ca = torch.empty(B, M, K, dtype=torch.uint8, device="cuda"); ea = torch.empty(B, M, K // 32, dtype=torch.uint8, device="cuda")
    cb = torch.empty(B, N, K, dtype=torch.uint8, device="cuda"); eb = torch.empty(B, N, K // 32, dtype=torch.uint8, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    nv.check(L.qt_mx_pack(qa.data_ptr(), sa.data_ptr(), 0, ca.data_ptr(), ea.data_ptr(), B, M, K, qa.stride(0), qa.stride(1), 1,
                          sa.stride(0), sa.stride(1), 1, 32, 0, bad.data_ptr(), stream()), "pack a")
    nv.check(L.qt_mx_pack(qb.data_ptr(), sb.data_ptr(), 0, cb.data_ptr(), eb.data_ptr(), B, N, K, qb.stride(0), 1, qb.stride(1),
                          sb.stride(0), 1, sb.stride(1), 32, 0, bad.data_ptr(), stream()), "pack b")
    assert int(bad.item()) == 0
    c = torch.empty(B, M, N, dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_mx_gemm(ca.data_ptr(), ea.data_ptr(), 0, cb.data_ptr(), eb.data_ptr(), 0, c.data_ptr(), 0, None, B, M, N, K,
                          M, N, stream()), "qt_mx_gemm")
    ref = torch.matmul((qa * sa.repeat_interleave(32, -1)).float(), (qb * sb.repeat_interleave(32, -2)).float())
    assert torch.allclose(c.float(), ref, rtol=2.0 ** -7, atol=1e-3)


# ---- fused quantize_mx (qt_quantize_mx_*) against the composite formulation on CPU tensors ------------------------
# The CPU ops are the reference's own sequence of torch calls (decomposed.py:365-448), pinned bit-exactly by the PT2E
# microscaling goldens (tests/test_pt2e_cpu.py).
def _quantize_mx_cpu(x, dtype_name, block, qmax, pow2, scale_dtype=None):
    from quantized_training.fake_quantize import get_quantization_map
    qmap = get_quantization_map(dtype_name, "cpu")
    smap = get_quantization_map(scale_dtype, "cpu") if scale_dtype else None
    xc = x.cpu()
    s = torch.ops.quantized_ops.calculate_mx_qparam(xc, [-1], block, qmax, pow2, smap)
    q = torch.ops.quantized_ops.quantize(xc, s, None, [-1], block, qmap)
    return s, q


def _same_bits(a, b):
    a, b = a.cpu().contiguous(), b.cpu().contiguous()
    if a.dtype == torch.bfloat16:
        ua, ub = a.view(torch.int16).numpy().view(np.uint16), b.view(torch.int16).numpy().view(np.uint16)
        na, nb = (ua & 0x7FFF) > 0x7F80, (ub & 0x7FFF) > 0x7F80
    else:
        ua, ub = a.view(torch.int32).numpy().view(np.uint32), b.view(torch.int32).numpy().view(np.uint32)
        na, nb = (ua & 0x7FFFFFFF) > 0x7F800000, (ub & 0x7FFFFFFF) > 0x7F800000
    return bool(np.array_equal(na, nb) and np.array_equal(ua[~na], ub[~nb]))


@pytest.mark.parametrize("dtype_name", ["fp8_e4m3", "fp6_e3m2", "fp4_e2m1", "int8", "fp8_e5m2"])
@pytest.mark.parametrize("tdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("pow2", [True, False])
def test_quantize_mx_fused_matches_composite(nv, dtype_name, tdtype, pow2):
    from quantized_training.fake_quantize import get_quantization_map
    from quantized_training.quantizer.quantizer import get_quant_min_max
    qmax = float(get_quant_min_max(dtype_name)[1])
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 256, generator=g) * torch.exp2(torch.randint(-20, 20, (64, 1), generator=g).float())
    x[3, :64] = 0.0                                   # an all-zero block
    x[4, 5] = 255.0; x[5, 7] = 0.998; x[6, 0] = 2.0 ** -126; x[7, 1] = 3.0e38      # logarithm-rounding and range edges
    x[8, 2] = float("inf"); x[9, 3] = float("nan"); x[10, 4] = -float("inf")
    x = x.to(tdtype)
    for block in (32, 64, 16):
        s_ref, q_ref = _quantize_mx_cpu(x, dtype_name, block, qmax, pow2)
        qmap = get_quantization_map(dtype_name, "cuda")
        s, q = torch.ops.quantized_ops.quantize_mx(x.cuda(), qmap, [-1], block, qmax, pow2, None, None)
        assert _same_bits(s, s_ref), (dtype_name, block)
        assert _same_bits(q, q_ref), (dtype_name, block)
        if pow2 and block % 32 == 0 and dtype_name in MX_FMT_ID:
            fid, bs, codes, e8 = q._qt_mx_packed_act
            finite = torch.isfinite(q_ref).all(dim=-1) & torch.isfinite(s_ref).all(dim=-1) & (s_ref.float() >= 2.0 ** -127).all(dim=-1)
            rows = torch.nonzero(finite).flatten().numpy()
            want_codes, want_e8 = o.mx_pack(q_ref.float().numpy()[rows], s_ref.float().numpy()[rows], dtype_name, block)
            assert np.array_equal(codes.cpu().numpy()[rows], want_codes)
            assert np.array_equal(e8.cpu().numpy()[rows], want_e8)


def test_quantize_mx_shared_exponent_for_every_bf16_amax(nv):
    """floor(log2(amax)) is taken of the bf16-ROUNDED logarithm in the reference; every positive finite bf16 amax."""
    from quantized_training.fake_quantize import get_quantization_map
    pats = np.arange(1, 0x7F80, dtype=np.uint16)
    x = torch.zeros(len(pats), 32, dtype=torch.bfloat16)
    x[:, 0] = torch.from_numpy(pats.view(np.int16)).view(torch.bfloat16)
    s_ref, q_ref = _quantize_mx_cpu(x, "fp8_e4m3", 32, 448.0, True)
    s, q = torch.ops.quantized_ops.quantize_mx(x.cuda(), get_quantization_map("fp8_e4m3", "cuda"), [-1], 32, 448.0, True, None, None)
    assert _same_bits(s, s_ref)
    assert _same_bits(q, q_ref)


def test_quantize_mx_with_scale_map(nv):
    from quantized_training.fake_quantize import get_quantization_map
    x = (torch.randn(32, 128, generator=torch.Generator().manual_seed(2)) * 3).bfloat16()
    s_ref, q_ref = _quantize_mx_cpu(x, "fp4_e2m1", 16, 6.0, False, scale_dtype="fp8_e4m3")
    s, q = torch.ops.quantized_ops.quantize_mx(x.cuda(), get_quantization_map("fp4_e2m1", "cuda"), [-1], 16, 6.0, False,
                                               get_quantization_map("fp8_e4m3", "cuda"), None)
    assert _same_bits(s, s_ref) and _same_bits(q, q_ref)


# ---- one-launch model ops (csrc/qt_model_ops.hip) against the torch chains of HF's modeling_llama ------------------
def test_silu_mul_and_rotary_are_bit_identical_to_the_torch_chains(nv):
    from quantized_training import model_fusions as mf
    from transformers.models.llama import modeling_llama as ml
    g = torch.Generator(device="cuda").manual_seed(0)
    gate = (torch.randn(1024, 11008, device="cuda", generator=g) * 3).bfloat16()
    up = torch.randn(1024, 11008, device="cuda", generator=g).bfloat16()
    with torch.no_grad():
        want = torch.nn.functional.silu(gate) * up
        got = mf.silu_mul(gate, up)
    assert torch.equal(want.view(torch.int16), got.view(torch.int16))
    for (B, H, Hk, S, D) in ((1, 32, 32, 1024, 128), (3, 8, 2, 77, 64)):
        q = torch.randn(B, S, H, D, device="cuda", generator=g).bfloat16().transpose(1, 2)
        k = torch.randn(B, S, Hk, D, device="cuda", generator=g).bfloat16().transpose(1, 2)
        ang = torch.rand(B, S, D, device="cuda", generator=g) * 6.28
        cos, sin = ang.cos().bfloat16(), ang.sin().bfloat16()
        orig = getattr(ml.apply_rotary_pos_emb, "_qt_original", ml.apply_rotary_pos_emb)
        with torch.no_grad():
            wq, wk = orig(q, k, cos, sin)
            gq, gk = mf.rope(q, k, cos, sin)
        assert gq.shape == wq.shape and gq.stride() == wq.stride()
        assert torch.equal(wq.contiguous().view(torch.int16), gq.contiguous().view(torch.int16))
        assert torch.equal(wk.contiguous().view(torch.int16), gk.contiguous().view(torch.int16))


def test_rmsnorm_within_bf16_rounding_of_the_torch_chain(nv):
    from quantized_training import model_fusions as mf
    from transformers.models.llama import modeling_llama as ml
    g = torch.Generator(device="cuda").manual_seed(1)
    for cols in (4096, 5120, 256, 11008):
        x = (torch.randn(512, cols, device="cuda", generator=g) * 2).bfloat16()
        norm = ml.LlamaRMSNorm(cols, eps=1e-5).cuda().bfloat16()
        with torch.no_grad():
            norm.weight.copy_(1 + 0.1 * torch.randn(cols, device="cuda", generator=g))
            want = ml.LlamaRMSNorm.forward(norm, x)
            got = mf.rmsnorm(x, norm.weight, norm.variance_epsilon)
        a, b = want.view(torch.int16).int(), got.view(torch.int16).int()
        # a last-bit difference of the mean can move h = bf16(x * r) to the neighbouring value on isolated elements;
        # the weight multiply then rounds once more, so y is at most two bf16 codes away, on isolated elements
        assert int((a - b).abs().max()) <= 2
        assert float((a != b).float().mean()) <= 2e-3


@pytest.mark.parametrize("dtype_name", ["int8", "e4m3", "posit8_1"])
@pytest.mark.parametrize("tdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape,axis", [((96, 1024), 0), ((5, 24, 640), 1), ((64, 100), 0)])
def test_per_channel_module_device_vs_cpu(nv, dtype_name, tdtype, shape, axis):
    """per_channel_symmetric fake-quant (fake_quantize.py:218-221): rows of whole 16-byte vectors take the vectorised
    kernel, the last shape the element-wise one; outputs, scales and amax history identical to the CPU formulas."""
    import quantized_training as qt
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    from quantized_training.quantizer.quantizer import QuantizationSpec
    from quantized_training.quantizer.quantizer import QScheme, get_quant_min_max
    if dtype_name == "e4m3":                      # the reference has no default range for the bare name
        qmin, qmax = -448.0, 448.0
    else:
        qmin, qmax = get_quant_min_max(dtype_name)
    g = torch.Generator().manual_seed(4)
    xs = [(torch.randn(*shape, generator=g) * (3.0 ** i)).to(tdtype) for i in range(3)]
    outs = {}
    for dev in ("cpu", "cuda"):
        fq = FusedAmaxObsFakeQuantize(dtype=dtype_name, qscheme=QScheme.PER_CHANNEL_SYMMETRIC, quant_min=qmin, quant_max=qmax,
                                      amax_history_len=3, ch_axis=axis).to(dev)
        ys = [fq(x.to(dev)) for x in xs]
        outs[dev] = (ys, fq.scale.clone(), fq.amax_history.clone())
    (y0, s0, a0), (y1, s1, a1) = outs["cpu"], outs["cuda"]
    for a, b in zip(y0, y1):
        assert _same_bits(a, b)
    assert torch.equal(s0, s1.cpu()) and torch.equal(a0, a1.cpu())


def test_producer_fused_fake_quant_matches_the_hook(nv):
    """SiLU * up with the down-projection's input fake-quantizer applied by the producing kernel: same bf16 values and
    the same FP8 bytes as the unfused kernel followed by the hook's own pass; the hook then returns the tensor as is
    and still counts the call."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize, STATS
    g = torch.Generator(device="cuda").manual_seed(3)
    gate = (torch.randn(512, 11008, device="cuda", generator=g) * 4).bfloat16()
    up = (torch.randn(512, 11008, device="cuda", generator=g) * 100).bfloat16()      # reaches the saturation range
    for dtype in ("e4m3", "e5m2"):
        fq = FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
        fq._emit_fp8 = "both"
        assert fq.producer_fusable()
        with torch.no_grad():
            want = fq(mf.silu_mul(gate, up))
            got = mf.silu_mul_fq(gate, up, fq)
            assert torch.equal(want.view(torch.int16), got.view(torch.int16))
            assert torch.equal(want._qt_fp8.view(torch.uint8), got._qt_fp8.view(torch.uint8))
            STATS.reset()
            again = fq(got)
            assert again is got and STATS.elements == got.numel()
            other = FusedAmaxObsFakeQuantize(dtype=dtype).cuda()               # a different fake-quantizer does its own pass
            assert other(got) is not got


def test_attention_output_fused_with_projection_input_fake_quant(nv):
    """[B, H, S, D] -> [B, S, H, D] with the output projection's input fake-quantizer applied in the layout pass; the
    hook receives a reshaped VIEW of that tensor (as HF's attention block does) and must hand it through with its FP8
    code -- once; any other tensor, or a second call, takes the normal pass."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize, STATS
    g = torch.Generator(device="cuda").manual_seed(9)
    out = (torch.randn(2, 8, 96, 64, device="cuda", generator=g) * 3).bfloat16()
    fq = FusedAmaxObsFakeQuantize(dtype="e4m3").cuda()
    fq._emit_fp8 = "both"
    with torch.no_grad():
        ref_in = out.transpose(1, 2).contiguous().reshape(2, 96, -1)
        want = fq(ref_in)
        y = mf.transpose_fq(out, fq)
        view = y.reshape(2, 96, -1).contiguous()                # what LlamaAttention.forward does next
        STATS.reset()
        got = fq(view)
        assert got is view and STATS.elements == view.numel()
        assert torch.equal(want.view(torch.int16), got.view(torch.int16))
        assert torch.equal(want._qt_fp8.view(torch.uint8).reshape(-1), got._qt_fp8.view(torch.uint8).reshape(-1))
        again = fq(view)                                        # the expectation was for one call only
        assert again is not view and torch.equal(again.view(torch.int16), got.view(torch.int16))
        mf.transpose_fq(out, fq)
        other = fq(ref_in)                                      # a different tensor: normal pass, expectation dropped
        assert torch.equal(other.view(torch.int16), want.view(torch.int16))
        assert fq.__dict__.get("_qt_expected") is None


def test_rotary_fused_with_qk_fake_quant(nv):
    """rope_fq == rotary kernel followed by the two qk_matmul input hooks' permuted-view passes (values and layout)."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    g = torch.Generator(device="cuda").manual_seed(12)
    for (B, H, Hk, S, D) in ((1, 32, 32, 1024, 128), (2, 8, 8, 77, 64)):
        q = (torch.randn(B, S, H, D, device="cuda", generator=g) * 20).bfloat16().transpose(1, 2)
        k = (torch.randn(B, S, Hk, D, device="cuda", generator=g) * 300).bfloat16().transpose(1, 2)
        ang = torch.rand(B, S, D, device="cuda", generator=g) * 6.28
        cos, sin = ang.cos().bfloat16(), ang.sin().bfloat16()
        fq_q, fq_k = FusedAmaxObsFakeQuantize(dtype="e4m3").cuda(), FusedAmaxObsFakeQuantize(dtype="e5m2").cuda()
        with torch.no_grad():
            rq, rk = mf.rope(q, k, cos, sin)
            want_q, want_k = fq_q(rq), fq_k(rk)
            got_q, got_k = mf.rope_fq(q, k, cos, sin, fq_q, fq_k)
        assert got_q.is_contiguous() and got_k.is_contiguous() and got_q.shape == q.shape
        assert torch.equal(want_q.contiguous().view(torch.int16), got_q.view(torch.int16))
        assert torch.equal(want_k.contiguous().view(torch.int16), got_k.view(torch.int16))
        assert fq_q(got_q) is got_q and fq_k(got_k) is got_k
        assert torch.equal(got_q._qt_fp8.float(), got_q.float()) and torch.equal(got_k._qt_fp8.float(), got_k.float())


def test_rotary_and_value_codes_in_one_launch(nv):
    """qt_rope_fq_value (rotary + qk fake-quant, codes only, and the attention kernel's value-code pass in one launch) == qt_rope_fq_bf16's
    codes and qt_value_codes_t's output, bit for bit; the bf16 tensors it leaves unwritten decode from the codes exactly when a
    fake-quantizer's hand-over is asked for them."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    L = nv.lib()
    g = torch.Generator(device="cuda").manual_seed(13)
    for (B, H, S, D, dt) in ((1, 32, 1024, 128, "e4m3"), (2, 4, 256, 64, "e5m2")):
        qkv = (torch.randn(B * S, 3 * H * D, device="cuda", generator=g) * 3).bfloat16()
        qkv.view(torch.int16)[5, :7] = torch.tensor([0x7F80, -128, 0x7FC0, 0, -32768, 0x7F7F, 1], dtype=torch.int16, device="cuda")   # inf, -inf, nan, 0, -0, max, tiny
        qkv.view(torch.int16)[6, 2 * H * D:2 * H * D + 4] = torch.tensor([0x7F80, 0x7FC0, -32768, 0x7F7F], dtype=torch.int16, device="cuda")
        q = qkv[:, :H * D].view(B, S, H, D).transpose(1, 2)
        k = qkv[:, H * D:2 * H * D].view(B, S, H, D).transpose(1, 2)
        v = qkv[:, 2 * H * D:].view(B, S, H, D).transpose(1, 2)
        ang = torch.rand(B, S, D, device="cuda", generator=g) * 6.28
        cos, sin = ang.cos().bfloat16(), ang.sin().bfloat16()
        fq_q, fq_k, fq_v = (FusedAmaxObsFakeQuantize(dtype=dt).cuda() for _ in range(3))

        class Holder:                                            # stands in for the attention module the codes are left on
            pass
        attn = Holder()
        with torch.no_grad():
            want_q, want_k = mf.rope_fq(q, k, cos, sin, fq_q, fq_k)
            got_q, got_k = mf.rope_fq(q, k, cos, sin, fq_q, fq_k, value_job=(attn, v, fq_v))
        want_vt = torch.empty(B, H, D, S, dtype=torch.uint8, device="cuda")
        nv.check(L.qt_value_codes_t(v.data_ptr(), want_vt.data_ptr(), B, H, S, D, v.stride(0), v.stride(1), v.stride(2),
                                    ctypes.byref(fq_v._qt_format), stream()), "qt_value_codes_t")
        assert torch.equal(attn._qt_vt8[2], want_vt)
        assert torch.equal(got_q._qt_fp8.view(torch.uint8), want_q._qt_fp8.view(torch.uint8))
        assert torch.equal(got_k._qt_fp8.view(torch.uint8), want_k._qt_fp8.view(torch.uint8))
        assert got_q._qt_lazy and got_k._qt_lazy
        with torch.no_grad():
            assert fq_q(got_q) is got_q and fq_k(got_k) is got_k     # the hand-over decodes the codes
        assert not got_q._qt_lazy and fq_q(got_q) is got_q
        for got, want in ((got_q, want_q), (got_k, want_k)):
            a, b = got.view(torch.int16), want.view(torch.int16)
            nan = torch.isnan(want)
            assert torch.equal(a[~nan], b[~nan]) and bool(torch.isnan(got)[nan].all())
        # without a rotation (cos = sin = NULL): the permuted-view codes-only passes of q and k (+ the value codes) in one launch
        q8, k8 = (torch.empty(B, H, S, D, dtype=torch.uint8, device="cuda") for _ in range(2))
        vt8 = torch.empty(B, H, D, S, dtype=torch.uint8, device="cuda")
        fmt = fq_q._qt_format
        nv.check(L.qt_rope_fq_value(q.data_ptr(), k.data_ptr(), None, None, None, None, q8.data_ptr(), k8.data_ptr(), B, S, H, H, D,
                                    3 * H * D, 3 * H * D, ctypes.byref(fmt), ctypes.byref(fmt), v.data_ptr(), vt8.data_ptr(), v.stride(0),
                                    v.stride(1), v.stride(2), ctypes.byref(fmt), stream()), "qt_rope_fq_value")
        for t, t8 in ((q, q8), (k, k8)):
            want8 = torch.empty_like(t8)
            nv.check(L.qt_fake_quant_rows_bf16_fp8(t.data_ptr(), None, want8.data_ptr(), B, H, S, D, t.stride(0), t.stride(1), t.stride(2),
                                                   ctypes.byref(fmt), stream()), "qt_fake_quant_rows_bf16_fp8")
            assert torch.equal(t8, want8)
        assert torch.equal(vt8, want_vt)


def test_rmsnorm_fused_with_first_consumer_fake_quant(nv):
    """rmsnorm_fq == the norm kernel followed by the consumer's pass, and a sibling consumer's pass over the already
    quantized tensor reproduces it (the stateless formats are idempotent), FP8 bytes included."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    g = torch.Generator(device="cuda").manual_seed(21)
    x = (torch.randn(1024, 4096, device="cuda", generator=g) * 50).bfloat16()
    w = (1 + 0.2 * torch.randn(4096, device="cuda", generator=g)).bfloat16()
    for dtype in ("e4m3", "e5m2"):
        fq, sibling = FusedAmaxObsFakeQuantize(dtype=dtype).cuda(), FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
        fq._emit_fp8 = sibling._emit_fp8 = "both"
        with torch.no_grad():
            plain = mf.rmsnorm(x, w, 1e-5)
            want = fq(plain)
            got = mf.rmsnorm_fq(x, w, 1e-5, fq)
            assert torch.equal(want.view(torch.int16), got.view(torch.int16))
            assert torch.equal(want._qt_fp8.view(torch.uint8), got._qt_fp8.view(torch.uint8))
            assert fq(got) is got
            sib_from_plain, sib_from_q = sibling(plain), sibling(got)
            assert sib_from_q is not got
            assert torch.equal(sib_from_plain.view(torch.int16), sib_from_q.view(torch.int16))
            assert torch.equal(sib_from_plain._qt_fp8.view(torch.uint8), sib_from_q._qt_fp8.view(torch.uint8))


@pytest.mark.parametrize("dtype", ["posit8_2", "posit8_1", "fp6_e3m2", "fp4_e2m1"])
def test_table_format_producers_equal_kernel_then_pass(nv, dtype):
    """Stateless TABLE formats: the RMSNorm (with and without the residual add), SiLU * up and rotary kernels with the consumers'
    fake-quantizer applied in its row form (qt_*_map_bf16) write exactly what the plain kernel followed by the fake-quantizer's own
    pass writes -- same arithmetic, then the same function -- and mark the result so that every consumer's call hands it through
    (siblings included: one format, idempotent)."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import STATS, FusedAmaxObsFakeQuantize
    g = torch.Generator(device="cuda").manual_seed(5)
    fqs = [FusedAmaxObsFakeQuantize(dtype=dtype).cuda() for _ in range(3)]
    x = (torch.randn(1024, 5120, device="cuda", generator=g) * 20).bfloat16()
    x.view(torch.int16)[:12, :5120] = torch.arange(12 * 5120, device="cuda", dtype=torch.int32).to(torch.int16).view(12, 5120)   # every bf16 pattern
    r = torch.randn(1024, 5120, device="cuda", generator=g).bfloat16()
    w = (1 + 0.2 * torch.randn(5120, device="cuda", generator=g)).bfloat16()
    same = lambda a, b: torch.equal(a.contiguous().view(torch.int16), b.contiguous().view(torch.int16)) or bool(  # noqa: E731
        ((a.float() == b.float()) | (a.float().isnan() & b.float().isnan())).all())
    with torch.no_grad():
        want = fqs[0](mf.rmsnorm(x.nan_to_num(0.0, 1e4, -1e4), w, 1e-5))
        total, got = mf.rmsnorm_map(x.nan_to_num(0.0, 1e4, -1e4), None, w, 1e-5, fqs)
        assert total is None and same(want, got)
        STATS.reset()
        assert fqs[0](got) is got and same(fqs[1](got), got) and same(fqs[2](got), got)
        assert STATS.calls == 3 and STATS.elements == 3 * got.numel()       # handed through, each counted once, no launch of its own
        xs = x.nan_to_num(0.0, 1e4, -1e4)
        s_want = xs + r
        want = fqs[0](mf.rmsnorm(s_want, w, 1e-5))
        total, got = mf.rmsnorm_map(xs, r, w, 1e-5, fqs[:2])
        assert same(total, s_want) and same(want, got)
        # SiLU * up, operands as column slices of one wider product
        gu = (torch.randn(1024, 2 * 13824, device="cuda", generator=g) * 3).bfloat16()
        gate, up = gu[:, :13824], gu[:, 13824:]
        want = fqs[0](mf.silu_mul(gate, up))
        got = mf.silu_mul_map(gate, up, fqs[0])
        assert same(want, got) and fqs[0](got) is got
        # rotary on [B, S, H, D] buffers seen as [B, H, S, D], row stride of a q / k / v product
        B, S, H, D = 2, 256, 8, 128
        qkv = torch.randn(B, S, 3 * H * D, device="cuda", generator=g).bfloat16()
        q = qkv[..., :H * D].view(B, S, H, D).transpose(1, 2)
        k = qkv[..., H * D:2 * H * D].view(B, S, H, D).transpose(1, 2)
        ang = torch.rand(B, S, D, device="cuda", generator=g) * 6.28
        cos, sin = ang.cos().bfloat16(), ang.sin().bfloat16()
        qp, kp = mf.rope(q, k, cos, sin)
        qm, km = mf.rope_map(q, k, cos, sin, fqs[0], fqs[1])
        assert qm.is_contiguous() and km.is_contiguous() and qm.shape == (B, H, S, D)
        assert same(fqs[0](qp), qm) and same(fqs[1](kp), km) and fqs[0](qm) is qm and fqs[1](km) is km


def test_table_format_producers_edge_shapes(nv):
    """The row-form producer kernels at the edges of what they take: one row, the narrowest and the widest rows (8 and 16 384 columns),
    fewer key heads than query heads in the rotary kernel, batch-shared tables, a token count that does not fill a workgroup; and what
    they refuse (no row form behind the map, misaligned buffers) comes back as an error code, not as a launch."""
    import quantized_training as qt
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize, _launch_format
    g = torch.Generator(device="cuda").manual_seed(8)
    fq = FusedAmaxObsFakeQuantize(dtype="posit8_1").cuda()
    same = lambda a, b: torch.equal(a.contiguous().view(torch.int16), b.contiguous().view(torch.int16))  # noqa: E731
    with torch.no_grad():
        for rows, cols in ((1, 8), (3, 16384), (257, 24), (5, 4096)):
            x = (torch.randn(rows, cols, device="cuda", generator=g) * 4).bfloat16()
            r = torch.randn(rows, cols, device="cuda", generator=g).bfloat16()
            w = (1 + 0.1 * torch.randn(cols, device="cuda", generator=g)).bfloat16()
            total, y = mf.rmsnorm_map(x, r, w, 1e-6, [fq])
            assert same(total, x + r) and same(y, fq(mf.rmsnorm(x + r, w, 1e-6))), (rows, cols)
            assert same(mf.silu_mul_map(x, r, fq), fq(mf.silu_mul(x, r))), (rows, cols)
        B, S, Hq, Hk, D = 3, 5, 8, 2, 64
        q = torch.randn(B, S, Hq * D, device="cuda", generator=g).bfloat16().view(B, S, Hq, D).transpose(1, 2)
        k = torch.randn(B, S, Hk * D, device="cuda", generator=g).bfloat16().view(B, S, Hk, D).transpose(1, 2)
        ang = torch.rand(B, S, D, device="cuda", generator=g) * 6.28
        cos, sin = ang.cos().bfloat16(), ang.sin().bfloat16()
        qp, kp = mf.rope(q, k, cos, sin)
        qm, km = mf.rope_map(q, k, cos, sin, fq, fq)
        assert km.shape == (B, Hk, S, D) and same(fq(qp), qm) and same(fq(kp), km)
    # refusals
    L = nv.lib()
    plain = nv.format_for("posit8_1")                       # p1 == 0: no row words announced
    m = qt.get_quantization_map("posit8_1", torch.device("cuda"))
    x = torch.zeros(4, 64, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    w = torch.ones(64, dtype=torch.bfloat16, device="cuda")
    assert L.qt_rmsnorm_map_bf16(x.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), 4, 64, 1e-6, ctypes.byref(plain), m.data_ptr(), 0, stream()) == nv.QT_ERR_BAD_ARG
    rows = _launch_format(plain, m)
    assert L.qt_rmsnorm_map_bf16(x.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), 4, 60, 1e-6, ctypes.byref(rows), m.data_ptr(), 0, stream()) == nv.QT_ERR_UNALIGNED
    assert L.qt_silu_mul_map_bf16(x.data_ptr() + 2, x.data_ptr(), y.data_ptr(), 4, 56, 64, 64, ctypes.byref(rows), m.data_ptr(), stream()) == nv.QT_ERR_UNALIGNED
    assert L.qt_rmsnorm_map_bf16(x.data_ptr(), x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 4, 64, 1e-6, ctypes.byref(rows), m.data_ptr(), 0, stream()) == nv.QT_ERR_BAD_ARG


@pytest.mark.parametrize("B,S,V,stride", [(1, 1024, 32000, 32000), (3, 37, 1003, 1008), (2, 5, 8, 8)])
def test_causal_lm_loss_from_bf16_logits(nv, B, S, V, stride):
    """qt_causal_lm_loss_bf16 == cross_entropy(logits.float()[:, :-1], labels[:, 1:], ignore_index=-100) (transformers' ForCausalLMLoss):
    fp32 arithmetic in another summation order (1e-6 relative), deterministic; nothing scored -> NaN like torch."""
    L = nv.lib()
    g = torch.Generator(device="cuda").manual_seed(31)
    buf = (torch.randn(B, S, stride, device="cuda", generator=g) * 4).bfloat16()
    logits = buf[:, :, :V]
    labels = torch.randint(0, V, (B, S), device="cuda", generator=g)
    labels[:, : S // 2] = -100
    labels[0, -1] = -100
    scratch = torch.empty(B * S + 1, dtype=torch.float32, device="cuda")

    def run(lab):
        nv.check(L.qt_causal_lm_loss_bf16(logits.data_ptr(), lab.data_ptr(), B, S, V, stride, -100, scratch.data_ptr(),
                                          scratch.data_ptr() + 4 * B * S, stream()), "qt_causal_lm_loss_bf16")
        torch.cuda.synchronize()
        return float(scratch[B * S])
    got = run(labels)
    want = float(torch.nn.functional.cross_entropy(logits.float()[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100))
    assert abs(got - want) <= 2e-6 * abs(want) + 1e-6, (got, want)
    assert run(labels) == got
    assert np.isnan(run(torch.full_like(labels, -100)))


def test_rmsnorm_with_all_consumers_fake_quant(nv):
    """One launch for the norm and the input fake-quantizers of all its consuming Linears (qt_rmsnorm_consumers_bf16), plain and with
    the residual add: values and the first consumer's codes as rmsnorm_fq / add_rmsnorm with one fake-quantizer; every further
    consumer's hand-over returns the codes its own pass over the tensor computes, counted once per call."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize, STATS
    g = torch.Generator(device="cuda").manual_seed(22)
    x = (torch.randn(1024, 4096, device="cuda", generator=g) * 50).bfloat16()
    res = (torch.randn(1024, 4096, device="cuda", generator=g) * 50).bfloat16()
    x.view(torch.int16)[3, :4] = torch.tensor([0x7F80, 0x7FC0, -32768, 0x7F7F], dtype=torch.int16, device="cuda")
    w = (1 + 0.2 * torch.randn(4096, device="cuda", generator=g)).bfloat16()

    class Norm:
        weight, variance_epsilon = w, 1e-5
    for dtype in ("e4m3", "e5m2"):
        for n in (2, 3):
            fqs = [FusedAmaxObsFakeQuantize(dtype=dtype).cuda() for _ in range(n)]
            loners = [FusedAmaxObsFakeQuantize(dtype=dtype).cuda() for _ in range(n)]
            for f in fqs + loners:
                f._emit_fp8 = "both"
            with torch.no_grad():
                for with_res in (False, True):
                    if with_res:
                        want_sum, want = mf.add_rmsnorm(x, res, Norm, loners[0])
                        got_sum, got = mf.add_rmsnorm(x, res, Norm, fqs)
                        assert torch.equal(want_sum.view(torch.int16), got_sum.view(torch.int16))
                    else:
                        want, got = mf.rmsnorm_fq(x, w, 1e-5, loners[0]), mf.rmsnorm_fq(x, w, 1e-5, fqs)
                    ok = ~torch.isnan(want)
                    assert torch.equal(want.view(torch.int16)[ok], got.view(torch.int16)[ok]) and bool(torch.isnan(got)[~ok].all())
                    assert torch.equal(want._qt_fp8.view(torch.uint8), got._qt_fp8.view(torch.uint8))
                    STATS.reset()
                    assert fqs[0](got) is got
                    for f, lone in zip(fqs[1:], loners[1:]):
                        mine, theirs = f(got), lone(want)                     # theirs: the separate pass over the quantized tensor
                        assert mine is not got and mine.data_ptr() == got.data_ptr()
                        assert torch.equal(mine._qt_fp8.view(torch.uint8), theirs._qt_fp8.view(torch.uint8))
                    assert (STATS.calls, STATS.elements) == (2 * n - 1, (2 * n - 1) * x.numel())


def test_layernorm_with_all_consumers_fake_quant(nv):
    """qt_layernorm_consumers_bf16: y, fq(y) and the first consumer's codes as the one-consumer kernel writes them; every consumer's next
    call on y hands the shared fake-quantized values and ITS codes through (the codes a pass of its own over y computes)."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize, STATS
    g = torch.Generator(device="cuda").manual_seed(23)
    for cols in (768, 4096):
        x = (torch.randn(512, cols, device="cuda", generator=g) * 5).bfloat16()
        res = (torch.randn(512, cols, device="cuda", generator=g) * 5).bfloat16()
        norm = torch.nn.LayerNorm(cols).cuda().bfloat16()
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(cols, device="cuda", generator=g))
            norm.bias.copy_(0.1 * torch.randn(cols, device="cuda", generator=g))
        for dtype in ("e4m3", "e5m2"):
            fqs = [FusedAmaxObsFakeQuantize(dtype=dtype).cuda() for _ in range(3)]
            lone = FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
            for f in fqs + [lone]:
                f._emit_fp8 = "both"
            with torch.no_grad():
                want_y = mf.layernorm(x, norm, res, lone)
                want_q = lone(want_y)
                got_y = mf.layernorm(x, norm, res, fqs)
                assert torch.equal(want_y.view(torch.int16), got_y.view(torch.int16))
                STATS.reset()
                for f in fqs:
                    out = f(got_y)
                    assert torch.equal(out.view(torch.int16), want_q.view(torch.int16))
                    assert torch.equal(out._qt_fp8.view(torch.uint8), want_q._qt_fp8.view(torch.uint8))
                assert (STATS.calls, STATS.elements) == (3, 3 * x.numel())


@pytest.mark.parametrize("M,N,K,bias", [(1024, 4096, 4096, False), (256, 768, 3072, True), (1000, 1008, 512, True)])
def test_lt_fp8_gemm_matches_scaled_mm(nv, M, N, K, bias):
    """qt_fp8_gemm (hipBLASLt; the suggestion a committed table names) against the exact product of the FP8 operands; bf16 output.
    The library's bias epilogue adds the bias to the already rounded product, so with a bias there are two bf16
    roundings (of the product and of the sum); torch._scaled_mm drives the same epilogue."""
    from quantized_training import fused
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.float8_e4m3fn)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.float8_e4m3fn)
    bv = torch.randn(N, device="cuda", generator=g).bfloat16() if bias else None
    y = fused.lt_fp8_gemm(a, b, bv)
    assert y is not None, "hipBLASLt route unavailable"
    ref = a.float().double() @ b.float().double().t() + (bv.double() if bias else 0.0)
    mag = a.float().double().abs() @ b.float().double().abs().t()
    err = (y.double() - ref).abs()
    prod = (ref - bv.double()).abs() if bias else 0.0
    assert bool((err <= 2.0 ** -8 * (ref.abs() + prod) + 2.0 ** -20 * mag + 1e-6).all()), float(err.max())
    y2 = fused.lt_fp8_gemm(a, b, bv)                  # cached plan, same result
    assert torch.equal(y, y2)


def test_lt_fp8_gemm_batched_both_layouts(nv):
    from quantized_training import fused
    g = torch.Generator(device="cuda").manual_seed(8)
    q = torch.randn(6, 128, 64, device="cuda", generator=g).to(torch.float8_e4m3fn)
    k = torch.randn(6, 96, 64, device="cuda", generator=g).to(torch.float8_e4m3fn)
    s = fused.lt_fp8_gemm(q, k)                        # Q . K^T, B given as [N, K] per batch
    assert s is not None and s.shape == (6, 128, 96)
    assert torch.allclose(s.float(), torch.matmul(q.float(), k.float().transpose(1, 2)), rtol=2 ** -7, atol=1e-2)
    p = torch.rand(6, 128, 96, device="cuda", generator=g).to(torch.float8_e4m3fn)
    v = torch.randn(6, 96, 64, device="cuda", generator=g).to(torch.float8_e4m3fn)
    o = fused.lt_fp8_gemm(p, v, b_is_kn=True)          # P . V, B given as [K, N] per batch
    assert o is not None and o.shape == (6, 128, 64)
    assert torch.allclose(o.float(), torch.matmul(p.float(), v.float()), rtol=2 ** -7, atol=1e-2)


def test_score_pass_fp8_only_equals_closed_form_path(nv):
    """qt_softmax_fq_bf16_fp8 with and without the bf16 output: the FP8-only variant converts the bf16 probability with
    the hardware's round-to-nearest-even instead of the closed form; same bytes for probabilities of every magnitude
    (rows of logits at several scales, with and without a causal mask), and the bytes decode to the bf16 output."""
    L = nv.lib()
    B, H, S = 2, 4, 512
    fmt8 = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}
    g = torch.Generator(device="cuda").manual_seed(4)
    mask = torch.full((S, S), torch.finfo(torch.bfloat16).min, device="cuda").triu(1).bfloat16()[None, None].contiguous()
    for dtype in ("e4m3", "e5m2"):
        fmt = nv.format_for(dtype)
        for sigma in (0.3, 3.0, 12.0, 40.0):
            scores = (torch.randn(B, H, S, S, device="cuda", generator=g) * sigma).bfloat16()
            for m in (None, mask):
                mp = m.data_ptr() if m is not None else None
                out = torch.empty_like(scores)
                a8 = torch.empty(B, H, S, S, dtype=torch.uint8, device="cuda")
                b8 = torch.empty_like(a8)
                nv.check(L.qt_softmax_fq_bf16_fp8(scores.data_ptr(), mp, out.data_ptr(), a8.data_ptr(), B, H, S, S, 0, 0,
                                                  S if m is not None else 0, 0.125, ctypes.byref(fmt), stream()), "both")
                nv.check(L.qt_softmax_fq_bf16_fp8(scores.data_ptr(), mp, None, b8.data_ptr(), B, H, S, S, 0, 0,
                                                  S if m is not None else 0, 0.125, ctypes.byref(fmt), stream()), "fp8 only")
                assert torch.equal(a8, b8), (dtype, sigma, m is not None)
                assert torch.equal(a8.view(fmt8[dtype]).float(), out.float())


@pytest.mark.parametrize("cols", [768, 1024, 3072, 4096, 264])
@pytest.mark.parametrize("with_residual", [True, False])
def test_layernorm_kernel_vs_torch_chain(nv, cols, with_residual):
    """qt_layernorm_bf16 against torch's add + layer_norm on bf16: same rounding points (the sum, then the affine
    result); the row statistics are summed in a different order, which moves isolated results to the neighbouring
    bf16 value.  With a consumer fake-quantizer: yq / y8 are exactly fq(y) of the y the kernel wrote, and the hand-over
    (fq's next call on y returns yq) is one-shot."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    g = torch.Generator(device="cuda").manual_seed(cols)
    rows = 518                                   # not a multiple of the 4 rows a workgroup takes at cols <= 1024
    x = (torch.randn(rows, cols, device="cuda", generator=g) * 3).bfloat16()
    res = (torch.randn(rows, cols, device="cuda", generator=g) * 2 + 0.5).bfloat16() if with_residual else None
    norm = torch.nn.LayerNorm(cols, eps=1e-12).cuda().bfloat16()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(cols, device="cuda", generator=g))
        norm.bias.copy_(0.1 * torch.randn(cols, device="cuda", generator=g))
        want = norm(x + res if with_residual else x)
        got = mf.layernorm(x, norm, res)
        # one bf16 ulp where the result is O(1); where w * xhat + b cancels, the fp32 difference of the two summation
        # orders (~1e-6 of the operands) is what remains
        assert bool(((want.float() - got.float()).abs() <= 2.0 ** -7 * want.float().abs() + 1e-5).all())
        assert float((want.view(torch.int16) != got.view(torch.int16)).float().mean()) <= 2e-3
        for dtype in ("e4m3", "e5m2"):
            fq, ref_fq = FusedAmaxObsFakeQuantize(dtype=dtype).cuda(), FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
            ref_fq._emit_fp8 = "both"
            y = mf.layernorm(x, norm, res, fq)
            assert torch.equal(y.view(torch.int16), got.view(torch.int16))            # y itself stays unquantized
            yq = fq(y.view(rows, cols))                                               # reaches the hook as a view
            assert yq is not y and yq.data_ptr() != y.data_ptr()
            wq = ref_fq(got)
            assert torch.equal(yq.view(torch.int16), wq.view(torch.int16))
            assert torch.equal(yq._qt_fp8.view(torch.uint8), wq._qt_fp8.view(torch.uint8))
            assert yq._qt_origin == (y.data_ptr(), y._version, tuple(y.shape))
            again = fq(y)                                                             # one-shot: now an ordinary pass
            assert again.data_ptr() not in (y.data_ptr(), yq.data_ptr())
            assert torch.equal(again.view(torch.int16), wq.view(torch.int16))


def test_gelu_kernel_vs_torch(nv):
    """qt_gelu_bf16 against torch's erf GELU on bf16 (same fp32 expression, one rounding): every finite bf16 input and a
    BERT-sized tensor; fused with the consumer's fake-quantizer == the two-step sequence, bit for bit."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    allbits = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).cuda().view(torch.bfloat16)
    allbits = allbits[torch.isfinite(allbits)]
    allbits = allbits[: allbits.numel() // 8 * 8].contiguous()
    g = torch.Generator(device="cuda").manual_seed(3)
    big = (torch.randn(6144, 3072, device="cuda", generator=g) * 1.5).bfloat16()
    for x in (allbits, big):
        want = torch.nn.functional.gelu(x)
        got = mf.gelu(x)
        a, b = want.view(torch.int16).int(), got.view(torch.int16).int()
        assert int((a - b).abs().max()) <= 1 and float((a != b).float().mean()) <= 1e-4, float((a != b).float().mean())
        for dtype in ("e4m3", "e5m2"):
            fq, ref_fq = FusedAmaxObsFakeQuantize(dtype=dtype).cuda(), FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
            ref_fq._emit_fp8 = "both"
            with torch.no_grad():
                y = mf.gelu(x, fq)
                wq = ref_fq(got)
                assert fq(y) is y
            assert torch.equal(y.view(torch.int16), wq.view(torch.int16))
            assert torch.equal(y._qt_fp8.view(torch.uint8), wq._qt_fp8.view(torch.uint8))


def test_outlier_side_path_operators_on_device(nv):
    """quantized_ops::filter_outlier / spmm_csr on device tensors (index arithmetic, no host loops) against the same
    operators on CPU tensors, which are pinned to the reference (tests/test_pt2e_cpu.py): identical inliers and CSR
    arrays; the product up to the accumulation order of the atomics."""
    ops = torch.ops.quantized_ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(64, 96, generator=g) * 2
    w = torch.randn(40, 96, generator=g)
    ws = 2.0 ** torch.randint(-3, 2, (40, 3), generator=g).float()
    c = ops.filter_outlier(x, 4.0, 0.05)
    d = ops.filter_outlier(x.cuda(), 4.0, 0.05)
    for a, b in zip(c, d):
        assert b.device.type == "cuda" and torch.equal(a, b.cpu())
    yc = ops.spmm_csr(c[1], c[2], c[3], w, ws, None, 32)
    yd = ops.spmm_csr(d[1], d[2], d[3], w.cuda(), ws.cuda(), None, 32)
    assert yd.device.type == "cuda" and float((yc - yd.cpu()).abs().max()) <= 1e-5 * float(yc.abs().max())


def test_add_rmsnorm_equals_add_then_rmsnorm(nv):
    """qt_add_rmsnorm_bf16: the sum is torch's bf16 add bit for bit, the norm (and, with a consumer, its fake-quantized
    image + FP8 code) is exactly what the norm kernel gives on that sum -- the residual add costs no launch of its own."""
    from quantized_training import model_fusions as mf
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    from transformers.models.llama import modeling_llama as ml
    g = torch.Generator(device="cuda").manual_seed(9)
    for cols in (4096, 5120, 264):
        x = (torch.randn(517, cols, device="cuda", generator=g) * 3).bfloat16()
        r = (torch.randn(517, cols, device="cuda", generator=g) * 40).bfloat16()
        norm = ml.LlamaRMSNorm(cols, eps=1e-5).cuda().bfloat16()
        with torch.no_grad():
            norm.weight.copy_(1 + 0.1 * torch.randn(cols, device="cuda", generator=g))
            total, y = mf.add_rmsnorm(x, r, norm)
            want_sum = x + r
            assert torch.equal(total.view(torch.int16), want_sum.view(torch.int16))
            assert torch.equal(y.view(torch.int16), mf.rmsnorm(want_sum, norm.weight, norm.variance_epsilon).view(torch.int16))
            for dtype in ("e4m3", "e5m2"):
                fq = FusedAmaxObsFakeQuantize(dtype=dtype).cuda()
                total2, yq = mf.add_rmsnorm(x, r, norm, fq)
                ref = mf.rmsnorm_fq(want_sum, norm.weight, norm.variance_epsilon, fq)
                assert torch.equal(total2.view(torch.int16), want_sum.view(torch.int16))
                assert torch.equal(yq.view(torch.int16), ref.view(torch.int16))
                assert torch.equal(yq._qt_fp8.view(torch.uint8), ref._qt_fp8.view(torch.uint8)) and fq(yq) is yq


# ---- qt_linear_fq8_bf16: FP8 GEMM with the weight fake-quantizer fused into its operand path ------------------------
F8_CODE = {"e4m3": 0, "e5m2": 1}
F8_TORCH = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}


def _codes_of(nv, t, dtype):
    """FP8 codes of fq(t) from the (oracle-pinned) elementwise pass, as a uint8 device tensor."""
    fmt = nv.format_for(dtype)
    y8 = torch.empty(t.shape, dtype=torch.uint8, device="cuda")
    one = torch.ones((), dtype=torch.float32, device="cuda")
    nv.check(nv.lib().qt_fake_quant_bf16_fp8(t.data_ptr(), None, y8.data_ptr(), t.numel(), ctypes.byref(fmt), one.data_ptr(), None,
                                             stream()), "fq8")
    return y8


def _linear_fq8(nv, x8, xdtype, ws, wdtype, biases=None):
    M, K = x8.shape
    n = len(ws)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in (biases or [None] * n)])
    ns = (ctypes.c_int * n)(*[w.shape[0] for w in ws])
    y = torch.empty((M, sum(w.shape[0] for w in ws)), dtype=torch.bfloat16, device="cuda")
    nv.check(nv.lib().qt_linear_fq8_bf16(x8.data_ptr(), F8_CODE[xdtype], wp, bp, ns, n, F8_CODE[wdtype], y.data_ptr(), M, K,
                                         stream()), "qt_linear_fq8_bf16")
    return y


@pytest.mark.parametrize("xdtype,wdtype,big", [("e4m3", "e4m3", 448.0), ("e5m2", "e5m2", 57344.0), ("e4m3", "e5m2", 448.0)])
@pytest.mark.parametrize("K", [128, 1024, 4096])
def test_linear_fq8_worst_case_cancellation(nv, xdtype, wdtype, big, K, capsys):
    """The scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4) adds its 128 products in a fixed-point tree that keeps fewer
    bits than the reference's fp32 chain (deviation 5).  Worst case for such a tree: products of the largest magnitude
    the format has with ALTERNATING signs -- they cancel exactly -- next to products thousands of times smaller that carry the
    result.  Rows: (a) pure cancellation (exact result 0), (b) cancellation + a small tail, (c) cancellation + one product at the
    bottom of the format.  The error must stay inside the bound the GEMM tests use, 2^-14 sum |a||b|; the measured ratio is printed
    (pytest -s) and recorded in DESIGN.md."""
    torch.manual_seed(K)
    M, N = 64, 64
    sign = torch.ones(K, device="cuda")
    sign[1::2] = -1.0
    x = (sign * big).repeat(M, 1)
    w = torch.full((N, K), 1.0, device="cuda") * (448.0 if wdtype == "e4m3" else 57344.0)
    # rows 16.. of x: the last 32 products are small (x tiny, w of order one): they carry the result
    x[:16, K - 32:] = 0.0
    x[16:, K - 32:] = torch.randn(M - 16, 32, device="cuda") * 2.0 ** -4
    w[:, K - 32:] = torch.randn(N, 32, device="cuda")
    # rows 32..: one more product at the bottom of the activation format
    x[32:, K - 31] = 2.0 ** -9 if xdtype == "e4m3" else 2.0 ** -16
    w[:, K - 31] = 1.0
    x, w = x.bfloat16(), w.bfloat16()
    x8 = _codes_of(nv, x, xdtype)
    y = _linear_fq8(nv, x8, xdtype, [w], wdtype).double()
    qx, qw = o.get_quantization_map(xdtype), o.get_quantization_map(wdtype)
    xa = torch.from_numpy(o.bf16_to_f32(o.vmap_bf16(host_u16(x.view(torch.int16)), qx))).cuda().double()
    wa = torch.from_numpy(o.bf16_to_f32(o.vmap_bf16(host_u16(w.view(torch.int16)), qw))).cuda().double()
    ref = xa @ wa.t()
    bound = xa.abs() @ wa.abs().t()
    assert bool((ref[:16] == 0).all())                                          # (a): exact cancellation
    err = (y - ref).abs()
    ratio = float((err / bound).max())
    with capsys.disabled():
        print(f"\n[fq8 cancellation] {xdtype} x {wdtype} K={K}: max |err| / sum|a||b| = 2^{math.log2(ratio) if ratio > 0 else float('-inf'):.1f}, "
              f"max |err| {float(err.max()):.4g} at |products| {big * float(wa.abs().max()):.4g}")
    tol = ref.abs() * 2.0 ** -8 + bound * 2.0 ** -14
    assert bool((err <= tol).all()), ratio
    assert bool((y[:16] == 0).all())                                            # the tree cancels equal magnitudes exactly


@pytest.mark.parametrize("xdtype,wdtype", [("e4m3", "e4m3"), ("e5m2", "e5m2"), ("e4m3", "e5m2"), ("e5m2", "e4m3")])
def test_linear_fq8_weight_codes_are_the_value_map(nv, xdtype, wdtype):
    """Identity activation: y[m][n] = fq(W)[n][m] -- one product per output, so the kernel's in-flight conversion of W is
    compared bit for bit with the oracle's value map on ALL 65 536 bf16 patterns (rows 0..127 of W), overflowing values
    included (the reference saturates, fp8.py:32); rows holding +-Inf / NaN must come out all-NaN (fp8.py:36 and 0 * NaN)."""
    K = 512
    torch.manual_seed(1)
    W = (torch.randn(400, K, device="cuda") * 3).bfloat16()
    W.view(torch.int16)[:128] = torch.arange(65536, device="cuda", dtype=torch.int32).to(torch.int16).view(128, 512)
    x8 = _codes_of(nv, torch.eye(K, device="cuda").bfloat16(), xdtype)
    qmap = o.get_quantization_map(wdtype)
    bias = torch.randn(400, device="cuda").bfloat16()
    for sanitize in (True, False):
        Wt = W.clone()
        if sanitize:
            Wt[~torch.isfinite(Wt.float())] = 0
        exp = o.canon_nan16(o.vmap_bf16(host_u16(Wt.view(torch.int16)), qmap)).reshape(400, K)
        got = o.canon_nan16(host_u16(_linear_fq8(nv, x8, xdtype, [Wt], wdtype).t().contiguous().view(torch.int16)))
        nan_rows = (exp == 0x7FC0).any(axis=1)
        assert nan_rows.sum() == (0 if sanitize else 2)
        same = (got == exp) | (((got | exp) & 0x7FFF) == 0)                 # the sign of a zero is not part of a product
        assert same[~nan_rows].all()
        assert (got[nan_rows] == 0x7FC0).all()
        gotb = _linear_fq8(nv, x8, xdtype, [Wt], wdtype, [bias]).t().contiguous()
        expb = (torch.from_numpy(o.bf16_to_f32(exp)).cuda() + bias.float()[:, None]).bfloat16()
        assert torch.equal(gotb[~torch.from_numpy(nan_rows).cuda()].view(torch.int16), expb[~torch.from_numpy(nan_rows).cuda()].view(torch.int16))


@pytest.mark.parametrize("M,Ns,K", [(1024, [4096], 1024), (1024, [176], 256), (300, [48, 64, 16], 384), (1, [16], 128),
                                    (777, [2048, 512, 512], 512), (520, [11008], 256), (257, [208, 4096 - 208], 128),
                                    # BASELINE.json's full LLaMA-2-7B sizes: gate / up, down (two k tiles per step), q / k / v
                                    (1024, [11008], 4096), (1024, [4096], 11008), (1024, [4096, 4096, 4096], 4096),
                                    # BERT-base batch [16, 384] and the LLaMA-2-13B widths (shapes the route table sends to the pair by default:
                                    # the fused kernel must be right there too, it is what QT_FQ8_GEMM=1 and the tuner run)
                                    (6144, [768], 768), (6144, [3072], 768), (1024, [13824], 5120), (1024, [5120], 13824)])
@pytest.mark.parametrize("xdtype,wdtype", [("e4m3", "e4m3"), ("e5m2", "e4m3")])
def test_linear_fq8_vs_fp64_product_of_the_codes(nv, M, Ns, K, xdtype, wdtype):
    """Ragged M, column tiles spanning two weights, several weights per launch, bias: against the fp64 product of the
    oracle-quantized operands.  Tolerance: one bf16 rounding of the result (2^-8 relative) plus the scaled matrix
    instruction's accumulation error, measured <= 2^-16 sum |a||b| (bound used: 2^-14)."""
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda").bfloat16()
    ws = [(torch.randn(n, K, device="cuda") * 0.05).bfloat16() for n in Ns]
    bs = [torch.randn(n, device="cuda").bfloat16() if i % 2 == 0 else None for i, n in enumerate(Ns)]
    x8 = _codes_of(nv, x, xdtype)
    y = _linear_fq8(nv, x8, xdtype, ws, wdtype, bs).double()
    qx, qw = o.get_quantization_map(xdtype), o.get_quantization_map(wdtype)
    xa = torch.from_numpy(o.bf16_to_f32(o.vmap_bf16(host_u16(x.view(torch.int16)), qx))).cuda().double()
    wa = torch.cat([torch.from_numpy(o.bf16_to_f32(o.vmap_bf16(host_u16(w.view(torch.int16)), qw))).cuda().double() for w in ws])
    bias = torch.cat([b.double() if b is not None else torch.zeros(n, device="cuda", dtype=torch.float64) for b, n in zip(bs, Ns)])
    ref = xa @ wa.t() + bias
    tol = ref.abs() * 2.0 ** -8 + (xa.abs() @ wa.abs().t()) * 2.0 ** -14 + 1e-30
    assert bool(((y - ref).abs() <= tol).all()), float(((y - ref).abs() / tol).max())


@pytest.mark.parametrize("dtype,scales", [("int8", (0.0173, 0.031, 1.0, 0.25)), ("fp8_e5m2", (3.1e-7, 1.7e-6, 2.0 ** -20, 1.0)),
                                          ("e4m3", (1.0, 1.0, 0.37, 2.0))])
@pytest.mark.parametrize("rows,cols", [(2048, 768), (2048, 3072), (77, 64), (1, 8), (4097, 1000)])
def test_fake_quant_chain_equals_the_single_passes(nv, dtype, scales, rows, cols):
    """qt_fake_quant_chain_bf16: four fake-quantizer calls over one tensor in ONE launch -- stage 0 on x, stages 1 and 2 on stage 0's
    result, stage 3 on stage 1's (the gradient chain behind a LayerNorm, quantize.py:116-179) -- each bit for bit the result and the
    amax of its own qt_fake_quant_bf16 launch (oracle-pinned elsewhere), for int8 (closed form), E5M2 as a table format in its row
    form and E4M3 (closed form), scaled and unscaled; the column sums of stage 3's result equal its fp64 column sums rounded to bf16
    within one bf16 step, and are run-to-run bit-identical."""
    import quantized_training as qt_pkg
    L = nv.lib()
    torch.manual_seed(rows + cols)
    x = (torch.randn(rows, cols, device="cuda") * (scales[0] * 40)).bfloat16()
    x.view(-1)[:: max(1, x.numel() // 7)] = 0.0
    fmt = nv.format_for(dtype)
    lut = qt_pkg.get_quantization_map(dtype, torch.device("cuda"))
    from quantized_training.fake_quantize import _launch_format
    fmt = _launch_format(fmt, lut)
    src = (-1, 0, 0, 1)
    sc = [torch.tensor([s_], dtype=torch.float32, device="cuda") for s_ in scales]
    # reference: one launch per stage
    want, want_amax, outs = [], [], []
    for i in range(4):
        inp = x if src[i] < 0 else want[src[i]]
        y = torch.empty_like(inp)
        am = torch.zeros(1, dtype=torch.float32, device="cuda")
        nv.check(L.qt_fake_quant_bf16(inp.data_ptr(), y.data_ptr(), inp.numel(), ctypes.byref(fmt), lut.data_ptr(), sc[i].data_ptr(), am.data_ptr(),
                                      stream()), "qt_fake_quant_bf16")
        want.append(y)
        want_amax.append(am)
    for nstage in (4, 1, 3):
        outs = [torch.full_like(x, float("nan")) for _ in range(nstage)]
        amax = [torch.zeros(1, dtype=torch.float32, device="cuda") for _ in range(nstage)]
        stages = (nv.QtChainStage * nstage)()
        for i in range(nstage):
            stages[i].scale_f32_dev = sc[i].data_ptr()
            stages[i].amax_bits_dev = amax[i].data_ptr()
            stages[i].out_dev = outs[i].data_ptr()
            stages[i].src = src[i]
        cs = nstage - 1
        wb = L.qt_fake_quant_chain_ws_bytes(rows, cols)
        assert wb > 0
        ws = torch.zeros(wb, dtype=torch.uint8, device="cuda")
        gb = torch.empty(cols, dtype=torch.bfloat16, device="cuda")
        fmax = float(np.abs(o.bf16_to_f32(o.get_quantization_map(dtype))[np.isfinite(o.bf16_to_f32(o.get_quantization_map(dtype)))]).max())
        nv.check(L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, nstage, ctypes.byref(fmt), lut.data_ptr(), cs, fmax, gb.data_ptr(),
                                            ws.data_ptr(), wb, stream()), "qt_fake_quant_chain_bf16")
        torch.cuda.synchronize()
        assert not bool(ws.any())                                   # accumulators and tickets are left zero
        for i in range(nstage):
            assert torch.equal(outs[i].view(torch.int16), want[i].view(torch.int16)), (nstage, i)
            assert torch.equal(amax[i].view(torch.int32), want_amax[i].view(torch.int32)), (nstage, i)
        ref = want[cs].double().sum(0)
        got = gb.double()
        tol = ref.abs() * 2.0 ** -7 + want[cs].double().abs().sum(0) * 2.0 ** -20 + 1e-30
        assert bool(((got - ref).abs() <= tol).all()), (nstage, float(((got - ref).abs() / tol).max()))
        for _ in range(3):
            gb2 = torch.empty_like(gb)
            nv.check(L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, nstage, ctypes.byref(fmt), lut.data_ptr(), cs, fmax, gb2.data_ptr(),
                                                ws.data_ptr(), wb, stream()), "qt_fake_quant_chain_bf16")
            assert torch.equal(gb.view(torch.int16), gb2.view(torch.int16))
    # what it refuses
    stages[0].src = 0
    assert L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, 1, ctypes.byref(fmt), lut.data_ptr(), -1, 0.0, None, None, 0,
                                      stream()) == nv.QT_ERR_BAD_ARG
    stages[0].src = -1
    if rows > 64:
        assert L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, 1, ctypes.byref(fmt), lut.data_ptr(), 0, fmax, gb.data_ptr(), ws.data_ptr(), 4,
                                          stream()) == nv.QT_ERR_BAD_ARG


@pytest.mark.parametrize("rows", [256, 1024, 2048, 4096])           # one 64-column strip: 4, 16, 32 and 32 bands of 64-row groups
def test_chain_column_sums_keep_non_finite_gradients(nv, rows):
    """The bias gradient the chain launch carries (grad_output.sum(0), run_glue_no_trainer.py:660-667) over a diverged step: a column whose
    every band holds a NaN must come out NaN, as torch's sum does -- round 5 recorded a non-finite band partial as 2^62 in the fixed-point
    accumulator, and 4, 8, 16 or 32 of those wrapped to exactly zero.  Also: one poisoned band among finite ones, a column
    whose finite partial sums are as large as the format allows (every element at the maximum: the finite total must stay clear of the
    poison threshold), and clean columns beside them unchanged."""
    L = nv.lib()
    cols = 64
    dtype = "fp8_e5m2"
    fmt = nv.format_for(dtype)
    import quantized_training as qt_pkg
    from quantized_training.fake_quantize import _launch_format
    lut = qt_pkg.get_quantization_map(dtype, torch.device("cuda"))
    fmt = _launch_format(fmt, lut)
    fmax = 57344.0
    torch.manual_seed(rows)
    x = torch.randn(rows, cols, device="cuda").bfloat16()
    x[:, 0] = float("nan")                      # every band poisoned
    x[:, 1] = float("inf")                      # (fp8_e5m2 keeps Inf: fake_quantize.py:63-80 -> the sum is Inf or NaN, never finite)
    x[rows // 2, 2] = float("nan")              # one band poisoned
    x[:, 3] = fmax                              # largest finite total: rows x max
    x[:, 4] = -fmax
    x[::2, 5] = float("nan")                    # half the rows
    sc = torch.ones(1, dtype=torch.float32, device="cuda")
    out = torch.empty_like(x)
    am = torch.zeros(1, dtype=torch.float32, device="cuda")
    stages = (nv.QtChainStage * 1)()
    stages[0].scale_f32_dev, stages[0].amax_bits_dev, stages[0].out_dev, stages[0].src = sc.data_ptr(), am.data_ptr(), out.data_ptr(), -1
    wb = L.qt_fake_quant_chain_ws_bytes(rows, cols)
    ws = torch.zeros(max(wb, 16), dtype=torch.uint8, device="cuda")
    gb = torch.empty(cols, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):                          # twice: the accumulators must be left zero by the poisoned launch too
        nv.check(L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, 1, ctypes.byref(fmt), lut.data_ptr(), 0, fmax, gb.data_ptr(),
                                            ws.data_ptr(), wb, stream()), "qt_fake_quant_chain_bf16")
        torch.cuda.synchronize()
        assert not bool(ws.any())
        ref = out.double().sum(0)
        got = gb.double()
        for c in (0, 2, 5):
            assert bool(torch.isnan(got[c])), (c, float(got[c]))
        assert not bool(torch.isfinite(got[1])), float(got[1])
        assert bool(torch.isfinite(got[3])) and bool(torch.isfinite(got[4]))
        clean = torch.ones(cols, dtype=torch.bool, device="cuda")
        clean[[0, 1, 2, 5]] = False
        tol = ref.abs() * 2.0 ** -7 + out.double().abs().sum(0) * 2.0 ** -20 + 1e-30
        assert bool(((got - ref).abs() <= tol)[clean].all())


def _chain_setup(nv, dtype, scales, like, src):
    """Stage array + outputs + amax slots for a producer kernel, and a checker: every stage equals its own qt_fake_quant_bf16 launch on the
    tensor it reads (bit for bit, amax included)."""
    import quantized_training as qt_pkg
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    lut = qt_pkg.get_quantization_map(dtype, torch.device("cuda"))
    fmt = _launch_format(nv.format_for(dtype), lut)
    n = len(scales)
    sc = [torch.tensor([s_], dtype=torch.float32, device="cuda") for s_ in scales]
    outs = [torch.full_like(like, float("nan")) for _ in range(n)]
    amax = [torch.zeros(1, dtype=torch.float32, device="cuda") for _ in range(n)]
    stages = (nv.QtChainStage * n)()
    for i in range(n):
        stages[i].scale_f32_dev, stages[i].amax_bits_dev, stages[i].out_dev, stages[i].src = sc[i].data_ptr(), amax[i].data_ptr(), outs[i].data_ptr(), src[i]

    def check(produced):
        for i in range(n):
            inp = produced if src[i] < 0 else outs[src[i]]
            want = torch.empty_like(inp)
            am = torch.zeros(1, dtype=torch.float32, device="cuda")
            nv.check(L.qt_fake_quant_bf16(inp.data_ptr(), want.data_ptr(), inp.numel(), ctypes.byref(fmt), lut.data_ptr(), sc[i].data_ptr(), am.data_ptr(),
                                          stream()), "qt_fake_quant_bf16")
            assert torch.equal(o.canon_nan16(host_u16(outs[i].view(torch.int16))) if False else outs[i].view(torch.int16), want.view(torch.int16)), i
            assert torch.equal(amax[i].view(torch.int32), am.view(torch.int32)), i
    return stages, fmt, lut, outs, check


def _ulp_close(a, b, share):
    """bf16 tensors equal up to one bf16 step on at most `share` of the elements (and never further)."""
    ia, ib = a.view(torch.int16).int(), b.view(torch.int16).int()
    ia = torch.where(ia < 0, -32768 - ia, ia)
    ib = torch.where(ib < 0, -32768 - ib, ib)
    d = (ia - ib).abs()
    ok = int(d.max()) <= 1 and float((d > 0).float().mean()) <= share
    if not ok:
        print(f"[_ulp_close] max code distance {int(d.max())}, share differing {float((d > 0).float().mean()):.5f} (allowed {share})")
    return ok


@pytest.mark.parametrize("rows,cols", [(2048, 768), (37, 1024), (300, 264)])
def test_layernorm_train_kernels_against_torch_and_the_single_passes(nv, rows, cols):
    """qt_layernorm_train_bf16 / _backward_bf16 (the LayerNorm of a training step with the fake-quantizer calls around it): y, mean, rstd
    against torch's layer_norm on the same bf16 input (one bf16 step on < 0.5 % of the elements: another summation order of the
    statistics); grad_in / grad_weight / grad_bias against torch's native_layer_norm_backward on the kernel's own mean / rstd; every
    stage bit for bit its own fake-quantizer launch on the tensor the kernel produced; the bias-gradient column sums against fp64."""
    L = nv.lib()
    torch.manual_seed(rows)
    x = (torch.randn(rows, cols, device="cuda") * 2 + 0.3).bfloat16()
    w = (1 + 0.1 * torch.randn(cols, device="cuda")).bfloat16()
    b = (0.1 * torch.randn(cols, device="cuda")).bfloat16()
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device="cuda")
    rstd = torch.empty(rows, dtype=torch.float32, device="cuda")
    stages, fmt, lut, outs, check = _chain_setup(nv, "int8", (0.031, 0.029, 0.04), x, (-1, -1, -1))
    nv.check(L.qt_layernorm_train_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, cols, 1e-5, stages, 3,
                                       ctypes.byref(fmt), lut.data_ptr(), None, None, stream()), "qt_layernorm_train_bf16")
    # the residual add in the same launch: x = h + r formed on the way equals the launch on torch's sum bit for bit (y, statistics, stages,
    # amax), and the sum it leaves is torch's
    h = (torch.randn(rows, cols, device="cuda") * 1.5).bfloat16()
    r_ = (x.float() - h.float()).bfloat16()
    xs = h + r_
    res_a = []
    for fusedadd in (False, True):
        st2, fmt2, lut2, outs2, _ = _chain_setup(nv, "int8", (0.031, 0.029, 0.04), x, (-1, -1, -1))
        y2, m2, r2, s2 = torch.empty_like(x), torch.empty_like(mean), torch.empty_like(rstd), torch.full_like(x, float("nan"))
        am2 = [torch.zeros(1, dtype=torch.float32, device="cuda") for _ in range(3)]
        for i in range(3):
            st2[i].amax_bits_dev = am2[i].data_ptr()
        nv.check(L.qt_layernorm_train_bf16((h if fusedadd else xs).data_ptr(), w.data_ptr(), b.data_ptr(), y2.data_ptr(), m2.data_ptr(), r2.data_ptr(), rows, cols,
                                           1e-5, st2, 3, ctypes.byref(fmt2), lut2.data_ptr(), r_.data_ptr() if fusedadd else None,
                                           s2.data_ptr() if fusedadd else None, stream()), "qt_layernorm_train_bf16")
        res_a.append([t.clone() for t in (y2, m2, r2, *outs2, *am2)] + ([s2.clone()] if fusedadd else []))
    for a_, b_ in zip(res_a[0], res_a[1]):
        assert torch.equal(a_.view(torch.int16 if a_.dtype == torch.bfloat16 else torch.int32), b_.view(torch.int16 if b_.dtype == torch.bfloat16 else torch.int32))
    assert torch.equal(res_a[1][-1].view(torch.int16), xs.view(torch.int16))
    ref = torch.nn.functional.layer_norm(x, (cols,), w, b, 1e-5)
    big = ref.float().abs() >= 2.0 ** -4                        # (outputs near zero come from a cancellation: compare those absolutely)
    assert _ulp_close(torch.where(big, y, ref), ref, 5e-3)
    assert float((y.float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())
    assert float((y.float() - ref.float())[~big].abs().max()) <= 2.0 ** -9
    xf = x.float()
    assert torch.allclose(mean, xf.mean(-1), rtol=1e-5, atol=1e-6) and torch.allclose(rstd, (xf.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-5)
    check(y)
    # backward
    dy = (torch.randn(rows, cols, device="cuda") * 3e-5).bfloat16()
    dx = torch.empty_like(x)
    gw, gb, gbias = (torch.empty(cols, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    stages, fmt, lut, outs, check = _chain_setup(nv, "fp8_e5m2", (2.1e-9, 1.7e-9, 2.6e-9, 1.0), dx, (-1, 0, 0, 1))
    groups = L.qt_layernorm_train_backward_groups(rows)
    part = torch.empty(groups * 3 * cols, dtype=torch.float32, device="cuda")
    nv.check(L.qt_layernorm_train_backward_bf16(dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), rows, cols, stages, 4,
                                                ctypes.byref(fmt), lut.data_ptr(), 3, part.data_ptr(), part.numel() * 4, gw.data_ptr(), gb.data_ptr(),
                                                gbias.data_ptr(), None, 0, stream()), "qt_layernorm_train_backward_bf16")
    rx, rw, rb = torch.ops.aten.native_layer_norm_backward(dy, x, [cols], mean.view(rows, 1), rstd.view(rows, 1), w, b, [True, True, True])
    scale = float(rx.float().abs().max())
    assert float((dx.float() - rx.float()).abs().max()) <= 2.0 ** -7 * scale
    assert float((dx.float() - rx.float()).abs().mean()) <= 2.0 ** -10 * scale
    xhat = (xf - mean[:, None]) * rstd[:, None]
    ref_w, ref_b = (dy.double() * xhat.double()).sum(0), dy.double().sum(0)
    tol_w = ref_w.abs() * 2.0 ** -7 + (dy.double().abs() * xhat.double().abs()).sum(0) * 2.0 ** -18
    tol_b = ref_b.abs() * 2.0 ** -7 + dy.double().abs().sum(0) * 2.0 ** -18
    assert bool(((gw.double() - ref_w).abs() <= tol_w).all()) and bool(((gb.double() - ref_b).abs() <= tol_b).all())
    check(dx)
    ref_c = outs[3].double().sum(0)
    assert bool(((gbias.double() - ref_c).abs() <= ref_c.abs() * 2.0 ** -7 + outs[3].double().abs().sum(0) * 2.0 ** -18).all())
    gw2, gb2 = torch.empty_like(gw), torch.empty_like(gb)
    nv.check(L.qt_layernorm_train_backward_bf16(dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), rows, cols, stages, 4,
                                                ctypes.byref(fmt), lut.data_ptr(), -1, part.data_ptr(), part.numel() * 4, gw2.data_ptr(), gb2.data_ptr(), None,
                                                None, 0, stream()), "qt_layernorm_train_backward_bf16")
    assert torch.equal(gw.view(torch.int16), gw2.view(torch.int16)) and torch.equal(gb.view(torch.int16), gb2.view(torch.int16))
    # further gradients arriving for the LayerNorm's result, summed while grad_out is loaded: the launch on (first, arrivals) equals the
    # launch on qt_grad_fanin_bf16(first, arrivals) bit for bit -- every output, every amax slot
    first = (torch.randn(rows, cols, device="cuda") * 2e-5).bfloat16()
    xs = [(torch.randn(rows, cols, device="cuda") * 1e-5).bfloat16() for _ in range(3)]
    fsc = [torch.tensor([v], dtype=torch.float32, device="cuda") for v in (1.3e-9, 0.0, 2.2e-9)]
    for use in (0, 1):
        fam = [torch.zeros(1, dtype=torch.float32, device="cuda") for _ in range(3)]
        items = (nv.QtFaninItem * 3)()
        for i in range(3):
            items[i].x_dev, items[i].fq = xs[i].data_ptr(), (0 if i == 1 else 1)
            items[i].scale_f32_dev, items[i].amax_bits_dev = (fsc[i].data_ptr(), fam[i].data_ptr()) if i != 1 else (None, None)
        stages, fmt, lut, outs, check = _chain_setup(nv, "fp8_e5m2", (2.1e-9, 1.7e-9, 2.6e-9, 1.0), dx, (-1, 0, 0, 1))
        dxs, gws, gbs, gcs = torch.empty_like(x), torch.empty_like(gw), torch.empty_like(gb), torch.empty_like(gb)
        if use == 0:
            total = torch.empty_like(first)
            nv.check(L.qt_grad_fanin_bf16(first.data_ptr(), items, 3, total.data_ptr(), first.numel(), ctypes.byref(fmt), lut.data_ptr(), stream()), "fanin")
            args = (total, None, 0)
        else:
            args = (first, items, 3)
        nv.check(L.qt_layernorm_train_backward_bf16(args[0].data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dxs.data_ptr(), rows, cols,
                                                    stages, 4, ctypes.byref(fmt), lut.data_ptr(), 3, part.data_ptr(), part.numel() * 4, gws.data_ptr(),
                                                    gbs.data_ptr(), gcs.data_ptr(), args[1], args[2], stream()), "qt_layernorm_train_backward_bf16")
        got = [t.clone() for t in (dxs, gws, gbs, gcs, *outs, *fam)]
        if use == 0:
            ref_run = got
    for a_, b_ in zip(ref_run, got):
        assert torch.equal(a_.view(torch.int16 if a_.dtype == torch.bfloat16 else torch.int32), b_.view(torch.int16 if b_.dtype == torch.bfloat16 else torch.int32))


@pytest.mark.parametrize("rows,cols", [(2048, 3072), (77, 64)])
def test_gelu_train_kernels_against_torch_and_the_single_passes(nv, rows, cols):
    """qt_gelu_chain_bf16 equals torch's erf GELU on bf16 bit for bit; qt_gelu_backward_chain_bf16 equals torch's gelu_backward up to one
    bf16 step on a small share (expf / erff of another library); stages bit for bit their own launches; column sums against fp64."""
    L = nv.lib()
    torch.manual_seed(cols)
    x = (torch.randn(rows, cols, device="cuda") * 1.5).bfloat16()
    y = torch.empty_like(x)
    stages, fmt, lut, outs, check = _chain_setup(nv, "int8", (0.023,), x, (-1,))
    nv.check(L.qt_gelu_chain_bf16(x.data_ptr(), y.data_ptr(), rows, cols, stages, 1, ctypes.byref(fmt), lut.data_ptr(), stream()), "qt_gelu_chain_bf16")
    assert torch.equal(y.view(torch.int16), torch.nn.functional.gelu(x).view(torch.int16))
    check(y)
    dy = (torch.randn(rows, cols, device="cuda") * 2e-5).bfloat16()
    dx = torch.empty_like(x)
    gb = torch.empty(cols, dtype=torch.bfloat16, device="cuda")
    stages, fmt, lut, outs, check = _chain_setup(nv, "fp8_e5m2", (1.5e-9,), dx, (-1,))
    wb = L.qt_fake_quant_chain_ws_bytes(rows, cols)
    ws = torch.zeros(wb, dtype=torch.uint8, device="cuda")
    nv.check(L.qt_gelu_backward_chain_bf16(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), rows, cols, stages, 1, ctypes.byref(fmt), lut.data_ptr(), 0, 57344.0,
                                           gb.data_ptr(), ws.data_ptr(), wb, stream()), "qt_gelu_backward_chain_bf16")
    assert _ulp_close(dx, torch.ops.aten.gelu_backward(dy, x, approximate="none"), 2e-2)
    check(dx)
    ref = outs[0].double().sum(0)
    assert bool(((gb.double() - ref).abs() <= ref.abs() * 2.0 ** -7 + outs[0].double().abs().sum(0) * 2.0 ** -20).all())
    assert not bool(ws.any())


@pytest.mark.parametrize("B,H,Q,C,masked", [(16, 12, 128, 128, True), (2, 3, 40, 384, False)])
def test_softmax_train_kernels_against_torch_and_the_single_passes(nv, B, H, Q, C, masked):
    """qt_softmax_fq_probs_bf16: the unquantized probabilities equal the module chain's (bf16(score * scaling), bf16(+ mask), softmax:
    one bf16 step on a small share) and the quantized ones are exactly the fake-quantizer of those; qt_softmax_backward_chain_bf16 against
    torch's _softmax_backward_data and the scaling's backward, its stage bit for bit its own launch."""
    import quantized_training as qt_pkg
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    torch.manual_seed(C)
    scores = (torch.randn(B, H, Q, C, device="cuda") * 3).bfloat16()
    mask = None
    if masked:
        mask = torch.zeros(B, 1, 1, C, device="cuda", dtype=torch.bfloat16)
        mask[::2, :, :, C - 24:] = torch.finfo(torch.bfloat16).min
    scaling = 0.125
    lut = qt_pkg.get_quantization_map("int8", torch.device("cuda"))
    fmt = _launch_format(nv.format_for("int8"), lut)
    sc = torch.tensor([0.0079], dtype=torch.float32, device="cuda")
    am = torch.zeros(1, dtype=torch.float32, device="cuda")
    out, probs = torch.empty_like(scores), torch.empty_like(scores)
    nv.check(L.qt_softmax_fq_probs_bf16(scores.data_ptr(), mask.data_ptr() if masked else None, out.data_ptr(), probs.data_ptr(), B, H, Q, C,
                                        mask.stride(0) if masked else 0, 0, 0, scaling, ctypes.byref(fmt), lut.data_ptr(), sc.data_ptr(), am.data_ptr(),
                                        stream()), "qt_softmax_fq_probs_bf16")
    t = scores * scaling
    if masked:
        t = t + mask
    ref = torch.softmax(t, -1)
    assert _ulp_close(probs, ref, 2e-2)
    want = torch.empty_like(probs)
    am2 = torch.zeros(1, dtype=torch.float32, device="cuda")
    nv.check(L.qt_fake_quant_bf16(probs.data_ptr(), want.data_ptr(), probs.numel(), ctypes.byref(fmt), lut.data_ptr(), sc.data_ptr(), am2.data_ptr(), stream()), "fq")
    assert torch.equal(out.view(torch.int16), want.view(torch.int16)) and torch.equal(am.view(torch.int32), am2.view(torch.int32))
    dp = (torch.randn(B, H, Q, C, device="cuda") * 1e-4).bfloat16()
    ds = torch.empty_like(scores)
    stages, fmt5, lut5, outs, check = _chain_setup(nv, "fp8_e5m2", (3.0e-10,), ds, (-1,))
    nv.check(L.qt_softmax_backward_chain_bf16(dp.data_ptr(), probs.data_ptr(), ds.data_ptr(), B * H * Q, C, scaling, stages, 1, ctypes.byref(fmt5), lut5.data_ptr(),
                                              stream()), "qt_softmax_backward_chain_bf16")
    rs = torch.ops.aten._softmax_backward_data(dp, probs, -1, probs.dtype) * scaling
    scale = float(rs.float().abs().max())
    assert float((ds.float() - rs.float()).abs().max()) <= 2.0 ** -7 * scale and float((ds.float() - rs.float()).abs().mean()) <= 2.0 ** -10 * scale
    check(ds)


def _products_close(got, ref64, share=1e-2):
    """A bf16 product tensor against the same product evaluated in fp64: all but `share` of the elements are the fp64 result rounded once,
    the others within one bf16 step of it (fp32 accumulation order), counted relative to each element with a floor for cancelling sums."""
    ref = ref64.float().bfloat16()
    same = float((got.view(torch.int16) == ref.view(torch.int16)).float().mean())
    err = (got.double() - ref64).abs()
    bound = 2.0 ** -7 * ref64.abs() + 2.0 ** -16 * float(ref64.abs().max())
    ok = same >= 1.0 - share and bool((err <= bound).all())
    if not ok:
        print(f"[_products_close] bit-equal share {same:.5f}, worst excess {float((err - bound).max()):.3e}")
    return ok


@pytest.mark.parametrize("B,H,S,masked,pow2,drop", [(16, 12, 128, True, True, 0.0), (2, 3, 64, False, True, 0.1), (3, 5, 96, True, False, 0.0),
                                                     (1, 2, 32, False, False, 0.25), (4, 12, 128, True, True, 0.1)])
def test_attention_train_kernels_against_the_launches_they_replace(nv, B, H, S, masked, pow2, drop):
    """qt_attention_train_bf16 / qt_attention_train_backward_bf16 against the launches of the unfused training step: the fake-quantizers
    (own launches of qt_fake_quant_bf16, amax included), torch.matmul for the four products, qt_softmax_fq_probs_bf16 and
    qt_softmax_backward_chain_bf16 for the softmax.  With power-of-two scales every dot of int8 values is exact in fp32 whatever the
    order of its additions, so the forward has to agree BIT FOR BIT (q', k', v', P, P', the result and its quantized form, every amax),
    and so do dV and the amax of both gradient quantizers; dQ / dK (E5M2 values of many binades: order-dependent in fp32) and every
    product under general scales are held to the fp64 product rounded once.  drop > 0: attention-probability dropout with a given keep
    mask -- torch's own dropout arithmetic (native_dropout_backward on the mask) between the softmax and the fake-quantizer, both ways."""
    import quantized_training as qt_pkg
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    D = 64
    assert L.qt_attention_train_supported(B, H, S, D) == 1 and L.qt_attention_train_supported(B, H, 384, D) == 0
    torch.manual_seed(S + B)
    dev = torch.device("cuda")

    def proj():
        return torch.randn(B, S, H * D, device=dev).bfloat16().view(B, S, H, D).permute(0, 2, 1, 3)
    q, k, v = proj(), proj(), proj()
    mask = None
    if masked:
        mask = torch.zeros(B, 1, 1, S, device=dev, dtype=torch.bfloat16)
        mask[::2, :, :, S - 24:] = torch.finfo(torch.bfloat16).min
    scaling = 0.125
    lut = qt_pkg.get_quantization_map("int8", dev)
    fmt = _launch_format(nv.format_for("int8"), lut)
    scales = (2.0 ** -5, 2.0 ** -5, 2.0 ** -5, 2.0 ** -7, 2.0 ** -4) if pow2 else (0.0317, 0.0291, 0.0333, 0.00787, 0.0171)
    sc = [torch.tensor([x], dtype=torch.float32, device=dev) for x in scales]
    am = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in scales]
    qq, kq, vq = (torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=dev) for _ in range(3))
    probs = torch.empty(B, H, S, S, dtype=torch.bfloat16, device=dev)
    pq = torch.empty_like(probs)
    out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device=dev)
    oq = torch.empty_like(out)
    outs = [qq, kq, vq, pq, oq]
    stages = (nv.QtChainStage * 5)()
    for i in range(5):
        stages[i].scale_f32_dev, stages[i].amax_bits_dev, stages[i].out_dev, stages[i].src = sc[i].data_ptr(), am[i].data_ptr(), outs[i].data_ptr(), -1
    keep = torch.empty(B, H, S, S, dtype=torch.uint8, device=dev).bernoulli_(1.0 - drop) if drop else None
    dscale = 1.0 / (1.0 - drop)

    def dropped(t):                    # torch's masked-scale arithmetic (what nn.functional.dropout computes for this mask)
        return torch.ops.aten.native_dropout_backward(t, keep.bool(), dscale) if drop else t
    nv.check(L.qt_attention_train_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(2), q.stride(1), mask.data_ptr() if masked else None,
                                       mask.stride(0) if masked else 0, 0, 0, stages, probs.data_ptr(), out.data_ptr(), keep.data_ptr() if drop else None,
                                       dscale, B, H, S, D, scaling, ctypes.byref(fmt), lut.data_ptr(), stream()), "qt_attention_train_bf16")

    def fq(x, i, f=fmt, lt=lut, scl=None):
        x = x.contiguous()
        want = torch.empty_like(x)
        a = torch.zeros(1, dtype=torch.float32, device=dev)
        nv.check(L.qt_fake_quant_bf16(x.data_ptr(), want.data_ptr(), x.numel(), ctypes.byref(f), lt.data_ptr(), (scl if scl is not None else sc[i]).data_ptr(),
                                      a.data_ptr(), stream()), "qt_fake_quant_bf16")
        return want, a
    for i, (x, y) in enumerate(((q, qq), (k, kq), (v, vq))):
        want, a = fq(x, i)
        assert torch.equal(y.contiguous().view(torch.int16), want.view(torch.int16)) and torch.equal(am[i].view(torch.int32), a.view(torch.int32)), i
    # the unfused chain from here on, on the kernel's own q', k', v'
    s_ref = torch.matmul(qq.contiguous(), kq.contiguous().transpose(2, 3))
    pq_ref, p_ref = torch.empty_like(probs), torch.empty_like(probs)
    a3 = torch.zeros(1, dtype=torch.float32, device=dev)
    nv.check(L.qt_softmax_fq_probs_bf16(s_ref.data_ptr(), mask.data_ptr() if masked else None, pq_ref.data_ptr(), p_ref.data_ptr(), B, H, S, S,
                                        mask.stride(0) if masked else 0, 0, 0, scaling, ctypes.byref(fmt), lut.data_ptr(), sc[3].data_ptr(), a3.data_ptr(),
                                        stream()), "qt_softmax_fq_probs_bf16")
    o_log = out.view(B, S, H, D).permute(0, 2, 1, 3)
    if drop:
        pq_ref, a3 = fq(dropped(p_ref), 3)
    if pow2:
        assert torch.equal(probs.view(torch.int16), p_ref.view(torch.int16)) and torch.equal(pq.view(torch.int16), pq_ref.view(torch.int16))
        assert torch.equal(am[3].view(torch.int32), a3.view(torch.int32))
        o_ref = torch.matmul(pq, vq.contiguous())
        assert torch.equal(o_log.contiguous().view(torch.int16), o_ref.view(torch.int16))
    else:
        d = (probs.float() - p_ref.float()).abs()
        assert float((d > 0).float().mean()) <= 2e-2 and float(d.max()) <= 2.0 ** -5, (float((d > 0).float().mean()), float(d.max()))
    # (whatever the scales) P' is the fake-quantizer of the kernel's own P, the result the product of its own P' and v', fq4 of its own result
    want, a = fq(dropped(probs), 3)
    assert torch.equal(pq.view(torch.int16), want.view(torch.int16)) and torch.equal(am[3].view(torch.int32), a.view(torch.int32))
    assert _products_close(o_log, torch.matmul(pq.double(), vq.contiguous().double()))
    want, a = fq(out, 4)
    assert torch.equal(oq.view(torch.int16), want.view(torch.int16)) and torch.equal(am[4].view(torch.int32), a.view(torch.int32))

    # ---- backward
    lut5 = qt_pkg.get_quantization_map("fp8_e5m2", dev)
    fmt5 = _launch_format(nv.format_for("fp8_e5m2"), lut5)
    if pow2:      # one binade, eight mantissa bits: E5M2 rounds them, and every sum of the products stays exact
        gy = (torch.rand(B, S, H, D, device=dev) + 1.0) * 2.0 ** -10 * torch.where(torch.rand(B, S, H, D, device=dev) < 0.5, -1.0, 1.0)
    else:
        gy = torch.randn(B, S, H, D, device=dev) * 1e-3
    gy = gy.bfloat16()
    esc = [torch.tensor([x], dtype=torch.float32, device=dev) for x in ((2.0 ** -24, 2.0 ** -24) if pow2 else (1.1e-7, 0.9e-7))]
    eam = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in esc]
    est = (nv.QtChainStage * 2)()
    for i in range(2):
        est[i].scale_f32_dev, est[i].amax_bits_dev, est[i].out_dev, est[i].src = esc[i].data_ptr(), eam[i].data_ptr(), None, -1
    dq, dk, dv = (torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(3))
    extras = masked                    # with and without the optional outputs (g, dS, dS')
    g_out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev) if extras else None
    ds_out, dsq_out = (torch.empty_like(probs), torch.empty_like(probs)) if extras else (None, None)
    if extras:
        est[0].out_dev, est[1].out_dev = g_out.data_ptr(), dsq_out.data_ptr()
    # the projections' own backward-pre quantizers riding on dQ / dK / dV, with the bias gradients (dK's without its column sums)
    gsc = [torch.tensor([x], dtype=torch.float32, device=dev) for x in ((2.0 ** -22, 2.0 ** -23, 2.0 ** -21) if pow2 else (2.3e-7, 1.9e-7, 3.3e-7))]
    gam = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in gsc]
    gout = [torch.full((B, S, H * D), float("nan"), dtype=torch.bfloat16, device=dev) for _ in gsc]
    gbias = [torch.full((H * D,), float("nan"), dtype=torch.bfloat16, device=dev) for _ in gsc]
    gst = (nv.QtChainStage * 3)()
    couts = (ctypes.c_void_p * 3)()
    for i in range(3):
        gst[i].scale_f32_dev, gst[i].amax_bits_dev, gst[i].out_dev, gst[i].src = gsc[i].data_ptr(), gam[i].data_ptr(), gout[i].data_ptr(), -1
        couts[i] = gbias[i].data_ptr() if i != 1 else None
    ws = torch.zeros(L.qt_attention_train_backward_ws_bytes(H), dtype=torch.uint8, device=dev)
    for rep_ in range(2):              # twice: the second launch finds the scratch as the first left it (zero) and must give the same sums
        first = [t.clone() for t in gbias]
        nv.check(L.qt_attention_train_backward_bf16(gy.data_ptr(), qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), qq.stride(0), qq.stride(2), qq.stride(1),
                                                    probs.data_ptr(), pq.data_ptr(), est, ds_out.data_ptr() if extras else None, dq.data_ptr(), dk.data_ptr(),
                                                    dv.data_ptr(), gst, couts, 57344.0, ws.data_ptr(), ws.numel(), keep.data_ptr() if drop else None, dscale,
                                                    B, H, S, D, scaling, ctypes.byref(fmt5), lut5.data_ptr(), stream()), "qt_attention_train_backward_bf16")
    assert int(ws.count_nonzero()) == 0
    for i, grad in enumerate((dq, dk, dv)):
        want, a = fq(grad, 0, fmt5, lut5, gsc[i])
        assert torch.equal(gout[i].view(torch.int16), want.view(B, S, H * D).view(torch.int16)), i
        if i != 1:
            assert torch.equal(gbias[i].view(torch.int16), first[i].view(torch.int16)), i
            ref = gout[i].double().sum((0, 1))
            err = (gbias[i].double() - ref).abs()
            assert bool((err <= 2.0 ** -7 * ref.abs() + 2.0 ** -12 * float(gout[i].double().abs().sum((0, 1)).max())).all()), (i, float(err.max()))
        else:
            assert bool(torch.isnan(gbias[i].float()).all())
    # (the amax slots saw two launches of the same data: max is idempotent)
    for i, grad in enumerate((dq, dk, dv)):
        assert float(gam[i]) == float(grad.float().abs().max()), i
    g, a0 = fq(gy.permute(0, 2, 1, 3), 0, fmt5, lut5, esc[0])
    assert torch.equal(eam[0].view(torch.int32), a0.view(torch.int32))
    if extras:
        assert torch.equal(g_out.permute(0, 2, 1, 3).contiguous().view(torch.int16), g.view(torch.int16))
        want, a = fq(ds_out, 1, fmt5, lut5, esc[1])
        assert torch.equal(dsq_out.view(torch.int16), want.view(torch.int16)) and torch.equal(eam[1].view(torch.int32), a.view(torch.int32))
    dv_log, dq_log, dk_log = (t.permute(0, 2, 1, 3) for t in (dv, dq, dk))
    assert _products_close(dv_log, torch.matmul(pq.double().transpose(2, 3), g.double()))
    dp_ref = dropped(torch.matmul(g, vq.contiguous().transpose(2, 3))).contiguous()
    ds_ref, dsq_ref = torch.empty_like(probs), torch.empty_like(probs)
    a1 = torch.zeros(1, dtype=torch.float32, device=dev)
    st1 = (nv.QtChainStage * 1)()
    st1[0].scale_f32_dev, st1[0].amax_bits_dev, st1[0].out_dev, st1[0].src = esc[1].data_ptr(), a1.data_ptr(), dsq_ref.data_ptr(), -1
    nv.check(L.qt_softmax_backward_chain_bf16(dp_ref.data_ptr(), probs.data_ptr(), ds_ref.data_ptr(), B * H * S, S, scaling, st1, 1, ctypes.byref(fmt5),
                                              lut5.data_ptr(), stream()), "qt_softmax_backward_chain_bf16")
    rq, rk = torch.matmul(dsq_ref.double(), kq.contiguous().double()), torch.matmul(dsq_ref.double().transpose(2, 3), qq.contiguous().double())
    if pow2:
        assert torch.equal(dv_log.contiguous().view(torch.int16), torch.matmul(pq.transpose(2, 3), g).view(torch.int16))
        assert torch.equal(eam[1].view(torch.int32), a1.view(torch.int32))
        if extras:
            assert torch.equal(ds_out.view(torch.int16), ds_ref.view(torch.int16)) and torch.equal(dsq_out.view(torch.int16), dsq_ref.view(torch.int16))
        assert _products_close(dq_log, rq) and _products_close(dk_log, rk)
    else:
        # a score gradient that rounds the other way moves E5M2 codes of dS': compare in the large
        for got, ref in ((dq_log, rq), (dk_log, rk)):
            e = (got.double() - ref).abs()
            assert float(e.max()) <= 0.05 * float(ref.abs().max()) and float(e.mean()) <= 2e-3 * float(ref.abs().mean()) + 1e-12, (float(e.max()), float(e.mean()))
    assert not any(bool(torch.isnan(t.float()).any()) for t in (dq, dk, dv))


@pytest.mark.parametrize("B", [4, 16, 32])
def test_attention_train_backward_bias_sums_keep_nan(nv, B):
    """The attention backward sums the projections' bias gradients over the batch in fixed point, one arrival per batch element
    (qt_attention_train.hip).  A fully-NaN incoming gradient (a diverged step) must leave NaN bias gradients, as torch's
    grad_output.sum(0) does, for 4, 16 and 32 arrivals (round 5's 2^62 marker wrapped to zero at exactly these counts), and the scratch
    must be left zero for the next launch."""
    import quantized_training as qt_pkg
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    H, S, D = 2, 64, 64
    dev = torch.device("cuda")
    torch.manual_seed(B)
    lut = qt_pkg.get_quantization_map("int8", dev)
    fmt = _launch_format(nv.format_for("int8"), lut)
    lut5 = qt_pkg.get_quantization_map("fp8_e5m2", dev)
    fmt5 = _launch_format(nv.format_for("fp8_e5m2"), lut5)

    def proj():
        return torch.randn(B, S, H * D, device=dev).bfloat16().view(B, S, H, D).permute(0, 2, 1, 3)
    q, k, v = proj(), proj(), proj()
    sc = [torch.tensor([x], dtype=torch.float32, device=dev) for x in (2.0 ** -5, 2.0 ** -5, 2.0 ** -5, 2.0 ** -7, 2.0 ** -4)]
    am = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in sc]
    qq, kq, vq = (torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=dev) for _ in range(3))
    probs = torch.empty(B, H, S, S, dtype=torch.bfloat16, device=dev)
    pq = torch.empty_like(probs)
    out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device=dev)
    oq = torch.empty_like(out)
    outs = [qq, kq, vq, pq, oq]
    stages = (nv.QtChainStage * 5)()
    for i in range(5):
        stages[i].scale_f32_dev, stages[i].amax_bits_dev, stages[i].out_dev, stages[i].src = sc[i].data_ptr(), am[i].data_ptr(), outs[i].data_ptr(), -1
    nv.check(L.qt_attention_train_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(2), q.stride(1), None, 0, 0, 0, stages,
                                       probs.data_ptr(), out.data_ptr(), None, 1.0, B, H, S, D, 0.125, ctypes.byref(fmt), lut.data_ptr(), stream()),
             "qt_attention_train_bf16")
    esc = [torch.tensor([2.0 ** -24], dtype=torch.float32, device=dev) for _ in range(2)]
    eam = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in esc]
    est = (nv.QtChainStage * 2)()
    for i in range(2):
        est[i].scale_f32_dev, est[i].amax_bits_dev, est[i].out_dev, est[i].src = esc[i].data_ptr(), eam[i].data_ptr(), None, -1
    gsc = [torch.tensor([2.0 ** -22], dtype=torch.float32, device=dev) for _ in range(3)]
    gam = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in gsc]
    gout = [torch.empty(B, S, H * D, dtype=torch.bfloat16, device=dev) for _ in gsc]
    gbias = [torch.zeros(H * D, dtype=torch.bfloat16, device=dev) for _ in gsc]
    gst = (nv.QtChainStage * 3)()
    couts = (ctypes.c_void_p * 3)()
    for i in range(3):
        gst[i].scale_f32_dev, gst[i].amax_bits_dev, gst[i].out_dev, gst[i].src = gsc[i].data_ptr(), gam[i].data_ptr(), gout[i].data_ptr(), -1
        couts[i] = gbias[i].data_ptr()
    dq, dk, dv = (torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev) for _ in range(3))
    ws = torch.zeros(L.qt_attention_train_backward_ws_bytes(H), dtype=torch.uint8, device=dev)
    for gy_kind in ("nan", "finite"):          # the finite launch afterwards finds clean scratch and gives finite sums
        gy = torch.full((B, S, H, D), float("nan"), device=dev).bfloat16() if gy_kind == "nan" else (torch.randn(B, S, H, D, device=dev) * 1e-3).bfloat16()
        nv.check(L.qt_attention_train_backward_bf16(gy.data_ptr(), qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), qq.stride(0), qq.stride(2), qq.stride(1),
                                                    probs.data_ptr(), pq.data_ptr(), est, None, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), gst, couts,
                                                    57344.0, ws.data_ptr(), ws.numel(), None, 1.0, B, H, S, D, 0.125, ctypes.byref(fmt5), lut5.data_ptr(),
                                                    stream()), "qt_attention_train_backward_bf16")
        torch.cuda.synchronize()
        assert int(ws.count_nonzero()) == 0
        for i in range(3):
            if gy_kind == "nan":
                assert bool(torch.isnan(gbias[i].float()).all()), (i, int(torch.isnan(gbias[i].float()).sum()))
            else:
                assert bool(torch.isfinite(gbias[i].float()).all()), i


@pytest.mark.parametrize("layout", ["forward", "dgrad", "wgrad", "tn"])
@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (2048, 3072, 768), (2048, 768, 3072), (768, 768, 2048), (3072, 768, 2048), (768, 3072, 2048),
                                   (200, 72, 256), (8, 8, 256), (136, 264, 320), (2048, 2304, 768), (1992, 3072, 320)])
def test_train_gemm_against_fp64_products(nv, layout, M, N, K):
    """qt_train_gemm_bf16, the three products of a QAT Linear under autograd (modules/qat/linear.py:40-41 and its backward): every layout
    (k-contiguous operands through ds_read_b128, operands with the contraction index as the row index through gfx950's transposing
    ds_read_b64_tr_b16), the shapes of a RoBERTa-base layer at [16, 128] and ragged ones (partial tiles in both directions; the wide forward
    products take 128 x 192 tiles -- 2048 / 1992 x 3072, 2048 x 2304, q / k / v together), one and three problems per launch, with and without bias: every element within one bf16 rounding of the fp64 product of the same operands
    plus fp32 accumulation (2^-8 |ref| + 2^-18 sum |a||b|), and run-to-run bit-identical (the order of an element's additions is fixed)."""
    ta, tb = {"forward": (0, 0), "dgrad": (0, 1), "wgrad": (1, 1), "tn": (1, 0)}[layout]
    torch.manual_seed(M + N + K)
    L = nv.lib()
    count = 3 if (M, N, K) in ((2048, 768, 768), (768, 768, 2048), (200, 72, 256)) else 1
    As = [(torch.randn((K, M) if ta else (M, K), device="cuda") * 0.5).bfloat16() for _ in range(count)]
    Bs = [(torch.randn((K, N) if tb else (N, K), device="cuda") * 0.05).bfloat16() for _ in range(count)]
    biases = [torch.randn(N, device="cuda").bfloat16() if i % 2 == 0 else None for i in range(count)]

    def run():
        Cs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(count)]
        arr = (nv.QtGemmProblem * count)()
        for i in range(count):
            arr[i].a, arr[i].b, arr[i].c = As[i].data_ptr(), Bs[i].data_ptr(), Cs[i].data_ptr()
            arr[i].bias = biases[i].data_ptr() if biases[i] is not None else None
        nv.check(L.qt_train_gemm_bf16(arr, count, ta, tb, M, N, K, As[0].stride(0), Bs[0].stride(0), N, stream()), "qt_train_gemm_bf16")
        return Cs
    first, again = run(), run()
    for i in range(count):
        a = (As[i].t() if ta else As[i]).double()
        b = (Bs[i] if tb else Bs[i].t()).double()
        ref = a @ b + (biases[i].double() if biases[i] is not None else 0.0)
        tol = ref.abs() * 2.0 ** -8 + (a.abs() @ b.abs()) * 2.0 ** -18 + 1e-30
        err = (first[i].double() - ref).abs()
        assert bool((err <= tol).all()), (i, float((err / tol).max()))
        assert torch.equal(first[i].view(torch.int16), again[i].view(torch.int16))
    # what it refuses (the caller keeps torch's GEMM)
    arr = (nv.QtGemmProblem * 1)()
    arr[0].a, arr[0].b, arr[0].c = As[0].data_ptr(), Bs[0].data_ptr(), first[0].data_ptr()
    assert L.qt_train_gemm_bf16(arr, 1, ta, tb, M, N, 96, As[0].stride(0), Bs[0].stride(0), N, stream()) == nv.QT_ERR_BAD_ARG       # K % 64, K < 256
    assert L.qt_train_gemm_bf16(arr, 5, ta, tb, M, N, K, As[0].stride(0), Bs[0].stride(0), N, stream()) == nv.QT_ERR_BAD_ARG
    arr[0].a = As[0].data_ptr() + 2
    assert L.qt_train_gemm_bf16(arr, 1, ta, tb, M, N, K, As[0].stride(0), Bs[0].stride(0), N, stream()) == nv.QT_ERR_UNALIGNED


@pytest.mark.parametrize("T,O,I,count", [(2048, 768, 768, 3), (2048, 768, 768, 1), (2048, 3072, 768, 1), (2048, 768, 3072, 1), (256, 320, 72, 2),
                                         (320, 256, 8, 4)])
def test_train_gemm_backward_pair_equals_the_two_single_launches(nv, T, O, I, count):
    """qt_train_gemm_backward_bf16: the input gradients gy . Wq and the weight gradients gy^T . x of one to four Linears of one shape in ONE
    launch (modules/qat/linear.py:40-41 under autograd; query / key / value of a RoBERTa layer, its three other Linears, ragged shapes
    with partial tiles): bit for bit what qt_train_gemm_bf16 gives for (trans_a 0, trans_b 1) and (1, 1) -- the same tiles in the same
    k order -- and, like them, within one bf16 rounding of the fp64 products."""
    torch.manual_seed(T + O + I + count)
    L = nv.lib()
    gys = [(torch.randn(T, O, device="cuda") * 0.05).bfloat16() for _ in range(count)]
    ws = [(torch.randn(O, I, device="cuda") * 0.05).bfloat16() for _ in range(count)]
    xs = [(torch.randn(T, I, device="cuda") * 0.5).bfloat16() for _ in range(count)]
    gxs = [torch.full((T, I), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(count)]
    gws = [torch.full((O, I), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(count)]
    items = (nv.QtLinearBackward * count)()
    for i in range(count):
        items[i].gy, items[i].wq, items[i].x, items[i].gx, items[i].gw = gys[i].data_ptr(), ws[i].data_ptr(), xs[i].data_ptr(), gxs[i].data_ptr(), gws[i].data_ptr()
    nv.check(L.qt_train_gemm_backward_bf16(items, count, T, O, I, O, I, I, I, I, stream()), "qt_train_gemm_backward_bf16")

    def single(As, Bs, ta, tb, M, N, K):
        Cs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(count)]
        arr = (nv.QtGemmProblem * count)()
        for i in range(count):
            arr[i].a, arr[i].b, arr[i].bias, arr[i].c = As[i].data_ptr(), Bs[i].data_ptr(), None, Cs[i].data_ptr()
        nv.check(L.qt_train_gemm_bf16(arr, count, ta, tb, M, N, K, As[0].stride(0), Bs[0].stride(0), N, stream()), "qt_train_gemm_bf16")
        return Cs
    d1 = single(gys, ws, 0, 1, T, I, O)
    w1 = single(gys, xs, 1, 1, O, I, T)
    for i in range(count):
        assert torch.equal(gxs[i].view(torch.int16), d1[i].view(torch.int16)), ("input gradient", i)
        assert torch.equal(gws[i].view(torch.int16), w1[i].view(torch.int16)), ("weight gradient", i)
        for got, a, b in ((gxs[i], gys[i].double(), ws[i].double()), (gws[i], gys[i].t().double(), xs[i].double())):
            ref = a @ b
            tol = ref.abs() * 2.0 ** -8 + (a.abs() @ b.abs()) * 2.0 ** -18 + 1e-30
            assert bool(((got.double() - ref).abs() <= tol).all())
    # what it refuses (the caller issues the two single launches)
    assert L.qt_train_gemm_backward_bf16(items, count, 96, O, I, O, I, I, I, I, stream()) == nv.QT_ERR_BAD_ARG       # T % 64, T < 256
    assert L.qt_train_gemm_backward_bf16(items, 5, T, O, I, O, I, I, I, I, stream()) == nv.QT_ERR_BAD_ARG
    items[0].x = xs[0].data_ptr() + 2
    assert L.qt_train_gemm_backward_bf16(items, count, T, O, I, O, I, I, I, I, stream()) == nv.QT_ERR_UNALIGNED


@pytest.mark.parametrize("M,N,K", [(16, 2, 768), (16, 3, 1024), (128, 5, 64), (1, 1, 8)])
def test_train_gemm_skinny_forward_is_deterministic_and_exact(nv, M, N, K):
    """A classifier head's forward product ([16, 768] -> [16, 2], run_glue_no_trainer.py's RobertaClassificationHead.out_proj): the library
    runs such a shape as a split-K kernel whose partial sums meet through atomics, so the last bit of a logit can change from launch to
    launch; qt_train_gemm_bf16 takes N < 8 (or N % 8 != 0) at M N <= 4096 on a one-wave-per-output kernel with a fixed order of additions:
    within one bf16 rounding of the fp64 product, and 50 launches bit-identical."""
    torch.manual_seed(M * N + K)
    L = nv.lib()
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    outs = []
    for i in range(50):
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        arr = (nv.QtGemmProblem * 1)()
        arr[0].a, arr[0].b, arr[0].bias, arr[0].c = a.data_ptr(), b.data_ptr(), bias.data_ptr(), c.data_ptr()
        nv.check(L.qt_train_gemm_bf16(arr, 1, 0, 0, M, N, K, K, K, N, stream()), "qt_train_gemm_bf16")
        outs.append(c)
    ref = a.double() @ b.double().t() + bias.double()
    tol = ref.abs() * 2.0 ** -8 + (a.double().abs() @ b.double().abs().t()) * 2.0 ** -18 + 1e-30
    assert bool(((outs[0].double() - ref).abs() <= tol).all())
    assert all(torch.equal(outs[0].view(torch.int16), o_.view(torch.int16)) for o_ in outs[1:])
    arr[0].a = a.data_ptr()
    assert L.qt_train_gemm_bf16(arr, 1, 0, 1, M, N, K, K, K, N, stream()) == nv.QT_ERR_BAD_ARG        # only the forward layout is taken skinny


def test_qat_linear_training_products_run_in_tree_and_match_autograd(nv, monkeypatch):
    """A QAT Linear under autograd on the device: forward, input gradient and weight gradient through qt_train_gemm_bf16 (routes_report
    says so), against the same layer with QT_TRAIN_GEMM=0 (torch's GEMMs): every result within two bf16 roundings of each other, the
    bias gradient identical (it does not come from a GEMM)."""
    import quantized_training as qt_pkg
    from quantized_training import fused
    from quantized_training.modules.qat import linear as qlin
    torch.manual_seed(5)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("QT_TRAIN_GEMM", mode)
        qlin.GEMM_ROUTES.clear()
        torch.manual_seed(5)
        lin = torch.nn.Linear(768, 3072).cuda().bfloat16()
        model = torch.nn.Sequential(lin)
        args = qt_pkg.add_qspec_args().parse_args(["--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric",
                                                   "--error", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm",
                                                   "--quantize_backprop", "gemm", "--bf16"])
        qt_pkg.quantize(model, args)
        model.train()
        x = torch.randn(16, 128, 768, device="cuda").bfloat16().requires_grad_(True)
        for _ in range(2):                              # the second step runs with real scales
            model.zero_grad(set_to_none=True)
            x.grad = None
            y = model(x)
            (y.float() * torch.linspace(-1, 1, 3072, device="cuda")).mean().backward()
        res[mode] = (y.detach().float(), x.grad.float(), model[0].weight.grad.float(), model[0].bias.grad.float(), dict(fused.routes_report()))
    routes = res["1"][4]
    assert routes.get("train:forward 2048x3072x768") == "in_tree_bf16_gemm", routes
    assert routes.get("train:dgrad + wgrad 2048x3072x768") == "in_tree_bf16_gemm, one launch", routes      # tokens x out x in
    assert not any(k.startswith("train:") and "optimizer" not in k for k in res["0"][4])       # (the optimizer's route entry is another module's)
    # the two backward products as two launches (QT_TRAIN_DEBUG bit 512): the same bits
    monkeypatch.setenv("QT_TRAIN_GEMM", "1")
    monkeypatch.setenv("QT_TRAIN_DEBUG", "512")
    qlin.GEMM_ROUTES.clear()
    torch.manual_seed(5)
    lin = torch.nn.Linear(768, 3072).cuda().bfloat16()
    model = torch.nn.Sequential(lin)
    qt_pkg.quantize(model, qt_pkg.add_qspec_args().parse_args(["--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric",
                                                               "--error", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm",
                                                               "--quantize_backprop", "gemm", "--bf16"]))
    model.train()
    x = torch.randn(16, 128, 768, device="cuda").bfloat16().requires_grad_(True)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        x.grad = None
        y = model(x)
        (y.float() * torch.linspace(-1, 1, 3072, device="cuda")).mean().backward()
    monkeypatch.delenv("QT_TRAIN_DEBUG")
    single = dict(fused.routes_report())
    for kind, shape in (("dgrad", "2048x768x3072"), ("wgrad", "3072x768x2048")):
        assert single.get(f"train:{kind} {shape}") == "in_tree_bf16_gemm", single
    assert torch.equal(x.grad.float(), res["1"][1]) and torch.equal(model[0].weight.grad.float(), res["1"][2])
    for i, name in enumerate(("output", "input gradient", "weight gradient")):
        a, b = res["1"][i], res["0"][i]
        assert bool(((a - b).abs() <= 2.0 ** -7 * b.abs() + 2.0 ** -10 * float(b.abs().max())).all()), (name, float((a - b).abs().max()))
    assert torch.equal(res["1"][3], res["0"][3])


@pytest.mark.parametrize("n,kinds", [(2048 * 768, (1, 1, 1)), (2048 * 768, (1,)), (37 * 264, (0, 1, 1, 0)), (8 * 8, (0, 0))])
def test_grad_fanin_equals_the_fake_quantizer_launches_and_torch_adds(nv, n, kinds):
    """qt_grad_fanin_bf16: sum = (((first + y_0) + y_1) + ...) with y_i = fq_i(x_i) (kind 1) or x_i (kind 0) is bit for bit what the
    separate launches give -- qt_fake_quant_bf16 per quantized item (its amax included) and torch's bf16 adds in the same order."""
    import quantized_training as qt_pkg
    from quantized_training.fake_quantize import _launch_format
    L = nv.lib()
    dev = torch.device("cuda")
    torch.manual_seed(n + len(kinds))
    lut = qt_pkg.get_quantization_map("fp8_e5m2", dev)
    fmt = _launch_format(nv.format_for("fp8_e5m2"), lut)
    first = (torch.randn(n, device=dev) * 1e-3).bfloat16()
    xs = [(torch.randn(n, device=dev) * 10.0 ** (-3 - i)).bfloat16() for i in range(len(kinds))]
    sc = [torch.tensor([3.0e-8 * (i + 1)], dtype=torch.float32, device=dev) for i in range(len(kinds))]
    am = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in kinds]
    outs = [torch.empty_like(first) if i % 2 == 0 else None for i in range(len(kinds))]
    items = (nv.QtFaninItem * len(kinds))()
    for i, kd in enumerate(kinds):
        items[i].x_dev, items[i].fq = xs[i].data_ptr(), kd
        items[i].scale_f32_dev, items[i].amax_bits_dev = (sc[i].data_ptr(), am[i].data_ptr()) if kd else (None, None)
        items[i].out_dev = outs[i].data_ptr() if (kd and outs[i] is not None) else None
    total = torch.empty_like(first)
    nv.check(L.qt_grad_fanin_bf16(first.data_ptr(), items, len(kinds), total.data_ptr(), n, ctypes.byref(fmt), lut.data_ptr(), stream()), "qt_grad_fanin_bf16")
    ref = first
    for i, kd in enumerate(kinds):
        y = xs[i]
        if kd:
            y = torch.empty_like(xs[i])
            a = torch.zeros(1, dtype=torch.float32, device=dev)
            nv.check(L.qt_fake_quant_bf16(xs[i].data_ptr(), y.data_ptr(), n, ctypes.byref(fmt), lut.data_ptr(), sc[i].data_ptr(), a.data_ptr(), stream()), "fq")
            assert torch.equal(am[i].view(torch.int32), a.view(torch.int32)), i
            if outs[i] is not None:
                assert torch.equal(outs[i].view(torch.int16), y.view(torch.int16)), i
        ref = ref + y
    assert torch.equal(total.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("n,rows,cols,kind,pad", [(2048, 50265, 768, "words", 1), (2048, 514, 768, "positions", 1), (2048, 1, 768, "one", -1),
                                                  (1000, 300, 256, "words", 7), (3072, 64, 128, "words", -1), (16, 5, 8, "one", -1)])
def test_embedding_backward_is_torchs_bit_for_bit(nv, n, rows, cols, kind, pad):
    """qt_embedding_backward_bf16 against torch.ops.aten.embedding_dense_backward (the <= 3072-index path): the same chunked fold, so the
    same bits -- tables with few and with many duplicates per index (a position table: every index once per sequence; a one-row
    token-type table: every token), padding rows, a token count that does not fill its last chunk."""
    L = nv.lib()
    dev = torch.device("cuda")
    torch.manual_seed(n + rows)
    if kind == "words":
        ids = torch.randint(0, rows, (n,), device=dev)
        ids[::7] = ids[3]                                        # a heavy hitter
        if pad >= 0:
            ids[5::11] = pad
    elif kind == "positions":
        ids = (torch.arange(n, device=dev) % 128) + 2
        ids[::9] = pad
    else:
        ids = torch.zeros(n, dtype=torch.int64, device=dev)
    grad = torch.randn(n, cols, device=dev).bfloat16()
    want = torch.ops.aten.embedding_dense_backward(grad, ids, rows, pad, False)
    part = torch.empty_like(grad)
    got = torch.zeros(rows, cols, dtype=torch.bfloat16, device=dev)
    nv.check(L.qt_embedding_backward_bf16(grad.data_ptr(), ids.data_ptr(), n, cols, pad, rows, part.data_ptr(), got.data_ptr(), stream()), "qt_embedding_backward_bf16")
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), int((got.view(torch.int16) != want.view(torch.int16)).sum())


def test_lt_fp8_gemm_algorithm_is_a_committed_table_and_runs_are_bit_equal_across_processes(nv):
    """The library FP8 GEMM runs the suggestion the committed table names (fused._LT_ALGO_TABLE; nothing is timed in the product):
    the choice is reported (routes_report), two fresh processes produce bit-identical outputs for a tabled and an untabled shape,
    and another suggestion of the library gives the same product within the accumulation bound."""
    import subprocess
    import sys
    from quantized_training import fused
    code = r"""
import sys, hashlib, torch
sys.path.insert(0, %r)
from quantized_training import fused
torch.manual_seed(0)
out = []
for (M, N, K, bias) in ((6144, 768, 3072, True), (777, 512, 640, False)):
    a = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.float8_e4m3fn)
    bv = torch.randn(N, device="cuda").bfloat16() if bias else None
    y = fused.lt_fp8_gemm(a, b, bv)
    out.append(hashlib.sha256(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest())
print("HASH", " ".join(out), fused.routes_report())
""" % os.path.join(ROOT, "quantized-training_amd")
    lines = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][0])
    assert lines[0] == lines[1]
    assert "'lt:1x6144x768x3072+bias': 3" in lines[0] and "'lt:1x777x512x640': 0" in lines[0]
    torch.manual_seed(0)
    a = torch.randn(6144, 3072, device="cuda").to(torch.float8_e4m3fn)
    b = (torch.randn(768, 3072, device="cuda") * 0.05).to(torch.float8_e4m3fn)
    y3 = fused.lt_fp8_gemm(a, b, None)
    os.environ["QT_LT_ALGO"] = "0"
    try:
        y0 = fused.lt_fp8_gemm(a, b, None)
    finally:
        del os.environ["QT_LT_ALGO"]
    bound = (a.float().abs() @ b.float().abs().t())
    assert bool(((y3.float() - y0.float()).abs() <= y0.float().abs() * 2.0 ** -7 + bound * 2.0 ** -14).all())
    # the table's indices belong to one build of the library: under any other the first suggestion runs, and the report says so
    assert nv.lib().qt_fp8_gemm_library_version() == fused._LT_ALGO_TABLE_LIBRARY, nv.lib().qt_fp8_gemm_library_version()
    saved = (fused._LT_ALGO_TABLE_LIBRARY, dict(fused._LT_LIBRARY), dict(fused.LT_ALGOS))
    try:
        fused._LT_ALGO_TABLE_LIBRARY, fused._LT_LIBRARY["version"] = saved[0] + 1, None
        fused.LT_ALGOS.clear()
        assert fused.lt_algo_index(1, 6144, 768, 3072, False, True) == 0
        assert "lt:library_version" in fused.LT_ALGOS and fused.LT_ALGOS["lt:1x6144x768x3072+bias"] == 0
    finally:
        fused._LT_ALGO_TABLE_LIBRARY = saved[0]
        fused._LT_LIBRARY.update(saved[1])
        fused.LT_ALGOS.clear()
        fused.LT_ALGOS.update(saved[2])


def test_splitk_scratch_is_per_stream_and_per_capture(nv):
    """fused.splitk_scratch: launches that share a split-K workspace must be ONE ordered sequence, so every eager stream and every
    stream capture gets buffers of its own (the capture's from the graph's pool), and the same stream gets the same ones again."""
    from quantized_training import fused
    dev = torch.device("cuda", torch.cuda.current_device())
    a_ws, a_tk = fused.splitk_scratch("test", 1 << 16, 64, dev)
    b_ws, b_tk = fused.splitk_scratch("test", 1 << 16, 64, dev)
    assert a_ws.data_ptr() == b_ws.data_ptr() and a_tk.data_ptr() == b_tk.data_ptr()
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        c_ws, c_tk = fused.splitk_scratch("test", 1 << 16, 64, dev)
    assert c_ws.data_ptr() != a_ws.data_ptr() and c_tk.data_ptr() != a_tk.data_ptr()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        d_ws, d_tk = fused.splitk_scratch("test", 1 << 16, 64, dev)
        d2_ws, _ = fused.splitk_scratch("test", 1 << 16, 64, dev)
        d_tk.add_(0)
    assert d_ws.data_ptr() == d2_ws.data_ptr() and d_ws.data_ptr() not in (a_ws.data_ptr(), c_ws.data_ptr())
    g.replay()
    torch.cuda.synchronize()
    assert not bool(d_tk.any()) and not bool(a_tk.any())
    e_ws, _ = fused.splitk_scratch("test", 1 << 16, 64, dev)           # eager again: the stream's own buffers, not the graph's
    assert e_ws.data_ptr() == a_ws.data_ptr()


@pytest.mark.parametrize("pair", [False, True])
def test_linear_fq8_wide_tiles_redo_overflowing_weights(nv, pair):
    """The widest tiles (twelve column groups: gate / up, q / k / v -- the two-register-set loop) on ALL 65 536 bf16 weight patterns:
    identity activation, N = 11008 at M = K = 1024.  Overflowing weights saturate (fp8.py:32), rows holding +-Inf / NaN come out
    all-NaN: the redo path of the kernels the headline window spends most of its time in, checked against the ORACLE's value map
    (test_mlp_fq8_equals_two_gemms_and_silu_mul compares two kernels that share that path).  pair: qt_mlp_fq8_bf16 whose gate weight
    holds every finite pattern, against the oracle-checked gate product + qt_silu_mul_fq8_bf16."""
    K = M = 1024
    N = 11008
    torch.manual_seed(3)
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    W.view(torch.int16)[:64] = torch.arange(65536, device="cuda", dtype=torch.int32).to(torch.int16).view(64, K)
    x8 = _codes_of(nv, torch.eye(K, device="cuda").bfloat16(), "e4m3")
    qmap = o.get_quantization_map("e4m3")

    def check_gate(Wt, n_nan_rows):
        exp = o.canon_nan16(o.vmap_bf16(host_u16(Wt.view(torch.int16)), qmap)).reshape(N, K)
        y = _linear_fq8(nv, x8, "e4m3", [Wt], "e4m3")
        got = o.canon_nan16(host_u16(y.t().contiguous().view(torch.int16)))
        nan_rows = (exp == 0x7FC0).any(axis=1)
        assert nan_rows.sum() == n_nan_rows
        same = (got == exp) | (((got | exp) & 0x7FFF) == 0)
        assert same[~nan_rows].all()
        assert (got[nan_rows] == 0x7FC0).all()
        return y

    Wf = W.clone()
    Wf[~torch.isfinite(Wf.float())] = 0
    g = check_gate(Wf, 0)
    if not pair:
        check_gate(W, 2)                                        # the rows holding +Inf / NaNs and -Inf / NaNs
        return
    Wu = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    Wu[7, 9] = 3.0e4                                            # beyond E4M3's range: saturates to 448
    u = _linear_fq8(nv, x8, "e4m3", [Wu], "e4m3")
    assert float(u[9, 7]) == 448.0
    fmt = nv.format_for("e4m3")
    want = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    want8 = torch.empty((M, N), dtype=torch.uint8, device="cuda")
    nv.check(nv.lib().qt_silu_mul_fq8_bf16(g.data_ptr(), u.data_ptr(), want.data_ptr(), want8.data_ptr(), M, N, N, N, ctypes.byref(fmt),
                                           stream()), "qt_silu_mul_fq8_bf16")
    h = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    h8 = torch.empty((M, N), dtype=torch.uint8, device="cuda")
    nv.check(nv.lib().qt_mlp_fq8_bf16(x8.data_ptr(), 0, Wf.data_ptr(), Wu.data_ptr(), None, None, N, 0, h.data_ptr(), h8.data_ptr(),
                                      ctypes.byref(fmt), M, K, stream()), "qt_mlp_fq8_bf16")
    assert torch.equal(h.view(torch.int16), want.view(torch.int16)) and torch.equal(h8, want8)


def _fq8_plan(nv, M, N, K, pair=False):
    v = [ctypes.c_int(0) for _ in range(5)]
    nv.check(nv.lib().qt_linear_fq8_plan(M, N, K, int(pair), *[ctypes.byref(x) for x in v]), "qt_linear_fq8_plan")
    return tuple(x.value for x in v)                   # tiles_m, tiles_n, groups_lo, groups_hi, variant


def _fq8_fused_shapes():
    """Every problem shape a DEFAULT route hands to the fused FP8 GEMM: the committed route table's fused entries (fused._FQ8_TABLE:
    BASELINE's configs) -- enumerated, not hand-picked -- plus shapes that reach the remaining (variant, tile width) classes of the
    planner (two weight pieces with an odd k-tile count: variant 2; four pieces: variant 4)."""
    from quantized_training import fused                       # (conftest.py put the package on sys.path)
    shapes = sorted(k for k, v in fused._FQ8_TABLE.items() if v)
    return shapes + [(1024, 4096, 4224), (512, 4096, 4096), (1024, 8192, 1152)]


@pytest.mark.parametrize("wdtype", ["e4m3", "e5m2"])
@pytest.mark.parametrize("shape", _fq8_fused_shapes(), ids=lambda s: "x".join(map(str, s)))
def test_linear_fq8_every_planned_tile_variant_on_all_weight_patterns(nv, shape, wdtype):
    """fp8.py:10-67 through qt_linear_fq8_bf16 for EVERY (kernel variant, tile width) the planner returns for the shapes of the route
    table (qt_linear_fq8_plan: two-k-tile narrow tiles, one-k-tile tiles with 2 / 4 / 6 weight pieces, the two-register-set loop) at
    the table's own M and N -- the planner's tile cut depends on those and on K only through K % 256, which the test's K keeps.  A
    wrong-result bug lived for two rounds in one variant's redo vote because the all-pattern tests ran other tile widths.
    Identity activation: y[m][n] = fq(W)[n][m], compared bit for bit with the ORACLE's value map.
      fast:  every weight row cycles through the bf16 patterns whose result is finite in the format -- no tile takes the redo path,
             every tile position converts every such pattern with the hardware conversion;
      redo:  every row cycles through ALL 65 536 patterns, non-finite ones zeroed -- every tile overflows and is redone with the
             closed form (saturation to the format maximum);
      raw:   all patterns as they are -- rows holding +-Inf / NaN must come out all-NaN (0 * NaN), the others exact."""
    M, N, Kreal = shape
    Kt = min(512, M) if Kreal % 256 == 0 else min(384, M)          # keeps K % 256 (the narrow-tile variant's condition)
    Kt = max(Kt - Kt % 128, 128)
    plan = _fq8_plan(nv, M, N, Kreal)
    assert plan == _fq8_plan(nv, M, N, Kt), (plan, _fq8_plan(nv, M, N, Kt))      # the test problem is cut like the real one
    qmap = o.get_quantization_map(wdtype)
    vals = o.bf16_to_f32(qmap)
    pats = np.arange(65536, dtype=np.uint16)
    raw = o.bf16_to_f32(pats)
    # patterns the HARDWARE conversion rounds to a finite code: below the midpoint between the format maximum and the next grid point
    inrange = pats[np.isfinite(raw) & (np.abs(raw) < (464.0 if wdtype == "e4m3" else 61440.0))]
    x8 = _codes_of(nv, torch.eye(Kt, device="cuda").bfloat16()[:M] if M < Kt else torch.cat(
        [torch.eye(Kt, device="cuda"), torch.zeros(M - Kt, Kt, device="cuda")]).bfloat16(), "e4m3")
    rows = min(M, Kt)

    def run(name, bits):
        W = torch.from_numpy(np.resize(bits, N * Kt).astype(np.uint16).view(np.int16).reshape(N, Kt).copy()).cuda().view(torch.bfloat16)
        exp = o.canon_nan16(o.vmap_bf16(np.resize(bits, N * Kt).astype(np.uint16), qmap)).reshape(N, Kt)[:, :rows]
        y = _linear_fq8(nv, x8, "e4m3", [W], wdtype)
        got = o.canon_nan16(host_u16(y[:rows].t().contiguous().view(torch.int16)))
        nan_rows = (o.canon_nan16(o.vmap_bf16(np.resize(bits, N * Kt).astype(np.uint16), qmap)).reshape(N, Kt) == 0x7FC0).any(axis=1)
        same = (got == exp) | (((got | exp) & 0x7FFF) == 0)         # the sign of a zero is not part of a product
        assert same[~nan_rows].all(), (name, plan, int((~same[~nan_rows]).sum()), np.argwhere(~same & ~nan_rows[:, None])[:4].tolist())
        assert (got[nan_rows] == 0x7FC0).all(), (name, plan)
        return int(nan_rows.sum())

    assert run("fast", inrange) == 0
    sane = pats.copy()
    sane[~np.isfinite(raw)] = 0
    assert run("redo", sane) == 0
    assert run("raw", pats) > 0
    print(f"[fq8 plan] {M}x{N}x{Kreal} {wdtype}: tiles {plan[0]}x{plan[1]}, {plan[2]}..{plan[3]} groups, variant {plan[4]}")


def test_linear_fq8_planned_variants_are_all_covered(nv):
    """The shapes of the test above reach every class the planner can return: each kernel variant (0, 2, 4, 6), and for the one-k-tile
    variants both the narrowest and the widest tile it serves; every fused entry of the route table is among them."""
    seen = {}
    for (M, N, K) in _fq8_fused_shapes():
        tm, tn, lo, hi, var = _fq8_plan(nv, M, N, K)
        seen.setdefault(var, set()).update((lo, hi))
    assert set(seen) == {0, 2, 4, 6}, seen
    assert max(seen[6]) == 12 and min(seen[6]) >= 9 and max(seen[4]) <= 8 and max(seen[0]) <= 4, seen


@pytest.mark.parametrize("M,N,K", [(1024, 11008, 4096), (512, 11008, 4096), (1024, 4096, 4096), (256, 1536, 512)])
def test_mlp_fq8_planned_pair_tiles_on_all_weight_patterns(nv, M, N, K):
    """qt_mlp_fq8_bf16 (pair mode) at the shapes of fused._MLP_TABLE and at narrower pair tiles (eight / four groups): the gate AND the
    up weight cycle through all 65 536 bf16 patterns (non-finite ones zeroed: every tile is redone) resp. the in-range ones (no tile
    is), identity activation -- the one launch must equal, bit for bit, the oracle-checked separate launches: two
    qt_linear_fq8_bf16 products (the test above pins them to the oracle) + qt_silu_mul_fq8_bf16."""
    from quantized_training import fused
    assert all(k in [(1024, 11008, 4096), (512, 11008, 4096)] for k in fused._MLP_TABLE)
    Kt = min(512, M)
    plan = _fq8_plan(nv, M, N, K, pair=True)
    assert plan == _fq8_plan(nv, M, N, Kt, pair=True)
    pats = np.arange(65536, dtype=np.uint16)
    raw = o.bf16_to_f32(pats)
    sane = pats.copy()
    sane[~np.isfinite(raw)] = 0
    inrange = pats[np.isfinite(raw) & (np.abs(raw) < 464.0)]
    x8 = _codes_of(nv, torch.cat([torch.eye(Kt, device="cuda"), torch.zeros(max(M - Kt, 0), Kt, device="cuda")]).bfloat16()[:M], "e4m3")
    fmt = nv.format_for("e4m3")
    for name, gbits, ubits in (("redo", sane, sane[::-1]), ("fast", inrange, inrange[::-1]), ("mixed", sane, inrange)):
        Wg = torch.from_numpy(np.resize(gbits, N * Kt).astype(np.uint16).view(np.int16).reshape(N, Kt).copy()).cuda().view(torch.bfloat16)
        Wu = torch.from_numpy(np.resize(ubits, N * Kt).astype(np.uint16).view(np.int16).reshape(N, Kt).copy()).cuda().view(torch.bfloat16)
        g = _linear_fq8(nv, x8, "e4m3", [Wg], "e4m3")
        u = _linear_fq8(nv, x8, "e4m3", [Wu], "e4m3")
        want = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        want8 = torch.empty((M, N), dtype=torch.uint8, device="cuda")
        nv.check(nv.lib().qt_silu_mul_fq8_bf16(g.data_ptr(), u.data_ptr(), want.data_ptr(), want8.data_ptr(), M, N, N, N, ctypes.byref(fmt),
                                               stream()), "qt_silu_mul_fq8_bf16")
        h = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        h8 = torch.empty((M, N), dtype=torch.uint8, device="cuda")
        nv.check(nv.lib().qt_mlp_fq8_bf16(x8.data_ptr(), 0, Wg.data_ptr(), Wu.data_ptr(), None, None, N, 0, h.data_ptr(), h8.data_ptr(),
                                          ctypes.byref(fmt), M, Kt, stream()), "qt_mlp_fq8_bf16")
        assert torch.equal(o_canon(h), o_canon(want)) and torch.equal(h8, want8), (name, plan)


def o_canon(t):
    """bf16 tensor as int16 bits with every NaN rewritten to one pattern (the payload is not part of the contract)."""
    b = t.view(torch.int16).clone()
    b[torch.isnan(t.float())] = 0x7FC0
    return b


# ---- qt_linear_fqt_bf16: bf16 GEMM with ANY value map applied to the weights in its operand path --------------------------
class _Fqt:
    """Device-side tables of one dtype: the value map, its row form (qt_build_rowparams) and the elementwise format."""
    _cache = {}

    def __new__(cls, nv, dtype):
        hit = cls._cache.get(dtype)
        if hit is None:
            hit = object.__new__(cls)
            hit.map_host = nv.build_map_u16(dtype)
            hit.rp = nv.build_rowparams(hit.map_host)
            rows = np.ctypeslib.as_array(hit.rp.row).reshape(2048).astype(np.uint32)
            hit.rows = torch.from_numpy(rows.view(np.int32).copy()).cuda()
            hit.map = torch.from_numpy(hit.map_host.view(np.int16).copy()).cuda()
            cls._cache[dtype] = hit
        return hit


_FQT_WS = {}


def _fqt_plan(nv, M, N, K):
    ks, wb, nt = ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    nv.check(nv.lib().qt_linear_fqt_plan(M, N, K, ctypes.byref(ks), ctypes.byref(wb), ctypes.byref(nt)), "qt_linear_fqt_plan")
    return ks.value, wb.value, nt.value


def _linear_fqt(nv, x, ws, dtype, biases=None, split=True):
    """split=True: qt_linear_fqt_ws_bf16 with the workspace qt_linear_fqt_plan asks for (split-K where the plan says so);
    split=False: qt_linear_fqt_bf16, the entry point that never splits K."""
    f = _Fqt(nv, dtype)
    M, K = x.shape
    n = len(ws)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in (biases or [None] * n)])
    ns = (ctypes.c_int * n)(*[w.shape[0] for w in ws])
    N = sum(w.shape[0] for w in ws)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    if not split:
        nv.check(nv.lib().qt_linear_fqt_bf16(x.data_ptr(), wp, bp, ns, n, f.rows.data_ptr(), f.rp.signed_rows, f.rp.sign_mask, f.map.data_ptr(),
                                             y.data_ptr(), M, K, stream()), "qt_linear_fqt_bf16")
        return y
    ks, wb, nt = _fqt_plan(nv, M, N, K)
    if "ws" not in _FQT_WS or _FQT_WS["ws"].numel() * 4 < wb or _FQT_WS["tickets"].numel() < nt:
        _FQT_WS["ws"] = torch.empty((max(wb, 16) // 4,), dtype=torch.float32, device="cuda")
        _FQT_WS["tickets"] = torch.zeros((max(nt, 1),), dtype=torch.int32, device="cuda")
    nv.check(nv.lib().qt_linear_fqt_ws_bf16(x.data_ptr(), wp, bp, ns, n, f.rows.data_ptr(), f.rp.signed_rows, f.rp.sign_mask, f.map.data_ptr(),
                                            y.data_ptr(), M, K, _FQT_WS["ws"].data_ptr(), _FQT_WS["ws"].numel() * 4,
                                            _FQT_WS["tickets"].data_ptr(), _FQT_WS["tickets"].numel(), stream()), "qt_linear_fqt_ws_bf16")
    assert not bool(_FQT_WS["tickets"].any())                  # every launch leaves its tickets zero
    return y


@pytest.mark.parametrize("wdtype", ["posit8_1", "posit8_2", "int8", "fp6_e3m2", "fp4_e2m1", "posit8_0", "uint8", "fp8_e5m3", "e4m3"])
@pytest.mark.parametrize("M,K", [(512, 512), (300, 512), (1536, 1536)])
def test_linear_fqt_weight_values_are_the_value_map(nv, wdtype, M, K):
    """Identity activation: y[m][n] = fq(W)[n][m] -- one product per output, so the kernel's in-flight conversion of W (the row form
    of the map, and the redo path for the rows it flags) is compared bit for bit with the oracle's value map on ALL 65 536 bf16
    patterns (the first 65 536 elements of W); rows holding +-Inf / NaN must come out all-NaN (0 * NaN).  M = 512 takes the 512-row
    tiles, M = 300 (ragged) the 256-row ones, M = K = 1536 the split-K path (two workgroups per tile, 24 k steps each: partial sums
    through the workspace, and the redo of a flagged tile decided by whichever workgroup draws the tile's last ticket)."""
    torch.manual_seed(1)
    W = (torch.randn(400, K, device="cuda") * 3).bfloat16()
    W.view(torch.int16).view(-1)[:65536] = torch.arange(65536, device="cuda", dtype=torch.int32).to(torch.int16)
    eye = torch.eye(K, device="cuda").bfloat16()[:M] if M <= K else None
    if K == 1536:
        assert _fqt_plan(nv, M, 400, K)[0] == 2
    qmap = o.get_quantization_map(wdtype)
    bias = torch.randn(400, device="cuda").bfloat16()
    for sanitize in (True, False):
        Wt = W.clone()
        if sanitize:
            Wt[~torch.isfinite(Wt.float())] = 0
        exp = o.canon_nan16(o.vmap_bf16(host_u16(Wt.view(torch.int16)), qmap)).reshape(400, K)[:, :M]
        got = o.canon_nan16(host_u16(_linear_fqt(nv, eye, [Wt], wdtype).t().contiguous().view(torch.int16)))
        nan_rows = (o.canon_nan16(o.vmap_bf16(host_u16(Wt.view(torch.int16)), qmap)).reshape(400, K) == 0x7FC0).any(axis=1)
        same = (got == exp) | (((got | exp) & 0x7FFF) == 0)                 # the sign of a zero is not part of a product
        assert same[~nan_rows].all(), (wdtype, int((~same[~nan_rows]).sum()))
        assert (got[nan_rows] == 0x7FC0).all()
        gotb = _linear_fqt(nv, eye, [Wt], wdtype, [bias]).t().contiguous()
        expb = (torch.from_numpy(o.bf16_to_f32(exp)).cuda() + bias.float()[:, None]).bfloat16()
        keep = ~torch.from_numpy(nan_rows).cuda()
        assert torch.equal(gotb[keep].view(torch.int16), expb[keep].view(torch.int16))


@pytest.mark.parametrize("M,Ns,K", [(1024, [4096], 1024), (1024, [176], 256), (300, [48, 64, 16], 384), (1, [16], 32), (777, [2048, 512, 512], 512),
                                    (257, [208, 4096 - 208], 128), (512, [128], 96),
                                    # BASELINE.json configs[3], full LLaMA-2-13B sizes: gate / up, down, o, q / k / v, lm head
                                    (1024, [13824], 5120), (1024, [5120], 13824), (1024, [5120], 5120), (1024, [5120, 5120, 5120], 5120),
                                    (1024, [32000], 5120),
                                    # BERT-base (configs[0] / [1] shapes)
                                    (6144, [768], 768), (6144, [3072], 768), (6144, [768], 3072),
                                    # split-K (round 4): 2, 4 and 3 workgroups per tile, ragged rows, two weights, uneven k ranges
                                    (1024, [1024], 2048), (1024, [2048], 3072), (777, [512, 512], 2400), (520, [1024], 3968),
                                    # LLaMA-2-7B o / down under a table format
                                    (1024, [4096], 4096), (1024, [4096], 11008)])
@pytest.mark.parametrize("dtype", ["posit8_2", "int8"])
def test_linear_fqt_vs_fp64_product_of_the_quantized_operands(nv, M, Ns, K, dtype):
    """Ragged M, both tile heights, column tiles spanning two weights, several weights per launch, bias: against the fp64 product of
    the oracle-quantized operands.  Tolerance: one bf16 rounding of the result (2^-8 relative) plus fp32 accumulation of exact
    products (measured <= 2^-20 sum |a||b|; bound used: 2^-18)."""
    torch.manual_seed(0)
    qm = o.get_quantization_map(dtype)

    def fq(t):
        return torch.from_numpy(o.vmap_bf16(host_u16(t.view(torch.int16)), qm).view(np.int16)).cuda().view(torch.bfloat16)

    x = fq((torch.randn(M, K, device="cuda") * (1.0 if dtype != "int8" else 20.0)).bfloat16())
    ws = [(torch.randn(n, K, device="cuda") * (0.05 if dtype != "int8" else 3.0)).bfloat16() for n in Ns]
    bs = [torch.randn(n, device="cuda").bfloat16() if i % 2 == 0 else None for i, n in enumerate(Ns)]
    y16 = _linear_fqt(nv, x, ws, dtype, bs)
    # split-K adds the partial sums in split order, whoever arrives last: run to run bit-identical
    assert torch.equal(y16.view(torch.int16), _linear_fqt(nv, x, ws, dtype, bs).view(torch.int16))
    if _fqt_plan(nv, M, sum(Ns), K)[0] > 1:
        # against the unsplit entry point: the same exact products, fp32 sums in another order
        y1_16 = _linear_fqt(nv, x, ws, dtype, bs, split=False)
        y1 = y1_16.double()
        assert bool(((y16.double() - y1).abs() <= y1.abs() * 2.0 ** -7 + (xa_abs := x.double().abs().sum(dim=1, keepdim=True)) * 2.0 ** -20).all())
        if dtype == "int8":
            assert torch.equal(y16.view(torch.int16), y1_16.view(torch.int16))               # integer products: every order is exact
    y = y16.double()
    xa = x.double()
    wa = torch.cat([fq(w).double() for w in ws])
    bias = torch.cat([b.double() if b is not None else torch.zeros(n, device="cuda", dtype=torch.float64) for b, n in zip(bs, Ns)])
    ref = xa @ wa.t() + bias
    tol = ref.abs() * 2.0 ** -8 + (xa.abs() @ wa.abs().t()) * 2.0 ** -18 + 1e-30
    assert bool(((y - ref).abs() <= tol).all()), float(((y - ref).abs() / tol).max())


# every problem shape the value-map GEMM's route rule (fused.fqt_route_is_fused) sends to it in BASELINE's configs: configs[3] (LLaMA-2-13B,
# posit(8,2)) and a 7B-shaped model under a table format -- (M, [n per weight], K)
_FQT_ROUTE_SHAPES = [(1024, [5120, 5120, 5120], 5120), (1024, [13824, 13824], 5120), (1024, [5120], 13824), (1024, [32000], 5120), (1024, [5120], 5120),
                     (1024, [4096, 4096, 4096], 4096), (1024, [11008, 11008], 4096), (1024, [4096], 11008), (1024, [4096], 4096), (1024, [32000], 4096)]


@pytest.mark.parametrize("wdtype", ["posit8_2", "int8"])
@pytest.mark.parametrize("M,Ns,K", _FQT_ROUTE_SHAPES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_linear_fqt_every_routed_shape_on_all_weight_patterns(nv, M, Ns, K, wdtype):
    """qt_linear_fqt_ws_bf16 at the FULL shapes the route rule hands to it (the shapes are enumerated, the rule decides which of them run
    here: the others are skipped) -- so the tile cut and the split-K plan are the ones the window runs (three workgroups per tile at the
    13B down projection, four at the 7B one, none for the wide outputs; sibling launches of two and three weights).  Identity activation
    placed at a k offset inside the LAST k range (so a split's partial sums travel through the workspace and the tile's redo is decided
    by whichever workgroup draws the last ticket): y[m][n] = fq(W)[n][k0 + m], compared bit for bit with the ORACLE's value map for
    weights that cycle through all 65 536 bf16 patterns (non-finite ones zeroed, then raw: rows holding +-Inf / NaN must be all-NaN)."""
    from quantized_training import fused
    if not fused.fqt_route_is_fused(M, Ns, K, torch.device("cuda")):
        pytest.skip("the route rule keeps this shape on the weight pass + library GEMM")
    N = sum(Ns)
    ks = _fqt_plan(nv, M, N, K)[0]
    k0 = K - M if K > M else 0
    rows = min(M, K)
    x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
    x[torch.arange(rows), k0 + torch.arange(rows)] = 1.0
    qmap = o.get_quantization_map(wdtype)
    pats = np.arange(65536, dtype=np.uint16)
    raw = o.bf16_to_f32(pats)
    sane = pats.copy()
    sane[~np.isfinite(raw)] = 0
    torch.manual_seed(K)
    for name, bits in (("sane", sane), ("raw", pats)):
        ws, exp_parts = [], []
        for i, n in enumerate(Ns):
            W = (torch.randn(n, K, device="cuda") * 0.05).bfloat16()
            blk = np.resize(np.roll(bits, 977 * i), n * rows).astype(np.uint16).reshape(n, rows)
            W[:, k0:k0 + rows] = torch.from_numpy(blk.view(np.int16).copy()).cuda().view(torch.bfloat16)
            ws.append(W)
            exp_parts.append(o.canon_nan16(o.vmap_bf16(blk, qmap)))
        exp = np.concatenate(exp_parts)                                  # [N][rows]
        y = _linear_fqt(nv, x, ws, wdtype)
        got = o.canon_nan16(host_u16(y[:rows].t().contiguous().view(torch.int16)))
        nan_rows = (exp == 0x7FC0).any(axis=1)
        same = (got == exp) | (((got | exp) & 0x7FFF) == 0)
        assert same[~nan_rows].all(), (name, ks, int((~same[~nan_rows]).sum()), np.argwhere(~same & ~nan_rows[:, None])[:4].tolist())
        assert (got[nan_rows] == 0x7FC0).all(), (name, ks)
        assert (nan_rows.sum() > 0) == (name == "raw" and bool((o.canon_nan16(qmap) == 0x7FC0).any()))
    print(f"[fqt route] {M}x{Ns}x{K} {wdtype}: split-K {ks}")


def test_linear_fqt_split_k_plan_and_workspace_contract(nv):
    """qt_linear_fqt_plan: no split for wide outputs or short K; 3 workgroups per tile at the 13B down projection, 4 at the 7B one;
    qt_linear_fqt_ws_bf16 refuses a workspace smaller than the plan's instead of writing past it."""
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("plan figures are for 256 CUs")
    assert _fqt_plan(nv, 1024, 15360, 5120) == (1, 0, 0)
    assert _fqt_plan(nv, 1024, 5120, 512)[0] == 1                       # 16 k steps: too short to split
    ks, wb, nt = _fqt_plan(nv, 1024, 5120, 13824)
    assert (ks, nt) == (3, 80) and wb == 80 * 3 * 8 * 32 * 1024
    assert _fqt_plan(nv, 1024, 4096, 11008)[0] == 4
    f = _Fqt(nv, "posit8_1")
    x = torch.zeros(1024, 2048, device="cuda", dtype=torch.bfloat16)
    W = torch.zeros(1024, 2048, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(1024, 1024, device="cuda", dtype=torch.bfloat16)
    ks, wb, nt = _fqt_plan(nv, 1024, 1024, 2048)
    assert ks == 2
    ws = torch.empty(wb // 4, dtype=torch.float32, device="cuda")
    tk = torch.zeros(nt, dtype=torch.int32, device="cuda")
    wp = (ctypes.c_void_p * 1)(W.data_ptr())
    ns = (ctypes.c_int * 1)(1024)
    args = (x.data_ptr(), wp, None, ns, 1, f.rows.data_ptr(), 0, f.rp.sign_mask, f.map.data_ptr(), y.data_ptr(), 1024, 2048)
    assert nv.lib().qt_linear_fqt_ws_bf16(*args, ws.data_ptr(), wb - 4, tk.data_ptr(), nt, stream()) == nv.QT_ERR_BAD_ARG
    assert nv.lib().qt_linear_fqt_ws_bf16(*args, ws.data_ptr(), wb, tk.data_ptr(), nt - 1, stream()) == nv.QT_ERR_BAD_ARG
    assert nv.lib().qt_linear_fqt_ws_bf16(*args, None, 0, None, 0, stream()) == nv.QT_ERR_BAD_ARG
    assert nv.lib().qt_linear_fqt_ws_bf16(*args, ws.data_ptr(), wb, tk.data_ptr(), nt, stream()) == 0
    torch.cuda.synchronize()
    assert not bool(tk.any()) and not bool(y.any())


def test_linear_fqt_rejects_what_it_does_not_take(nv):
    f = _Fqt(nv, "posit8_1")
    x = torch.zeros(64, 48, device="cuda", dtype=torch.bfloat16)        # K % 32 != 0
    W = torch.zeros(32, 48, device="cuda", dtype=torch.bfloat16)
    wp = (ctypes.c_void_p * 1)(W.data_ptr())
    ns = (ctypes.c_int * 1)(32)
    y = torch.empty(64, 32, device="cuda", dtype=torch.bfloat16)
    rc = nv.lib().qt_linear_fqt_bf16(x.data_ptr(), wp, None, ns, 1, f.rows.data_ptr(), 0, f.rp.sign_mask, f.map.data_ptr(), y.data_ptr(), 64, 48, stream())
    assert rc == nv.QT_ERR_BAD_ARG
    ns = (ctypes.c_int * 1)(24)                                          # N % 16 != 0
    rc = nv.lib().qt_linear_fqt_bf16(x.data_ptr(), wp, None, ns, 1, f.rows.data_ptr(), 0, f.rp.sign_mask, f.map.data_ptr(), y.data_ptr(), 64, 64, stream())
    assert rc == nv.QT_ERR_BAD_ARG


@pytest.mark.parametrize("dtype", ["posit8_2", "fp4_e2m1"])
def test_qat_linear_takes_the_fused_value_map_gemm(nv, dtype, monkeypatch):
    """modules/qat/linear.py:40-41 through fused.fqt_linear_or_none: same result as the weight pass + library GEMM up to the
    accumulation order (both are fp32 sums of the same exact products), one fake-quant call counted for the weight, and the fixed
    routing rule is what QT_FQT_GEMM=auto follows."""
    import quantized_training as qt
    from quantized_training import fused
    from quantized_training.fake_quantize import STATS
    from quantized_training.modules.qat.linear import Linear as QATLinear
    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 1024, bias=True).cuda().bfloat16()
    lin.qconfig = qt.QConfig(activation=None, weight=lambda **kw: qt.FusedAmaxObsFakeQuantize(dtype=dtype, **{k: v for k, v in kw.items() if k == "device"}),
                             error=None)
    q = QATLinear.from_float(lin).cuda()
    x = (torch.randn(4, 128, 512, device="cuda")).bfloat16()
    with torch.no_grad():
        monkeypatch.setenv("QT_FQT_GEMM", "0")
        ref = q(x)
        monkeypatch.setenv("QT_FQT_GEMM", "1")
        before = STATS.elements
        got = q(x)
        assert STATS.elements - before == q.weight.numel()
    assert got.shape == ref.shape
    err = (got.float() - ref.float()).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.float().abs().max()) + 1e-6
    # the routing rule (a table of shapes, never a timing race) on the 256-CU part, for the five Linear shapes of BASELINE configs[3]
    monkeypatch.setenv("QT_FQT_GEMM", "auto")
    if torch.cuda.get_device_properties(x.device).multi_processor_count == 256:
        assert fused.fqt_route_is_fused(1024, [5120, 5120, 5120], 5120, x.device) is True       # q / k / v in one launch
        assert fused.fqt_route_is_fused(1024, [15360], 5120, x.device) is True
        assert fused.fqt_route_is_fused(1024, [32000], 5120, x.device) is True                  # lm head
        assert fused.fqt_route_is_fused(1024, [5120], 13824, x.device) is True                  # down: split-K, 3 x 144 k steps
        assert fused.fqt_plan(1024, 5120, 13824)[0] == 3
        assert fused.fqt_route_is_fused(1024, [13824], 5120, x.device) is True                  # gate / up: ties alone, wins inside the window
        assert fused.fqt_route_is_fused(1024, [5120], 5120, x.device) is False                  # o: 53 k steps per split do not pay
        assert fused.fqt_route_is_fused(256, [15360], 5120, x.device) is False
    assert isinstance(fused.fqt_route_is_fused(1024, [4096], 11008, x.device), bool)


@pytest.mark.parametrize("M,N,K", [(1024, 11008, 512), (300, 96, 256), (520, 2064, 384), (1, 16, 128), (64, 4096, 1024),
                                   (1024, 11008, 4096)])
@pytest.mark.parametrize("xdtype,wdtype,odtype", [("e4m3", "e4m3", "e4m3"), ("e4m3", "e5m2", "e5m2")])
def test_mlp_fq8_equals_two_gemms_and_silu_mul(nv, M, N, K, xdtype, wdtype, odtype):
    """qt_mlp_fq8_bf16 = gate GEMM, up GEMM (qt_linear_fq8_bf16: same tiles of the matrix instruction in the same k order, so the
    same fp32 sums), SiLU * up and the consumer's fake-quantizer (qt_silu_mul_fq8_bf16) -- values and FP8 codes bit for bit, with
    bias, ragged M, and weights beyond the format's range / non-finite ones (the kernel's redo path)."""
    L = nv.lib()
    torch.manual_seed(3)
    x = torch.randn(M, K, device="cuda").bfloat16()
    wg = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    wu = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bg, bu = torch.randn(N, device="cuda").bfloat16(), None
    x8 = _codes_of(nv, x, xdtype)
    fo = nv.format_for(odtype)
    for poison in (False, True):
        if poison:                                   # a tile with an overflowing and an infinite weight: the exact redo path
            wg[min(3, N - 1), 5] = 30000.0 if wdtype == "e4m3" else 1e30
            wu[N // 2, 7] = float("inf")
        yg = _linear_fq8(nv, x8, xdtype, [wg], wdtype, [bg])
        yu = _linear_fq8(nv, x8, xdtype, [wu], wdtype, [bu])
        want = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        want8 = torch.empty(M, N, dtype=torch.uint8, device="cuda")
        nv.check(L.qt_silu_mul_fq8_bf16(yg.data_ptr(), yu.data_ptr(), want.data_ptr(), want8.data_ptr(), M, N, N, N, ctypes.byref(fo),
                                        stream()), "qt_silu_mul_fq8_bf16")
        h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        h8 = torch.full((M, N), 0x55, dtype=torch.uint8, device="cuda")
        nv.check(L.qt_mlp_fq8_bf16(x8.data_ptr(), 0 if xdtype == "e4m3" else 1, wg.data_ptr(), wu.data_ptr(), bg.data_ptr(), None, N,
                                   0 if wdtype == "e4m3" else 1, h.data_ptr(), h8.data_ptr(), ctypes.byref(fo), M, K, stream()), "qt_mlp_fq8_bf16")
        a, b = o.canon_nan16(host_u16(h.view(torch.int16))), o.canon_nan16(host_u16(want.view(torch.int16)))
        assert np.array_equal(a, b), (poison, int((a != b).sum()))
        nan = (b == 0x7FC0)
        assert np.array_equal(h8.cpu().numpy()[~nan.reshape(M, N)], want8.cpu().numpy()[~nan.reshape(M, N)]), poison


@pytest.mark.parametrize("kind", ["causal", "padding", "one_row_fully_masked", "scattered"])
def test_softmax_row_live_shortcut_changes_nothing(nv, kind):
    """qt_softmax_fq_bf16_fp8_live (pieces of 512 columns beyond a row's last unmasked column are not loaded) writes the same FP8
    codes as qt_softmax_fq_bf16_fp8 for causal and padding masks, a fully masked row (uniform distribution over all columns) and a
    mask with holes; scores beyond the unmasked part hold NaN / Inf patterns to prove they are not read."""
    L = nv.lib()
    B, H, Q, C = 2, 3, 1024, 1024
    torch.manual_seed(0)
    scores = (torch.randn(B, H, Q, C, device="cuda") * 3).bfloat16()
    neg = torch.finfo(torch.bfloat16).min
    mask = torch.zeros(B, 1, Q, C, device="cuda", dtype=torch.bfloat16)
    if kind == "causal":
        mask[:] = torch.full((Q, C), neg, device="cuda", dtype=torch.bfloat16).triu(1)
    elif kind == "padding":
        mask[0, :, :, 700:] = neg
        mask[1, :, :, 130:] = neg
    elif kind == "one_row_fully_masked":
        mask[:] = torch.full((Q, C), neg, device="cuda", dtype=torch.bfloat16).triu(1)
        mask[1, 0, 5, :] = neg
    else:
        mask[:, :, :, 5::7] = neg
        mask[0, :, :300, 600:] = neg
    rows = B * Q
    live = torch.empty(rows, dtype=torch.int32, device="cuda")
    nv.check(L.qt_mask_row_live(mask.data_ptr(), rows, C, C, live.data_ptr(), stream()), "qt_mask_row_live")
    want_live = ((mask.view(rows, C).float() > -1e30) * torch.arange(1, C + 1, device="cuda")).amax(dim=1).int()
    assert torch.equal(live, want_live)
    fmt = nv.format_for("e4m3")
    ref = torch.empty(B, H, Q, C, dtype=torch.uint8, device="cuda")
    nv.check(L.qt_softmax_fq_bf16_fp8(scores.data_ptr(), mask.data_ptr(), None, ref.data_ptr(), B, H, Q, C, Q * C, 0, C, 0.125,
                                      ctypes.byref(fmt), stream()), "qt_softmax_fq_bf16_fp8")
    poisoned = scores.clone()
    beyond = (torch.arange(C, device="cuda")[None, :] >= ((live.view(B, 1, Q, 1) + 511) // 512 * 512)) & (live.view(B, 1, Q, 1) > 0)
    poisoned.view(torch.int16)[beyond.expand(B, H, Q, C)] = 0x7FC0          # NaN wherever the shortcut promises not to look
    got = torch.full((B, H, Q, C), 0x55, dtype=torch.uint8, device="cuda")
    nv.check(L.qt_softmax_fq_bf16_fp8_live(poisoned.data_ptr(), mask.data_ptr(), got.data_ptr(), B, H, Q, C, Q * C, 0, C, 0.125,
                                           ctypes.byref(fmt), live.data_ptr(), Q, 0, 1, stream()), "qt_softmax_fq_bf16_fp8_live")
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,H,Sq,Sk,mask_kind,live,D", [
    (1, 4, 128, 128, "causal", True, 128), (2, 3, 200, 256, "padding", True, 128), (1, 2, 64, 384, None, False, 128),
    (1, 8, 1024, 1024, "causal", True, 128), (1, 2, 256, 256, "causal", False, 128),
    (1, 2, 1000, 640, "causal", True, 128),      # ragged rows, an odd block count, rows 0 .. 359 fully masked
    (2, 2, 1, 256, "padding", True, 128),        # one query row
    (2, 12, 384, 384, "padding", True, 64),      # the BERT-base head shape
    (1, 3, 200, 256, "causal", True, 64), (1, 2, 64, 128, None, False, 64), (1, 2, 1000, 640, "causal", True, 64)])
@pytest.mark.parametrize("fmt_name", ["e4m3", "e5m2"])
@pytest.mark.parametrize("simple", [False, True])
@pytest.mark.parametrize("variant", [2, 1])
def test_attention_fp8_kernel(nv, monkeypatch, B, H, Sq, Sk, mask_kind, live, D, fmt_name, simple, variant):
    """qt_attention_fp8 (+ qt_value_codes_t) against oracle.attention_fq, head_dim 128 and 64, all four matmul inputs in one stateless FP8
    format: the module chain with every rounding point explicit and exp / sums in float64.  Almost every output element is
    identical (measured <= 6e-3 differ: a score is ONE matrix instruction's sum of 128 products rounded to fp32 once, where the bf16
    kernel adds four partial sums -- another rounding pattern around the bf16 boundaries; the instruction's adder tree itself loses
    nothing measurable, test_linear_fq8_worst_case_cancellation); where a probability lands on the other side of a boundary of its 8-bit format one output row moves by at most that
    probability's step.  Also: the permuted transposed value codes against a torch restatement."""
    L = nv.lib()
    torch.manual_seed(B * 7 + H + Sk)
    qmap_in = torch.from_numpy(o.get_quantization_map(fmt_name).view(np.int16)).cuda().view(torch.bfloat16)
    fqin = lambda t: qmap_in[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q = fqin((torch.randn(B, H, Sq, D, device="cuda")).bfloat16())
    k = fqin((torch.randn(B, H, Sk, D, device="cuda")).bfloat16())
    v_raw = torch.randn(B, Sk, H, D, device="cuda").bfloat16().transpose(1, 2)        # [B, H, Sk, D] view of a [B, S, H, D] buffer, unquantized
    v = fqin(v_raw.contiguous())
    scaling = D ** -0.5
    minv = torch.finfo(torch.bfloat16).min
    mask, msb, msq = None, 0, 0
    if mask_kind == "causal":
        mask = torch.full((Sq, Sk), minv, device="cuda").triu(1 + Sk - Sq).bfloat16()[None, None]
        msq = mask.stride(2)
    elif mask_kind == "padding":
        mask = torch.zeros(B, 1, 1, Sk, device="cuda", dtype=torch.bfloat16)
        mask[:, :, :, Sk - 29:] = minv
        mask[0, :, :, Sk - 150:] = minv
        msb = mask.stride(0)
    fmt = nv.format_for(fmt_name)
    fcode = 0 if fmt_name == "e4m3" else 1
    q8, k8 = _codes_of(nv, q, fmt_name), _codes_of(nv, k, fmt_name)
    vt8 = torch.empty(B, H, D, Sk, dtype=torch.uint8, device="cuda")
    nv.check(L.qt_value_codes_t(v_raw.data_ptr(), vt8.data_ptr(), B, H, Sk, D, v_raw.stride(0), v_raw.stride(1), v_raw.stride(2), ctypes.byref(fmt),
                                stream()), "qt_value_codes_t")
    # the permutation: position p of a 128-block holds key ((p >> 2) & 3) * 16 + ((p >> 4) & 3) * 4 + (p & 3) (+ 64 for the upper half)
    pos = torch.arange(128, device="cuda")
    key_of = (pos & 64) | (((pos >> 2) & 3) << 4) | (((pos >> 4) & 3) << 2) | (pos & 3)
    want_vt = _codes_of(nv, v, fmt_name).view(B, H, Sk // 128, 128, D)[:, :, :, key_of, :].permute(0, 1, 4, 2, 3).reshape(B, H, D, Sk)
    assert torch.equal(vt8, want_vt)
    simple = simple and live and mask is not None              # both test masks are "zeros, then the minimum" row by row
    rl, lsb, lsq = None, 0, 0
    if live and mask is not None:
        mrows = mask.shape[0] * mask.shape[2]
        rl = torch.empty(mrows, dtype=torch.int32, device="cuda")
        nv.check(L.qt_mask_row_live(mask.data_ptr(), mrows, Sk, Sk, rl.data_ptr(), stream()), "qt_mask_row_live")
        lsb, lsq = (mask.shape[2] if mask.shape[0] > 1 else 0), (1 if mask.shape[2] > 1 else 0)
    out = torch.full((B, Sq, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), fcode, mask.data_ptr() if mask is not None else None, msb, 0, msq,
                                rl.data_ptr() if rl is not None else None, lsb, 0, lsq, int(simple), None, out.data_ptr(), None, None, B, H, Sq, Sk, D, scaling, stream()),
             "qt_attention_fp8")
    torch.cuda.synchronize()
    u16 = lambda t: host_u16(t.contiguous().view(torch.int16))  # noqa: E731
    exp, _ = o.attention_fq(u16(q), u16(k), u16(v), u16(mask) if mask is not None else None, scaling, o.get_quantization_map(fmt_name))
    got = u16(out.permute(0, 2, 1, 3))
    assert float((got != exp).mean()) <= (1.2e-2 if fmt_name == "e4m3" else 3e-2), float((got != exp).mean())      # e5m2: two mantissa bits
    ev = o.bf16_to_f32(exp)
    err = np.abs(o.bf16_to_f32(got) - ev) / (np.abs(ev).max(axis=-1, keepdims=True) + 1e-30)
    assert float(err.max()) <= 0.08, float(err.max())
    # the differences are isolated single-step flips of a probability (the readout test below): as a whole the result is the oracle's
    l2 = float(np.sqrt(((o.bf16_to_f32(got).astype(np.float64) - ev) ** 2).sum() / ((ev.astype(np.float64) ** 2).sum() + 1e-30)))
    assert l2 <= (4e-3 if fmt_name == "e4m3" else 1.2e-2), l2
    # with the consumer's fake-quantizer on the epilogue: fq(out) and its codes, from the very same result
    fo = nv.format_for("e4m3")
    outq = torch.empty_like(out)
    out8 = torch.empty(B, Sq, H, D, dtype=torch.uint8, device="cuda")
    # ... and with the device-side verdict on the mask's rows (qt_mask_row_live_checked) in place of the host's
    flag = None
    if rl is not None:
        rl2 = torch.empty(rl.numel() + 1, dtype=torch.int32, device="cuda")
        nv.check(L.qt_mask_row_live_checked(mask.data_ptr(), rl.numel(), Sk, Sk, rl2.data_ptr(), rl2.data_ptr() + 4 * rl.numel(), stream()),
                 "qt_mask_row_live_checked")
        assert torch.equal(rl2[:-1], rl) and int(rl2[-1]) == 0                  # both test masks are regular
        flag = rl2.data_ptr() + 4 * rl.numel()
    nv.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), fcode, mask.data_ptr() if mask is not None else None, msb, 0, msq,
                                rl.data_ptr() if rl is not None else None, lsb, 0, lsq, 0 if flag else int(simple), flag, outq.data_ptr(), out8.data_ptr(),
                                ctypes.byref(fo), B, H, Sq, Sk, D, scaling, stream()), "qt_attention_fp8")
    want8 = _codes_of(nv, out, "e4m3")
    assert torch.equal(out8, want8)
    qm = torch.from_numpy(o.get_quantization_map("e4m3").view(np.int16)).cuda()
    assert torch.equal(outq.view(torch.int16), qm[(out.view(torch.int16).to(torch.int32) & 0xFFFF).long()])


@pytest.mark.parametrize("fmt_name", ["e4m3", "e5m2"])
@pytest.mark.parametrize("D,Sq,Sk,mask_kind", [(128, 256, 128, "causal"), (128, 300, 256, None), (64, 256, 128, "causal"), (128, 1024, 1024, "causal")])
def test_attention_fp8_kernel_probabilities_are_one_code_step_from_the_oracle(nv, fmt_name, D, Sq, Sk, mask_kind):
    """what the kernel's differences from the oracle ARE.  With V = identity on the first D keys (zero elsewhere) the
    output IS the kernel's fake-quantized probability of keys 0 .. D-1 (one exact product per element), so the probabilities the kernel
    forms inside can be compared with oracle.attention_fq's -- the oracle that tests/test_oracle_golden.py::test_attention_chain pins
    to upstream's twin.  Every difference is exactly ONE step of the format's value grid (a score that lands on the neighbouring
    bf16 value moves exp() by a few per cent, less than a grid step), on a bounded share of the elements; nothing else differs."""
    L = nv.lib()
    B, H = 1, 2
    torch.manual_seed(D + Sk)
    qmap_np = o.get_quantization_map(fmt_name)
    qmap_in = torch.from_numpy(qmap_np.view(np.int16)).cuda().view(torch.bfloat16)
    fqin = lambda t: qmap_in[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q = fqin((torch.randn(B, H, Sq, D, device="cuda")).bfloat16())
    k = fqin((torch.randn(B, H, Sk, D, device="cuda")).bfloat16())
    v_raw = torch.zeros(B, Sk, H, D, device="cuda", dtype=torch.bfloat16)
    idx = torch.arange(D, device="cuda")
    v_raw[:, idx, :, idx] = 1.0
    v_raw = v_raw.transpose(1, 2)
    scaling = D ** -0.5
    mask, msq = None, 0
    if mask_kind == "causal":
        mask = torch.full((Sq, Sk), torch.finfo(torch.bfloat16).min, device="cuda").triu(1 + Sk - Sq).bfloat16()[None, None]
        msq = mask.stride(2)
    fmt = nv.format_for(fmt_name)
    q8, k8 = _codes_of(nv, q, fmt_name), _codes_of(nv, k, fmt_name)
    vt8 = torch.empty(B, H, D, Sk, dtype=torch.uint8, device="cuda")
    nv.check(L.qt_value_codes_t(v_raw.data_ptr(), vt8.data_ptr(), B, H, Sk, D, v_raw.stride(0), v_raw.stride(1), v_raw.stride(2), ctypes.byref(fmt),
                                stream()), "qt_value_codes_t")
    out = torch.full((B, Sq, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), 0 if fmt_name == "e4m3" else 1, mask.data_ptr() if mask is not None else None,
                                0, 0, msq, None, 0, 0, 0, 0, None, out.data_ptr(), None, None, B, H, Sq, Sk, D, scaling, stream()), "qt_attention_fp8")
    torch.cuda.synchronize()
    u16 = lambda t: host_u16(t.contiguous().view(torch.int16))  # noqa: E731
    _, pq = o.attention_fq(u16(q), u16(k), u16(v_raw.contiguous()), u16(mask) if mask is not None else None, scaling, qmap_np)
    want = o.bf16_to_f32(pq[..., :D]).astype(np.float64)                  # [B, H, Sq, D]: probabilities of keys 0 .. D-1
    got = o.bf16_to_f32(u16(out.permute(0, 2, 1, 3))).astype(np.float64)
    values = np.unique(o.bf16_to_f32(qmap_np[np.isfinite(o.bf16_to_f32(qmap_np))]).astype(np.float64))
    assert np.isin(got, values).all(), "every output is a value of the format (V = identity reads the probabilities out)"
    steps = np.abs(np.searchsorted(values, got) - np.searchsorted(values, want))
    share = float((steps > 0).mean())
    assert steps.max() <= 1, int(steps.max())
    assert share <= (2.5e-2 if fmt_name == "e4m3" else 1.5e-2), share     # measured: see the docstring of test_attention_fp8_kernel


def test_attention_fp8_kernel_reads_an_irregular_mask(nv):
    """A mask that is not "zeros, then the minimum" (a relative-position bias inside the extents, a hole of masked columns): the scan
    flags it on the device and the kernel reads it; the result matches the oracle as for the regular masks."""
    L = nv.lib()
    B, H, Sq, Sk, D = 1, 2, 192, 256, 128
    torch.manual_seed(5)
    qmap_in = torch.from_numpy(o.get_quantization_map("e4m3").view(np.int16)).cuda().view(torch.bfloat16)
    fqin = lambda t: qmap_in[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q = fqin(torch.randn(B, H, Sq, D, device="cuda").bfloat16())
    k = fqin(torch.randn(B, H, Sk, D, device="cuda").bfloat16())
    v_raw = torch.randn(B, Sk, H, D, device="cuda").bfloat16().transpose(1, 2)
    v = fqin(v_raw.contiguous())
    minv = torch.finfo(torch.bfloat16).min
    mask = (torch.randn(Sq, Sk, device="cuda") * 0.5).bfloat16()
    mask[:, 200:] = minv
    mask[:, 40:57] = minv                                                     # a hole
    mask = mask[None, None].contiguous()
    fmt = nv.format_for("e4m3")
    q8, k8 = _codes_of(nv, q, "e4m3"), _codes_of(nv, k, "e4m3")
    vt8 = torch.empty(B, H, D, Sk, dtype=torch.uint8, device="cuda")
    nv.check(L.qt_value_codes_t(v_raw.data_ptr(), vt8.data_ptr(), B, H, Sk, D, v_raw.stride(0), v_raw.stride(1), v_raw.stride(2), ctypes.byref(fmt),
                                stream()), "qt_value_codes_t")
    rl = torch.empty(Sq + 1, dtype=torch.int32, device="cuda")
    nv.check(L.qt_mask_row_live_checked(mask.data_ptr(), Sq, Sk, Sk, rl.data_ptr(), rl.data_ptr() + 4 * Sq, stream()), "qt_mask_row_live_checked")
    assert int(rl[-1]) == 1 and bool((rl[:-1] == 200).all())
    out = torch.full((B, Sq, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), 0, mask.data_ptr(), 0, 0, Sk, rl.data_ptr(), 0, 0, 1, 0,
                                rl.data_ptr() + 4 * Sq, out.data_ptr(), None, None, B, H, Sq, Sk, D, D ** -0.5, stream()), "qt_attention_fp8")
    torch.cuda.synchronize()
    u16 = lambda t: host_u16(t.contiguous().view(torch.int16))  # noqa: E731
    exp, _ = o.attention_fq(u16(q), u16(k), u16(v), u16(mask), D ** -0.5, o.get_quantization_map("e4m3"))
    got = u16(out.permute(0, 2, 1, 3))
    # 2e-2 (measured 1.45e-2): with a bias added to every score there is one more bf16 rounding at which the matrix instruction's
    # fixed-point sum and the oracle's exact sum can part than with a 0 / minimum mask (1.2e-2 there)
    assert float((got != exp).mean()) <= 2e-2, float((got != exp).mean())
    ev = o.bf16_to_f32(exp)
    err = np.abs(o.bf16_to_f32(got) - ev) / (np.abs(ev).max(axis=-1, keepdims=True) + 1e-30)
    assert float(err.max()) <= 0.08, float(err.max())


def test_linear_fq8_rejects_what_it_does_not_take(nv):
    x8 = torch.zeros(16, 192, dtype=torch.uint8, device="cuda")
    w = torch.zeros(32, 192, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(16, 32, dtype=torch.bfloat16, device="cuda")
    wp, ns = (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_int * 1)(32)
    L = nv.lib()
    assert L.qt_linear_fq8_bf16(x8.data_ptr(), 0, wp, None, ns, 1, 0, y.data_ptr(), 16, 192, stream()) == nv.QT_ERR_BAD_ARG   # K % 128
    ns2 = (ctypes.c_int * 1)(24)
    assert L.qt_linear_fq8_bf16(x8.data_ptr(), 0, wp, None, ns2, 1, 0, y.data_ptr(), 16, 128, stream()) == nv.QT_ERR_BAD_ARG  # n % 16
    assert L.qt_linear_fq8_bf16(x8.data_ptr(), 2, wp, None, ns, 1, 0, y.data_ptr(), 16, 128, stream()) == nv.QT_ERR_BAD_DTYPE
    assert L.qt_linear_fq8_bf16(x8.data_ptr(), 0, wp, None, ns, 1, 0, y.data_ptr(), 0, 128, stream()) == 0                    # empty


def test_fq8_route_through_quantize(nv, monkeypatch):
    """QT_FQ8_GEMM=1: the QAT Linears of a quantize()d model (single layers and a q / k / v sibling group) run the fused
    kernel; logits equal the default route's within the accumulation bound, element counts are identical."""
    import quantized_training as qt
    from quantized_training import fake_quantize
    torch.manual_seed(0)

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v = (torch.nn.Linear(256, 128, bias=True) for _ in range(3))
            self.o = torch.nn.Linear(128, 256, bias=False)

        def forward(self, x):
            return self.o(self.q(x) * self.k(x) + self.v(x))

    model = Block().cuda().bfloat16()
    args = qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16"])
    qt.quantize(model, args)
    x = torch.randn(3, 70, 256, device="cuda").bfloat16()
    outs, counts = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("QT_FQ8_GEMM", mode)
        with torch.no_grad():
            model(x)                                                       # creates the activation fake-quantizers
            fake_quantize.STATS.reset()
            outs[mode] = model(x).double()
            counts[mode] = (fake_quantize.STATS.elements, fake_quantize.STATS.calls)
    assert counts["0"] == counts["1"]
    scale = outs["0"].abs().max()
    assert float((outs["0"] - outs["1"]).abs().max()) <= 2.0 ** -6 * float(scale)


# ---- converted per-tensor PT2E graphs on the integer matrix cores (qt_q8_gemm, pt2e_native.py) -----------------------------
@pytest.mark.parametrize("batch,M,N,K", [(1, 128, 128, 128), (1, 1, 16, 16), (1, 300, 200, 48), (3, 70, 96, 256), (1, 1024, 512, 1024),
                                         (1, 2048, 3072, 512), (2, 520, 1040, 384)])
def test_q8_gemm_is_exact(nv, batch, M, N, K):
    """int8 codes x int8 codes with int32 accumulation: equal to the integer product (computed in int64 on the device by
    torch) whenever it fits fp32 exactly; bias, the bf16 rounding of the sum and the dequantize multiply keep the rounding
    points of `aten.linear` -> `dequantize` (quantize_pt2e.py:323-446)."""
    torch.manual_seed(batch + M + N + K)
    a = torch.randint(-128, 128, (batch, M, K), device="cuda", dtype=torch.int8)
    b = torch.randint(-128, 128, (batch, N, K), device="cuda", dtype=torch.int8)
    L = nv.lib()
    exact = torch.matmul(a.double(), b.double().transpose(1, 2))        # integers < 2^53: exact in float64
    y = torch.empty((batch, M, N), dtype=torch.float32, device="cuda")
    nv.check(L.qt_q8_gemm(a.data_ptr(), b.data_ptr(), y.data_ptr(), 1, None, None, 0, 0, batch, M, N, K, M, N, stream()), "q8")
    assert torch.equal(y.double(), exact)                          # |sum| <= 128 * 128 * 1024 = 2^24: exact in fp32
    bias = torch.randint(-2000, 2000, (N,), device="cuda").float()
    scale = torch.rand(N, device="cuda") * 1e-3 + 1e-4
    for per_col in (0, 1):
        s = scale if per_col else scale[:1].clone()
        nv.check(L.qt_q8_gemm(a.data_ptr(), b.data_ptr(), y.data_ptr(), 1, bias.data_ptr(), s.data_ptr(), per_col, 0, batch, M, N, K, M, N,
                              stream()), "q8")
        assert torch.equal(y, (exact.float() + bias) * s)
        nv.check(L.qt_q8_gemm(a.data_ptr(), b.data_ptr(), y.data_ptr(), 1, bias.data_ptr(), s.data_ptr(), per_col, 1, batch, M, N, K, M, N,
                              stream()), "q8")
        pre = (exact.float() + bias).view(torch.int32)
        folded = ((pre & -65536) | ((pre & 0xFFFF) != 0).int() * 65536).view(torch.float32)     # identity map on an fp32 tensor
        assert torch.equal(y, folded * s)
        yb = torch.empty((batch, M, N), dtype=torch.bfloat16, device="cuda")
        bias16, s16 = bias.bfloat16(), s.bfloat16()
        nv.check(L.qt_q8_gemm(a.data_ptr(), b.data_ptr(), yb.data_ptr(), 0, bias16.data_ptr(), s16.data_ptr(), per_col, 0,
                              batch, M, N, K, M, N, stream()), "q8")
        expb = ((exact.float() + bias16.float()).bfloat16().float() * s16.float()).bfloat16()
        assert torch.equal(yb, expb)


def test_pt2e_converted_graph_runs_natively(nv):
    """The reference's PT2E toy flow (tests/golden/pt2e.*: prepare, calibrate, convert) with the model on the device: the
    int8 configuration's two Linears and its y . y^T matmul run through qt_q8_gemm (counted); the FP8 configuration is an
    fp32 model whose GEMM outputs are re-rounded through a bfloat16 map, which stays on the value-tensor route.  Outputs match
    upstream's converted-graph output within the accumulation bound."""
    from test_pt2e_cpu import META, _model
    from quantized_training import pt2e_native, quantize_pt2e as qp
    arr = np.load(os.path.join(G, "pt2e.npz"))
    for name in ("int8", "fp8"):
        info = META[name]
        xs = [torch.from_numpy(arr[f"x{i}"].view(np.float32)).reshape(4, 8, 16).cuda() for i in range(4)]
        gm = qp.prepare_pt2e(_model(arr).cuda(), qp.get_default_quantizer(**info["kw"]), (xs[0],))
        with torch.no_grad():
            for i in range(4):
                gm(xs[i])
        before = dict(pt2e_native.STATS)
        gc = qp.convert_pt2e(gm, info["output_dtype"]) if info["output_dtype"] else qp.convert_pt2e(gm)
        targets = [str(n.target) for n in gc.graph.nodes]
        if name == "int8":
            assert any("linear_q" in t for t in targets) and not any(t == "aten.linear.default" for t in targets)
        with torch.no_grad():
            y = gc(xs[3]).float().cpu()
        exp = torch.from_numpy(arr[f"{name}__y_converted"].view(np.float32).copy()).reshape(y.shape)
        ran = {k: pt2e_native.STATS[k] - before[k] for k in before}
        if name == "int8":
            assert ran["linear_int8"] == 2 and ran["matmul_int8"] >= 1, ran
        g, e = y.reshape(-1, y.shape[-1]), exp.reshape(-1, y.shape[-1])
        rel = ((g - e).norm(dim=1) / e.norm(dim=1).clamp_min(1e-6)).max()
        assert float(rel) <= 0.05, (name, float(rel))


def test_pt2e_native_matches_value_tensor_route_at_size(nv, monkeypatch):
    """A wider model (K = 512 / 1024, 1024 rows: the LDS-DMA and 256 x 256 int8 kernels): the natively executed converted
    graph against the same converted graph run as upstream runs it (bf16 `aten.linear` on the value tensors)."""
    from quantized_training import pt2e_native, quantize_pt2e as qp

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(512, 1024)
            self.fc2 = torch.nn.Linear(1024, 512)

        def forward(self, x):
            return self.fc2(torch.relu(self.fc1(x))) + x

    torch.manual_seed(0)
    x = torch.randn(4, 256, 512, device="cuda").bfloat16()
    outs = {}
    for native in ("1", "0"):
        monkeypatch.setenv("QT_PT2E_NATIVE", native)
        torch.manual_seed(1)
        net = Net().cuda().bfloat16().eval()
        gm = qp.prepare_pt2e(net, qp.get_default_quantizer("int8,qs=per_tensor_symmetric", None, "int8,qs=per_tensor_symmetric", "int24"), (x,))
        with torch.no_grad():
            gm(x), gm(x * 0.5)
        before = pt2e_native.STATS["linear_int8"]
        gc = qp.convert_pt2e(gm)
        with torch.no_grad():
            outs[native] = gc(x).float()
        assert (pt2e_native.STATS["linear_int8"] - before) == (2 if native == "1" else 0)
    a, b = outs["1"], outs["0"]
    rel = ((a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-6)).max()
    assert float(rel) <= 0.02, float(rel)


def test_pt2e_fp8_linears_run_on_the_fp8_matrix_cores(nv, monkeypatch):
    """bf16 model, `fp8_e4m3,qs=per_tensor_symmetric` activations and weights: the converted graph's Linears narrow both
    operands to OCP FP8 codes and run the in-tree FP8 GEMM (qt_mx_gemm at unit block scales; counted), against the value-tensor route."""
    from quantized_training import pt2e_native, quantize_pt2e as qp

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(256, 512)
            self.fc2 = torch.nn.Linear(512, 256)

        def forward(self, x):
            return self.fc2(torch.relu(self.fc1(x)))

    x = torch.randn(2, 128, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)).bfloat16()
    outs = {}
    for native in ("1", "0"):
        monkeypatch.setenv("QT_PT2E_NATIVE", native)
        torch.manual_seed(1)
        gm = qp.prepare_pt2e(Net().cuda().bfloat16().eval(),
                             qp.get_default_quantizer("fp8_e4m3,qs=per_tensor_symmetric", None, "fp8_e4m3,qs=per_tensor_symmetric", "float32"), (x,))
        with torch.no_grad():
            gm(x), gm(x * 0.5)
        before = pt2e_native.STATS["linear_fp8"]
        before_native = pt2e_native.STATS.get("linear_fp8_native", 0)
        gc = qp.convert_pt2e(gm)
        with torch.no_grad():
            outs[native] = gc(x).float()
        assert (pt2e_native.STATS["linear_fp8"] - before) == (2 if native == "1" else 0)
        # ... and on the in-tree scaled-MFMA kernel (qt_mx_gemm at unit block scales), not the library GEMM
        assert (pt2e_native.STATS.get("linear_fp8_native", 0) - before_native) == (2 if native == "1" else 0)
    rel = ((outs["1"] - outs["0"]).norm(dim=-1) / outs["0"].norm(dim=-1).clamp_min(1e-6)).max()
    assert float(rel) <= 0.02, float(rel)


@pytest.mark.parametrize("rows,cols", [(2048, 768), (2048, 3072), (1, 8), (777, 40), (64, 2304)])
def test_colsum_is_the_bias_gradient(nv, rows, cols):
    """qt_colsum_bf16 = grad_output.sum(0) (autograd's bias gradient of F.linear, modules/qat/linear.py:40-41): against an fp64 column
    sum, within one bf16 rounding of the result plus fp32 accumulation; deterministic; and through the QAT Linear's backward the three
    gradients equal autograd's own up to that rounding."""
    torch.manual_seed(rows + cols)
    x = (torch.randn(rows, cols, device="cuda") * 3).bfloat16()
    out = torch.empty(cols, dtype=torch.bfloat16, device="cuda")
    nv.check(nv.lib().qt_colsum_bf16(x.data_ptr(), out.data_ptr(), rows, cols, stream()), "qt_colsum_bf16")
    again = torch.empty_like(out)
    nv.check(nv.lib().qt_colsum_bf16(x.data_ptr(), again.data_ptr(), rows, cols, stream()), "qt_colsum_bf16")
    assert torch.equal(out.view(torch.int16), again.view(torch.int16))
    ref = x.double().sum(0)
    tol = ref.abs() * 2.0 ** -8 + x.double().abs().sum(0) * 2.0 ** -20 + 1e-30
    assert bool(((out.double() - ref).abs() <= tol).all())
    assert nv.lib().qt_colsum_bf16(x.data_ptr(), out.data_ptr(), rows, 12, stream()) == nv.QT_ERR_BAD_ARG       # cols % 8


def test_qat_linear_training_backward_uses_colsum(nv):
    import quantized_training as qt
    from quantized_training.modules.qat.linear import Linear as QATLinear
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 384, bias=True).cuda().bfloat16()
    lin.qconfig = qt.QConfig(activation=None, weight=lambda **kw: qt.FusedAmaxObsFakeQuantize(dtype="posit8_1", **{k: v for k, v in kw.items() if k == "device"}),
                             error=None)                                     # stateless: the second call below quantizes W to the same values
    q = QATLinear.from_float(lin).cuda().train()
    x = torch.randn(8, 64, 256, device="cuda").bfloat16().requires_grad_(True)
    y = q(x)
    assert y.grad_fn is not None and "LinearColsumBias" in type(y.grad_fn).__name__
    gy = torch.randn_like(y)
    y.backward(gy)
    got = (x.grad.clone(), q.weight.grad.clone(), q.bias.grad.clone())
    # autograd's own backward on the same quantized weight
    x2 = x.detach().clone().requires_grad_(True)
    wq = q.weight_fake_quant(q.weight.detach()).detach().requires_grad_(True)
    b2 = q.bias.detach().clone().requires_grad_(True)
    torch.nn.functional.linear(x2, wq, b2).backward(gy)
    assert torch.equal(got[0], x2.grad) and torch.equal(got[1], wq.grad)
    ref = gy.reshape(-1, 384).double().sum(0)
    assert bool(((got[2].double() - ref).abs() <= ref.abs() * 2.0 ** -8 + gy.double().abs().sum((0, 1)) * 2.0 ** -20).all())
