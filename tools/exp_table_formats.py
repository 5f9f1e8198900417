"""GPU experiment: table-format (LDS value map) streaming kernels, per-tensor and per-channel, on a LLaMA-2 weight
(bf16 [4096, 11008], pool of 8 = 720 MB, beyond the Infinity Cache); algorithmic traffic 4 B/element.
Run under QT_LUT_HALF=0 / QT_PC_LDS=0 for the previous kernels."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import quantized_training as qt  # noqa: E402
from quantized_training import _native as nv  # noqa: E402

L = nv.lib()
dev = torch.device("cuda")
rows, cols, pool = 4096, 11008, 8
n = rows * cols
x = torch.empty(pool, rows, cols, device=dev, dtype=torch.bfloat16).normal_(0, 0.02)
y = torch.empty_like(x)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def per_tensor(dtype, scale=None, amax=None, rows_form=True):
    from quantized_training.fake_quantize import _launch_format
    fmt = nv.format_for(dtype)
    lut = qt.get_quantization_map(dtype, dev)
    if rows_form:
        fmt = _launch_format(fmt, lut)
    ms = ctypes.c_float()
    for iters in (pool, 6 * pool):
        nv.check(L.qt_bench_fake_quant_bf16(x.data_ptr(), y.data_ptr(), n, ctypes.byref(fmt), lut.data_ptr(),
                                            scale.data_ptr() if scale is not None else None, amax.data_ptr() if amax is not None else None,
                                            iters, n, pool, st, ctypes.byref(ms)), "bench")
    return ms.value * 1e3


def per_channel(dtype, observe, rows_form=True):
    from quantized_training.fake_quantize import _launch_format
    fmt = nv.format_for(dtype)
    lut = qt.get_quantization_map(dtype, dev)
    if rows_form:
        fmt = _launch_format(fmt, lut)
    scale = torch.rand(rows, device=dev) * 0.01 + 0.001
    amax = torch.zeros(rows, dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for iters in (pool, 6 * pool):
        e0.record()
        for i in range(iters):
            nv.check(L.qt_fake_quant_pc_bf16(x[i % pool].data_ptr(), y[i % pool].data_ptr(), 1, rows, cols, ctypes.byref(fmt), lut.data_ptr(),
                                             scale.data_ptr(), amax.data_ptr() if observe else None, st), "pc")
        e1.record()
        e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


s = torch.tensor([0.013], device=dev)
h = torch.zeros(16, device=dev)
print(f"QT_LUT_HALF={os.environ.get('QT_LUT_HALF', '1')} QT_PC_LDS={os.environ.get('QT_PC_LDS', '1')}")
for dt in ("posit8_1", "posit8_2", "fp8_e4m3", "fp4_e2m1", "int8"):
    f = nv.format_for(dt)
    for label, sc, am in (("unit", None, None), ("scale", s, None), ("scale+obs", s, h)):
        us = per_tensor(dt, sc, am, rows_form=False)
        ur = per_tensor(dt, sc, am, rows_form=True)
        print(f"per-tensor  {dt:9s} {label:10s} kind {f.kind}: table in LDS {us:6.1f} us  {n * 4 / us / 1e6:5.2f} TB/s   row form {ur:6.1f} us  "
              f"{n * 4 / ur / 1e6:5.2f} TB/s", flush=True)
for dt in ("posit8_1", "fp8_e4m3", "int8"):
    for obs in (False, True):
        us = per_channel(dt, obs, rows_form=False)
        ur = per_channel(dt, obs, rows_form=True)
        print(f"per-channel {dt:9s} observer={obs}: table in LDS {us:6.1f} us  {n * 4 / us / 1e6:5.2f} TB/s   row form {ur:6.1f} us  {n * 4 / ur / 1e6:5.2f} TB/s", flush=True)
