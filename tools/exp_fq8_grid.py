"""Launch-geometry sweep for the FP8-output pass (weight pass of the FP8 GEMM route)."""
import ctypes, sys, torch
sys.path.insert(0, "quantized-training_amd")
from quantized_training import _native as nv
L = nv.lib()
L.qt_internal_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
L.qt_internal_set_variant.restype = None
pool = 8
for rows, cols in ((4096, 11008), (4096, 4096)):
    n = rows * cols
    x = torch.empty(pool, rows, cols, device="cuda", dtype=torch.bfloat16).normal_(0, 0.02)
    y8 = torch.empty(pool, rows, cols, device="cuda", dtype=torch.uint8)
    y = torch.empty_like(x)
    fmt = nv.format_for("e4m3")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for bpc in (4, 8, 16, 32, 64, 128, 1024):
        L.qt_internal_set_variant(0, bpc)
        ms = ctypes.c_float(0)
        for iters in (pool, 6 * pool):
            nv.check(L.qt_bench_fake_quant_bf16_fp8(x.data_ptr(), None, y8.data_ptr(), n, ctypes.byref(fmt), None, None, iters, n, pool, st, ctypes.byref(ms)), "b")
        ms2 = ctypes.c_float(0)
        for iters in (pool, 6 * pool):
            nv.check(L.qt_bench_fake_quant_bf16_fp8(x.data_ptr(), y.data_ptr(), y8.data_ptr(), n, ctypes.byref(fmt), None, None, iters, n, pool, st, ctypes.byref(ms2)), "b")
        print(f"{rows}x{cols} blocks/CU {bpc:5d}: fp8-only {ms.value*1e3:6.1f} us {n*3/ms.value/1e9:6.2f} TB/s | bf16+fp8 {ms2.value*1e3:6.1f} us {n*5/ms2.value/1e9:6.2f} TB/s")
L.qt_internal_set_variant(0, 32)
