"""Times qt_mlp_fq8_bf16 (gate GEMM + up GEMM + SiLU * up + the consumer's fake-quantizer in one launch) against the launches it
replaces, weights rotating over a pool larger than the Infinity Cache.   python tools/exp_mlp_fq8.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
from quantized_training import _native  # noqa: E402
from quantized_training.fused import lt_fp8_gemm  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")


def st():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


def timeit(fn, iters=40):
    for i in range(6):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    fmt = _native.format_for("e4m3")
    for M, N, K in ((1024, 11008, 4096), (512, 11008, 4096), (1024, 13824, 5120), (1024, 8192, 4096)):
        pool = 6
        x8 = torch.randn(M, K, device=DEV).to(torch.float8_e4m3fn)
        wg = (torch.randn(pool, N, K, device=DEV) * 0.02).bfloat16()
        wu = (torch.randn(pool, N, K, device=DEV) * 0.02).bfloat16()
        yg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        yu = torch.empty_like(yg)
        h = torch.empty_like(yg)
        h8 = torch.empty(M, N, dtype=torch.uint8, device=DEV)
        w8 = torch.empty(N, K, dtype=torch.uint8, device=DEV)
        one = torch.ones((), dtype=torch.float32, device=DEV)

        def lin(w, y):
            wp = (ctypes.c_void_p * 1)(w.data_ptr())
            _native.check(L.qt_linear_fq8_bf16(x8.data_ptr(), 0, wp, None, (ctypes.c_int * 1)(N), 1, 0, y.data_ptr(), M, K, st()), "lin")

        def silu():
            _native.check(L.qt_silu_mul_fq8_bf16(yg.data_ptr(), yu.data_ptr(), h.data_ptr(), h8.data_ptr(), M, N, N, N, ctypes.byref(fmt), st()), "silu")

        def three(i):
            lin(wg[i % pool], yg); lin(wu[i % pool], yu); silu()

        def pair_route(i):
            for w, y in ((wg[i % pool], yg), (wu[i % pool], yu)):
                _native.check(L.qt_fake_quant_bf16_fp8(w.data_ptr(), None, w8.data_ptr(), w.numel(), ctypes.byref(fmt), one.data_ptr(), None, st()), "pass")
                lt_fp8_gemm(x8, w8.view(torch.float8_e4m3fn))
            silu()

        def one_launch(i):
            _native.check(L.qt_mlp_fq8_bf16(x8.data_ptr(), 0, wg[i % pool].data_ptr(), wu[i % pool].data_ptr(), None, None, N, 0, h.data_ptr(),
                                            h8.data_ptr(), ctypes.byref(fmt), M, K, st()), "mlp")

        t1, t3, tp, ts = timeit(one_launch), timeit(three), timeit(pair_route), timeit(lambda i: silu())
        print(f"{M}x{N}x{K}: one launch {t1:7.1f} us | 2 fused GEMMs + SiLU*up {t3:7.1f} us (SiLU*up alone {ts:5.1f}) | 2 x (pass + hipBLASLt) + SiLU*up {tp:7.1f} us",
              flush=True)
        del wg, wu
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
