"""One block-scaled linear layer end to end, as a converted PT2E graph runs it:
    scale, q = quantize_mx(x);  y = linear_mx(q, w_codes, bias, input_scale=scale, weight_scale=w_scale, block_size=32)
native path (fused quantize_mx + scaled-MFMA GEMM) against the reference formulation (QT_MX_GEMM=0 and the composite
quantize_mx).  Run on the GPU box:  python tools/exp_mx_linear.py"""
import os
import sys
import time

import torch

sys.path.insert(0, "quantized-training_amd")
import quantized_training  # noqa: E402,F401
from quantized_training import decomposed, mx_gemm  # noqa: E402
from quantized_training.fake_quantize import get_quantization_map  # noqa: E402
from quantized_training.quantizer.quantizer import get_quant_min_max  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    for fmt in ("fp8_e4m3", "fp6_e3m2", "fp4_e2m1"):
        qmap = get_quantization_map(fmt, "cuda")
        qmax = float(get_quant_min_max(fmt)[1])
        for M, N, K in ((1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (8192, 8192, 8192)):
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
            bias = torch.randn(N, device="cuda").bfloat16()
            ws, wq = torch.ops.quantized_ops.quantize_mx(w, qmap, [-1], 32, qmax, True, None, None)

            def layer():
                s, q = torch.ops.quantized_ops.quantize_mx(x, qmap, [-1], 32, qmax, True, None, None)
                return torch.ops.quantized_ops.linear_mx(q, wq, bias, input_scale=s, weight_scale=ws, block_size=32)

            def quant_only():
                return torch.ops.quantized_ops.quantize_mx(x, qmap, [-1], 32, qmax, True, None, None)

            y = layer()
            t_native, t_q = timeit(layer), timeit(quant_only)
            os.environ["QT_MX_GEMM"] = "0"
            fused = decomposed._quantize_mx_hip_or_none
            decomposed._quantize_mx_hip_or_none = lambda *a, **k: None
            try:
                y_ref = layer()
                t_ref = timeit(layer, 5)
            finally:
                decomposed._quantize_mx_hip_or_none = fused
                del os.environ["QT_MX_GEMM"]
            rel = float((y.float() - y_ref.float()).norm() / y_ref.float().norm())
            print(f"{fmt} M{M} N{N} K{K}: native {t_native*1e6:8.1f} us (quantize_mx {t_q*1e6:6.1f} us, "
                  f"{2.0*M*N*K/t_native/1e12:6.0f} TFLOP/s whole layer) | reference formulation {t_ref*1e6:9.1f} us | "
                  f"speed-up {t_ref/t_native:5.1f}x | rel diff {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
