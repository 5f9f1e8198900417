"""Times the fused quantize_mx pass (bf16, block 32, power-of-two scales).  python tools/exp_mx_quant.py"""
import sys
import time

import torch

sys.path.insert(0, "quantized-training_amd")
import quantized_training  # noqa: E402,F401
from quantized_training.fake_quantize import get_quantization_map  # noqa: E402

for fmt, qmax, bits in (("fp8_e4m3", 448.0, 8), ("fp4_e2m1", 6.0, 4)):
    qmap = get_quantization_map(fmt, "cuda")
    for rows, cols in ((1024, 4096), (4096, 4096), (4096, 11008), (8192, 8192), (32000, 4096)):
        xs = [torch.randn(rows, cols, device="cuda").bfloat16() for _ in range(4)]
        fn = lambda i: torch.ops.quantized_ops.quantize_mx(xs[i % 4], qmap, [-1], 32, qmax, True, None, None)
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 40
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        nbytes = rows * cols * (2 + 2 + bits / 8 + 2 / 32 + 1 / 32)
        print(f"{fmt} {rows}x{cols}: {dt*1e6:7.1f} us  {rows*cols/dt/1e9:6.1f} G elem/s  {nbytes/dt/1e12:5.2f} TB/s (read x, write q + scales + packed)")
