"""Times qt_mx_gemm against (a) torch._scaled_mm with e8m0 block scales (hipBLASLt) and (b) the reference
formulation: two dequantize passes + a bf16 GEMM.  Run on the GPU box:  python tools/exp_mx_gemm.py"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, "quantized-training_amd")
from quantized_training import _native  # noqa: E402

L = _native.lib()
BITS = {0: 8, 1: 8, 2: 6, 3: 6, 4: 4}


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (4096, 4096, 4096), (8192, 8192, 8192)]
    for M, N, K in shapes:
        flop = 2.0 * M * N * K
        line = f"M{M} N{N} K{K}:"
        for f in (0, 2, 4):
            a = torch.randint(0, 255, (M, K * BITS[f] // 8), device="cuda", dtype=torch.uint8)
            b = torch.randint(0, 255, (N, K * BITS[f] // 8), device="cuda", dtype=torch.uint8)
            if f == 0:                                   # keep e4m3 codes finite
                a &= 0x77
                b &= 0x77
            sa = torch.full((M, K // 32), 127, device="cuda", dtype=torch.uint8)
            sb = torch.full((N, K // 32), 127, device="cuda", dtype=torch.uint8)
            c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            def run():
                _native.check(L.qt_mx_gemm(a.data_ptr(), sa.data_ptr(), f, b.data_ptr(), sb.data_ptr(), f, c.data_ptr(), 0, None,
                                           1, M, N, K, 0, 0, st), "mx")
            dt = timeit(run)
            line += f"  fmt{f} {dt*1e6:7.1f} us {flop/dt/1e12:6.0f} TF"
        try:
            a8 = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
            b8 = torch.randn(N, K, device="cuda").to(torch.float8_e4m3fn)
            s1 = torch.ones(M, K // 32, device="cuda").to(torch.float8_e8m0fnu)
            s2 = torch.ones(N, K // 32, device="cuda").to(torch.float8_e8m0fnu)
            dt = timeit(lambda: torch._scaled_mm(a8, b8.t(), scale_a=s1, scale_b=s2, out_dtype=torch.bfloat16))
            line += f" | lib mxfp8 {dt*1e6:7.1f} us {flop/dt/1e12:6.0f} TF"
        except Exception as e:  # noqa: BLE001
            line += f" | lib mxfp8 failed {type(e).__name__}"
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16()
        sx = torch.ones(M, K // 32, device="cuda").bfloat16()
        sw = torch.ones(N, K // 32, device="cuda").bfloat16()
        def ref():
            return torch.nn.functional.linear(x * sx.repeat_interleave(32, 1), w * sw.repeat_interleave(32, 1))
        dt = timeit(ref, 10)
        dt2 = timeit(lambda: torch.nn.functional.linear(x, w), 10)
        line += f" | dequant+bf16 {dt*1e6:7.1f} us (GEMM alone {dt2*1e6:7.1f} us {flop/dt2/1e12:5.0f} TF)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
