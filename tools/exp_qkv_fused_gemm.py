"""Is one N = 12288 FP8 GEMM faster than three N = 4096 ones (q, k, v projections of a LLaMA-2-7B layer, M = 1024)?"""
import sys, time, torch
sys.path.insert(0, "quantized-training_amd")
from quantized_training import fused
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
a = torch.randn(1024, 4096, device="cuda").to(torch.float8_e4m3fn)
ws = [(torch.randn(4096, 4096, device="cuda") * 0.02).to(torch.float8_e4m3fn) for _ in range(3)]
wcat = torch.cat(ws, 0).contiguous()
print("3 x N=4096 :", round(t(lambda: [fused.lt_fp8_gemm(a, w) for w in ws]), 1), "us")
print("1 x N=12288:", round(t(lambda: fused.lt_fp8_gemm(a, wcat)), 1), "us")
g = [(torch.randn(11008, 4096, device="cuda") * 0.02).to(torch.float8_e4m3fn) for _ in range(2)]
gcat = torch.cat(g, 0).contiguous()
print("2 x N=11008:", round(t(lambda: [fused.lt_fp8_gemm(a, w) for w in g]), 1), "us")
print("1 x N=22016:", round(t(lambda: fused.lt_fp8_gemm(a, gcat)), 1), "us")
