"""Q.K^T of a LLaMA-2-7B layer (32 heads, S = 1024, D = 128): bf16 torch.matmul vs the FP8 batched GEMM."""
import sys, time, torch
sys.path.insert(0, "quantized-training_amd")
from quantized_training import fused
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
q = torch.randn(1, 32, 1024, 128, device="cuda").bfloat16(); k = torch.randn(1, 32, 1024, 128, device="cuda").bfloat16()
q8 = q.to(torch.float8_e4m3fn).view(32, 1024, 128); k8 = k.to(torch.float8_e4m3fn).view(32, 1024, 128)
print("bf16 matmul:", round(t(lambda: torch.matmul(q, k.transpose(2, 3))), 1), "us")
print("fp8 batched:", round(t(lambda: fused.lt_fp8_gemm(q8, k8)), 1), "us")
p8 = torch.rand(32, 1024, 1024, device="cuda").to(torch.float8_e4m3fn); v8 = torch.randn(32, 1024, 128, device="cuda").to(torch.float8_e4m3fn)
p = p8.to(torch.bfloat16).view(1, 32, 1024, 1024); v = v8.to(torch.bfloat16).view(1, 32, 1024, 128)
print("P.V bf16:", round(t(lambda: torch.matmul(p, v)), 1), "us;  fp8:", round(t(lambda: fused.lt_fp8_gemm(p8, v8, b_is_kn=True)), 1), "us")
