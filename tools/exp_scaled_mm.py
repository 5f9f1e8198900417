"""GPU experiment: library FP8 GEMM (torch._scaled_mm, OCP e4m3) vs bf16 GEMM on the LLaMA shapes."""
import torch, json, os
dev = torch.device("cuda")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
one = torch.ones((), device=dev)
for (M, N, K) in [(1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (1024, 32000, 4096)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    x8 = x.to(torch.float8_e4m3fn); w8 = w.to(torch.float8_e4m3fn)
    ms_bf = t(lambda: torch.nn.functional.linear(x, w))
    try:
        y8 = torch._scaled_mm(x8, w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        ref = torch.nn.functional.linear(x8.to(torch.bfloat16), w8.to(torch.bfloat16))
        err = (y8.float() - ref.float()).abs().max().item()
        ms_8 = t(lambda: torch._scaled_mm(x8, w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
        print(f"{M}x{N}x{K}: bf16 {ms_bf*1e3:.1f} us {2*M*N*K/ms_bf/1e9:.0f} TF | fp8 {ms_8*1e3:.1f} us {2*M*N*K/ms_8/1e9:.0f} TF | max diff vs bf16 gemm of same values {err:.4f}", flush=True)
    except Exception as e:
        print("scaled_mm failed:", repr(e)[:300], flush=True)
