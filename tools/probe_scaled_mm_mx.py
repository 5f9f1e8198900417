"""Probe: does torch._scaled_mm accept MX (e8m0 block-32) scales on this box?  Prints a line per attempt."""
import torch, time
M, N, K = 1024, 4096, 4096
a = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
b = torch.randn(N, K, device="cuda").to(torch.float8_e4m3fn)
for name in ("float8_e8m0fnu",):
    print(name, hasattr(torch, name))
try:
    sa = torch.full((M, K // 32), 1.0, device="cuda").to(torch.float8_e8m0fnu)
    sb = torch.full((N, K // 32), 1.0, device="cuda").to(torch.float8_e8m0fnu)
    y = torch._scaled_mm(a, b.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
    ref = (a.float() @ b.float().t())
    print("mxfp8 _scaled_mm ok, rel err", float((y.float() - ref).abs().max() / ref.abs().max()))
    for _ in range(5): torch._scaled_mm(a, b.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): torch._scaled_mm(a, b.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(f"mxfp8 {M}x{N}x{K}: {dt*1e6:.1f} us, {2*M*N*K/dt/1e12:.0f} TFLOP/s")
except Exception as e:  # noqa: BLE001
    print("mxfp8 _scaled_mm failed:", type(e).__name__, str(e)[:300])
try:
    a4 = torch.randint(0, 255, (M, K // 2), device="cuda", dtype=torch.uint8).view(torch.float4_e2m1fn_x2)
    b4 = torch.randint(0, 255, (N, K // 2), device="cuda", dtype=torch.uint8).view(torch.float4_e2m1fn_x2)
    y = torch._scaled_mm(a4, b4.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): torch._scaled_mm(a4, b4.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(f"mxfp4 {M}x{N}x{K}: {dt*1e6:.1f} us, {2*M*N*K/dt/1e12:.0f} TFLOP/s")
except Exception as e:  # noqa: BLE001
    print("mxfp4 _scaled_mm failed:", type(e).__name__, str(e)[:300])
