import torch
from torch.profiler import profile, ProfilerActivity
a = torch.randn(1024, 4096, device="cuda").to(torch.float8_e4m3fn)
b = torch.randn(4096, 4096, device="cuda").to(torch.float8_e4m3fn)
one = torch.ones((), device="cuda")
for _ in range(3):
    torch._scaled_mm(a, b.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        torch._scaled_mm(a, b.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=80))
