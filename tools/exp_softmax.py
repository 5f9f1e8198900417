"""Score pass (scale + mask + softmax + fake-quant) on a LLaMA-2-7B window: causal vs no mask, bf16 vs FP8 output."""
import ctypes, sys, time, torch
sys.path.insert(0, "quantized-training_amd")
from quantized_training import _native as nv
L = nv.lib()
B, H, S = 1, 32, 1024
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scores = [torch.randn(B, H, S, S, device="cuda").bfloat16() * 4 for _ in range(4)]
mask = torch.full((S, S), torch.finfo(torch.bfloat16).min, device="cuda").triu(1).bfloat16()[None, None].contiguous()
out = torch.empty_like(scores[0]); out8 = torch.empty(B, H, S, S, dtype=torch.uint8, device="cuda")
fmt = nv.format_for("e4m3")
def run(kind, m):
    mp = m.data_ptr() if m is not None else None
    def f(i):
        if kind == "bf16":
            nv.check(L.qt_softmax_fq_bf16(scores[i % 4].data_ptr(), mp, out.data_ptr(), B, H, S, S, 0, 0, S if m is not None else 0, 0.088, ctypes.byref(fmt), None, None, None, st), "s")
        else:
            nv.check(L.qt_softmax_fq_bf16_fp8(scores[i % 4].data_ptr(), mp, None, out8.data_ptr(), B, H, S, S, 0, 0, S if m is not None else 0, 0.088, ctypes.byref(fmt), st), "s8")
    for i in range(4): f(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(40): f(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 40 * 1e6
for kind in ("bf16", "fp8"):
    for name, m in (("causal", mask), ("no mask", None)):
        print(f"{kind} output, {name}: {run(kind, m):.1f} us")
