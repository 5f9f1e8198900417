"""GPU experiment: fused attention kernel vs the torch chain it replaces (LLaMA-2-7B and BERT-base shapes)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import torch
import quantized_training as qt
from quantized_training import _native as nv
L = nv.lib()
dev = torch.device("cuda")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, H, S, D, causal) in [(1, 32, 1024, 128, True), (16, 12, 384, 64, False), (1, 40, 1024, 128, True)]:
    q = torch.randn(B, H, S, D, device=dev).bfloat16(); k = torch.randn(B, H, S, D, device=dev).bfloat16(); v = torch.randn(B, H, S, D, device=dev).bfloat16()
    minv = torch.finfo(torch.bfloat16).min
    mask = torch.full((S, S), minv, device=dev).triu(1).bfloat16()[None, None] if causal else torch.zeros(B, 1, 1, S, device=dev, dtype=torch.bfloat16)
    msb = 0 if causal else mask.stride(0); msq = mask.stride(2) if causal else 0
    fmt = nv.format_for("e4m3"); lut = qt.get_quantization_map("e4m3", dev)
    out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scaling = D ** -0.5
    def fused():
        nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr(), out.data_ptr(), B, H, S, S, D,
                                        msb, 0, msq, scaling, ctypes.byref(fmt), lut.data_ptr(), None, None, st), "attn")
    probs = torch.empty(B, H, S, S, dtype=torch.bfloat16, device=dev)
    def chain():
        s = torch.matmul(q, k.transpose(2, 3))
        nv.check(L.qt_softmax_fq_bf16(s.data_ptr(), mask.data_ptr(), probs.data_ptr(), B, H, S, S, msb, 0, msq, scaling,
                                      ctypes.byref(fmt), lut.data_ptr(), None, None, st), "sm")
        return torch.matmul(probs, v).transpose(1, 2).contiguous()
    flops = 4.0 * B * H * S * S * D
    tf, tc = t(fused), t(chain)
    print(f"B{B} H{H} S{S} D{D} causal={causal}: fused {tf:.1f} us ({flops/tf/1e6:.0f} TF useful) | chain {tc:.1f} us", flush=True)
