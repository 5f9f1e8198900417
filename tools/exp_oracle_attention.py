"""GPU experiment: how far are qt_softmax_fq_bf16 / qt_attention_fq_bf16 from the oracle's float64 restatement of the chain
(oracle.qt_oracle.softmax_fq / attention_fq)?  Prints the share of differing elements and the largest difference in bf16 ULPs;
the thresholds in tests/test_gpu_parity.py come from these figures."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
sys.path.insert(0, ROOT)
from oracle import qt_oracle as o  # noqa: E402
from quantized_training import _native as nv  # noqa: E402

L = nv.lib()


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def u16(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def ulp(a, b):
    key = lambda t: np.where(t >= 0x8000, 0x8000 - t.astype(np.int32), t.astype(np.int32))  # noqa: E731
    return np.abs(key(a) - key(b))


def masks(kind, B, Q, C):
    minv = torch.finfo(torch.bfloat16).min
    if kind == "causal":
        m = torch.full((Q, C), minv, device="cuda").triu(1 + C - Q).bfloat16()[None, None]
        return m, 0, m.stride(2)
    if kind == "padding":
        m = torch.zeros(B, 1, 1, C, device="cuda", dtype=torch.bfloat16)
        m[:, :, :, C - 37:] = minv
        return m, m.stride(0), 0
    return None, 0, 0


def softmax_case(shape, kind):
    B, H, Q, C = shape
    torch.manual_seed(1)
    scores = (torch.randn(shape, device="cuda") * 3).bfloat16()
    scaling = 0.08838834764831845
    mask, msb, msq = masks(kind, B, Q, C)
    fmt = nv.format_for(None)
    out = torch.empty_like(scores)
    nv.check(L.qt_softmax_fq_bf16(scores.data_ptr(), mask.data_ptr() if mask is not None else None, out.data_ptr(), B, H, Q, C, msb, 0,
                                  msq, scaling, ctypes.byref(fmt), None, None, None, stream()), "softmax")
    pb, _ = o.softmax_fq(u16(scores), u16(mask) if mask is not None else None, scaling, None)
    d = ulp(u16(out), pb)
    print(f"softmax {shape} {kind}: differing {np.mean(d > 0):.2e}, max {d.max()} ULP", flush=True)


def attention_case(B, H, Sq, Sk, D, kind, pdtype):
    torch.manual_seed(B * 7 + H)
    qm = torch.from_numpy(o.get_quantization_map("e4m3").view(np.int16)).cuda().view(torch.bfloat16)
    fq = lambda t: qm[(t.view(torch.int16).to(torch.int32) & 0xFFFF).long()]  # noqa: E731
    q, k, v = (fq(torch.randn(B, H, s, D, device="cuda").bfloat16()) for s in (Sq, Sk, Sk))
    scaling = D ** -0.5
    mask, msb, msq = masks(kind, B, Sq, Sk)
    fmt = nv.format_for(pdtype)
    lut = torch.from_numpy(nv.build_map_u16(pdtype).view(np.int16)).cuda()
    out = torch.empty(B, Sq, H, D, dtype=torch.bfloat16, device="cuda")
    nv.check(L.qt_attention_fq_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mask.data_ptr() if mask is not None else None, out.data_ptr(),
                                    B, H, Sq, Sk, D, msb, 0, msq, scaling, ctypes.byref(fmt), lut.data_ptr(), None, None, stream()), "attn")
    exp, _ = o.attention_fq(u16(q), u16(k), u16(v), u16(mask) if mask is not None else None, scaling,
                            o.get_quantization_map(pdtype) if pdtype else None)
    got = u16(out.permute(0, 2, 1, 3))
    d = ulp(got, exp)
    ev = o.bf16_to_f32(exp)
    rel = np.abs(o.bf16_to_f32(got) - ev) / (np.abs(ev).max(axis=-1, keepdims=True) + 1e-30)
    print(f"attention B{B} H{H} Sq{Sq} Sk{Sk} D{D} {kind} {pdtype}: differing {np.mean(d > 0):.2e}, >1 ULP {np.mean(d > 1):.2e}, "
          f"max {d.max()} ULP, max |err| / row max {rel.max():.2e}", flush=True)


if __name__ == "__main__":
    for shape, kind in (((2, 4, 128, 128), "causal"), ((1, 32, 1024, 1024), "causal"), ((16, 12, 384, 384), "padding"), ((2, 3, 40, 72), None)):
        softmax_case(shape, kind)
    for case in ((1, 4, 128, 128, 128, "causal"), (2, 3, 200, 200, 64, "padding"), (1, 2, 64, 320, 128, None), (1, 32, 1024, 1024, 128, "causal")):
        for pd in (None, "e4m3", "posit8_1"):
            attention_case(*case, pd)
