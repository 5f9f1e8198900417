"""GPU experiment: qt_attention_rows_bf16 (+ qt_value_t_rows) -- the round-4 attention core for table formats -- against round 3's two-pass
kernel (qt_attention_fq_live_bf16) at the LLaMA-2-13B shape (40 heads x 1024 x 1024 x 128, causal) and smaller ones.

    python tools/exp_attention_rows.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import quantized_training as qt  # noqa: E402
from quantized_training import _native  # noqa: E402
from quantized_training.fake_quantize import _launch_format  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")


def st():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


def timeit(fn, iters=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    dt = "posit8_2"
    m = qt.get_quantization_map(dt, DEV)
    f = _launch_format(_native.format_for(dt), m)
    for (B, H, S, causal) in [(1, 40, 1024, True), (1, 32, 1024, True), (1, 40, 1024, False), (1, 40, 512, True), (4, 40, 256, True)]:
        D = 128
        q = m[(torch.randn(B, H, S, D, device=DEV).bfloat16().view(torch.int16).to(torch.int32) & 0xFFFF).long()]
        k = m[(torch.randn(B, H, S, D, device=DEV).bfloat16().view(torch.int16).to(torch.int32) & 0xFFFF).long()]
        v_raw = torch.randn(B, S, H, D, device=DEV).bfloat16().transpose(1, 2)
        v = m[(v_raw.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF).long()]
        mask = torch.full((S, S), torch.finfo(torch.bfloat16).min, device=DEV).triu(1).bfloat16()[None, None] if causal else None
        rl = None
        if mask is not None:
            rl = torch.empty(S + 1, dtype=torch.int32, device=DEV)
            _native.check(L.qt_mask_row_live_checked(mask.data_ptr(), S, S, S, rl.data_ptr(), rl.data_ptr() + 4 * S, st()), "live")
        vt = torch.empty(B, H, D, S, dtype=torch.bfloat16, device=DEV)
        out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=DEV)
        mp = mask.data_ptr() if mask is not None else None
        msq = mask.stride(2) if mask is not None else 0

        def value():
            _native.check(L.qt_value_t_rows(v_raw.data_ptr(), vt.data_ptr(), B, H, S, D, v_raw.stride(0), v_raw.stride(1), v_raw.stride(2), ctypes.byref(f),
                                            m.data_ptr(), st()), "value")

        def new():
            _native.check(L.qt_attention_rows_bf16(q.data_ptr(), k.data_ptr(), vt.data_ptr(), mp, 0, 0, msq, rl.data_ptr() if rl is not None else None, 0, 0, 1,
                                                   rl.data_ptr() + 4 * S if rl is not None else None, out.data_ptr(), 1, ctypes.byref(f), m.data_ptr(), B, H, S, S,
                                                   D, D ** -0.5, st()), "new")

        def old():
            if rl is not None:
                _native.check(L.qt_attention_fq_live_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(), B, H, S, S, D, 0, 0, msq, D ** -0.5,
                                                          ctypes.byref(f), m.data_ptr(), None, None, 1, rl.data_ptr(), 0, 0, 1, rl.data_ptr() + 4 * S, st()), "old")
            else:
                _native.check(L.qt_attention_fq_out_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), B, H, S, S, D, 0, 0, 0, D ** -0.5,
                                                         ctypes.byref(f), m.data_ptr(), st()), "old")
        value()
        print(f"B {B} H {H} S {S} {'causal' if causal else 'no mask'}: strip kernel {timeit(new):6.1f} us + value pass {timeit(value):5.1f} us    "
              f"two-pass kernel {timeit(old):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
