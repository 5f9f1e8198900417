"""Times the one-launch FP8 attention core (qt_value_codes_t + qt_attention_fp8) against the chain it replaces -- Q.K^T as a batched
FP8 GEMM, the score pass (qt_softmax_fq_bf16_fp8), the value pass (qt_fake_quant_rows_bf16_fp8) and P.V as a batched FP8 GEMM -- at the
LLaMA-2-7B window shape (B 1, H 32, S 1024, head_dim 128, causal mask) and a few others.   python tools/exp_attention_fp8.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
from quantized_training import _native  # noqa: E402
from quantized_training.fused import lt_fp8_gemm  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")


def st():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    fmt = _native.format_for("e4m3")
    for (B, H, S, causal, D) in ((1, 32, 1024, True, 128), (1, 32, 1024, False, 128), (1, 32, 512, True, 128), (4, 32, 256, True, 128),
                                 (16, 12, 384, False, 64), (16, 12, 128, False, 64)):
        q8 = torch.randn(B, H, S, D, device=DEV).to(torch.float8_e4m3fn)
        k8 = torch.randn(B, H, S, D, device=DEV).to(torch.float8_e4m3fn)
        v = torch.randn(B, S, H, D, device=DEV).bfloat16().transpose(1, 2)           # as the v projection leaves it
        mask = torch.full((S, S), torch.finfo(torch.bfloat16).min, device=DEV).triu(1).bfloat16()[None, None] if causal else None
        live = None
        if mask is not None:
            live = torch.empty(S, dtype=torch.int32, device=DEV)
            _native.check(L.qt_mask_row_live(mask.data_ptr(), S, S, S, live.data_ptr(), st()), "live")
        vt8 = torch.empty(B, H, D, S, dtype=torch.uint8, device=DEV)
        out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=DEV)
        v8 = torch.empty(B, H, S, D, dtype=torch.uint8, device=DEV)
        p8 = torch.empty(B, H, S, S, dtype=torch.uint8, device=DEV)
        scaling = D ** -0.5
        mp = mask.data_ptr() if mask is not None else None

        def vpass_t():
            _native.check(L.qt_value_codes_t(v.data_ptr(), vt8.data_ptr(), B, H, S, D, v.stride(0), v.stride(1), v.stride(2), ctypes.byref(fmt), st()), "vt")

        simple = True

        def core(with_live=True):
            _native.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), 0, mp, 0, 0, S if mask is not None else 0,
                                             live.data_ptr() if (live is not None and with_live) else None, 0, 0, 1, int(simple and with_live), None, out.data_ptr(), None, None, B, H, S, S,
                                             D, scaling, st()), "attn")

        def one_launch():
            vpass_t(); core()

        def chain():
            sc = lt_fp8_gemm(q8.view(B * H, S, D), k8.view(B * H, S, D))                                  # [BH, S, S] bf16
            _native.check(L.qt_softmax_fq_bf16_fp8(sc.data_ptr(), mp, None, p8.data_ptr(), B, H, S, S, 0, 0, S if mask is not None else 0,
                                                   scaling, ctypes.byref(fmt), st()), "softmax")
            _native.check(L.qt_fake_quant_rows_bf16_fp8(v.data_ptr(), None, v8.data_ptr(), B, H, S, D, v.stride(0), v.stride(1), v.stride(2),
                                                        ctypes.byref(fmt), st()), "vpass")
            return lt_fp8_gemm(p8.view(torch.float8_e4m3fn).view(B * H, S, S), v8.view(torch.float8_e4m3fn).view(B * H, S, D), None, b_is_kn=True)

        if chain() is None:
            print("library route unavailable"); continue
        t1, tc = timeit(one_launch), timeit(chain)
        tv, ta, tn = timeit(vpass_t), timeit(core), timeit(lambda: core(False))
        if mask is not None and B == 1 and S == 1024:
            # the call as the model issues it: the output projection's fake-quantizer on the epilogue, the mask's regularity as a device
            # flag; and the same call with a large GEMM-like stream of other work in between (cold caches, sustained clocks)
            out8 = torch.empty(B, S, H, D, dtype=torch.uint8, device=DEV)
            rl = torch.empty(S + 1, dtype=torch.int32, device=DEV)
            _native.check(L.qt_mask_row_live_checked(mask.data_ptr(), S, S, S, rl.data_ptr(), rl.data_ptr() + 4 * S, st()), "live")

            def as_issued():
                _native.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), 0, mp, 0, 0, S, rl.data_ptr(), 0, 0, 1, 0,
                                                 rl.data_ptr() + 4 * S, out.data_ptr(), out8.data_ptr(), ctypes.byref(fmt), B, H, S, S, D, scaling,
                                                 st()), "attn")
            big_a = torch.randn(4096, 4096, device=DEV).bfloat16()
            big_b = torch.randn(4096, 8192, device=DEV).bfloat16()

            def busy():
                torch.mm(big_a, big_b)

            def busy_then_attn():
                busy(); as_issued()
            # inputs rewritten by another kernel in front of every launch (as the rotary kernel does in the window): are the core's reads
            # served from the writer's L2, or from the Infinity Cache?
            q_src, k_src, v_src = q8.clone(), k8.clone(), vt8.clone()

            def rewrite():
                q8.copy_(q_src); k8.copy_(k_src); vt8.copy_(v_src)

            def rewrite_then_attn():
                rewrite(); as_issued()
            tr, tra = timeit(rewrite), timeit(rewrite_then_attn)
            print(f"    behind kernels that rewrite q / k / v codes: {tra - tr:6.1f} us (the copies alone {tr:6.1f})", flush=True)
            ti, tb, tba = timeit(as_issued), timeit(busy), timeit(busy_then_attn)
            print(f"    as issued by the model (epilogue codes, device flag): {ti:6.1f} us; behind a 275-GFLOP bf16 GEMM: {tba - tb:6.1f} us "
                  f"(GEMM alone {tb:6.1f})", flush=True)
        print(f"B{B} H{H} S{S} D{D} {'causal' if causal else 'no mask'}: one launch {t1:6.1f} us (value codes {tv:5.1f} + core {ta:6.1f}; core without row extents {tn:6.1f})"
              f" | chain {tc:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
