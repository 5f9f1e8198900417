"""GPU experiment: qt_linear_fq8_bf16 (FP8 GEMM with the weight fake-quantizer fused into its operand path) against
the two-kernel route it replaces (qt_fake_quant_bf16_fp8 weight pass + qt_fp8_gemm through hipBLASLt).

    python tools/exp_linear_fq8.py [--iters 50] [--shapes llama|bert|all]

Prints, per shape: exactness through an identity activation (y = fq(W)^T bit for bit, incl. overflow / Inf / NaN
weights), the accumulation error against an fp64 product of the decoded codes, and microseconds per call of both routes
(weights rotate over a pool larger than the Infinity Cache, as in the real window where every layer has its own).
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
sys.path.insert(0, ROOT)

from quantized_training import _native  # noqa: E402
from quantized_training.fused import lt_fp8_gemm  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")
F8 = {0: torch.float8_e4m3fn, 1: torch.float8_e5m2}
NAME = {0: "e4m3", 1: "e5m2"}
POOL = int(os.environ.get("POOL", "0"))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


def fq8(x, fmt_id):
    """bf16 tensor -> FP8 codes of fq(x) (uint8) with the exact elementwise pass."""
    fmt = _native.format_for(NAME[fmt_id])
    y8 = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    one = torch.ones((), dtype=torch.float32, device=DEV)
    _native.check(L.qt_fake_quant_bf16_fp8(x.data_ptr(), None, y8.data_ptr(), x.numel(), ctypes.byref(fmt), one.data_ptr(), None,
                                           stream()), "fq8")
    return y8


def linear_fq8(x8, fx, ws, fw, biases=None):
    M, K = x8.shape
    n = len(ws)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in (biases or [None] * n)])
    ns = (ctypes.c_int * n)(*[w.shape[0] for w in ws])
    y = torch.empty((M, sum(w.shape[0] for w in ws)), dtype=torch.bfloat16, device=DEV)
    _native.check(L.qt_linear_fq8_bf16(x8.data_ptr(), fx, wp, bp, ns, n, fw, y.data_ptr(), M, K, stream()), "qt_linear_fq8_bf16")
    return y


def exact_check(fx, fw):
    """x = identity: y[m][n] = fq(W)[n][m] exactly."""
    K = 512
    torch.manual_seed(1)
    W = (torch.randn(400, K, device=DEV) * 3).bfloat16()
    Wb = W.view(torch.int16)
    # every bf16 pattern appears somewhere: rows 0..127 hold patterns 0 .. 65535
    Wb[:128] = torch.arange(65536, device=DEV, dtype=torch.int32).to(torch.int16).view(128, 512)
    eye = torch.eye(K, device=DEV).bfloat16()
    x8 = fq8(eye, fx)
    bias = torch.randn(W.shape[0], device=DEV).bfloat16()
    ok = True
    for sanitize in (True, False):
        Wt = W.clone()
        if sanitize:                                      # non-finite patterns -> 0: every row can be compared element by element
            Wt[~torch.isfinite(Wt.float())] = 0
        w8 = fq8(Wt, fw).view(F8[fw]).float().bfloat16()     # decoded codes = fq(W)
        y = linear_fq8(x8, fx, [Wt], fw).t().contiguous()
        bad_rows = torch.isnan(w8).any(dim=1)              # 0 * NaN: a NaN weight poisons its whole output column
        same = (y.view(torch.int16) == w8.view(torch.int16)) | ((y.float() == 0) & (w8.float() == 0))
        ok1 = bool(same[~bad_rows].all()) and bool(torch.isnan(y[bad_rows]).all())
        yb = linear_fq8(x8, fx, [Wt], fw, [bias]).t().contiguous()
        ok2 = torch.equal((w8.float() + bias.float()[:, None]).bfloat16()[~bad_rows].view(torch.int16), yb[~bad_rows].view(torch.int16))
        print(f"exact x={NAME[fx]} w={NAME[fw]} sanitized={sanitize}: identity-activation parity {ok1}, with bias {ok2}, "
              f"rows holding NaN {int(bad_rows.sum())}")
        ok = ok and ok1 and ok2
    return ok


def accuracy(M, Ns, K, fx, fw):
    torch.manual_seed(0)
    x = torch.randn(M, K, device=DEV).bfloat16()
    ws = [(torch.randn(n, K, device=DEV) * 0.05).bfloat16() for n in Ns]
    x8 = fq8(x, fx)
    y = linear_fq8(x8, fx, ws, fw).float()
    xa = x8.view(F8[fx]).double()
    wa = torch.cat([fq8(w, fw).view(F8[fw]).double() for w in ws])
    ref = xa @ wa.t()
    bound = (xa.abs() @ wa.abs().t())
    err = (y.double() - ref).abs()
    # bf16 rounding of the result (2^-9 relative) + the instruction's accumulation error (<= 2^-14 sum |a||b|)
    tol = ref.abs() * 2.0 ** -8 + bound * 2.0 ** -14 + 1e-30
    rel = float((err / tol).max())
    print(f"accuracy {M}x{sum(Ns)}x{K} x={NAME[fx]} w={NAME[fw]}: max err / tolerance = {rel:.3f}  (max |err| {float(err.max()):.3e})")
    return rel <= 1.0


def timeit(fn, iters):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def bench(M, Ns, K, iters, fx=0, fw=0):
    torch.manual_seed(0)
    N = sum(Ns)
    pool = POOL or max(2, min(16, int(600e6 // (N * K * 2)) + 1))
    x = torch.randn(M, K, device=DEV).bfloat16()
    x8 = fq8(x, fx)
    wsets = [[(torch.randn(n, K, device=DEV) * 0.05).bfloat16() for n in Ns] for _ in range(pool)]
    wcat = [torch.cat(ws) for ws in wsets]
    w8buf = torch.empty((N, K), dtype=torch.uint8, device=DEV)
    fmt = _native.format_for(NAME[fw])
    one = torch.ones((), dtype=torch.float32, device=DEV)

    def fused(i):
        return linear_fq8(x8, fx, wsets[i % pool], fw)

    def two_kernel(i):
        W = wcat[i % pool]
        _native.check(L.qt_fake_quant_bf16_fp8(W.data_ptr(), None, w8buf.data_ptr(), W.numel(), ctypes.byref(fmt), one.data_ptr(), None,
                                               stream()), "fq8")
        return lt_fp8_gemm(x8.view(F8[fx]), w8buf.view(F8[fw]), None)

    def gemm_only(i):
        return lt_fp8_gemm(x8.view(F8[fx]), w8buf.view(F8[fw]), None)

    two_kernel(0)
    t_f = timeit(fused, iters)
    flops = 2.0 * M * N * K
    if os.environ.get("FUSED_ONLY"):
        print(f"bench {M}x{N}x{K}: fused {t_f:7.1f} us ({flops / t_f / 1e6:6.0f} TFLOP/s)", flush=True)
        return t_f, t_f
    t_2 = timeit(two_kernel, iters)
    t_g = timeit(gemm_only, iters)
    print(f"bench {M}x{N}x{K} (segments {Ns}): fused {t_f:7.1f} us ({flops / t_f / 1e6:6.0f} TFLOP/s)   "
          f"pass + hipBLASLt {t_2:7.1f} us   hipBLASLt alone {t_g:7.1f} us   speed-up {t_2 / t_f:.2f}x", flush=True)
    return t_f, t_2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--skip-checks", action="store_true")
    args = ap.parse_args()
    ok = True
    if not args.skip_checks:
        for fx, fw in ((0, 0), (1, 1), (0, 1), (1, 0)):
            ok &= exact_check(fx, fw)
        for (M, Ns, K) in ((1024, [4096], 4096), (1024, [176], 256), (300, [48, 64, 16], 384), (1024, [4096, 4096, 4096], 1024),
                           (520, [11008], 512)):
            ok &= accuracy(M, Ns, K, 0, 0)
        ok &= accuracy(640, [768], 768, 1, 1)
    llama = [(1024, [11008], 4096), (1024, [4096], 11008), (1024, [4096], 4096), (1024, [4096, 4096, 4096], 4096), (1024, [32000], 4096)]
    bert = [(6144, [768], 768), (6144, [3072], 768), (6144, [768], 3072), (6144, [768, 768, 768], 768)]
    probe = [(1024, [11008], 4096), (1024, [11008], 4224), (1024, [4096], 11008)]
    # other window lengths and the 13B widths: where does the fused kernel beat the pair?
    sweep = [(512, [11008], 4096), (2048, [11008], 4096), (2048, [4096], 4096), (4096, [4096], 4096), (1024, [13824], 5120), (1024, [5120], 13824),
             (1024, [5120], 5120), (1024, [5120, 5120, 5120], 5120), (1024, [8192], 4096), (1024, [6144], 4096), (256, [11008], 4096)]
    if "x" in args.shapes:      # custom: "1024x4096x5504,1024x4096x11008"
        shapes = [(int(m), [int(n)], int(k)) for m, n, k in (t.split("x") for t in args.shapes.split(","))]
    else:
        shapes = {"llama": llama, "bert": bert, "all": llama + bert, "probe": probe, "sweep": sweep}[args.shapes]
    for (M, Ns, K) in shapes:
        bench(M, Ns, K, args.iters)
    print("ALL CHECKS", "PASSED" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
