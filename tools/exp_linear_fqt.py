"""GPU experiment: qt_linear_fqt_bf16 (bf16 GEMM with ANY value map applied to the weights in its operand path) against the
two-kernel route it replaces (LDS-table weight pass qt_fake_quant_bf16 + library bf16 GEMM).

    python tools/exp_linear_fqt.py [--iters 30] [--shapes 13b|7b|bert|all] [--dtypes posit8_2,...] [--skip-checks]

Prints, per dtype: exactness through an identity activation (y = fq(W)^T bit for bit on all 65 536 bf16 patterns, rows that
meet flagged table rows included), the accumulation error against an fp64 product of the quantized operands, and microseconds
per call of both routes (weights rotate over a pool larger than the Infinity Cache).
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

L = _native.lib()
DEV = torch.device("cuda:0")
POOL = int(os.environ.get("POOL", "0"))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


class Fmt:
    """Device-side description of one dtype: the value map, its row form and the elementwise format."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.map_host = _native.build_map_u16(dtype)
        self.rp = _native.build_rowparams(self.map_host)
        rows = np.ctypeslib.as_array(self.rp.row).reshape(512 * 4).astype(np.uint32)
        self.rows = torch.from_numpy(rows.view(np.int32).copy()).to(DEV)
        self.map = torch.from_numpy(self.map_host.view(np.int16).copy()).to(DEV)
        self.fmt = _native.format_for(dtype)

    def fq(self, x):
        """bf16 tensor -> bf16 values of fq(x) with the (oracle-pinned) elementwise pass."""
        y = torch.empty_like(x)
        one = torch.ones((), dtype=torch.float32, device=DEV)
        _native.check(L.qt_fake_quant_bf16(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(self.fmt), self.map.data_ptr(), one.data_ptr(),
                                           None, stream()), "qt_fake_quant_bf16")
        return y


_WS = {}


def plan(M, N, K):
    ks, wb, nt = ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(L.qt_linear_fqt_plan(M, N, K, ctypes.byref(ks), ctypes.byref(wb), ctypes.byref(nt)), "qt_linear_fqt_plan")
    return ks.value, wb.value, nt.value


def linear_fqt(x, ws, f, biases=None, split=True):
    """split=True: qt_linear_fqt_ws_bf16 with the workspace its plan asks for; False: the entry point that never splits K."""
    M, K = x.shape
    n = len(ws)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in (biases or [None] * n)])
    ns = (ctypes.c_int * n)(*[w.shape[0] for w in ws])
    N = sum(w.shape[0] for w in ws)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    if not split:
        _native.check(L.qt_linear_fqt_bf16(x.data_ptr(), wp, bp, ns, n, f.rows.data_ptr(), f.rp.signed_rows, f.rp.sign_mask, f.map.data_ptr(),
                                           y.data_ptr(), M, K, stream()), "qt_linear_fqt_bf16")
        return y
    ks, wb, nt = plan(M, N, K)
    if "ws" not in _WS or _WS["ws"].numel() * 4 < wb or _WS["tickets"].numel() < nt:
        _WS["ws"] = torch.empty((max(wb, 16) // 4,), dtype=torch.float32, device=DEV)
        _WS["tickets"] = torch.zeros((max(nt, 1),), dtype=torch.int32, device=DEV)
    _native.check(L.qt_linear_fqt_ws_bf16(x.data_ptr(), wp, bp, ns, n, f.rows.data_ptr(), f.rp.signed_rows, f.rp.sign_mask, f.map.data_ptr(),
                                          y.data_ptr(), M, K, _WS["ws"].data_ptr(), _WS["ws"].numel() * 4, _WS["tickets"].data_ptr(),
                                          _WS["tickets"].numel(), stream()), "qt_linear_fqt_ws_bf16")
    return y


def exact_check(f, K=512):
    """x = identity: y[m][n] = fq(W)[n][m] exactly, for every bf16 pattern.  K = 1536 takes the split-K path (two workgroups per tile,
    the flagged rows' redo decided by the tile's last arriver)."""
    torch.manual_seed(1)
    W = (torch.randn(400, K, device=DEV) * 3).bfloat16()
    W.view(torch.int16).view(-1)[:65536] = torch.arange(65536, device=DEV, dtype=torch.int32).to(torch.int16)
    eye = torch.eye(K, device=DEV).bfloat16()
    bias = torch.randn(W.shape[0], device=DEV).bfloat16()
    ok = True
    for sanitize in (True, False):
        Wt = W.clone()
        if sanitize:
            Wt[~torch.isfinite(Wt.float())] = 0
        want = f.fq(Wt)
        y = linear_fqt(eye, [Wt], f).t().contiguous()
        bad_rows = torch.isnan(want.float()).any(dim=1)
        same = (y.view(torch.int16) == want.view(torch.int16)) | ((y.float() == 0) & (want.float() == 0))
        ok1 = bool(same[~bad_rows].all()) and bool(torch.isnan(y[bad_rows].float()).all())
        yb = linear_fqt(eye, [Wt], f, [bias]).t().contiguous()
        ok2 = torch.equal((want.float() + bias.float()[:, None]).bfloat16()[~bad_rows].view(torch.int16), yb[~bad_rows].view(torch.int16))
        print(f"exact {f.dtype} K={K} (ksplit {plan(K, 400, K)[0]}) sanitized={sanitize}: identity-activation parity {ok1} (mismatches {int((~same[~bad_rows]).sum())}), with bias {ok2}, "
              f"rows holding NaN {int(bad_rows.sum())}, flagged table rows {f.rp.n_flagged}", flush=True)
        ok = ok and ok1 and ok2
    return ok


def accuracy(M, Ns, K, f, scale=0.05):
    torch.manual_seed(0)
    x = f.fq(torch.randn(M, K, device=DEV).bfloat16())
    ws = [(torch.randn(n, K, device=DEV) * scale).bfloat16() for n in Ns]
    bs = [torch.randn(n, device=DEV).bfloat16() if i % 2 == 0 else None for i, n in enumerate(Ns)]
    y16 = linear_fqt(x, ws, f, bs)
    again = linear_fqt(x, ws, f, bs)
    det = torch.equal(y16.view(torch.int16), again.view(torch.int16))           # split-K sums in split order: run-to-run identical
    clean = "tickets" not in _WS or not bool(_WS["tickets"].any())              # every launch leaves its tickets zero
    y = y16.double()
    xa = x.double()
    wa = torch.cat([f.fq(w).double() for w in ws])
    bias = torch.cat([b.double() if b is not None else torch.zeros(n, device=DEV, dtype=torch.float64) for b, n in zip(bs, Ns)])
    ref = xa @ wa.t() + bias
    bound = xa.abs() @ wa.abs().t()
    err = (y - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + bound * 2.0 ** -18 + 1e-30       # one bf16 rounding + fp32 accumulation
    rel = float((err / tol).max())
    print(f"accuracy {f.dtype} {M}x{sum(Ns)}x{K} (ksplit {plan(M, sum(Ns), K)[0]}): max err / tolerance = {rel:.3f}  (max |err| {float(err.max()):.3e})  "
          f"deterministic {det}  tickets zero {clean}", flush=True)
    return rel <= 1.0 and det and clean


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


def bench(M, Ns, K, iters, f):
    torch.manual_seed(0)
    N = sum(Ns)
    pool = POOL or max(2, min(16, int(600e6 // (N * K * 2)) + 1))
    x = f.fq(torch.randn(M, K, device=DEV).bfloat16())
    wsets = [[(torch.randn(n, K, device=DEV) * 0.02).bfloat16() for n in Ns] for _ in range(pool)]
    wcat = [torch.cat(ws) for ws in wsets]
    wq = torch.empty((N, K), dtype=torch.bfloat16, device=DEV)
    one = torch.ones((), dtype=torch.float32, device=DEV)

    def fused(i):
        return linear_fqt(x, wsets[i % pool], f)

    def unsplit(i):
        return linear_fqt(x, wsets[i % pool], f, split=False)

    def two_kernel(i):
        W = wcat[i % pool]
        _native.check(L.qt_fake_quant_bf16(W.data_ptr(), wq.data_ptr(), W.numel(), ctypes.byref(f.fmt), f.map.data_ptr(), one.data_ptr(), None,
                                           stream()), "fq")
        return torch.nn.functional.linear(x, wq)

    def gemm_only(i):
        return torch.nn.functional.linear(x, wcat[i % pool])

    t_f = timeit(fused, iters)
    ks = plan(M, N, K)[0]
    t_u = timeit(unsplit, iters) if ks > 1 else t_f
    t_2 = timeit(two_kernel, iters)
    t_g = timeit(gemm_only, iters)
    flops = 2.0 * M * N * K
    print(f"bench {f.dtype} {M}x{N}x{K} (segments {Ns}, ksplit {ks}): unsplit {t_u:7.1f} us   fused {t_f:7.1f} us ({flops / t_f / 1e6:6.0f} TFLOP/s = {flops / t_f / 1e6 / 2500:.3f} of bf16 peak)   "
          f"pass + library {t_2:7.1f} us   library GEMM alone {t_g:7.1f} us   speed-up {t_2 / t_f:.2f}x", flush=True)
    return t_f, t_2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shapes", default="13b")
    ap.add_argument("--dtypes", default="posit8_2,posit8_1,int8,fp6_e3m2,fp4_e2m1,e4m3")
    ap.add_argument("--bench-dtypes", default="posit8_2")
    ap.add_argument("--skip-checks", action="store_true")
    args = ap.parse_args()
    ok = True
    fmts = {d: Fmt(d) for d in set(args.dtypes.split(",")) | set(args.bench_dtypes.split(","))}
    if not args.skip_checks:
        for d in args.dtypes.split(","):
            ok &= exact_check(fmts[d])
            ok &= exact_check(fmts[d], K=1536)
        for d in args.dtypes.split(",")[:3]:
            for (M, Ns, K) in ((1024, [4096], 1024), (1024, [176], 256), (300, [48, 64, 16], 384), (1, [16], 64), (777, [2048, 512, 512], 512),
                               (257, [208, 4096 - 208], 128),
                               # split-K: 2, 4 and 3 workgroups per tile, ragged rows, several weights, uneven k ranges
                               (1024, [1024], 2048), (1024, [2048], 3072), (777, [512, 512], 2400), (520, [1024], 4000 - 32)):
                ok &= accuracy(M, Ns, K, fmts[d])
    s13 = [(1024, [13824], 5120), (1024, [5120], 13824), (1024, [5120], 5120), (1024, [5120, 5120, 5120], 5120), (1024, [32000], 5120)]
    s7 = [(1024, [11008], 4096), (1024, [4096], 11008), (1024, [4096], 4096), (1024, [4096, 4096, 4096], 4096), (1024, [32000], 4096)]
    bert = [(6144, [768], 768), (6144, [3072], 768), (6144, [768], 3072), (6144, [768, 768, 768], 768)]
    if "x" in args.shapes:      # custom: "1024x14336x5120,1024x12288x5120"
        shapes = [(int(m), [int(n)], int(k)) for m, n, k in (t.split("x") for t in args.shapes.split(","))]
    else:
        shapes = {"13b": s13, "7b": s7, "bert": bert, "all": s13 + s7 + bert, "one": s13[:1], "none": []}[args.shapes]
    for d in args.bench_dtypes.split(","):
        for (M, Ns, K) in shapes:
            bench(M, Ns, K, args.iters, fmts[d])
    print("ALL CHECKS", "PASSED" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
